/*
 * scene_gen.c -- deterministic synthetic marker-mapping scenes.
 *
 * Bench/test input generator, not part of the hot path.  It stands in for the two things the
 * reference needs before its bundle adjustment can run and that cannot run here: the AprilTag
 * detector (marker_detections.json) and the OpenCV PnP initialisation
 * (/root/reference/src/TagReconstructor.cpp:156,167-230).  Output follows the reference's data
 * model: tag quad corners LL,LR,UR,UL (include/visual_marker_mapping/TagReconstructor.h:47-50),
 * camera pose world->camera, tag pose tag->world, quaternions (w,x,y,z).
 *
 * PRNG: splitmix64, uniforms from the top 53 bits, normals by Box-Muller (both values used).
 * Draw order (fixed): tags 1..N-1 {axis(3n), angle(n), offset(n)}; cameras {az(u), el(u), roll(n),
 * dist(u)} -- with neighbors_min > 0 instead {point x(u), point y(u), count(u), az(u), el(u), roll(n), dist(u)} and no
 * visibility draws; visibility (u per camera-tag pair, only when visibility < 1); per kept observation
 * 8 noise normals, then (only when outlier_frac > 0) per corner {u, u, u}; initial-guess
 * perturbation: cameras {axis(3n), angle(n), trans(3n)}, tags 1..N-1 likewise.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct vmm_scene_cfg {
    int n_cams;
    int n_tags;
    uint64_t seed;
    double visibility;    /* 1.0 = every image sees every tag */
    double noise_px;      /* sigma of the corner noise */
    double outlier_frac;  /* fraction of corners displaced by U(-outlier_px, outlier_px) per axis */
    double outlier_px;
    int use_distortion;   /* 0: zero distortion; 1: README.md:139-141 coefficients */
    double cam_rot_deg, cam_trans_m; /* initial-guess perturbation sigmas */
    double tag_rot_deg, tag_trans_m;
    /* > 0: "close-up" scenes like real projects (README.md:155-216: a handful of tags per image) -- every camera
     * stands in front of a random point of the wall and sees the neighbors_min..neighbors_max tags nearest to it
     * (`visibility` is ignored).  The reduced systems of such scenes are block-sparse. */
    int neighbors_min, neighbors_max;
    /* > 0: the tags hang in this many rows (a corridor: 1 or 2) instead of a wall about twice as wide as high */
    int wall_rows;
} vmm_scene_cfg;

typedef struct rng {
    uint64_t s;
    int have;
    double spare;
} rng;

static uint64_t next_u64(rng* r)
{
    uint64_t z = (r->s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static double uni(rng* r) { return (double)(next_u64(r) >> 11) * (1.0 / 9007199254740992.0); }
static double uni_ab(rng* r, double a, double b) { return a + (b - a) * uni(r); }
static double nrm(rng* r)
{
    if (r->have) {
        r->have = 0;
        return r->spare;
    }
    double u1 = uni(r), u2 = uni(r);
    if (u1 < 1e-300)
        u1 = 1e-300;
    const double m = sqrt(-2.0 * log(u1));
    r->spare = m * sin(6.283185307179586 * u2);
    r->have = 1;
    return m * cos(6.283185307179586 * u2);
}

static void q_mul(const double a[4], const double b[4], double o[4])
{
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}
static void q_axis_angle(const double ax[3], double ang, double q[4])
{
    const double n = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    const double s = (n > 0) ? sin(ang / 2) / n : 0.0;
    q[0] = cos(ang / 2);
    q[1] = s * ax[0];
    q[2] = s * ax[1];
    q[3] = s * ax[2];
}
static void q_to_R(const double q[4], double R[9])
{
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}
/* unit quaternion from a proper rotation matrix (row-major), w >= 0 */
static void R_to_q(const double R[9], double q[4])
{
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        double s = sqrt(tr + 1.0) * 2;
        q[0] = 0.25 * s; q[1] = (R[7] - R[5]) / s; q[2] = (R[2] - R[6]) / s; q[3] = (R[3] - R[1]) / s;
    } else if (R[0] > R[4] && R[0] > R[8]) {
        double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2;
        q[0] = (R[7] - R[5]) / s; q[1] = 0.25 * s; q[2] = (R[1] + R[3]) / s; q[3] = (R[2] + R[6]) / s;
    } else if (R[4] > R[8]) {
        double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2;
        q[0] = (R[2] - R[6]) / s; q[1] = (R[1] + R[3]) / s; q[2] = 0.25 * s; q[3] = (R[5] + R[7]) / s;
    } else {
        double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2;
        q[0] = (R[3] - R[1]) / s; q[1] = (R[2] + R[6]) / s; q[2] = (R[5] + R[7]) / s; q[3] = 0.25 * s;
    }
    if (q[0] < 0)
        for (int k = 0; k < 4; ++k)
            q[k] = -q[k];
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int k = 0; k < 4; ++k)
        q[k] /= n;
}

static void project(const double intr[4], const double d[5], const double cam[7], const double tag[7],
                    const double cl[3], double uv[2])
{
    double Rt[9], Rc[9], pw[3], pc[3];
    q_to_R(tag, Rt);
    q_to_R(cam, Rc);
    for (int k = 0; k < 3; ++k)
        pw[k] = Rt[3 * k] * cl[0] + Rt[3 * k + 1] * cl[1] + Rt[3 * k + 2] * cl[2] + tag[4 + k];
    for (int k = 0; k < 3; ++k)
        pc[k] = Rc[3 * k] * pw[0] + Rc[3 * k + 1] * pw[1] + Rc[3 * k + 2] * pw[2] + cam[4 + k];
    const double x = pc[0] / pc[2], y = pc[1] / pc[2], r2 = x * x + y * y;
    const double rad = 1 + r2 * (d[0] + r2 * (d[1] + r2 * d[4]));
    const double xd = x * rad + 2 * d[2] * x * y + d[3] * (r2 + 2 * x * x);
    const double yd = y * rad + 2 * d[3] * x * y + d[2] * (r2 + 2 * y * y);
    uv[0] = intr[0] * xd + intr[2];
    uv[1] = intr[1] * yd + intr[3];
}

static void perturb(rng* r, const double in[7], double rot_deg, double trans, double out[7])
{
    double ax[3] = { nrm(r), nrm(r), nrm(r) };
    const double ang = nrm(r) * rot_deg * 0.017453292519943295;
    double dq[4];
    q_axis_angle(ax, ang, dq);
    q_mul(dq, in, out);
    for (int k = 0; k < 3; ++k)
        out[4 + k] = in[4 + k] + nrm(r) * trans;
}

void vmm_scene_default_cfg(vmm_scene_cfg* c, int config_index)
{
    /* BASELINE.json configs[config_index-1]; SURVEY.md 8(d). */
    memset(c, 0, sizeof(*c));
    c->seed = 0x564D4D00ULL + (uint64_t)config_index;
    c->visibility = 1.0;
    c->noise_px = 0.3;
    c->cam_rot_deg = 2.0; c->cam_trans_m = 0.05;
    c->tag_rot_deg = 1.0; c->tag_trans_m = 0.02;
    switch (config_index) {
    case 1: c->n_cams = 20; c->n_tags = 10; break;
    case 4: c->n_cams = 2000; c->n_tags = 1000; break;
    case 5:
        c->n_cams = 500; c->n_tags = 200; c->use_distortion = 1; c->noise_px = 0.5;
        c->outlier_frac = 0.02; c->outlier_px = 20.0;
        break;
    default: c->n_cams = 500; c->n_tags = 200; break;
    }
}

/* Returns the number of observations written, or -(needed) when max_obs is too small. */
int vmm_scene_generate(const vmm_scene_cfg* cfg, double intr[4], double dist[5], double* cam_gt,
                       double* tag_gt, double* tag_wh, double* cam_init, double* tag_init,
                       int* obs_cam, int* obs_tag, double* obs_px, int max_obs)
{
    const int nc = cfg->n_cams, nt = cfg->n_tags;
    rng r = { cfg->seed, 0, 0.0 };
    /* README.md:132-145 */
    intr[0] = 8.0752937867635346e+03; intr[1] = 8.0831676114192869e+03;
    intr[2] = 3.0163896805084278e+03; intr[3] = 1.9962896554785455e+03;
    memset(dist, 0, 5 * sizeof(double));
    if (cfg->use_distortion) {
        dist[0] = -1.8618183262669760e-01; dist[1] = 3.7018092365577054e-01;
        dist[2] = -2.9390604003594177e-04; dist[3] = 4.1533180829908799e-04;
        dist[4] = 5.7043887874185996e-02;
    }
    const int gw = cfg->wall_rows > 0 ? (nt + cfg->wall_rows - 1) / cfg->wall_rows : (int)ceil(sqrt(2.0 * nt));
    const int gh = (nt + gw - 1) / gw;
    const double pitch = 0.30, side = 0.1285; /* main_detection.cpp:43,48 default marker size */
    for (int t = 0; t < nt; ++t) {
        double* g = tag_gt + 7 * t;
        tag_wh[2 * t] = side;
        tag_wh[2 * t + 1] = side;
        g[0] = 1; g[1] = g[2] = g[3] = 0;
        g[4] = pitch * (t % gw); g[5] = pitch * (t / gw); g[6] = 0;
        if (t > 0) {
            double ax[3] = { nrm(&r), nrm(&r), nrm(&r) };
            const double ang = nrm(&r) * 8.0 * 0.017453292519943295;
            q_axis_angle(ax, ang, g);
            g[6] = nrm(&r) * 0.03;
        }
    }
    const double wall_w = pitch * (gw - 1), wall_h = pitch * (gh - 1);
    const double ctr[3] = { wall_w / 2, wall_h / 2, 0 };
    const double d0 = (wall_w > 0.6 ? wall_w : 0.6) / (2.0 * tan(20.4 * 0.017453292519943295));
    const int close_up = cfg->neighbors_min > 0;
    double* look = (double*)malloc(sizeof(double) * 2 * (size_t)nc);   /* wall point each camera looks at */
    int* n_see = (int*)malloc(sizeof(int) * (size_t)nc);
    const double d0_close = (4.0 * pitch) / (2.0 * tan(20.4 * 0.017453292519943295));
    for (int c = 0; c < nc; ++c) {
        double at[3] = { ctr[0], ctr[1], ctr[2] };
        n_see[c] = nt;
        if (close_up) {
            at[0] = uni_ab(&r, 0.0, wall_w);
            at[1] = uni_ab(&r, 0.0, wall_h);
            const int span = cfg->neighbors_max - cfg->neighbors_min + 1;
            n_see[c] = cfg->neighbors_min + (int)(uni(&r) * (span > 0 ? span : 1));
            if (n_see[c] > nt)
                n_see[c] = nt;
        }
        look[2 * c] = at[0];
        look[2 * c + 1] = at[1];
        const double az = uni_ab(&r, -35, 35) * 0.017453292519943295;
        const double el = uni_ab(&r, -20, 20) * 0.017453292519943295;
        const double roll = nrm(&r) * 5.0 * 0.017453292519943295;
        const double dd = uni_ab(&r, 1.25, 1.75) * (close_up ? d0_close : d0);
        /* camera centre on a viewing cap in front of the wall (+z side) */
        const double C[3] = { at[0] + dd * sin(az) * cos(el), at[1] + dd * sin(el),
                              dd * cos(az) * cos(el) };
        double f[3] = { at[0] - C[0], at[1] - C[1], at[2] - C[2] };
        const double fn = sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
        for (int k = 0; k < 3; ++k)
            f[k] /= fn;
        /* camera axes in world: z = forward, y = down (world -y projected), x = y x z */
        double up[3] = { 0, -1, 0 };
        const double du = up[0] * f[0] + up[1] * f[1] + up[2] * f[2];
        double yv[3] = { up[0] - du * f[0], up[1] - du * f[1], up[2] - du * f[2] };
        const double yn = sqrt(yv[0] * yv[0] + yv[1] * yv[1] + yv[2] * yv[2]);
        for (int k = 0; k < 3; ++k)
            yv[k] /= yn;
        double xv[3] = { yv[1] * f[2] - yv[2] * f[1], yv[2] * f[0] - yv[0] * f[2],
                         yv[0] * f[1] - yv[1] * f[0] };
        /* world->camera rotation has the camera axes as rows */
        double Rwc[9] = { xv[0], xv[1], xv[2], yv[0], yv[1], yv[2], f[0], f[1], f[2] };
        double q0[4], qr[4], zax[3] = { 0, 0, 1 };
        R_to_q(Rwc, q0);
        q_axis_angle(zax, roll, qr);
        double* g = cam_gt + 7 * c;
        q_mul(qr, q0, g);
        double R[9];
        q_to_R(g, R);
        for (int k = 0; k < 3; ++k)
            g[4 + k] = -(R[3 * k] * C[0] + R[3 * k + 1] * C[1] + R[3 * k + 2] * C[2]);
    }
    /* visibility mask */
    unsigned char* vis = (unsigned char*)malloc((size_t)nc * nt);
    memset(vis, 1, (size_t)nc * nt);
    if (close_up) {
        /* the n_see[c] tags nearest to the point the camera looks at; then every tag in >= 2 images
         * (TagReconstructor.cpp:189-194), added to the cameras standing nearest to it */
        memset(vis, 0, (size_t)nc * nt);
        double* d2 = (double*)malloc(sizeof(double) * (size_t)(nt > nc ? nt : nc));
        for (int c = 0; c < nc; ++c) {
            for (int t = 0; t < nt; ++t) {
                const double dx = tag_gt[7 * t + 4] - look[2 * c], dy = tag_gt[7 * t + 5] - look[2 * c + 1];
                d2[t] = dx * dx + dy * dy;
            }
            for (int k = 0; k < n_see[c]; ++k) {
                int best = -1;
                for (int t = 0; t < nt; ++t)
                    if (!vis[(size_t)c * nt + t] && (best < 0 || d2[t] < d2[best]))
                        best = t;
                vis[(size_t)c * nt + best] = 1;
            }
        }
        for (int t = 0; t < nt; ++t) {
            int n = 0;
            for (int c = 0; c < nc; ++c) {
                n += vis[(size_t)c * nt + t];
                const double dx = tag_gt[7 * t + 4] - look[2 * c], dy = tag_gt[7 * t + 5] - look[2 * c + 1];
                d2[c] = dx * dx + dy * dy;
            }
            while (n < 2 && n < nc) {
                int best = -1;
                for (int c = 0; c < nc; ++c)
                    if (!vis[(size_t)c * nt + t] && (best < 0 || d2[c] < d2[best]))
                        best = c;
                vis[(size_t)best * nt + t] = 1;
                ++n;
            }
        }
        free(d2);
    } else if (cfg->visibility < 1.0) {
        for (int c = 0; c < nc; ++c)
            for (int t = 0; t < nt; ++t)
                vis[(size_t)c * nt + t] = uni(&r) < cfg->visibility;
        for (int t = 0; t < nt; ++t) { /* every tag in >= 2 images (TagReconstructor.cpp:189-194) */
            int n = 0;
            for (int c = 0; c < nc; ++c)
                n += vis[(size_t)c * nt + t];
            for (int k = 0; n < 2 && k < nc; ++k) {
                const int c = (int)(((long)t * 7 + k) % nc);
                if (!vis[(size_t)c * nt + t]) {
                    vis[(size_t)c * nt + t] = 1;
                    ++n;
                }
            }
        }
        for (int c = 0; c < nc; ++c) { /* every image sees >= 1 tag (TagReconstructor.cpp:689-690) */
            int n = 0;
            for (int t = 0; t < nt; ++t)
                n += vis[(size_t)c * nt + t];
            if (!n)
                vis[(size_t)c * nt + (c % nt)] = 1;
        }
    }
    long need = 0;
    for (size_t i = 0; i < (size_t)nc * nt; ++i)
        need += vis[i];
    free(look);
    free(n_see);
    if (need > max_obs) {
        free(vis);
        return (int)-need;
    }
    static const double sx[4] = { -1, 1, 1, -1 }, sy[4] = { -1, -1, 1, 1 };
    int n = 0;
    for (int c = 0; c < nc; ++c)
        for (int t = 0; t < nt; ++t) {
            if (!vis[(size_t)c * nt + t])
                continue;
            obs_cam[n] = c;
            obs_tag[n] = t;
            for (int k = 0; k < 4; ++k) {
                const double cl[3] = { sx[k] * side / 2, sy[k] * side / 2, 0 };
                project(intr, dist, cam_gt + 7 * c, tag_gt + 7 * t, cl, obs_px + 8 * (size_t)n + 2 * k);
            }
            for (int k = 0; k < 8; ++k)
                obs_px[8 * (size_t)n + k] += nrm(&r) * cfg->noise_px;
            if (cfg->outlier_frac > 0)
                for (int k = 0; k < 4; ++k) {
                    const double u = uni(&r);
                    const double ox = uni_ab(&r, -cfg->outlier_px, cfg->outlier_px);
                    const double oy = uni_ab(&r, -cfg->outlier_px, cfg->outlier_px);
                    if (u < cfg->outlier_frac) {
                        obs_px[8 * (size_t)n + 2 * k] += ox;
                        obs_px[8 * (size_t)n + 2 * k + 1] += oy;
                    }
                }
            ++n;
        }
    free(vis);
    for (int c = 0; c < nc; ++c)
        perturb(&r, cam_gt + 7 * c, cfg->cam_rot_deg, cfg->cam_trans_m, cam_init + 7 * c);
    memcpy(tag_init, tag_gt, 7 * sizeof(double)); /* origin tag exact (TagReconstructor.cpp:134-141) */
    for (int t = 1; t < nt; ++t)
        perturb(&r, tag_gt + 7 * t, cfg->tag_rot_deg, cfg->tag_trans_m, tag_init + 7 * t);
    return n;
}
