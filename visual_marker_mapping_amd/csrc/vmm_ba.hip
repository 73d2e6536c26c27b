// libvmm_ba.so -- C-ABI (include/vmm_ba.h) and host driver of the device-resident LM loop.
//
// Replaces the body of TagReconstructor::doBundleAdjustment
// (/root/reference/src/TagReconstructor.cpp:646-743): vmm_ba_create() takes the place of the
// ceres::Problem construction (:657-724), vmm_ba_solve() of ceres::Solve (:737-738).  No CPU
// fallback exists: without a HIP device every entry point fails with VMM_BA_ERR_HIP.
#include <dlfcn.h>
#include <math.h>
#include <rccl/rccl.h>   // types and prototypes only: librccl.so is resolved with dlopen when a communicator is asked for
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <string>
#include <functional>
#include <vector>

#include "engine.hpp"

namespace vmm {

static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                         \
            return VMM_BA_ERR_HIP;                                                                 \
        }                                                                                          \
    } while (0)

template <typename T>
static int dev_alloc(Engine& e, T** p, size_t count, bool zero = true)
{
    *p = nullptr;
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    HIP_TRY(hipMalloc((void**)p, bytes));
    e.allocs.push_back(*p);
    if (zero)
        HIP_TRY(hipMemsetAsync(*p, 0, bytes, e.stream));
    return VMM_BA_OK;
}

template <typename T>
static int upload(Engine& e, T* dst, const std::vector<T>& src)
{
    if (!src.empty())
        HIP_TRY(hipMemcpyAsync(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, e.stream));
    return VMM_BA_OK;
}

static int round_up(int64_t v, int m) { return (int)(((v + m - 1) / m) * m); }

// ---- optional libraries, resolved at run time (libvmm_ba.so itself links only the HIP runtime) ----
// roctx ranges around the solve and its iterations (rocprofv3 --marker-trace / the reference prints
// summary.FullReport(), src/TagReconstructor.cpp:741-742; the numeric phase report is in vmm_ba_summary).
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        void* h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h)
            h = dlopen("/opt/rocm/lib/libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (h) {
            push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
            pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        }
    }
};
static Roctx& roctx()
{
    static Roctx r;
    return r;
}
struct Range {
    explicit Range(const char* name)
    {
        if (roctx().push)
            roctx().push(name);
    }
    ~Range()
    {
        if (roctx().pop)
            roctx().pop();
    }
};

// RCCL (north_star: "RCCL all-reduce over xGMI of the reduced camera system")
struct Rccl {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
    Rccl()
    {
        void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h)
            h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h)
            h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h)
            return;
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(h, "ncclAllReduce"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        ok = GetUniqueId && CommInitRank && AllReduce && CommDestroy && GetErrorString;
    }
};
static Rccl& rccl()
{
    static Rccl r;
    return r;
}

// VMM_BA_DEBUG_SPIN_LIMIT=<polls> [VMM_BA_DEBUG_SPIN_KERNEL=df|chain|both] [VMM_BA_DEBUG_SPIN_ONCE=1] [VMM_BA_DEBUG_SPIN_WG=<b>]:
// shrink the bounded spins of k_chol_dataflow / k_backsolve_chain so that they give up (tests of the recovery path only).
// A limit of 1 makes every wait give up at its first poll, arrived data or not: every pass is then redone,
// deterministically.  With _WG only workgroup b of the launch gets the shrunk limit -- a give-up that the other workgroups
// learn of through the abort word only (or not at all, when their part of the factor does not depend on b's).
static void read_spin_debug_env(Engine& e)
{
    const char* sl = getenv("VMM_BA_DEBUG_SPIN_LIMIT");
    if (!sl)
        return;
    const unsigned lim = (unsigned)std::max(1L, atol(sl));
    const char* sk = getenv("VMM_BA_DEBUG_SPIN_KERNEL");
    const std::string which = sk ? sk : "both";
    if (which == "df" || which == "both")
        e.dbg_spin_df = lim;
    if (which == "chain" || which == "both")
        e.dbg_spin_chain = lim;
    const char* so = getenv("VMM_BA_DEBUG_SPIN_ONCE");
    e.dbg_spin_once = so && so[0] == '1';
    const char* sw = getenv("VMM_BA_DEBUG_SPIN_WG");
    e.dbg_spin_wg = sw ? atoi(sw) : -1;
}

// Sorts the observations by one pose family (stable counting sort) and cuts each pose's run into
// wave-sized tasks.
static int build_order(Engine& e, ObsOrder& o, int n_own, const int32_t* own_idx, const int32_t* other_idx,
                       const double* px, int64_t n, std::vector<int32_t>* caller_out = nullptr,
                       std::vector<int32_t>* start_out = nullptr, std::vector<int32_t>* other_out = nullptr)
{
    std::vector<int64_t> start((size_t)n_own + 1, 0);
    for (int64_t i = 0; i < n; ++i)
        start[own_idx[i] + 1]++;
    for (int p = 0; p < n_own; ++p)
        start[p + 1] += start[p];
    std::vector<int64_t> pos(start.begin(), start.end() - 1);
    const int64_t n_pad = std::max<int64_t>(64, round_up(n, 64));
    std::vector<int32_t> own((size_t)n), other((size_t)n), caller((size_t)n);
    std::vector<double> pxs((size_t)8 * n_pad, 0.0);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t d = pos[own_idx[i]]++;
        own[d] = own_idx[i];
        other[d] = other_idx[i];
        caller[d] = (int32_t)i;
        for (int k = 0; k < 8; ++k)
            pxs[(size_t)k * n_pad + d] = px[8 * i + k];
    }
    std::vector<Task> tasks;
    std::vector<int32_t> pose_task((size_t)n_own + 1, 0);
    for (int p = 0; p < n_own; ++p) {
        pose_task[p] = (int32_t)tasks.size();
        for (int64_t b = start[p]; b < start[p + 1]; b += kWave) {
            Task t;
            t.pose = p;
            t.begin = (int32_t)b;
            t.end = (int32_t)std::min<int64_t>(b + kWave, start[p + 1]);
            tasks.push_back(t);
        }
    }
    pose_task[n_own] = (int32_t)tasks.size();
    o.n = n;
    o.n_pad = n_pad;
    o.n_tasks = (int32_t)tasks.size();
    int rc;
    if ((rc = dev_alloc(e, &o.own, (size_t)n))) return rc;
    if ((rc = dev_alloc(e, &o.other, (size_t)n))) return rc;
    if ((rc = dev_alloc(e, &o.caller, (size_t)n))) return rc;
    if ((rc = dev_alloc(e, &o.px, (size_t)8 * n_pad))) return rc;
    if ((rc = dev_alloc(e, &o.tasks, tasks.size()))) return rc;
    if ((rc = dev_alloc(e, &o.pose_task, pose_task.size()))) return rc;
    if ((rc = dev_alloc(e, &o.part, tasks.size() * kPart))) return rc;
    std::vector<int32_t> start32(start.begin(), start.end());
    if ((rc = dev_alloc(e, &o.start, start32.size()))) return rc;
    if ((rc = upload(e, o.start, start32))) return rc;
    if ((rc = upload(e, o.own, own))) return rc;
    if ((rc = upload(e, o.other, other))) return rc;
    if ((rc = upload(e, o.caller, caller))) return rc;
    if ((rc = upload(e, o.px, pxs))) return rc;
    if ((rc = upload(e, o.tasks, tasks))) return rc;
    if ((rc = upload(e, o.pose_task, pose_task))) return rc;
    HIP_TRY(hipStreamSynchronize(e.stream));  // host vectors go out of scope
    if (caller_out)
        caller_out->swap(caller);
    if (start_out)
        start_out->swap(start32);
    if (other_out)
        other_out->swap(other);
    return VMM_BA_OK;
}

// Work plan of the rank-k update: lower 128x128 tiles with row blocks 0..n_row_blk-1 and column blocks
// 0..n_col_blk-1 (bj <= bi), K stages of 16 rows; the unit of work is one K stage of one tile.
//   * At most slots / 8 tiles (500 x 200: 55 tiles, 512 slots): one K slice per XCD -- workgroup b takes the
//     (b % 8)-th eighth of K of tile b / 8 (see below).
//   * Otherwise fewer tiles than workgroup slots: "stream-K" -- all units, tile-major, are cut into equal contiguous
//     ranges, one per workgroup.
//   * More tiles than slots (2000 x 1000: 1128 tiles): whole rounds of one-tile-per-workgroup first, XCD-aware:
//     workgroup b runs on XCD b % 8, so the 64 workgroups an XCD holds at a time get 64 CONSECUTIVE tiles of
//     the row-major tile list -- one or two block rows -- and sweep K in step: the A panel of a block row and
//     the B panels of its columns are fetched into that XCD's L2 once per K stage and shared (with the plain
//     stream-K order every workgroup streams its own two panels from HBM: 16 flop/B, measured HBM-bound at
//     63 TFLOP/s).  The tiles left over after the last full round are split stream-K over one more round.
// Every workgroup gets its unit range and first segment id by blockIdx; segments (one partial tile each) are
// numbered in unit order, so a tile's partials are consecutive and summed in that order.
static int make_syrk_plan(Engine& e, SyrkPlan& p, int n_row_blk, int n_col_blk, int k_pad)
{
    std::vector<int32_t> bi, bj;
    for (int r = 0; r < n_row_blk; ++r)
        for (int c = 0; c <= std::min(r, n_col_blk - 1); ++c) {
            bi.push_back(r);
            bj.push_back(c);
        }
    p.n_tiles = (int)bi.size();
    p.n_kt = k_pad / kKT;
    if (const char* v = getenv("VMM_BA_DEBUG_SYRK_KT"))   // timing experiments only: a product over the first rows of Z
        p.n_kt = std::max(2, std::min(atoi(v), p.n_kt));
    // two workgroups (72 KB of LDS each) per CU
    int hw = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, e.device) == hipSuccess)
        hw = prop.multiProcessorCount;
    if (hw <= 0)
        hw = 256;
    int per_cu = 2;
    if (const char* v = getenv("VMM_BA_SYRK_WG_PER_CU"))
        per_cu = std::max(1, atoi(v));
    const int64_t slots = per_cu * (int64_t)hw;
    const bool xcd_rounds = !(getenv("VMM_BA_SYRK_NO_XCD") && getenv("VMM_BA_SYRK_NO_XCD")[0] == '1');
    const int n_xcd = 8;
    std::vector<int64_t> wg_u0, wg_u1;
    std::vector<int32_t> wg_seg0, tile_seg0((size_t)p.n_tiles + 1, 0);
    int seg = 0;
    if (xcd_rounds && (int64_t)p.n_tiles * n_xcd <= slots && p.n_kt >= n_xcd) {
        // Few tiles (500 x 200: 55): one K slice per XCD.  Workgroup b runs on XCD b % 8 (round-robin dispatch)
        // and owns the (b % 8)-th eighth of K of tile b / 8, so the 55 workgroups of an XCD sweep the SAME rows of
        // Z in step: every row is fetched into that XCD's L2 once and shared (Z crosses the fabric once per launch
        // instead of once per workgroup), and every tile leaves exactly eight partials.  Off-diagonal tiles come
        // first in the tile list: the second workgroup a CU receives is then one of the cheaper diagonal tiles.
        std::vector<int> order;
        for (int t = 0; t < p.n_tiles; ++t)
            if (bi[t] != bj[t])
                order.push_back(t);
        for (int t = 0; t < p.n_tiles; ++t)
            if (bi[t] == bj[t])
                order.push_back(t);
        std::vector<int32_t> bi2(bi.size()), bj2(bj.size());
        for (int t = 0; t < p.n_tiles; ++t) {
            bi2[t] = bi[order[t]];
            bj2[t] = bj[order[t]];
        }
        bi.swap(bi2);
        bj.swap(bj2);
        // Round 4: one 8-wave workgroup per CU (k_syrk_wide), one K slice of ONE tile each; a diagonal tile costs 9/16 of
        // an off-diagonal one there and gets as many fewer workgroups.  500 x 200: 45 x 5 + 10 x 3 = 255 workgroups on
        // 256 CUs, 255 partial tiles instead of 495.  VMM_BA_SYRK_WIDE=0: the two-workgroups-per-CU kernel below.
        const bool want_wide = !(getenv("VMM_BA_SYRK_WIDE") && getenv("VMM_BA_SYRK_WIDE")[0] == '0');
        if (want_wide && !getenv("VMM_BA_SYRK_SLICES") && !getenv("VMM_BA_SYRK_WG_PER_CU")) {
            int n_diag = 0;
            for (int t = 0; t < p.n_tiles; ++t)
                n_diag += bi[t] == bj[t];
            const int n_off = p.n_tiles - n_diag;
            const double kDiagCost = 9.0 / 16.0;
            const int max_w = std::max(1, p.n_kt / 2);   // at least one 32-row stage per workgroup
            int w_off = (int)std::floor(hw / (n_off + kDiagCost * n_diag));
            w_off = std::max(1, std::min(w_off, max_w));
            int w_diag = std::max(1, std::min((int)std::lround(kDiagCost * w_off), max_w));
            while (w_off > 1 && (int64_t)w_off * n_off + (int64_t)w_diag * n_diag > hw) {
                --w_off;
                w_diag = std::max(1, std::min((int)std::lround(kDiagCost * w_off), max_w));
            }
            std::vector<int> w_of((size_t)p.n_tiles);
            int n_items = 0;
            for (int t = 0; t < p.n_tiles; ++t) {
                w_of[t] = bi[t] == bj[t] ? w_diag : w_off;
                tile_seg0[t] = n_items;
                n_items += w_of[t];
            }
            tile_seg0[p.n_tiles] = n_items;
            // items slice-major (all tiles' first slices, then the second ones, ...): XCD x takes the x-th run of them, so
            // the workgroups an XCD holds sweep the same rows of Z
            std::vector<std::pair<int, int>> items;
            for (int sl = 0; sl < std::max(w_off, w_diag); ++sl)
                for (int t = 0; t < p.n_tiles; ++t)
                    if (sl < w_of[t])
                        items.emplace_back(t, sl);
            const int per_x = (n_items + n_xcd - 1) / n_xcd;
            p.n_wg = per_x * n_xcd;
            p.wide = true;
            wg_u0.assign((size_t)p.n_wg, 0);
            wg_u1.assign((size_t)p.n_wg, 0);
            wg_seg0.assign((size_t)p.n_wg, 0);
            for (int b = 0; b < p.n_wg; ++b) {
                const int x = b % n_xcd, j = b / n_xcd;
                const int it = x * per_x + j;
                if (j >= per_x || it >= n_items)
                    continue;
                const int t = items[(size_t)it].first, sl = items[(size_t)it].second;
                wg_u0[b] = (int64_t)t * p.n_kt + (int64_t)p.n_kt * sl / w_of[t];
                wg_u1[b] = (int64_t)t * p.n_kt + (int64_t)p.n_kt * (sl + 1) / w_of[t];
                wg_seg0[b] = tile_seg0[t] + sl;
            }
            seg = n_items;
        } else {
        // K slices per tile: as many as fit the workgroup slots (two per CU), so that every SIMD carries about the
        // same number of MFMAs (8 slices on 440 of 512 slots left 184 CUs with two workgroups and 72 with one)
        int n_sl = (int)std::min<int64_t>(slots / p.n_tiles, p.n_kt);
        if (const char* v = getenv("VMM_BA_SYRK_SLICES"))
            n_sl = std::max(1, std::min(atoi(v), p.n_kt));
        const int n_items = p.n_tiles * n_sl;
        const int per_x = (n_items + n_xcd - 1) / n_xcd;
        p.n_wg = per_x * n_xcd;
        wg_u0.assign((size_t)p.n_wg, 0);
        wg_u1.assign((size_t)p.n_wg, 0);
        wg_seg0.assign((size_t)p.n_wg, 0);
        for (int b = 0; b < p.n_wg; ++b) {
            // items in slice-major order; XCD x (blockIdx % 8) takes the x-th run of per_x items: one or two slices of K
            const int x = b % n_xcd, j = b / n_xcd;
            const int it = x * per_x + j;
            if (j >= per_x || it >= n_items)
                continue;
            const int sl = it / p.n_tiles, t = it % p.n_tiles;
            wg_u0[b] = (int64_t)t * p.n_kt + (int64_t)p.n_kt * sl / n_sl;
            wg_u1[b] = (int64_t)t * p.n_kt + (int64_t)p.n_kt * (sl + 1) / n_sl;
            wg_seg0[b] = n_sl * t + sl;
        }
        for (int t = 0; t <= p.n_tiles; ++t)
            tile_seg0[t] = n_sl * t;
        seg = n_sl * p.n_tiles;
        }
    } else {
    const int64_t full_rounds = (xcd_rounds && slots % n_xcd == 0) ? p.n_tiles / slots : 0;
    const int64_t tiles_a = full_rounds * slots;                      // one tile per workgroup
    const int64_t units_b = (int64_t)(p.n_tiles - tiles_a) * p.n_kt;  // the rest: stream-K
    int64_t n_wg_b = std::min<int64_t>(units_b, slots);
    if (full_rounds == 0 && units_b > 64 * slots)
        n_wg_b = 4 * slots;   // xcd_rounds switched off: several waves of stream-K workgroups
    const int64_t upw_b = n_wg_b > 0 ? (units_b + n_wg_b - 1) / n_wg_b : 0;
    n_wg_b = upw_b > 0 ? (units_b + upw_b - 1) / upw_b : 0;
    p.n_wg = (int)(tiles_a + n_wg_b);
    // logical workgroup l (unit order) -> [u0, u1); segments numbered in unit order
    std::vector<int64_t> lu0((size_t)p.n_wg + 1, 0);
    for (int64_t l = 0; l < tiles_a; ++l)
        lu0[(size_t)l] = l * p.n_kt;
    for (int64_t l = 0; l <= n_wg_b; ++l)
        lu0[(size_t)(tiles_a + l)] = std::min<int64_t>(tiles_a * p.n_kt + l * upw_b, (int64_t)p.n_tiles * p.n_kt);
    std::vector<int32_t> lseg0((size_t)p.n_wg + 1, 0);
    for (int l = 0; l < p.n_wg; ++l) {
        lseg0[l] = seg;
        int64_t u = lu0[l];
        const int64_t u_end = lu0[l + 1];
        while (u < u_end) {
            const int t = (int)(u / p.n_kt);
            const int kt0 = (int)(u % p.n_kt);
            const int64_t take = std::min<int64_t>(p.n_kt - kt0, u_end - u);
            if (kt0 == 0)
                tile_seg0[t] = seg;
            u += take;
            ++seg;
        }
    }
    tile_seg0[p.n_tiles] = seg;
    // blockIdx -> logical workgroup: inside a full round, XCD x (blockIdx % 8) takes the x-th run of slots/8 tiles
    wg_u0.resize((size_t)p.n_wg);
    wg_u1.resize((size_t)p.n_wg);
    wg_seg0.resize((size_t)p.n_wg);
    for (int b = 0; b < p.n_wg; ++b) {
        int64_t l = b;
        if (b < tiles_a) {
            const int64_t r = b / slots, o = b % slots;
            l = r * slots + (o % n_xcd) * (slots / n_xcd) + o / n_xcd;
        }
        wg_u0[b] = lu0[(size_t)l];
        wg_u1[b] = lu0[(size_t)l + 1];
        wg_seg0[b] = lseg0[(size_t)l];
    }
    }
    p.n_segments = seg;
    int rc;
    if ((rc = dev_alloc(e, &p.tile_bi, bi.size()))) return rc;
    if ((rc = dev_alloc(e, &p.tile_bj, bj.size()))) return rc;
    if ((rc = dev_alloc(e, &p.wg_u0, wg_u0.size()))) return rc;
    if ((rc = dev_alloc(e, &p.wg_u1, wg_u1.size()))) return rc;
    if ((rc = dev_alloc(e, &p.wg_seg0, wg_seg0.size()))) return rc;
    if ((rc = dev_alloc(e, &p.tile_seg0, tile_seg0.size()))) return rc;
    if ((rc = dev_alloc(e, &p.partials, (size_t)std::max(seg, 1) * kST * kST, false))) return rc;
    if ((rc = upload(e, p.tile_bi, bi))) return rc;
    if ((rc = upload(e, p.tile_bj, bj))) return rc;
    if ((rc = upload(e, p.wg_u0, wg_u0))) return rc;
    if ((rc = upload(e, p.wg_u1, wg_u1))) return rc;
    if ((rc = upload(e, p.wg_seg0, wg_seg0))) return rc;
    if ((rc = upload(e, p.tile_seg0, tile_seg0))) return rc;
    HIP_TRY(hipStreamSynchronize(e.stream));
    return VMM_BA_OK;
}

// the two alternating transposed-panel buffers of the look-ahead Cholesky; P must be allocated
static int setup_lookahead(Engine& e, int n_blk_max, int ld)
{
    (void)n_blk_max;
    for (int i = 0; i < 4; ++i)
        e.P4[i] = e.P + (size_t)i * kNB * ld;
    return VMM_BA_OK;
}

// ---- point landmarks: tag pose <-> its four world corners ----
// computeMarkerCorners3D (include/visual_marker_mapping/TagReconstructor.h:33-52, called at
// src/TagReconstructor.cpp:483): R = Eigen::Quaterniond::toRotationMatrix() (no normalisation), corner = R local + t,
// corners LL, LR, UR, UL.
static void tag_to_points(const double* qt, const double* wh, double* pts)
{
    const double w = qt[0], x = qt[1], y = qt[2], z = qt[3];
    const double R[9] = { 1.0 - 2.0 * (y * y + z * z), 2.0 * (x * y - w * z), 2.0 * (x * z + w * y),
                          2.0 * (x * y + w * z), 1.0 - 2.0 * (x * x + z * z), 2.0 * (y * z - w * x),
                          2.0 * (x * z - w * y), 2.0 * (y * z + w * x), 1.0 - 2.0 * (x * x + y * y) };
    static const double sx[4] = { -1.0, 1.0, 1.0, -1.0 }, sy[4] = { -1.0, -1.0, 1.0, 1.0 };
    for (int k = 0; k < 4; ++k) {
        const double lx = sx[k] * wh[0] / 2.0, ly = sy[k] * wh[1] / 2.0;
        for (int a = 0; a < 3; ++a)
            pts[3 * k + a] = (R[3 * a] * lx + R[3 * a + 1] * ly) + qt[4 + a];
    }
}

// The tag pose of four optimised corners, src/TagReconstructor.cpp:608-639: x along corner 0 -> 1, y along corner
// 0 -> 3, z = x cross y, t = the mean of the corners.  The dead code stores R.col(0) = y, R.col(1) = -x (its comments
// name the corners ul, ur, or, ol: an older corner order); with today's order LL, LR, UR, UL
// (TagReconstructor.h:47-50) the tag frame is (x, y, z), and y is re-orthogonalised (z cross x) so that the matrix is
// a rotation before it becomes the quaternion the current ReconstructedTag stores.
static void points_to_tag(const double* pts, double* qt)
{
    double ex[3], ey[3], ez[3];
    for (int a = 0; a < 3; ++a) {
        ex[a] = pts[3 + a] - pts[a];
        ey[a] = pts[9 + a] - pts[a];
    }
    auto normalise = [](double* v) {
        const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        if (n > 0.0)
            for (int a = 0; a < 3; ++a)
                v[a] /= n;
    };
    normalise(ex);
    normalise(ey);
    ez[0] = ex[1] * ey[2] - ex[2] * ey[1];
    ez[1] = ex[2] * ey[0] - ex[0] * ey[2];
    ez[2] = ex[0] * ey[1] - ex[1] * ey[0];
    normalise(ez);
    ey[0] = ez[1] * ex[2] - ez[2] * ex[1];
    ey[1] = ez[2] * ex[0] - ez[0] * ex[2];
    ey[2] = ez[0] * ex[1] - ez[1] * ex[0];
    const double m00 = ex[0], m01 = ey[0], m02 = ez[0], m10 = ex[1], m11 = ey[1], m12 = ez[1], m20 = ex[2], m21 = ey[2],
                 m22 = ez[2];
    // Eigen::Quaterniond(Matrix3d)
    double q[4];
    const double tr = m00 + m11 + m22;
    if (tr > 0.0) {
        double t = sqrt(tr + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (m21 - m12) * t;
        q[2] = (m02 - m20) * t;
        q[3] = (m10 - m01) * t;
    } else {
        const double m[3][3] = { { m00, m01, m02 }, { m10, m11, m12 }, { m20, m21, m22 } };
        int i = 0;
        if (m[1][1] > m[0][0])
            i = 1;
        if (m[2][2] > m[i][i])
            i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double t = sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
        q[1 + i] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[k][j] - m[j][k]) * t;
        q[1 + j] = (m[j][i] + m[i][j]) * t;
        q[1 + k] = (m[k][i] + m[i][k]) * t;
    }
    for (int a = 0; a < 4; ++a)
        qt[a] = q[a];
    for (int a = 0; a < 3; ++a)
        qt[4 + a] = (pts[a] + pts[3 + a] + pts[6 + a] + pts[9 + a]) / 4.0;
}

// caller's tag poses -> the device's point pairs [2 * n_tags][7] (slot 6 unused)
static std::vector<double> pairs_from_tags(const double* tag_qt, const double* tag_wh, int n_tags)
{
    std::vector<double> pairs((size_t)14 * n_tags, 0.0);
    for (int t = 0; t < n_tags; ++t) {
        double pts[12];
        tag_to_points(tag_qt + 7 * (size_t)t, tag_wh + 2 * (size_t)t, pts);
        for (int k = 0; k < 6; ++k) {
            pairs[(size_t)14 * t + k] = pts[k];
            pairs[(size_t)14 * t + 7 + k] = pts[6 + k];
        }
    }
    return pairs;
}

// Dense Z and the plan of its rank-k update (the default at high visibility; a handle on the block-sparse path
// builds them the first time the covariance report asks for Z as a matrix).
static int ensure_dense_schur(Engine& e)
{
    if (e.Z)
        return VMM_BA_OK;
    int rc;
    if ((rc = dev_alloc(e, &e.Z, (size_t)e.k_pad * e.ldz))) return rc;
    // 128-row blocks covering rows 0..n_pad (the last one holds the rhs row) and columns 0..n_pad-1
    return make_syrk_plan(e, e.syrk, (e.n_pad + kST) / kST, (e.n_pad + kST - 1) / kST, e.k_pad);
}

static int do_allreduce(Engine& e, double* buf, size_t count)
{
    if (!e.multi)
        return VMM_BA_OK;
    if (e.rccl_comm) {
        // in place, on the engine's stream: ordered behind the kernels that filled the buffer, capturable
        const ncclResult_t r = rccl().AllReduce(buf, buf, count, ncclDouble, ncclSum,
                                                reinterpret_cast<ncclComm_t>(e.rccl_comm), e.stream);
        if (r != ncclSuccess) {
            set_error(std::string("ncclAllReduce: ") + rccl().GetErrorString(r));
            return VMM_BA_ERR_COLLECTIVE;
        }
        return VMM_BA_OK;
    }
    if (!e.allreduce) {
        set_error("world_size > 1 but no all-reduce callback was set (vmm_ba_set_allreduce)");
        return VMM_BA_ERR_STATE;
    }
    if (e.allreduce(e.allreduce_user, buf, count, (void*)e.stream) != 0) {
        set_error("all-reduce callback failed");
        return VMM_BA_ERR_COLLECTIVE;
    }
    return VMM_BA_OK;
}

// World > 1 with the library's own communicator: the ranks agree (minimum) on a yes/no each of them holds, with an
// eager ncclAllReduce on the engine's stream.  Used where a per-rank decision would make the ranks enqueue different
// sequences of collectives (graph or no graph: ADVICE.md round 2).
static int rccl_agree(Engine& e, bool mine, bool* all)
{
    unsigned* word = e.flags + 260;
    const unsigned v = mine ? 1u : 0u;
    unsigned out = 0;
    HIP_TRY(hipMemcpyAsync(word, &v, sizeof(v), hipMemcpyHostToDevice, e.stream));
    const ncclResult_t r = rccl().AllReduce(word, word, 1, ncclUint32, ncclMin, reinterpret_cast<ncclComm_t>(e.rccl_comm),
                                            e.stream);
    if (r != ncclSuccess) {
        set_error(std::string("ncclAllReduce (agreement): ") + rccl().GetErrorString(r));
        return VMM_BA_ERR_COLLECTIVE;
    }
    HIP_TRY(hipMemcpyAsync(&out, word, sizeof(out), hipMemcpyDeviceToHost, e.stream));
    HIP_TRY(hipStreamSynchronize(e.stream));
    *all = out != 0;
    return VMM_BA_OK;
}

static void destroy_engine(Engine* e)
{
    if (!e)
        return;
    (void)hipSetDevice(e->device);
    if (e->stream)
        (void)hipStreamSynchronize(e->stream);
    if (e->iter_graph)
        (void)hipGraphExecDestroy(e->iter_graph);
    for (auto& g : e->iter_graph_seg)
        if (g)
            (void)hipGraphExecDestroy(g);
    if (e->rccl_comm)
        (void)rccl().CommDestroy(reinterpret_cast<ncclComm_t>(e->rccl_comm));
    for (void* p : e->allocs)
        (void)hipFree(p);
    if (e->ctl_host)
        (void)hipHostFree(e->ctl_host);
    if (e->pose_stage)
        (void)hipHostFree(e->pose_stage);
    if (e->pose_ev)
        (void)hipEventDestroy(e->pose_ev);
    if (e->stream)
        (void)hipStreamDestroy(e->stream);
    delete e;
}

// One LM iteration = four groups of kernels separated (world > 1) by the three sum-all-reduces.
//
// Every iteration evaluates residuals AND Jacobians at the CANDIDATE x + delta of the previous iteration (iteration
// zero: the candidate buffers start as a copy of x).  Its cost is the cost Ceres evaluates at the candidate; when the
// step is accepted its blocks simply become the blocks at the new x (staging copy -> working copy, the other W
// buffer), so an accepted step -- the common case -- costs one evaluation instead of a cost pass plus a Jacobian
// pass, and the accept/reject decision sits directly in front of the next iteration's set-up.  A rejected step
// wastes the Jacobian part of its evaluation (21 us of a 0.5 ms iteration at 500 x 200).
constexpr int kNumSeg = 4;
// redo: the host's recovery of a pass whose one-launch factorisation gave up waiting (LmCtl::sync_timeout): the
// reduced system is rebuilt from Z (the factorisation overwrote it) and factored on the path without
// inter-workgroup waits; the control kernel and the elimination of that pass have run and are not repeated.
static void enqueue_segment(Engine& e, const vmm_ba_options& o, int seg, bool redo = false)
{
    switch (seg) {
    case 0:
        if (!redo)
            launch_eval_passes(e, o.robustify, o.huber_a, true);
        break;
    case 1:
        if (!redo) {
            launch_control(e);
            launch_elim(e);
        }
        launch_syrk_reduced(e);
        launch_pack_lower(e, false);
        break;
    case 2:
        launch_pack_lower(e, true);   // world > 1: unpack + the kept family's diagonal blocks
        launch_cholesky_solve(e, e.S, e.n_pad, e.ldz, e.yf, e.ctl, redo);
        launch_backsub(e);
        break;
    default:
        launch_candidate(e);   // world > 1 only; one GPU forms the candidates in k_backsub
        break;
    }
}

// the all-reduce that follows segment `seg` when world > 1 (none after the last one)
static int allreduce_after(Engine& e, int seg)
{
    switch (seg) {
    case 0: return do_allreduce(e, e.small_stage, e.small_count + (size_t)e.n_e);   // blocks, gradient, per-pose costs at the candidate
    case 1: return do_allreduce(e, e.S_packed, (size_t)(e.n_pad + 1) * (e.n_pad + 2) / 2);
    case 2: return do_allreduce(e, e.step_comm, 7 * (size_t)e.n_e + 1);   // steps | per-pose cross terms | sync-time-out votes
    default: return VMM_BA_OK;
    }
}

static int enqueue_iteration(Engine& e, const vmm_ba_options& o)
{
    static const char* const names[kNumSeg] = { "vmm_ba evaluation at the candidate", "vmm_ba decide + eliminate + rank-k",
                                                "vmm_ba factor + solve + step", "vmm_ba candidate" };
    int rc;
    for (int seg = 0; seg < kNumSeg; ++seg) {
        if (seg == kNumSeg - 1 && !e.multi)
            break;   // one GPU: the candidates are formed by k_backsub
        Range r(names[seg]);
        enqueue_segment(e, o, seg);
        if (e.multi && (rc = allreduce_after(e, seg)))
            return rc;
    }
    HIP_TRY(hipGetLastError());
    return VMM_BA_OK;
}

// Records `body` (kernel launches, and ncclAllReduce calls when a communicator is attached) into a hipGraph.
// Nothing may fail silently here: a sticky error from before the capture, or a call that invalidates the capture,
// is reported with the place where it was noticed (`where` is set by the body after every group).
static int capture_graph(Engine& e, hipGraphExec_t* out, const std::function<int(const char**)>& body)
{
    const hipError_t pre = hipGetLastError();
    if (pre != hipSuccess) {
        set_error(std::string("error pending before hipStreamBeginCapture: ") + hipGetErrorString(pre));
        return VMM_BA_ERR_HIP;
    }
    hipGraph_t g = nullptr;
    HIP_TRY(hipStreamBeginCapture(e.stream, hipStreamCaptureModeThreadLocal));
    const char* where = "start";
    const int rc = body(&where);
    const hipError_t ee = hipStreamEndCapture(e.stream, &g);
    if (rc != VMM_BA_OK) {
        if (g)
            (void)hipGraphDestroy(g);
        return rc;
    }
    if (ee != hipSuccess || !g) {
        set_error(std::string("hipStreamEndCapture: ") + hipGetErrorString(ee) + " (capture invalidated at or before: " + where
                  + ")");
        (void)hipGetLastError();
        return VMM_BA_ERR_HIP;
    }
    const hipError_t ei = hipGraphInstantiate(out, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ei != hipSuccess) {
        *out = nullptr;
        set_error(std::string("hipGraphInstantiate: ") + hipGetErrorString(ei));
        return VMM_BA_ERR_HIP;
    }
    return VMM_BA_OK;
}

// after a group of launches under capture: is the capture still alive?
static bool capture_alive(Engine& e)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(e.stream, &st) != hipSuccess)
        return false;
    return st == hipStreamCaptureStatusActive;
}

static void drop_graphs(Engine& e)
{
    for (auto& g : e.iter_graph_seg)
        if (g) {
            (void)hipGraphExecDestroy(g);
            g = nullptr;
        }
    if (e.iter_graph) {
        (void)hipGraphExecDestroy(e.iter_graph);
        e.iter_graph = nullptr;
    }
}

static const char* const kSegName[kNumSeg] = { "evaluation at the candidate", "decide + control + eliminate + rank-k update",
                                               "factor + solve + back-substitution", "candidate" };

// The iteration is captured once into hipGraphs and replayed (eager enqueueing is host-bound).
//   one GPU:                      one graph;
//   world > 1, RCCL communicator: one graph, the three ncclAllReduce calls recorded between the kernel groups
//                                 (VMM_BA_RCCL_GRAPH=0: four graphs with the collectives enqueued between them);
//   world > 1, host callback:     the callbacks are host calls, so the four kernel groups around them are four graphs.
// No kernel is launched for the first time under capture: vmm_ba_create touches every kernel (preload_*_kernels).
// Round 1 ran the first iteration of a handle eagerly because of an intermittent "operation failed due to a previous
// error during capture" at 2000 x 1000; VMM_BA_EAGER_FIRST=1 brings that back.
static int run_iteration(Engine& e, const vmm_ba_options& o)
{
    e.last_passes = 1;
    if (!e.use_graph)
        return enqueue_iteration(e, o);
    if (e.eager_first && !e.launched_eagerly) {
        e.launched_eagerly = true;
        return enqueue_iteration(e, o);
    }
    const bool one_graph = !e.multi || (e.rccl_comm && e.rccl_graph);
    const bool have = one_graph ? e.iter_graph != nullptr : e.iter_graph_seg[0] != nullptr;
    if (!have || e.graph_robustify != o.robustify || e.graph_huber_a != o.huber_a) {
        drop_graphs(e);
        int rc;
        if (one_graph) {
            // graph_passes passes in one graph: nothing separates two passes inside a graph (~9 us between two graph
            // launches); the passes behind the terminating one find `done` set and return at once
            rc = capture_graph(e, &e.iter_graph, [&](const char** where) -> int {
                for (int pass = 0; pass < e.graph_passes; ++pass)
                    for (int seg = 0; seg < kNumSeg; ++seg) {
                        enqueue_segment(e, o, seg);
                        int r;
                        if (e.multi && (r = allreduce_after(e, seg)))
                            return r;
                        if (capture_alive(e))
                            *where = kSegName[seg];
                    }
                return VMM_BA_OK;
            });
            if (e.multi && e.rccl_comm) {
                // Graph or eagerly enqueued collectives is ONE decision for all ranks: a rank that fell back on its own
                // would enqueue one pass per launch while the others enqueue graph_passes, and the surplus
                // ncclAllReduce calls of one side would never complete.  So the ranks agree on "everybody captured".
                bool all_ok = false;
                const std::string why = rc ? g_err : std::string();
                // VMM_BA_DEBUG_CAPTURE_FAIL=<rank>: that rank votes "my capture failed" although it did not -- the path of a
                // rank whose RCCL refuses the capture, without such an RCCL (tests/test_gpu_distributed.py)
                bool mine = rc == VMM_BA_OK;
                if (const char* v = getenv("VMM_BA_DEBUG_CAPTURE_FAIL"))
                    if (v[0] && atoi(v) == e.rank)
                        mine = false;
                int arc;
                if ((arc = rccl_agree(e, mine, &all_ok)))
                    return arc;
                if (!all_ok) {
                    // a collective that cannot be recorded (here or on another rank): graphs around eagerly
                    // enqueued collectives, on every rank
                    drop_graphs(e);
                    e.rccl_graph = false;
                    if (getenv("VMM_BA_DEBUG"))
                        fprintf(stderr, "[vmm_ba debug] rank %d: RCCL not recorded into the iteration graph (%s)\n", e.rank,
                                rc ? why.c_str() : "another rank could not");
                    return run_iteration(e, o);
                }
            }
            if (rc)
                return rc;
        } else {
            for (int seg = 0; seg < kNumSeg; ++seg)
                if ((rc = capture_graph(e, &e.iter_graph_seg[seg], [&](const char** where) -> int {
                         enqueue_segment(e, o, seg);
                         if (capture_alive(e))
                             *where = kSegName[seg];
                         return VMM_BA_OK;
                     }))) {
                    drop_graphs(e);
                    return rc;
                }
        }
        e.graph_robustify = o.robustify;
        e.graph_huber_a = o.huber_a;
    }
    Range r("vmm_ba lm_iteration");
    if (one_graph) {
        HIP_TRY(hipGraphLaunch(e.iter_graph, e.stream));
        e.last_passes = e.graph_passes;
        return VMM_BA_OK;
    }
    int rc;
    for (int seg = 0; seg < kNumSeg; ++seg) {
        HIP_TRY(hipGraphLaunch(e.iter_graph_seg[seg], e.stream));
        if ((rc = allreduce_after(e, seg)))
            return rc;
    }
    return VMM_BA_OK;
}

static void init_ctl(Engine& e, LmCtl& c, const vmm_ba_options& o, int trace_capacity);

// The control block (already read back into e.ctl_host) says done == 2: a workgroup of k_chol_dataflow or
// k_backsolve_chain gave up waiting for another one in the pass that paused.  That is a scheduling event (several
// processes time-slicing one GPU, a profiler serialising workgroups), not a property of the matrix: the pass is
// completed here on the launch-per-block-column factorisation and the per-block back-substitution, which wait for
// nothing inside a launch, and the loop resumes -- the trust-region policy never learns of it.  Every kernel that
// was enqueued behind the give-up has returned at once (done != 0).  World > 1: all ranks pause in the same pass
// (the votes ride on the step all-reduce, k_candidate) and make the same collective calls here.
static int recover_sync_timeout(Engine& e, const vmm_ba_options& o)
{
    LmCtl& c = *e.ctl_host;
    c.num_sync_timeouts++;
    c.sync_kernels |= c.sync_timeout;
    c.sync_timeout = 0;
    c.lin_fail = 0;
    c.done = 0;
    if (e.dbg_spin_once)
        c.spin_limit_df = c.spin_limit_chain = 0;
    HIP_TRY(hipMemcpyAsync(e.ctl, e.ctl_host, sizeof(LmCtl), hipMemcpyHostToDevice, e.stream));
    Range r("vmm_ba sync time-out recovery");
    int rc;
    for (int seg = 1; seg < kNumSeg; ++seg) {
        if (seg == kNumSeg - 1 && !e.multi)
            break;
        enqueue_segment(e, o, seg, true);
        if (e.multi && (rc = allreduce_after(e, seg)))
            return rc;
    }
    HIP_TRY(hipGetLastError());
    return VMM_BA_OK;
}

// Control block and candidate buffers at the start of an LM loop: iteration zero evaluates "the candidate" = x.
static int begin_lm_loop(Engine& e, const vmm_ba_options& o, int trace_capacity)
{
    init_ctl(e, *e.ctl_host, o, trace_capacity);
    launch_begin_loop(e, *e.ctl_host);   // control block, staged poses -> state, state -> candidate: one launch
    HIP_TRY(hipGetLastError());
    if (e.dirty_cam || e.dirty_tag) {
        HIP_TRY(hipEventRecord(e.pose_ev, e.stream));   // the staging buffer is read by that launch
        e.pose_ev_pending = true;
        e.dirty_cam = e.dirty_tag = false;
    }
    return VMM_BA_OK;
}

// Poses staged by vmm_ba_set_state go to the device (every entry point that reads them there calls this first;
// vmm_ba_solve does it inside k_begin_loop).
static int flush_state(Engine& e)
{
    if (!e.dirty_cam && !e.dirty_tag)
        return VMM_BA_OK;
    if (e.dirty_cam)
        HIP_TRY(hipMemcpyAsync(e.cam_qt, e.pose_stage, sizeof(double) * 7 * e.n_cams, hipMemcpyHostToDevice, e.stream));
    if (e.dirty_tag)
        HIP_TRY(hipMemcpyAsync(e.tag_qt, e.pose_stage + (size_t)7 * e.n_cams, sizeof(double) * 7 * e.n_tags,
                               hipMemcpyHostToDevice, e.stream));
    HIP_TRY(hipEventRecord(e.pose_ev, e.stream));
    e.pose_ev_pending = true;
    e.dirty_cam = e.dirty_tag = false;
    return VMM_BA_OK;
}

static void init_ctl(Engine& e, LmCtl& c, const vmm_ba_options& o, int trace_capacity)
{
    memset(&c, 0, sizeof(c));
    c.spin_limit_df = e.dbg_spin_df;
    c.spin_limit_chain = e.dbg_spin_chain;
    c.spin_wg = e.dbg_spin_wg;
    c.max_num_iterations = o.max_num_iterations;
    c.robustify = o.robustify;
    c.jacobi_scaling = o.jacobi_scaling;
    c.max_invalid = o.max_num_consecutive_invalid_steps;
    c.huber_a = o.huber_a;
    c.function_tolerance = o.function_tolerance;
    c.gradient_tolerance = o.gradient_tolerance;
    c.parameter_tolerance = o.parameter_tolerance;
    c.max_radius = o.max_trust_region_radius;
    c.min_radius = o.min_trust_region_radius;
    c.min_relative_decrease = o.min_relative_decrease;
    c.min_lm_diagonal = o.min_lm_diagonal;
    c.max_lm_diagonal = o.max_lm_diagonal;
    c.radius = o.initial_trust_region_radius;
    c.decrease_factor = 2.0;
    c.first_eval = 1;
    c.termination = VMM_BA_NO_CONVERGENCE;
    c.trace_capacity = trace_capacity;
}

} // namespace vmm

using namespace vmm;

extern "C" {

const char* vmm_ba_last_error(void) { return g_err.c_str(); }
int vmm_ba_abi_version(void) { return VMM_BA_ABI_VERSION; }

void vmm_ba_default_options(vmm_ba_options* o)
{
    // src/TagReconstructor.cpp:725-735 + Ceres Solver::Options defaults
    memset(o, 0, sizeof(*o));
    o->max_num_iterations = 400;
    o->robustify = 1;
    o->huber_a = 1.0;
    o->function_tolerance = 1e-6;
    o->gradient_tolerance = 1e-10;
    o->parameter_tolerance = 1e-8;
    o->initial_trust_region_radius = 1e4;
    o->max_trust_region_radius = 1e16;
    o->min_trust_region_radius = 1e-32;
    o->min_relative_decrease = 1e-3;
    o->min_lm_diagonal = 1e-6;
    o->max_lm_diagonal = 1e32;
    o->max_num_consecutive_invalid_steps = 5;
    o->jacobi_scaling = 1;
    o->num_threads = 1;
    o->poll_interval = 1;
}

void vmm_ba_default_create_options(vmm_ba_create_options* o)
{
    memset(o, 0, sizeof(*o));
    o->device = 0;
    o->elimination = VMM_BA_ELIM_AUTO;
    o->rank = 0;
    o->world_size = 1;
}

// ---- block structure of the factor under a tree ordering ---------------------------------------------------------
// One mask per 64-row block row (kDfMaskWords words: up to 255 block columns), bit k = block (i, k) of the factor may be
// non-zero: the blocks the kept poses' 6x6 blocks touch (rows = first row of every kept pose, nbr = the co-observation
// graph), then symbolic fill (eliminating block column k couples every two block rows with an entry in it).  Row nb is the
// right-hand side row: all ones.
struct BlkMask {
    unsigned long long w[kDfMaskWords] = {};
    void set(int k) { w[k >> 6] |= 1ull << (k & 63); }
    bool test(int k) const { return (w[k >> 6] >> (k & 63)) & 1ull; }
};

static std::vector<BlkMask> symbolic_factor(int nb, const std::vector<int32_t>& rows,
                                            const std::vector<std::vector<int32_t>>& nbr)
{
    std::vector<BlkMask> nz((size_t)nb + 1);
    for (int i = 0; i < nb; ++i)
        nz[(size_t)i].set(i);
    const int n_f = (int)rows.size();
    auto touch = [&](int f1, int f2) {
        const int r1 = rows[(size_t)f1], r2 = rows[(size_t)f2];
        for (int bi = r1 / kNB; bi <= (r1 + 5) / kNB; ++bi)
            for (int bj = r2 / kNB; bj <= (r2 + 5) / kNB; ++bj)
                nz[(size_t)std::max(bi, bj)].set(std::min(bi, bj));
    };
    for (int fq = 0; fq < n_f; ++fq) {
        touch(fq, fq);
        for (const int32_t f2 : nbr[(size_t)fq])
            touch(fq, f2);
    }
    for (int k = 0; k < nb; ++k)
        for (int i = k + 1; i < nb; ++i)
            if (nz[(size_t)i].test(k))
                for (int j2 = k + 1; j2 <= i; ++j2)
                    if (nz[(size_t)j2].test(k))
                        nz[(size_t)i].set(j2);
    for (int k = 0; k < nb; ++k)
        nz[(size_t)nb].set(k);
    return nz;
}

// The model of the one-launch kernel the ordering decisions use (microseconds; measured in round 4: a block column's own
// eight rounds 7.1; a panel's eight slices are 8 of work for the workgroup that applies them, which it does while they are
// produced -- it is 2.7 behind when the panel ends): when block column j is done if it takes its panels in the order they
// finish.  Also fills the order and the longest chain of dependent columns.
static double model_tree_factorisation(int nb, const std::vector<BlkMask>& nz, std::vector<unsigned char>* ord, int* path_max)
{
    std::vector<double> t_done((size_t)nb, 0.0);
    std::vector<int> path((size_t)nb, 1);
    int pm = 0;
    static const bool old_model = [] {
        const char* v = getenv("VMM_BA_TREE_MODEL");
        return v && !strcmp(v, "r3");
    }();
    for (int j2 = 0; j2 < nb; ++j2) {
        std::vector<int> ks;
        for (int k = 0; k < j2; ++k)
            if (nz[(size_t)j2].test(k))
                ks.push_back(k);
        std::stable_sort(ks.begin(), ks.end(), [&](int x, int y) { return t_done[(size_t)x] < t_done[(size_t)y]; });
        double t = 0.0;
        for (size_t q = 0; q < ks.size(); ++q) {
            if (ord)
                (*ord)[(size_t)j2 * kDfMaxBlk + q] = (unsigned char)ks[q];
            t = old_model ? std::max(t, t_done[(size_t)ks[q]]) + 5.6 : std::max(t + 8.0, t_done[(size_t)ks[q]] + 2.7);
            path[(size_t)j2] = std::max(path[(size_t)j2], path[(size_t)ks[q]] + 1);
        }
        t_done[(size_t)j2] = t + (old_model ? 11.0 : 7.1);
        pm = std::max(pm, path[(size_t)j2]);
    }
    if (path_max)
        *path_max = pm;
    double t_end = 0.0;
    for (const double t : t_done)
        t_end = std::max(t_end, t);
    return t_end;
}

// ---- tree ordering of the kept family (block-sparse path) -------------------------------------------------------
// Nested dissection of the co-observation graph of the kept poses (two kept poses are neighbours when one eliminated
// pose sees both: exactly the non-zero blocks of the reduced system).  A part is cut at the breadth-first level (from a
// pseudo-peripheral vertex) that balances the two sides; the level is the separator and is ordered BEHIND both sides,
// recursively.  `nodes` comes out in elimination order (children before their separator); parts of one level do not
// touch each other, so their block columns of the factor do not depend on each other.
static void nd_dissect(const std::vector<std::vector<int32_t>>& nbr, std::vector<int32_t> verts, int leaf_max, int depth,
                       std::vector<int32_t>& stamp, int32_t& stamp_next, std::vector<std::vector<int32_t>>& nodes)
{
    if (verts.empty())
        return;
    if ((int)verts.size() <= leaf_max || depth <= 0) {
        nodes.push_back(std::move(verts));
        return;
    }
    // connected components of the induced subgraph
    const int32_t in_set = stamp_next++;
    for (const int32_t v : verts)
        stamp[(size_t)v] = in_set;
    std::vector<std::vector<int32_t>> comps;
    {
        const int32_t seen = stamp_next++;
        for (const int32_t v0 : verts) {
            if (stamp[(size_t)v0] != in_set)
                continue;
            comps.emplace_back();
            std::vector<int32_t>& c = comps.back();
            c.push_back(v0);
            stamp[(size_t)v0] = seen;
            for (size_t h = 0; h < c.size(); ++h)
                for (const int32_t w : nbr[(size_t)c[h]])
                    if (stamp[(size_t)w] == in_set) {
                        stamp[(size_t)w] = seen;
                        c.push_back(w);
                    }
        }
    }
    if (comps.size() > 1) {
        // independent already: two groups of about equal size, no separator
        std::sort(comps.begin(), comps.end(),
                  [](const std::vector<int32_t>& a, const std::vector<int32_t>& b) { return a.size() > b.size(); });
        std::vector<int32_t> A, B;
        for (auto& c : comps) {
            std::vector<int32_t>& dst = A.size() <= B.size() ? A : B;
            dst.insert(dst.end(), c.begin(), c.end());
        }
        nd_dissect(nbr, std::move(A), leaf_max, depth - 1, stamp, stamp_next, nodes);
        nd_dissect(nbr, std::move(B), leaf_max, depth - 1, stamp, stamp_next, nodes);
        return;
    }
    // level structure from a pseudo-peripheral vertex (two sweeps)
    std::vector<int32_t> order, level_of_pos;
    int32_t root = verts[0];
    for (int sweep = 0; sweep < 2; ++sweep) {
        const int32_t mark = stamp_next++, todo = stamp_next++;
        for (const int32_t v : verts)
            stamp[(size_t)v] = todo;
        order.assign(1, root);
        level_of_pos.assign(1, 0);
        stamp[(size_t)root] = mark;
        for (size_t h = 0; h < order.size(); ++h)
            for (const int32_t w : nbr[(size_t)order[h]])
                if (stamp[(size_t)w] == todo) {
                    stamp[(size_t)w] = mark;
                    order.push_back(w);
                    level_of_pos.push_back(level_of_pos[h] + 1);
                }
        root = order.back();
    }
    const int n_levels = level_of_pos.back() + 1;
    if (n_levels < 3) {   // (nearly) complete graph: nothing to cut
        nodes.push_back(std::move(verts));
        return;
    }
    std::vector<int32_t> cnt((size_t)n_levels, 0);
    for (const int32_t l : level_of_pos)
        cnt[(size_t)l]++;
    int best = 1;
    long long best_cost = -1;
    for (int l = 1, below = cnt[0]; l + 1 < n_levels; below += cnt[(size_t)l], ++l) {
        const int above = (int)order.size() - below - cnt[(size_t)l];
        const long long cost = (long long)std::abs(below - above) * 4 + cnt[(size_t)l];
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = l;
        }
    }
    std::vector<int32_t> A, B, S;
    for (size_t h = 0; h < order.size(); ++h)
        (level_of_pos[h] < best ? A : level_of_pos[h] > best ? B : S).push_back(order[h]);
    nd_dissect(nbr, std::move(A), leaf_max, depth - 1, stamp, stamp_next, nodes);
    nd_dissect(nbr, std::move(B), leaf_max, depth - 1, stamp, stamp_next, nodes);
    nodes.push_back(std::move(S));
}

int vmm_ba_create(const vmm_ba_problem* p, const vmm_ba_create_options* copt, vmm_ba_handle* out)
{
    if (!p || !out) {
        set_error("null argument");
        return VMM_BA_ERR_ARGUMENT;
    }
    *out = nullptr;
    vmm_ba_create_options co;
    if (copt)
        co = *copt;
    else
        vmm_ba_default_create_options(&co);
    if (p->n_cams <= 0 || p->n_tags <= 0 || p->n_obs < 0 || !p->cam_qt || !p->tag_qt || !p->tag_wh
        || (p->n_obs > 0 && (!p->obs_cam || !p->obs_tag || !p->obs_px))) {
        set_error("problem needs >= 1 camera, >= 1 tag and non-null arrays");
        return VMM_BA_ERR_ARGUMENT;
    }
    if (p->n_obs >= (int64_t)1 << 31) {
        set_error("n_obs must fit in 31 bits");
        return VMM_BA_ERR_ARGUMENT;
    }
    for (int64_t i = 0; i < p->n_obs; ++i)
        if (p->obs_cam[i] < 0 || p->obs_cam[i] >= p->n_cams || p->obs_tag[i] < 0 || p->obs_tag[i] >= p->n_tags) {
            set_error("observation " + std::to_string(i) + " references a camera or tag index out of range");
            return VMM_BA_ERR_ARGUMENT;
        }
    if (p->fixed_tag >= p->n_tags) {
        set_error("fixed_tag out of range");
        return VMM_BA_ERR_ARGUMENT;
    }
    if (co.landmarks != VMM_BA_LANDMARK_TAG_POSES && co.landmarks != VMM_BA_LANDMARK_POINTS) {
        set_error("create_options.landmarks must be VMM_BA_LANDMARK_TAG_POSES or VMM_BA_LANDMARK_POINTS");
        return VMM_BA_ERR_ARGUMENT;
    }
    if (co.landmarks == VMM_BA_LANDMARK_POINTS && co.precision != VMM_BA_PRECISION_F64) {
        set_error("point landmarks run in f64 only");
        return VMM_BA_ERR_ARGUMENT;
    }
    // Point landmarks (doBundleAdjustment_points, src/TagReconstructor.cpp:457-644): every tag becomes its four world
    // corners (:483-491), kept as two 6-dof blocks of two points each; every tag observation becomes the two corner-pair
    // observations of those blocks (:549-560).  From here on `p` is that expanded problem.
    const vmm_ba_problem* const user = p;
    vmm_ba_problem expanded;
    std::vector<double> x_tag, x_wh, x_px;
    std::vector<int32_t> x_cam, x_tagidx;
    if (co.landmarks == VMM_BA_LANDMARK_POINTS) {
        if (p->n_obs >= (int64_t)1 << 30 || p->n_tags >= 1 << 30) {
            set_error("problem too large for point landmarks");
            return VMM_BA_ERR_ARGUMENT;
        }
        x_tag = pairs_from_tags(p->tag_qt, p->tag_wh, p->n_tags);
        x_wh.assign((size_t)4 * p->n_tags, 0.0);
        x_cam.resize((size_t)2 * p->n_obs);
        x_tagidx.resize((size_t)2 * p->n_obs);
        x_px.assign((size_t)16 * p->n_obs, 0.0);
        for (int64_t i = 0; i < p->n_obs; ++i)
            for (int h2 = 0; h2 < 2; ++h2) {
                x_cam[(size_t)2 * i + h2] = p->obs_cam[i];
                x_tagidx[(size_t)2 * i + h2] = 2 * p->obs_tag[i] + h2;
                for (int k = 0; k < 4; ++k)
                    x_px[(size_t)8 * (2 * i + h2) + k] = p->obs_px[8 * i + 4 * h2 + k];
            }
        expanded = *p;
        expanded.n_tags = 2 * p->n_tags;
        expanded.tag_qt = x_tag.data();
        expanded.tag_wh = x_wh.data();
        expanded.n_obs = 2 * p->n_obs;
        expanded.obs_cam = x_cam.data();
        expanded.obs_tag = x_tagidx.data();
        expanded.obs_px = x_px.data();
        p = &expanded;
    }
    if (co.precision != VMM_BA_PRECISION_F64 && co.precision != VMM_BA_PRECISION_F32_ACCUM) {
        set_error("create_options.precision must be VMM_BA_PRECISION_F64 or VMM_BA_PRECISION_F32_ACCUM");
        return VMM_BA_ERR_ARGUMENT;
    }
    if (co.world_size < 1 || co.rank < 0 || co.rank >= co.world_size) {
        set_error("bad rank / world_size");
        return VMM_BA_ERR_ARGUMENT;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device available: libvmm_ba has no CPU fallback");
        return VMM_BA_ERR_HIP;
    }
    if (co.device < 0 || co.device >= ndev) {
        set_error("device ordinal out of range");
        return VMM_BA_ERR_ARGUMENT;
    }
    Engine* ep = new Engine();
    Engine& e = *ep;
    int rc = VMM_BA_OK;
    auto fail = [&](int code) {
        destroy_engine(ep);
        return code;
    };
    e.device = co.device;
    if (hipSetDevice(e.device) != hipSuccess || hipStreamCreateWithFlags(&e.stream, hipStreamNonBlocking) != hipSuccess) {
        set_error("hipSetDevice / hipStreamCreate failed");
        e.stream = nullptr;
        return fail(VMM_BA_ERR_HIP);
    }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, e.device) == hipSuccess && prop.multiProcessorCount > 0)
            e.n_cu = prop.multiProcessorCount;
    }
    e.rank = co.rank;
    e.world = co.world_size;
    e.f32_accum = co.precision == VMM_BA_PRECISION_F32_ACCUM;
    e.multi = e.world > 1;
    if (const char* fw = getenv("VMM_BA_FORCE_COLLECTIVES"))
        if (fw[0] == '1')
            e.multi = true;   // test hook: one rank, but staging buffers + eager launches + all-reduce callbacks
    {
        // every kernel is touched once per process and device before anything is captured
        static bool preloaded[64] = {};
        const char* np = getenv("VMM_BA_NO_PRELOAD");   // diagnosis of the round-1 capture failure only
        if (e.device < 64 && !preloaded[e.device] && !(np && np[0] == '1')) {
            const int bad = preload_eval_kernels() + preload_schur_kernels() + preload_chol_kernels() + preload_lm_kernels()
                + preload_cov_kernels();
            if (bad) {
                set_error("hipFuncGetAttributes failed for " + std::to_string(bad) + " kernels (code object not loadable on this device)");
                return fail(VMM_BA_ERR_HIP);
            }
            preloaded[e.device] = true;
        }
        const char* ef = getenv("VMM_BA_EAGER_FIRST");
        e.eager_first = ef && ef[0] == '1';
        const char* rg = getenv("VMM_BA_RCCL_GRAPH");
        e.rccl_graph = !(rg && rg[0] == '0');
        const char* ng = getenv("VMM_BA_NO_GRAPH");
        e.use_graph = !(ng && ng[0] == '1');
        if (const char* gp = getenv("VMM_BA_GRAPH_PASSES"))
            e.graph_passes = std::min(std::max(atoi(gp), 1), 8);
        const char* nc = getenv("VMM_BA_NO_CHAIN");
        e.no_chain = nc && nc[0] == '1';
        const char* nd = getenv("VMM_BA_NO_DATAFLOW");
        e.no_dataflow = nd && nd[0] == '1';
        read_spin_debug_env(e);
    }
    e.K.fx = p->intr[0]; e.K.fy = p->intr[1]; e.K.cx = p->intr[2]; e.K.cy = p->intr[3];
    e.K.k1 = p->dist[0]; e.K.k2 = p->dist[1]; e.K.p1 = p->dist[2]; e.K.p2 = p->dist[3]; e.K.k3 = p->dist[4];
    e.n_cams = p->n_cams;
    e.n_tags = p->n_tags;
    e.fixed_tag = user->fixed_tag;
    e.n_obs = p->n_obs;
    e.points = co.landmarks == VMM_BA_LANDMARK_POINTS;
    e.n_tags_user = user->n_tags;
    e.n_obs_user = user->n_obs;
    if (e.points) {
        std::vector<double>(user->tag_wh, user->tag_wh + 2 * (size_t)user->n_tags).swap(e.user_tag_wh);
    }
    int elim = co.elimination;
    if (elim == VMM_BA_ELIM_AUTO)
        elim = (p->n_cams >= p->n_tags) ? VMM_BA_ELIM_CAMERAS : VMM_BA_ELIM_TAGS;
    if (elim != VMM_BA_ELIM_CAMERAS && elim != VMM_BA_ELIM_TAGS) {
        set_error("bad elimination mode");
        return fail(VMM_BA_ERR_ARGUMENT);
    }
    e.elim_cams = (elim == VMM_BA_ELIM_CAMERAS);
    e.n_e = e.elim_cams ? e.n_cams : e.n_tags;
    e.n_f = e.elim_cams ? e.n_tags : e.n_cams;
    const int n_pose = e.n_cams + e.n_tags;

    // poses
    if ((rc = dev_alloc(e, &e.cam_qt, (size_t)7 * e.n_cams))) return fail(rc);
    if ((rc = dev_alloc(e, &e.tag_qt, (size_t)7 * e.n_tags))) return fail(rc);
    if ((rc = dev_alloc(e, &e.cam_cand, (size_t)7 * e.n_cams))) return fail(rc);
    if ((rc = dev_alloc(e, &e.tag_cand, (size_t)7 * e.n_tags))) return fail(rc);
    if ((rc = dev_alloc(e, &e.tag_wh, (size_t)2 * e.n_tags))) return fail(rc);
    if (hipMemcpyAsync(e.cam_qt, p->cam_qt, sizeof(double) * 7 * e.n_cams, hipMemcpyHostToDevice, e.stream) != hipSuccess
        || hipMemcpyAsync(e.tag_qt, p->tag_qt, sizeof(double) * 7 * e.n_tags, hipMemcpyHostToDevice, e.stream) != hipSuccess
        || hipMemcpyAsync(e.tag_wh, p->tag_wh, sizeof(double) * 2 * e.n_tags, hipMemcpyHostToDevice, e.stream) != hipSuccess
        || hipStreamSynchronize(e.stream) != hipSuccess) {
        set_error("pose upload failed");
        return fail(VMM_BA_ERR_HIP);
    }

    // observation orders
    const int32_t* own_e = e.elim_cams ? p->obs_cam : p->obs_tag;
    const int32_t* own_f = e.elim_cams ? p->obs_tag : p->obs_cam;
    std::vector<int32_t> callerE, callerF, startE, startF, otherE;
    if ((rc = build_order(e, e.ordE, e.n_e, own_e, own_f, p->obs_px, p->n_obs, &callerE, &startE, &otherE))) return fail(rc);
    if ((rc = build_order(e, e.ordF, e.n_f, own_f, own_e, p->obs_px, p->n_obs, &callerF, &startF))) return fail(rc);

    // Fused evaluation (k_eval_fused: one evaluation per observation, lane = kept pose, both families' sums in
    // registers).  Opt-in, VMM_BA_EVAL=fused: it needs ~370 registers per lane, so one wave per SIMD, and measured
    // SLOWER than the two-pass kernel at two waves per SIMD (500 x 200: 59-68 us against 28.4; 2000 x 1000 f32:
    // 291 against 308).
    {
        int n_act = 0;
        for (int q = 0; q < e.n_e; ++q)
            n_act += startE[q + 1] > startE[q];
        e.fused_eval = false;
        if (const char* ev = getenv("VMM_BA_EVAL"))
            e.fused_eval = !strcmp(ev, "fused") && n_act > 0;
        if (e.fused_eval) {
            e.fused_n_e_act = n_act;
            e.fused_f_pad = round_up(e.n_f, 64);
            e.fused_chunks = e.fused_f_pad / 64;
            // eliminated poses per wave: about one wave per SIMD in flight (1024 SIMDs), at most 16
            e.fused_group = std::min(16, std::max(1, (int)(((int64_t)n_act * e.fused_chunks + 500) / 1000)));
            if (const char* gv = getenv("VMM_BA_FUSED_GROUP"))
                e.fused_group = std::min(64, std::max(1, atoi(gv)));
            e.fused_groups = (n_act + e.fused_group - 1) / e.fused_group;
            std::vector<int32_t> pair((size_t)e.n_e * e.fused_f_pad, -1), e_list, part0((size_t)e.n_e, 0),
                ptask((size_t)e.n_e + 1, 0);
            int slot = 0;
            for (int q = 0; q < e.n_e; ++q) {
                ptask[(size_t)q] = slot;
                part0[(size_t)q] = slot;
                if (startE[q + 1] > startE[q]) {
                    e_list.push_back(q);
                    slot += e.fused_chunks;
                }
                // a pair observed twice keeps its last observation here: such input is not supported by the dense
                // elimination either (one Z block per pair)
                for (int32_t d = startE[q]; d < startE[q + 1]; ++d)
                    pair[(size_t)q * e.fused_f_pad + otherE[(size_t)d]] = d;
            }
            ptask[(size_t)e.n_e] = slot;
            // the lookup table cannot hold two observations of one pair: fall back to the two-pass kernel then
            int64_t n_in_table = 0;
            for (const int32_t v : pair)
                n_in_table += v >= 0;
            if (n_in_table != e.n_obs) {
                e.fused_eval = false;
            } else {
                if ((rc = dev_alloc(e, &e.pair_obs, pair.size()))) return fail(rc);
                if ((rc = dev_alloc(e, &e.fused_e_list, e_list.size()))) return fail(rc);
                if ((rc = dev_alloc(e, &e.fused_e_part0, part0.size()))) return fail(rc);
                if ((rc = dev_alloc(e, &e.fused_pose_task, ptask.size()))) return fail(rc);
                if ((rc = dev_alloc(e, &e.fused_partE, (size_t)std::max(slot, 1) * kPart))) return fail(rc);
                if ((rc = dev_alloc(e, &e.fused_partF, (size_t)e.fused_groups * 28 * e.fused_f_pad))) return fail(rc);
                if ((rc = upload(e, e.pair_obs, pair))) return fail(rc);
                if ((rc = upload(e, e.fused_e_list, e_list))) return fail(rc);
                if ((rc = upload(e, e.fused_e_part0, part0))) return fail(rc);
                if ((rc = upload(e, e.fused_pose_task, ptask))) return fail(rc);
                if (hipStreamSynchronize(e.stream) != hipSuccess) {   // host vectors go out of scope
                    set_error("create: upload of the fused-evaluation tables failed");
                    return fail(VMM_BA_ERR_HIP);
                }
            }
        }
    }

    // normal-equation blocks
    e.small_count = (size_t)42 * n_pose + 2;
    if ((rc = dev_alloc(e, &e.small, e.small_count))) return fail(rc);
    e.H_cam = e.small;
    e.H_tag = e.H_cam + (size_t)36 * e.n_cams;
    e.g_cam = e.H_tag + (size_t)36 * e.n_tags;
    e.g_tag = e.g_cam + (size_t)6 * e.n_cams;
    e.cost_slot = e.g_tag + (size_t)6 * e.n_tags;
    // the LM loop evaluates at the candidate into this staging copy (all-reduced when world > 1); it replaces the
    // working copy when the step is accepted
    // (+ n_e per-pose costs of the eliminated family behind it: world > 1 sums them over the ranks with the blocks)
    if ((rc = dev_alloc(e, &e.small_stage, e.small_count + (size_t)e.n_e))) return fail(rc);
    e.ev_pose_cost = e.small_stage + e.small_count;
    e.small_alt_off = e.multi ? 0 : (int64_t)(e.small_stage - e.small);
    e.ev_H_cam = e.small_stage;
    e.ev_H_tag = e.ev_H_cam + (size_t)36 * e.n_cams;
    e.ev_g_cam = e.ev_H_tag + (size_t)36 * e.n_tags;
    e.ev_g_tag = e.ev_g_cam + (size_t)6 * e.n_cams;
    e.ev_cost = e.ev_g_tag + (size_t)6 * e.n_tags;
    if ((rc = dev_alloc(e, &e.obs_mask, (size_t)std::max<int64_t>(e.n_obs, 1), false))) return fail(rc);
    if (hipMemsetAsync(e.obs_mask, 1, (size_t)std::max<int64_t>(e.n_obs, 1), e.stream) != hipSuccess) {
        set_error("hipMemsetAsync(obs_mask) failed");
        return fail(VMM_BA_ERR_HIP);
    }
    {
        const size_t n_stat = (size_t)e.ordE.n_tasks + (size_t)e.ordF.n_tasks + 1;
        if ((rc = dev_alloc(e, &e.stats_part, n_stat))) return fail(rc);
        if ((rc = dev_alloc(e, &e.stats_cnt, n_stat))) return fail(rc);
        if ((rc = dev_alloc(e, &e.stats_pose, (size_t)2 * n_pose))) return fail(rc);
    }
    if (e.f32_accum) {
        if ((rc = dev_alloc(e, &e.Wf, (size_t)36 * e.ordE.n_pad))) return fail(rc);
        if ((rc = dev_alloc(e, &e.Wf2, (size_t)36 * e.ordE.n_pad))) return fail(rc);
    } else {
        if ((rc = dev_alloc(e, &e.W, (size_t)36 * e.ordE.n_pad))) return fail(rc);
        if ((rc = dev_alloc(e, &e.W2, (size_t)36 * e.ordE.n_pad))) return fail(rc);
    }
    if ((rc = dev_alloc(e, &e.scale, (size_t)6 * n_pose))) return fail(rc);
    if ((rc = dev_alloc(e, &e.diag, (size_t)6 * n_pose))) return fail(rc);
    if ((rc = dev_alloc(e, &e.D2, (size_t)6 * n_pose))) return fail(rc);
    if ((rc = dev_alloc(e, &e.delta, (size_t)6 * n_pose))) return fail(rc);
    if ((rc = dev_alloc(e, &e.active, (size_t)n_pose))) return fail(rc);

    // elimination / reduced system geometry
    e.n_red = 6 * e.n_f;
    e.n_pad = round_up(e.n_red, kNB);
    e.n_blk = e.n_pad / kNB;
    // >= n_pad + 64; the rank-k update reads whole 128-wide tiles.  The extra 32 doubles (256 B) make the row
    // stride an odd multiple of 256 B, so the 64 rows of a tile spread over the HBM channels instead of
    // hitting a few of them (a 48 KB stride at 2000 x 1000 does)
    e.ldz = round_up(e.n_pad + 1, kST) + 32;
    e.k_dim = 6 * e.n_e;
    e.k_pad = round_up(e.k_dim, kKT);
    if ((rc = dev_alloc(e, &e.Le, (size_t)36 * e.n_e))) return fail(rc);
    if ((rc = dev_alloc(e, &e.ze, (size_t)6 * e.n_e))) return fail(rc);
    // Reduced-system formation: dense Z + MFMA rank-k update, or compressed Z + the pair-list kernel.  Both are
    // priced per launch from the block structure (measured at 500 x 200, profiles/r03_sparse_*: the dense update + its
    // partial-tile sum take 118 us whatever the fill = 44 TFLOP/s; k_schur_pairs 16 us at 6-10 tags per image, 57 us at
    // 25 % and 116 us at 50 % visibility, priced as 18 us + 9 TFLOP/s of useful 6x6x6 block products);
    // VMM_BA_SCHUR=dense|sparse overrides.
    // World > 1: every decision that shapes the reduced system (its form is free per rank, its LAYOUT is not: the ranks'
    // systems are summed) is taken from the structure of ALL ranks' observations when the caller passes it
    // (vmm_ba_create_options.structure_obs_*), else from this rank's own -- and then the layout stays the natural one.
    // gStart / gOther: per eliminated pose the kept poses it sees (family-local indices), this rank's or everybody's.
    std::vector<int32_t> gStartV, gOtherV;
    const bool have_structure = co.n_structure_obs > 0 && co.structure_obs_cam && co.structure_obs_tag;
    if (e.multi && have_structure) {
        gStartV.assign((size_t)e.n_e + 1, 0);
        for (int64_t d = 0; d < co.n_structure_obs; ++d) {
            const int32_t c = co.structure_obs_cam[d], t = co.structure_obs_tag[d];
            if (c < 0 || c >= p->n_cams || t < 0 || t >= p->n_tags) {
                set_error("structure_obs index out of range");
                return fail(VMM_BA_ERR_ARGUMENT);
            }
            if (e.points) {
                set_error("structure_obs is not supported with point landmarks");
                return fail(VMM_BA_ERR_ARGUMENT);
            }
            gStartV[(size_t)(e.elim_cams ? c : t) + 1]++;
        }
        for (int q = 0; q < e.n_e; ++q)
            gStartV[(size_t)q + 1] += gStartV[(size_t)q];
        gOtherV.resize((size_t)co.n_structure_obs);
        std::vector<int32_t> fillg(gStartV.begin(), gStartV.end() - 1);
        for (int64_t d = 0; d < co.n_structure_obs; ++d) {
            const int32_t c = co.structure_obs_cam[d], t = co.structure_obs_tag[d];
            gOtherV[(size_t)fillg[(size_t)(e.elim_cams ? c : t)]++] = e.elim_cams ? t : c;
        }
    }
    const bool global_lists = !gStartV.empty();
    const std::vector<int32_t>& gStart = global_lists ? gStartV : startE;
    const std::vector<int32_t>& gOther = global_lists ? gOtherV : otherE;
    // one layout for all ranks: this rank alone (also the one-rank test hook), or everybody's structure in hand
    const bool layout_free = !e.multi || e.world == 1 || global_lists;
    {
        double pairs = 0.0;   // 6x6 block products of the lower triangle: sum over e of deg (deg + 1) / 2
        e.co_terms = 0.0;
        for (int q = 0; q < e.n_e; ++q) {
            const double deg = (double)(gStart[q + 1] - gStart[q]);
            pairs += 0.5 * deg * (deg + 1.0);
            e.co_terms += deg * deg;   // entries the host's adjacency lists of the kept family would hold before merging
        }
        const double dense_flops = (double)(e.n_pad + 1) * (e.n_pad + 2) * e.k_dim;
        const double n_obs_model = global_lists ? (double)co.n_structure_obs : (double)e.n_obs;
        const double sparse_flops = 432.0 * pairs + 72.0 * n_obs_model;
        const double dense_us = dense_flops / 44e6 + 12.0, sparse_us = sparse_flops / 9e6 + 18.0;
        // The plan of the block-sparse form lists every term: one (left, right) position pair per product plus the
        // right-hand-side term of every observation -- known from the degrees alone, before anything is allocated.  16 bytes
        // per term on the host while it is built, 8 on the device: the automatic choice stays below 4e7 terms (0.64 GB
        // transient, 0.32 GB resident; 2000 x 1000 at 25 % visibility would be 6e7), a forced one below the 2^31 the
        // 32-bit positions can address.
        const double plan_terms = pairs + n_obs_model;
        e.sparse_schur = n_obs_model > 0 && sparse_us < dense_us && plan_terms <= 4e7;
        if (const char* sv = getenv("VMM_BA_SCHUR")) {
            if (!strcmp(sv, "dense"))
                e.sparse_schur = false;
            else if (!strcmp(sv, "sparse"))
                e.sparse_schur = n_obs_model > 0;
        }
        if (e.sparse_schur && plan_terms >= 2147483647.0) {
            set_error("block-sparse elimination: more than 2^31 block products (set VMM_BA_SCHUR=dense)");
            return fail(VMM_BA_ERR_ARGUMENT);
        }
        e.schur_flops = e.sparse_schur ? sparse_flops : dense_flops;
    }
    std::vector<std::vector<int32_t>> tree_nbr;   // co-observation graph of the kept family (all ranks' when known)
    // Tree ordering of the kept family (VMM_BA_ORDER=nd; block-sparse path; world > 1: with the global structure): every node of the dissection tree
    // starts on a 64-row boundary of the reduced system (padding rows with a unit diagonal in between), so that whole
    // block columns of the factor belong to one node and the block columns of two parts of one level are independent.
    {
        const char* ov = getenv("VMM_BA_ORDER");
        const bool forced = ov && !strcmp(ov, "nd"), forbidden = ov && !strcmp(ov, "natural");
        // (the host-side graph work is bounded: 5e7 list entries, and a natural order beyond the one-launch kernel's 48 block
        // columns cannot become a tree order within them)
        if (e.sparse_schur && layout_free && !forbidden && e.n_f > 1 && e.n_blk >= 4 && e.co_terms <= 5e7) {
            std::vector<std::vector<int32_t>>& nbr = tree_nbr;
            nbr.assign((size_t)e.n_f, {});
            for (int q = 0; q < e.n_e; ++q)
                for (int32_t d1 = gStart[q]; d1 < gStart[q + 1]; ++d1)
                    for (int32_t d2 = gStart[q]; d2 < gStart[q + 1]; ++d2)
                        if (d1 != d2)
                            nbr[(size_t)gOther[(size_t)d1]].push_back(gOther[(size_t)d2]);
            for (auto& v : nbr) {
                std::sort(v.begin(), v.end());
                v.erase(std::unique(v.begin(), v.end()), v.end());
            }
            std::vector<int32_t> all((size_t)e.n_f), stamp((size_t)e.n_f, 0);
            for (int fq = 0; fq < e.n_f; ++fq)
                all[(size_t)fq] = fq;
            int32_t stamp_next = 1;
            int leaf_max = 42;   // tags per leaf (four 64-row blocks; 21 and 10 measured slower on the close-up scene)
            if (const char* lv = getenv("VMM_BA_ND_LEAF"))
                leaf_max = std::max(1, atoi(lv));
            std::vector<std::vector<int32_t>> nodes;
            nd_dissect(nbr, all, leaf_max, 12, stamp, stamp_next, nodes);
            std::vector<int32_t> rows((size_t)e.n_f, -1);
            int row = 0;
            for (const auto& nd : nodes) {
                row = round_up(row, kNB);
                for (const int32_t v : nd) {
                    rows[(size_t)v] = row;
                    row += 6;
                }
            }
            const int n_pad_nd = round_up(row, kNB);
            // Worth it?  The factorisation is a chain of dependent block columns (~11 us each): the longest chain under
            // the tree ordering (block structure after symbolic fill, nodes as dense blocks: an upper bound) against the
            // n_blk of the natural order.  Taken when it is at most 0.7 of it (VMM_BA_ORDER=nd: always).
            // (the block structure is kept as kDfMaskWords 64-bit words per block row, the panel order in bytes: at most 255
            // block columns; only the non-zero blocks of the factor get a workgroup, so the one-launch kernel takes the
            // system whatever its order -- counted below)
            const int nb = n_pad_nd / kNB;
            bool take = nodes.size() > 2 && nb <= kDfMaxBlk - 1;
            if (take) {
                const std::vector<BlkMask> nzr = symbolic_factor(nb, rows, nbr);
                int n_wg = nb;
                for (int i = 0; i < nb; ++i)
                    for (int k = 0; k < i; ++k)
                        n_wg += nzr[(size_t)i].test(k) ? 1 : 0;
                n_wg += nb;   // the right-hand side row's block of every column
                int path_max = 1;
                const double t_tree = model_tree_factorisation(nb, nzr, nullptr, &path_max);
                // natural order: the one-launch kernel up to 48 block columns (~9.8 us each), one launch per column beyond
                // (~32 us each at 94 columns)
                const double t_nat = e.n_blk <= 48 ? 9.8 * e.n_blk : 32.0 * e.n_blk;
                static const int max_wg = [] {
                    const char* v = getenv("VMM_BA_TREE_MAX_WG");
                    return v ? atoi(v) : 16384;
                }();
                take = n_wg <= max_wg && (forced || (path_max * 10 <= e.n_blk * 7 && t_tree <= 0.9 * t_nat));
                if (getenv("VMM_BA_DEBUG"))
                    fprintf(stderr, "[vmm_ba debug] tree ordering candidate: longest chain %d of %d block columns against %d in "
                                    "natural order, %d workgroups, modelled %.0f against %.0f us -> %s\n", path_max, nb, e.n_blk,
                            n_wg, t_tree, t_nat, take ? "taken" : "not taken");
            }
            if (take) {
                e.h_row_of = rows;
                e.nd_node_first_blk.clear();
                int r2 = 0;
                for (const auto& nd : nodes) {
                    r2 = round_up(r2, kNB);
                    e.nd_node_first_blk.push_back(r2 / kNB);
                    r2 += 6 * (int)nd.size();
                }
                e.n_pad = n_pad_nd;
                e.n_blk = e.n_pad / kNB;
                e.ldz = round_up(e.n_pad + 1, kST) + 32;
                if (getenv("VMM_BA_DEBUG")) {
                    fprintf(stderr, "[vmm_ba debug] tree ordering: %zu nodes, %d rows (%d blocks) for %d kept poses; node sizes:",
                            nodes.size(), e.n_pad, e.n_blk, e.n_f);
                    for (const auto& nd : nodes)
                        fprintf(stderr, " %zu", nd.size());
                    fprintf(stderr, "\n");
                }
            }
        }
    }
    if (e.sparse_schur) {
        if ((rc = dev_alloc(e, &e.Zc, (size_t)36 * std::max<int64_t>(e.n_obs, 1)))) return fail(rc);
        // F-order <-> E-order positions of an observation
        std::vector<int32_t> posE((size_t)e.n_obs), f2e((size_t)e.n_obs), row_pos((size_t)e.n_obs);
        for (int64_t d = 0; d < e.n_obs; ++d)
            posE[(size_t)callerE[(size_t)d]] = (int32_t)d;
        for (int fq = 0; fq < e.n_f; ++fq)
            for (int32_t d = startF[fq]; d < startF[fq + 1]; ++d) {
                f2e[(size_t)d] = posE[(size_t)callerF[(size_t)d]];
                row_pos[(size_t)f2e[(size_t)d]] = d - startF[fq];   // position of the observation in its kept pose's row
            }
        // The symbolic structure of S -= Z^T Z, once per problem (the counterpart of the symbolic phase of the sparse
        // Cholesky behind ceres::Solve): row f owns the pairs (f, f' = 0..f) and, last, its right-hand side entry.
        // A term of pair (f, f') = two observations (e, f), (e, f') of one eliminated pose; terms are listed in e
        // order (that is the summation order), the left block by its position in f's row (the kernel stages the
        // row's blocks in LDS), the right block by its E-order index.
        // Two forms of the pair list.  Implicit (every pair of the lower triangle, the empty ones written as zeros:
        // pair j of row f is f' = j): visibility-type scenes, where nearly every pair exists.  Explicit (only the pairs
        // that share an eliminated pose, their column in `pair_col`; S is zero-filled by a kernel of its own first):
        // scenes where an image sees a handful of tags -- at 6-10 tags per image 3.4 k of the 20.1 k pairs exist.  The
        // explicit form also carries the position of every kept pose in the reduced system (`row_of`), which need not
        // be 6 f (tree orderings of the kept family).
        std::vector<int32_t> rank_of((size_t)e.n_f);                  // order of the kept poses in the reduced system
        for (int fq = 0; fq < e.n_f; ++fq)
            rank_of[(size_t)fq] = e.h_row_of.empty() ? fq : e.h_row_of[(size_t)fq];
        std::vector<std::vector<int32_t>> partners;                    // explicit form: f' of every pair of row f
        {
            double co = 0.0;   // co-observed pairs incl. the diagonal, counted once
            std::vector<int32_t> mark((size_t)e.n_f, -1);
            std::vector<std::vector<int32_t>> adj((size_t)e.n_f);
            const bool list_pairs = !e.h_row_of.empty() || e.co_terms <= 5e7;   // else: the implicit form
            for (int q = 0; q < e.n_e && list_pairs; ++q)
                for (int32_t d1 = startE[q]; d1 < startE[q + 1]; ++d1)
                    for (int32_t d2 = startE[q]; d2 < startE[q + 1]; ++d2) {
                        const int f1 = otherE[(size_t)d1], f2 = otherE[(size_t)d2];
                        if (rank_of[(size_t)f2] < rank_of[(size_t)f1])
                            adj[(size_t)f1].push_back(f2);
                    }
            for (int fq = 0; fq < e.n_f; ++fq) {
                std::vector<int32_t>& a = adj[(size_t)fq];
                std::sort(a.begin(), a.end(), [&](int32_t x, int32_t y) { return rank_of[(size_t)x] < rank_of[(size_t)y]; });
                a.erase(std::unique(a.begin(), a.end()), a.end());
                a.push_back(fq);   // the diagonal pair: always there (it carries the kept pose's own block)
                co += (double)a.size();
            }
            const double all = 0.5 * (double)e.n_f * (e.n_f + 1.0);
            e.explicit_pairs = !e.h_row_of.empty() || (list_pairs && co < 0.5 * all);
            if (const char* pv = getenv("VMM_BA_PAIRS"))
                e.explicit_pairs = !e.h_row_of.empty() || (list_pairs && !strcmp(pv, "explicit"));
            if (e.explicit_pairs)
                partners.swap(adj);
            (void)mark;
        }
        std::vector<int32_t> pstart((size_t)e.n_f + 1, 0), pcol;
        for (int fq = 0; fq < e.n_f; ++fq)
            pstart[(size_t)fq + 1] = pstart[(size_t)fq] + (e.explicit_pairs ? (int32_t)partners[(size_t)fq].size() + 1 : fq + 2);
        const size_t n_pairs = (size_t)pstart[(size_t)e.n_f];
        // pair id of (f1, f2) inside row f1; explicit form: through a per-row look-up table
        std::vector<int32_t> slot_of;
        if (e.explicit_pairs) {
            pcol.assign(n_pairs, -1);
            for (int fq = 0; fq < e.n_f; ++fq)
                for (size_t k = 0; k < partners[(size_t)fq].size(); ++k)
                    pcol[(size_t)pstart[(size_t)fq] + k] = e.h_row_of.empty() ? 6 * partners[(size_t)fq][k]
                                                                              : e.h_row_of[(size_t)partners[(size_t)fq][k]];
        }
        // (f1, f2) -> pair id or -1.  Explicit: binary search in row f1's partner list (sorted by rank).
        auto pair_id = [&](int f1, int f2) -> int64_t {
            if (!e.explicit_pairs)
                return f2 <= f1 ? (int64_t)pstart[(size_t)f1] + f2 : -1;
            if (rank_of[(size_t)f2] > rank_of[(size_t)f1])
                return -1;
            const std::vector<int32_t>& a = partners[(size_t)f1];
            const auto it = std::lower_bound(a.begin(), a.end(), f2,
                                             [&](int32_t x, int32_t y) { return rank_of[(size_t)x] < rank_of[(size_t)y]; });
            return (int64_t)pstart[(size_t)f1] + (it - a.begin());
        };
        auto rhs_id = [&](int f1) -> int64_t { return (int64_t)pstart[(size_t)f1 + 1] - 1; };
        std::vector<int32_t> tstart(n_pairs + 1, 0);
        for (int q = 0; q < e.n_e; ++q)
            for (int32_t d1 = startE[q]; d1 < startE[q + 1]; ++d1) {
                const int f1 = otherE[(size_t)d1];
                tstart[(size_t)rhs_id(f1) + 1]++;   // rhs pair of row f1
                for (int32_t d2 = startE[q]; d2 < startE[q + 1]; ++d2) {
                    const int64_t id = pair_id(f1, otherE[(size_t)d2]);
                    if (id >= 0)
                        tstart[(size_t)id + 1]++;
                }
            }
        for (size_t k = 0; k < n_pairs; ++k)
            tstart[k + 1] += tstart[k];
        const size_t n_terms = (size_t)tstart[n_pairs];
        if (n_terms >= ((size_t)1 << 31)) {
            set_error("block-sparse elimination: more than 2^31 block products (set VMM_BA_SCHUR=dense)");
            return fail(VMM_BA_ERR_ARGUMENT);
        }
        std::vector<int32_t> ta(n_terms), tb(n_terms), fill(tstart.begin(), tstart.end() - 1);
        for (int q = 0; q < e.n_e; ++q)
            for (int32_t d1 = startE[q]; d1 < startE[q + 1]; ++d1) {
                const int f1 = otherE[(size_t)d1];
                {
                    const int32_t k = fill[(size_t)rhs_id(f1)]++;
                    ta[(size_t)k] = row_pos[(size_t)d1];
                    tb[(size_t)k] = q;
                }
                for (int32_t d2 = startE[q]; d2 < startE[q + 1]; ++d2) {
                    const int64_t id = pair_id(f1, otherE[(size_t)d2]);
                    if (id < 0)
                        continue;
                    const int32_t k = fill[(size_t)id]++;
                    ta[(size_t)k] = row_pos[(size_t)d1];
                    tb[(size_t)k] = d2;
                }
            }
        // the kernel walks a pair's terms pass by pass of 128 left blocks: terms must be ordered by left position.
        // They are listed in e order; a row's positions follow the caller's order, which need not be e order.
        for (size_t k = 0; k < n_pairs; ++k) {
            const int32_t t0 = tstart[k], t1 = tstart[k + 1];
            bool sorted = true;
            for (int32_t t = t0 + 1; t < t1 && sorted; ++t)
                sorted = ta[(size_t)t - 1] <= ta[(size_t)t];
            if (!sorted) {
                std::vector<std::pair<int32_t, int32_t>> tmp;
                for (int32_t t = t0; t < t1; ++t)
                    tmp.emplace_back(ta[(size_t)t], tb[(size_t)t]);
                std::stable_sort(tmp.begin(), tmp.end(),
                                 [](const std::pair<int32_t, int32_t>& x, const std::pair<int32_t, int32_t>& y) { return x.first < y.first; });
                for (int32_t t = t0; t < t1; ++t) {
                    ta[(size_t)t] = tmp[(size_t)(t - t0)].first;
                    tb[(size_t)t] = tmp[(size_t)(t - t0)].second;
                }
            }
        }
        // work items: up to pairs_per_item() consecutive pairs of one row; rows with many pairs first
        std::vector<int32_t> item_row, item_p0;
        const int ppi = schur_pairs_per_item();
        std::vector<int32_t> rows_by_len((size_t)e.n_f);
        for (int fq = 0; fq < e.n_f; ++fq)
            rows_by_len[(size_t)fq] = e.n_f - 1 - fq;
        if (e.explicit_pairs)
            std::stable_sort(rows_by_len.begin(), rows_by_len.end(), [&](int32_t x, int32_t y) {
                return pstart[(size_t)x + 1] - pstart[(size_t)x] > pstart[(size_t)y + 1] - pstart[(size_t)y];
            });
        for (const int32_t fq : rows_by_len)
            for (int j0 = 0; j0 < pstart[(size_t)fq + 1] - pstart[(size_t)fq]; j0 += ppi) {
                item_row.push_back(fq);
                item_p0.push_back(pstart[(size_t)fq] + j0);
            }
        e.n_row_items = (int)item_row.size();
        std::vector<int32_t> items(item_row);
        items.insert(items.end(), item_p0.begin(), item_p0.end());
        if ((rc = dev_alloc(e, &e.f2e, f2e.size()))) return fail(rc);
        if ((rc = dev_alloc(e, &e.pair_start, pstart.size()))) return fail(rc);
        if ((rc = dev_alloc(e, &e.pair_tstart, tstart.size()))) return fail(rc);
        std::vector<int32_t> tt(2 * std::max<size_t>(n_terms, 1), 0);
        for (size_t k = 0; k < n_terms; ++k) {
            tt[2 * k] = ta[k];
            tt[2 * k + 1] = tb[k];
        }
        if ((rc = dev_alloc(e, &e.pair_terms, tt.size()))) return fail(rc);
        if ((rc = dev_alloc(e, &e.row_items, items.size()))) return fail(rc);
        if ((rc = upload(e, e.f2e, f2e))) return fail(rc);
        if ((rc = upload(e, e.pair_start, pstart))) return fail(rc);
        if ((rc = upload(e, e.pair_tstart, tstart))) return fail(rc);
        if ((rc = upload(e, e.pair_terms, tt))) return fail(rc);
        if ((rc = upload(e, e.row_items, items))) return fail(rc);
        if (!e.h_row_of.empty()) {
            // Block structure of the factor under the tree ordering (symbolic_factor): from the co-observation graph the
            // ordering was made from -- ALL observations (an observation mask only removes entries), all ranks' with
            // world > 1 (the summed system has an entry wherever any rank has one).
            const std::vector<BlkMask> nzr = symbolic_factor(e.n_blk, e.h_row_of, tree_nbr);
            std::vector<unsigned long long> nzw((size_t)(e.n_blk + 1) * kDfMaskWords);
            for (int i = 0; i <= e.n_blk; ++i)
                for (int w = 0; w < kDfMaskWords; ++w)
                    nzw[(size_t)i * kDfMaskWords + w] = nzr[(size_t)i].w[w];
            if ((rc = dev_alloc(e, &e.chol_nz, nzw.size()))) return fail(rc);
            if ((rc = upload(e, e.chol_nz, nzw))) return fail(rc);
            {
                // what this factorisation computes: per block column the diagonal block (d^3 / 3), a triangular solve per
                // non-zero block below it (d^3), a product per pair of them (2 d^3; d^3 on the diagonal), the right-hand
                // side row with them (2 d^2 per block), and the back-substitution (2 d^2 per block)
                const double d = (double)kNB;
                double fl = 0.0;
                for (int j2 = 0; j2 < e.n_blk; ++j2) {
                    int nb = 0;
                    for (int r = j2 + 1; r < e.n_blk; ++r)
                        nb += nzr[(size_t)r].test(j2) ? 1 : 0;
                    fl += d * d * d / 3.0 + nb * d * d * d + (double)nb * nb * d * d * d + 4.0 * (nb + 1) * d * d;
                }
                e.chol_flops = fl;
            }
            // In which order does block column j take the panels it depends on?  In the order they are expected to be
            // finished, from the model of the kernel.
            std::vector<unsigned char> ord((size_t)e.n_blk * kDfMaxBlk, 0);
            int path_max = 0;
            const double t_model = model_tree_factorisation(e.n_blk, nzr, &ord, &path_max);
            if ((rc = dev_alloc(e, &e.chol_order, ord.size()))) return fail(rc);
            if ((rc = upload(e, e.chol_order, ord))) return fail(rc);
            // The launch: one workgroup per non-zero block below the diagonal (the right-hand side row's last) and the
            // diagonal-only workgroup, panel-major -- a workgroup only ever waits for workgroups in front of it -- and one
            // slot of published slices per block that has a workgroup.
            std::vector<int32_t> wg, slot((size_t)e.n_blk * (e.n_blk + 1), -1);
            int32_t n_slots = 0;
            for (int j2 = 0; j2 < e.n_blk; ++j2) {
                for (int r = j2 + 1; r <= e.n_blk; ++r)
                    if (r == e.n_blk || nzr[(size_t)r].test(j2)) {
                        wg.push_back(j2);
                        wg.push_back(r);
                        slot[(size_t)j2 * (e.n_blk + 1) + r] = n_slots++;
                    }
                wg.push_back(j2);
                wg.push_back(j2);
            }
            e.n_df_wg = (int)(wg.size() / 2);
            e.df_tree_slots = (size_t)n_slots;
            if ((rc = dev_alloc(e, &e.df_wg, wg.size()))) return fail(rc);
            if ((rc = upload(e, e.df_wg, wg))) return fail(rc);
            if ((rc = dev_alloc(e, &e.df_slot, slot.size()))) return fail(rc);
            if ((rc = upload(e, e.df_slot, slot))) return fail(rc);
            e.chol_nz_on = true;
            if (getenv("VMM_BA_DEBUG")) {
                fprintf(stderr, "[vmm_ba debug] factor: longest chain %d of %d block columns, modelled %.0f us, %d workgroups, "
                                "%d of %d lower blocks\n", path_max, e.n_blk, t_model, e.n_df_wg, n_slots - e.n_blk + e.n_blk,
                        e.n_blk * (e.n_blk + 1) / 2);
                if (e.n_blk <= 64)
                    for (int i = 0; i < e.n_blk; ++i) {
                        fprintf(stderr, "[vmm_ba debug]   %2d ", i);
                        for (int k = 0; k <= i; ++k)
                            fputc(nzr[(size_t)i].test(k) ? 'x' : '.', stderr);
                        fputc('\n', stderr);
                    }
            }
        }
        if (e.explicit_pairs) {
            std::vector<int32_t> rows((size_t)e.n_f);
            for (int fq = 0; fq < e.n_f; ++fq)
                rows[(size_t)fq] = e.h_row_of.empty() ? 6 * fq : e.h_row_of[(size_t)fq];
            if ((rc = dev_alloc(e, &e.pair_col, pcol.size()))) return fail(rc);
            if ((rc = dev_alloc(e, &e.row_of, rows.size()))) return fail(rc);
            if ((rc = upload(e, e.pair_col, pcol))) return fail(rc);
            if ((rc = upload(e, e.row_of, rows))) return fail(rc);
            if (e.multi && !e.h_row_of.empty()) {
                // world > 1 with a tree ordering: which kept pose a row of the reduced system belongs to (-1: padding), for the
                // kernel that adds the kept family's diagonal blocks behind the all-reduce (k_unpack_diag)
                std::vector<int32_t> pose_of((size_t)e.n_pad, -1);
                for (int fq = 0; fq < e.n_f; ++fq)
                    for (int k = 0; k < 6; ++k)
                        pose_of[(size_t)rows[(size_t)fq] + k] = fq;
                if ((rc = dev_alloc(e, &e.pose_of_row, pose_of.size()))) return fail(rc);
                if ((rc = upload(e, e.pose_of_row, pose_of))) return fail(rc);
            }
        }
        if (hipStreamSynchronize(e.stream) != hipSuccess) {   // host vectors go out of scope
            set_error("create: upload of the block-sparse plan failed");
            return fail(VMM_BA_ERR_HIP);
        }
    } else {
        if ((rc = ensure_dense_schur(e))) return fail(rc);
    }
    if ((rc = dev_alloc(e, &e.S, (size_t)e.ldz * e.ldz))) return fail(rc);
    if (e.multi && (rc = dev_alloc(e, &e.S_packed, (size_t)(e.n_pad + 1) * (e.n_pad + 2) / 2))) return fail(rc);
    if ((rc = dev_alloc(e, &e.P, (size_t)4 * kNB * e.ldz))) return fail(rc);
    if ((rc = setup_lookahead(e, e.n_blk, e.ldz))) return fail(rc);
    if ((rc = dev_alloc(e, &e.dinv, (size_t)e.ldz))) return fail(rc);
    if ((rc = dev_alloc(e, &e.Ldiag, (size_t)(e.n_blk + 1) * 4096))) return fail(rc);
    if ((rc = dev_alloc(e, &e.Linv, (size_t)(e.n_blk + 1) * 4096))) return fail(rc);
    if ((rc = dev_alloc(e, &e.flags, 264))) return fail(rc);
    if ((rc = dev_alloc(e, &e.gran, (size_t)2 * e.ldz))) return fail(rc);
    {
        // published slices of the one-launch factorisation: 64 KB per block below the diagonal (dense: of the block columns
        // that kernel takes; tree ordering: of the non-zero blocks, and of the dense kernel's share when the handle is
        // switched to the natural dense system for a call)
        const size_t nd = e.no_dataflow ? 0 : (size_t)dataflow_blocks(e.n_blk, e.n_cu);
        const size_t slots = std::max(nd * (nd + 1) / 2, e.no_dataflow ? (size_t)0 : e.df_tree_slots);
        if (slots > 0 && (rc = dev_alloc(e, &e.df_gran, slots * 8 * 1024)))
            return fail(rc);
        if (slots > 0 && (rc = dev_alloc(e, &e.df_compact, slots * 4096, false)))
            return fail(rc);
        if (slots > 0 && (rc = dev_alloc(e, &e.df_done, slots)))
            return fail(rc);
    }
    if ((rc = dev_alloc(e, &e.yf, (size_t)e.ldz))) return fail(rc);
    if ((rc = dev_alloc(e, &e.step_comm, (size_t)7 * e.n_e + 1))) return fail(rc);
    if ((rc = dev_alloc(e, &e.cost_comm, 2))) return fail(rc);
    const size_t n_part = (size_t)std::max<int>(e.ordE.n_tasks, e.n_e) + 1;
    if ((rc = dev_alloc(e, &e.part_cost, n_part))) return fail(rc);
    if ((rc = dev_alloc(e, &e.part_cross, n_part))) return fail(rc);
    if ((rc = dev_alloc(e, &e.part_k1, n_part))) return fail(rc);
    if ((rc = dev_alloc(e, &e.pose_part, (size_t)5 * n_pose))) return fail(rc);
    if ((rc = dev_alloc(e, &e.pose_gm, (size_t)n_pose))) return fail(rc);
    if ((rc = dev_alloc(e, &e.ctl, 1))) return fail(rc);
    if (hipHostMalloc((void**)&e.ctl_host, sizeof(LmCtl)) != hipSuccess) {
        set_error("hipHostMalloc failed");
        e.ctl_host = nullptr;
        return fail(VMM_BA_ERR_HIP);
    }
    memset(e.ctl_host, 0, sizeof(LmCtl));
    if (hipHostMalloc((void**)&e.pose_stage, sizeof(double) * 7 * (size_t)n_pose, hipHostMallocMapped) != hipSuccess
        || hipHostGetDevicePointer((void**)&e.pose_stage_dev, e.pose_stage, 0) != hipSuccess
        || hipEventCreateWithFlags(&e.pose_ev, hipEventDisableTiming) != hipSuccess) {
        set_error("hipHostMalloc / hipEventCreate (pose staging) failed");
        return fail(VMM_BA_ERR_HIP);
    }
    e.trace_capacity = 0;
    if (hipStreamSynchronize(e.stream) != hipSuccess) {
        set_error("create: device synchronisation failed");
        return fail(VMM_BA_ERR_HIP);
    }
    *out = reinterpret_cast<vmm_ba_handle>(ep);
    return VMM_BA_OK;
}

void vmm_ba_destroy(vmm_ba_handle h) { destroy_engine(reinterpret_cast<Engine*>(h)); }

int vmm_ba_set_allreduce(vmm_ba_handle h, vmm_ba_allreduce_fn fn, void* user)
{
    if (!h) {
        set_error("null handle");
        return VMM_BA_ERR_ARGUMENT;
    }
    Engine& e = *reinterpret_cast<Engine*>(h);
    e.allreduce = fn;
    e.allreduce_user = user;
    return VMM_BA_OK;
}

int vmm_ba_rccl_available(void) { return rccl().ok ? 1 : 0; }

int vmm_ba_rccl_unique_id(void* id128)
{
    if (!id128) {
        set_error("null argument");
        return VMM_BA_ERR_ARGUMENT;
    }
    if (!rccl().ok) {
        set_error("librccl.so could not be loaded");
        return VMM_BA_ERR_COLLECTIVE;
    }
    static_assert(sizeof(ncclUniqueId) == VMM_BA_RCCL_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    const ncclResult_t r = rccl().GetUniqueId(&id);
    if (r != ncclSuccess) {
        set_error(std::string("ncclGetUniqueId: ") + rccl().GetErrorString(r));
        return VMM_BA_ERR_COLLECTIVE;
    }
    memcpy(id128, &id, sizeof(id));
    return VMM_BA_OK;
}

int vmm_ba_enable_rccl(vmm_ba_handle h, const void* id128)
{
    if (!h || !id128) {
        set_error("null argument");
        return VMM_BA_ERR_ARGUMENT;
    }
    Engine& e = *reinterpret_cast<Engine*>(h);
    if (!e.multi) {
        set_error("vmm_ba_enable_rccl: the handle was created with world_size 1");
        return VMM_BA_ERR_STATE;
    }
    if (!rccl().ok) {
        set_error("librccl.so could not be loaded");
        return VMM_BA_ERR_COLLECTIVE;
    }
    HIP_TRY(hipSetDevice(e.device));
    if (e.rccl_comm) {
        (void)rccl().CommDestroy(reinterpret_cast<ncclComm_t>(e.rccl_comm));
        e.rccl_comm = nullptr;
    }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    const ncclResult_t r = rccl().CommInitRank(&comm, e.world, id, e.rank);
    if (r != ncclSuccess) {
        set_error(std::string("ncclCommInitRank: ") + rccl().GetErrorString(r));
        return VMM_BA_ERR_COLLECTIVE;
    }
    e.rccl_comm = comm;
    drop_graphs(e);
    return VMM_BA_OK;
}

int vmm_ba_set_state(vmm_ba_handle h, const double* cam_qt, const double* tag_qt)
{
    if (!h) {
        set_error("null handle");
        return VMM_BA_ERR_ARGUMENT;
    }
    Engine& e = *reinterpret_cast<Engine*>(h);
    HIP_TRY(hipSetDevice(e.device));
    std::vector<double> pairs;
    if (tag_qt && e.points) {   // tag poses -> their corners, as at create
        pairs = pairs_from_tags(tag_qt, e.user_tag_wh.data(), e.n_tags_user);
        tag_qt = pairs.data();
    }
    // Staged in pinned memory only: the next call that needs the poses on the device uploads them (vmm_ba_solve inside
    // the launch that starts its loop), so the call costs a 39 KB memcpy on the host and no device command.
    if (e.pose_ev_pending) {
        HIP_TRY(hipEventSynchronize(e.pose_ev));   // an earlier upload has left the staging buffer
        e.pose_ev_pending = false;
    }
    if (cam_qt) {
        memcpy(e.pose_stage, cam_qt, sizeof(double) * 7 * e.n_cams);
        e.dirty_cam = true;
    }
    if (tag_qt) {
        memcpy(e.pose_stage + (size_t)7 * e.n_cams, tag_qt, sizeof(double) * 7 * e.n_tags);
        e.dirty_tag = true;
    }
    return VMM_BA_OK;
}

int vmm_ba_get_points(vmm_ba_handle h, double* points)
{
    if (!h || !points) {
        set_error("null argument");
        return VMM_BA_ERR_ARGUMENT;
    }
    Engine& e = *reinterpret_cast<Engine*>(h);
    if (!e.points) {
        set_error("vmm_ba_get_points: the handle was not created with VMM_BA_LANDMARK_POINTS");
        return VMM_BA_ERR_STATE;
    }
    HIP_TRY(hipSetDevice(e.device));
    {
        int frc;
        if ((frc = flush_state(e))) return frc;
    }
    std::vector<double> pairs((size_t)7 * e.n_tags);
    HIP_TRY(hipMemcpyAsync(pairs.data(), e.tag_qt, sizeof(double) * pairs.size(), hipMemcpyDeviceToHost, e.stream));
    HIP_TRY(hipStreamSynchronize(e.stream));
    for (int t = 0; t < e.n_tags_user; ++t)
        for (int k = 0; k < 6; ++k) {
            points[(size_t)12 * t + k] = pairs[(size_t)14 * t + k];
            points[(size_t)12 * t + 6 + k] = pairs[(size_t)14 * t + 7 + k];
        }
    return VMM_BA_OK;
}

int vmm_ba_get_state(vmm_ba_handle h, double* cam_qt, double* tag_qt)
{
    if (!h) {
        set_error("null handle");
        return VMM_BA_ERR_ARGUMENT;
    }
    Engine& e = *reinterpret_cast<Engine*>(h);
    HIP_TRY(hipSetDevice(e.device));
    {
        int frc;
        if ((frc = flush_state(e))) return frc;
    }
    if (cam_qt)
        HIP_TRY(hipMemcpyAsync(cam_qt, e.cam_qt, sizeof(double) * 7 * e.n_cams, hipMemcpyDeviceToHost, e.stream));
    if (tag_qt && e.points) {   // tag poses rebuilt from the optimised corners (src/TagReconstructor.cpp:608-639)
        std::vector<double> pts((size_t)12 * e.n_tags_user);
        HIP_TRY(hipStreamSynchronize(e.stream));
        int rc;
        if ((rc = vmm_ba_get_points(h, pts.data()))) return rc;
        for (int t = 0; t < e.n_tags_user; ++t)
            points_to_tag(pts.data() + (size_t)12 * t, tag_qt + (size_t)7 * t);
        return VMM_BA_OK;
    }
    if (tag_qt)
        HIP_TRY(hipMemcpyAsync(tag_qt, e.tag_qt, sizeof(double) * 7 * e.n_tags, hipMemcpyDeviceToHost, e.stream));
    HIP_TRY(hipStreamSynchronize(e.stream));
    return VMM_BA_OK;
}

int vmm_ba_solve(vmm_ba_handle h, const vmm_ba_options* opt, vmm_ba_summary* s)
{
    if (!h || !s) {
        set_error("null argument");
        return VMM_BA_ERR_ARGUMENT;
    }
    Engine& e = *reinterpret_cast<Engine*>(h);
    vmm_ba_options o;
    if (opt)
        o = *opt;
    else
        vmm_ba_default_options(&o);
    if (o.max_num_iterations < 0 || o.poll_interval < 0 || !(o.initial_trust_region_radius > 0.0)) {
        set_error("bad solver options");
        return VMM_BA_ERR_ARGUMENT;
    }
    HIP_TRY(hipSetDevice(e.device));
    Range solve_range("vmm_ba_solve");
    const auto t0 = std::chrono::steady_clock::now();
    vmm_ba_iteration* user_trace = s->trace;
    const int user_cap = s->trace ? s->trace_capacity : 0;
    memset(s, 0, sizeof(*s));
    s->trace = user_trace;
    s->trace_capacity = user_cap;

    // device trace buffer sized for this call
    const int need_cap = std::max(user_cap, 1);
    if (need_cap > e.trace_capacity) {
        int rc;
        drop_graphs(e);   // the captured k_control holds the old trace pointer
        if (e.trace) {
            HIP_TRY(hipStreamSynchronize(e.stream));
            e.allocs.erase(std::remove(e.allocs.begin(), e.allocs.end(), (void*)e.trace), e.allocs.end());
            (void)hipFree(e.trace);
            e.trace = nullptr;
            e.trace_capacity = 0;
        }
        if ((rc = dev_alloc(e, &e.trace, (size_t)need_cap, false))) return rc;
        e.trace_capacity = need_cap;
    }
    {
        int rc;
        if ((rc = begin_lm_loop(e, o, user_cap))) return rc;
    }

    const int poll = std::max(1, o.poll_interval);
    const int64_t max_steps = (int64_t)o.max_num_iterations + 2;
    int64_t enq = 0;
    for (;;) {
        int rc;
        for (int k = 0; k < poll && enq < max_steps; ++k) {
            if ((rc = run_iteration(e, o))) return rc;
            enq += e.last_passes;
        }
        HIP_TRY(hipMemcpyAsync(e.ctl_host, e.ctl, sizeof(LmCtl), hipMemcpyDeviceToHost, e.stream));
        HIP_TRY(hipStreamSynchronize(e.stream));
        if (e.ctl_host->done == 2) {
            // a spin give-up, not a numerical failure: finish that pass on the fallback path and go on
            if (e.ctl_host->num_sync_timeouts >= 1000) {
                set_error("the one-launch factorisation gave up waiting in 1000 passes of one solve (kernels: "
                          + std::to_string(e.ctl_host->sync_kernels | e.ctl_host->sync_timeout) + ")");
                return VMM_BA_ERR_HIP;
            }
            if ((rc = recover_sync_timeout(e, o))) return rc;
            enq = e.ctl_host->num_lm_iterations;   // the passes enqueued behind the paused one did nothing
            continue;
        }
        if (e.ctl_host->done)
            break;
        if (enq >= max_steps) {
            set_error("LM loop did not terminate within max_num_iterations + 2 passes");
            return VMM_BA_ERR_STATE;
        }
    }
    const LmCtl& c = *e.ctl_host;
    s->termination_type = c.termination;
    s->iterations = c.records;
    s->num_successful_steps = c.num_successful;
    s->num_unsuccessful_steps = c.num_unsuccessful;
    s->num_lm_iterations = c.num_lm_iterations;
    s->num_jacobian_evals = c.num_jac_evals;
    s->num_cost_evals = c.num_cost_evals;
    s->elimination = e.elim_cams ? VMM_BA_ELIM_CAMERAS : VMM_BA_ELIM_TAGS;
    s->initial_cost = c.initial_cost;
    s->final_cost = c.x_cost;
    s->time_eval_s = 1e-8 * (double)c.phase_ticks[0];          // 100 MHz ticks
    s->time_control_s = 1e-8 * (double)c.phase_ticks[1];
    s->time_eliminate_s = 1e-8 * (double)c.phase_ticks[2];
    s->time_factor_solve_s = 1e-8 * (double)c.phase_ticks[3];
    s->time_step_s = 1e-8 * (double)c.phase_ticks[4];
    s->num_sync_timeouts = c.num_sync_timeouts;
    s->sync_timeout_kernels = c.sync_kernels;
    s->block_sparse = e.sparse_schur ? 1 : 0;
    s->tree_ordering = (e.sparse_schur && !e.h_row_of.empty()) ? (int32_t)e.nd_node_first_blk.size() : 0;
    if (user_trace && user_cap > 0) {
        const int n = std::min(c.records, user_cap);
        if (n > 0) {
            HIP_TRY(hipMemcpyAsync(user_trace, e.trace, sizeof(vmm_ba_iteration) * n, hipMemcpyDeviceToHost, e.stream));
            HIP_TRY(hipStreamSynchronize(e.stream));
        }
    }
    s->time_solve_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return VMM_BA_OK;
}

int vmm_ba_cost(vmm_ba_handle h, int robustify, double huber_a, double* cost)
{
    if (!h || !cost) {
        set_error("null argument");
        return VMM_BA_ERR_ARGUMENT;
    }
    Engine& e = *reinterpret_cast<Engine*>(h);
    HIP_TRY(hipSetDevice(e.device));
    {
        int frc;
        if ((frc = flush_state(e))) return frc;
    }
    launch_cost(e, e.cam_qt, e.tag_qt, false, robustify, huber_a, e.cost_comm);
    int rc;
    if ((rc = do_allreduce(e, e.cost_comm, 1))) return rc;
    HIP_TRY(hipMemcpyAsync(cost, e.cost_comm, sizeof(double), hipMemcpyDeviceToHost, e.stream));
    HIP_TRY(hipStreamSynchronize(e.stream));
    return VMM_BA_OK;
}

int vmm_ba_set_observation_mask(vmm_ba_handle h, const uint8_t* mask)
{
    if (!h) {
        set_error("null handle");
        return VMM_BA_ERR_ARGUMENT;
    }
    Engine& e = *reinterpret_cast<Engine*>(h);
    HIP_TRY(hipSetDevice(e.device));
    if (e.n_obs == 0)
        return VMM_BA_OK;
    if (mask) {
        std::vector<uint8_t> m((size_t)e.n_obs);
        const int rep = e.points ? 2 : 1;   // a tag observation is two point-pair observations
        for (int64_t i = 0; i < e.n_obs; ++i)
            m[(size_t)i] = mask[i / rep] ? 1 : 0;
        HIP_TRY(hipMemcpyAsync(e.obs_mask, m.data(), m.size(), hipMemcpyHostToDevice, e.stream));
        HIP_TRY(hipStreamSynchronize(e.stream));   // the staging vector goes out of scope
    } else {
        HIP_TRY(hipMemsetAsync(e.obs_mask, 1, (size_t)e.n_obs, e.stream));
    }
    return VMM_BA_OK;
}

int vmm_ba_reprojection_stats(vmm_ba_handle h, double* per_cam_mean, double* per_tag_mean, double* avg,
                              double* per_corner)
{
    if (!h) {
        set_error("null handle");
        return VMM_BA_ERR_ARGUMENT;
    }
    Engine& e = *reinterpret_cast<Engine*>(h);
    if (e.points) {
        set_error("reprojection_stats works on tag poses: read them with vmm_ba_get_state and use a tag-pose handle");
        return VMM_BA_ERR_STATE;
    }
    HIP_TRY(hipSetDevice(e.device));
    {
        int frc;
        if ((frc = flush_state(e))) return frc;
    }
    const int n_pose = e.n_cams + e.n_tags;
    if (per_corner && e.n_obs > 0 && !e.stats_corner) {
        int rc;
        if ((rc = dev_alloc(e, &e.stats_corner, (size_t)8 * e.n_obs, false))) return rc;
    }
    launch_stats(e, (per_corner && e.n_obs > 0) ? e.stats_corner : nullptr);
    HIP_TRY(hipGetLastError());
    // per-pose sums and counts come back in one copy; the means and the average are formed here in the
    // reference's order (per tag, ascending: src/TagReconstructor.cpp:416-426)
    std::vector<double> pose((size_t)2 * n_pose);
    HIP_TRY(hipMemcpyAsync(pose.data(), e.stats_pose, sizeof(double) * pose.size(), hipMemcpyDeviceToHost, e.stream));
    if (per_corner && e.n_obs > 0)
        HIP_TRY(hipMemcpyAsync(per_corner, e.stats_corner, sizeof(double) * 8 * e.n_obs, hipMemcpyDeviceToHost, e.stream));
    HIP_TRY(hipStreamSynchronize(e.stream));
    const double* sc = pose.data();
    const double* st = sc + e.n_cams;
    const double* nc = sc + n_pose;
    const double* nt = nc + e.n_cams;
    if (per_cam_mean)
        for (int c = 0; c < e.n_cams; ++c)
            per_cam_mean[c] = nc[c] > 0.0 ? sc[c] / nc[c] : -1.0;   // :371-383
    double a = 0.0, tot = 0.0;
    for (int t = 0; t < e.n_tags; ++t) {
        if (nt[t] > 0.0) {
            a += st[t];
            tot += nt[t];
        }
        if (per_tag_mean)
            per_tag_mean[t] = nt[t] > 0.0 ? st[t] / nt[t] : NAN;
    }
    if (avg)
        *avg = tot > 0.0 ? a / tot : 0.0;                            // :416-426
    return VMM_BA_OK;
}

int vmm_ba_tag_translation_covariance(vmm_ba_handle h, int robustify, double huber_a, double* cov)
{
    if (!h || !cov) {
        set_error("bad argument");
        return VMM_BA_ERR_ARGUMENT;
    }
    Engine& e = *reinterpret_cast<Engine*>(h);
    if (e.multi || e.points) {
        set_error("tag_translation_covariance needs a single-GPU handle with tag-pose landmarks");
        return VMM_BA_ERR_STATE;
    }
    HIP_TRY(hipSetDevice(e.device));
    {
        int frc;
        if ((frc = flush_state(e))) return frc;
    }
    if (e.n_obs == 0 || e.n_tags == 0) {
        memset(cov, 0, sizeof(double) * 9 * (size_t)e.n_tags);
        return VMM_BA_OK;
    }
    vmm_ba_options o;
    vmm_ba_default_options(&o);
    o.robustify = robustify;
    o.huber_a = huber_a;
    // the covariance kernels read Z as a dense matrix: a handle on the block-sparse path switches over for this call
    const bool was_sparse = e.sparse_schur;
    if (was_sparse) {
        int drc;
        if ((drc = ensure_dense_schur(e))) return drc;
        e.sparse_schur = false;
    }
    struct Restore {
        Engine& e;
        bool v, nz;
        ~Restore()
        {
            e.sparse_schur = v;
            e.chol_nz_on = nz;
        }
    } restore_path{ e, was_sparse, e.chol_nz_on };
    e.chol_nz_on = false;   // the dense, naturally ordered system has no block structure to follow
    // L X = B with B = I (tags kept) or Z^T (tags eliminated), then per-tag Gram blocks
    const bool identity_rhs = e.elim_cams;
    const int n_rhs = identity_rhs ? e.n_pad : e.k_dim;
    const int ldb = round_up(n_rhs, 64);
    double *B = nullptr, *cov_dev = nullptr;
    hipError_t err = hipMalloc((void**)&B, sizeof(double) * (size_t)e.n_pad * ldb);
    if (err == hipSuccess) err = hipMalloc((void**)&cov_dev, sizeof(double) * 9 * (size_t)e.n_tags);
    // second attempt only after a spin give-up of the one-launch factorisation: the same on the fallback path
    for (int attempt = 0; attempt < 2 && err == hipSuccess; ++attempt) {
        init_ctl(e, *e.ctl_host, o, 0);
        err = hipMemcpyAsync(e.ctl, e.ctl_host, sizeof(LmCtl), hipMemcpyHostToDevice, e.stream);
        if (err != hipSuccess)
            break;
        // the iteration's kernels on the undamped, unscaled system: H blocks, Z, S = L L^T (+ block inverses)
        launch_eval_passes(e, robustify, huber_a, false);
        launch_cov_prepare(e);
        launch_elim(e);
        launch_syrk_reduced(e);
        launch_cholesky_solve(e, e.S, e.n_pad, e.ldz, e.yf, e.ctl, attempt > 0);
        launch_chol_inverse(e, e.n_blk - 1);
        err = hipMemsetAsync(B, 0, sizeof(double) * (size_t)e.n_pad * ldb, e.stream);
        if (err == hipSuccess) {
            launch_cov_rhs(e, B, ldb, identity_rhs);
            launch_cov_trsm(e, B, ldb, ldb / 64, identity_rhs);
            launch_cov_gram(e, B, ldb, cov_dev);
            err = hipGetLastError();
        }
        if (err == hipSuccess)
            err = hipMemcpyAsync(cov, cov_dev, sizeof(double) * 9 * (size_t)e.n_tags, hipMemcpyDeviceToHost, e.stream);
        if (err == hipSuccess)
            err = hipMemcpyAsync(e.ctl_host, e.ctl, sizeof(LmCtl), hipMemcpyDeviceToHost, e.stream);
        if (err == hipSuccess)
            err = hipStreamSynchronize(e.stream);
        if (err != hipSuccess || e.ctl_host->done != 2)
            break;
    }
    (void)hipFree(B);
    (void)hipFree(cov_dev);
    if (err != hipSuccess) {
        set_error(std::string("tag_translation_covariance: ") + hipGetErrorString(err));
        return VMM_BA_ERR_HIP;
    }
    if (e.ctl_host->lin_fail) {
        set_error("tag_translation_covariance: J^T J is not positive definite (rank-deficient Jacobian)");
        return VMM_BA_ERR_NUMERIC;
    }
    return VMM_BA_OK;
}

constexpr int kMaxProjectDevices = 64;

int vmm_ba_project_points(const double intr[4], const double dist[5], int64_t n, const double* points_cam,
                          double* uv, int device)
{
    if (!intr || !dist || n < 0 || (n > 0 && (!points_cam || !uv))) {
        set_error("bad argument");
        return VMM_BA_ERR_ARGUMENT;
    }
    if (n == 0)
        return VMM_BA_OK;
    HIP_TRY(hipSetDevice(device));
    Intrinsics K;
    K.fx = intr[0]; K.fy = intr[1]; K.cx = intr[2]; K.cy = intr[3];
    K.k1 = dist[0]; K.k2 = dist[1]; K.p1 = dist[2]; K.p2 = dist[3]; K.k3 = dist[4];
    // CameraModel::projectPoint is called point by point by its users: the device buffer of small calls is kept per
    // device (grown on demand, up to 1 M points = 40 MB; larger calls allocate and free), calls are serialised
    static std::mutex mu;
    static double* cache[kMaxProjectDevices] = {};
    static int64_t cache_cap[kMaxProjectDevices] = {};
    std::lock_guard<std::mutex> lock(mu);
    const bool cached = n <= (int64_t)1 << 20 && device >= 0 && device < kMaxProjectDevices;
    double* buf = nullptr;
    if (cached && cache_cap[device] >= n) {
        buf = cache[device];
    } else {
        const int64_t cap = cached ? std::max<int64_t>(n, 1024) : n;
        HIP_TRY(hipMalloc((void**)&buf, sizeof(double) * 5 * (size_t)cap));
        if (cached) {
            if (cache[device])
                (void)hipFree(cache[device]);
            cache[device] = buf;
            cache_cap[device] = cap;
        }
    }
    double *d_p = buf, *d_uv = buf + 3 * n;
    hipError_t err = hipMemcpy(d_p, points_cam, sizeof(double) * 3 * n, hipMemcpyHostToDevice);
    if (err == hipSuccess) {
        launch_project(nullptr, K, n, d_p, d_uv);
        err = hipMemcpy(uv, d_uv, sizeof(double) * 2 * n, hipMemcpyDeviceToHost);
    }
    if (!cached)
        (void)hipFree(buf);
    if (err != hipSuccess) {
        set_error(std::string("project_points: ") + hipGetErrorString(err));
        return VMM_BA_ERR_HIP;
    }
    return VMM_BA_OK;
}

int vmm_ba_pose_plus(int64_t n, const double* qt, const double* delta, double* out, int device)
{
    if (n < 0 || (n > 0 && (!qt || !delta || !out))) {
        set_error("bad argument");
        return VMM_BA_ERR_ARGUMENT;
    }
    if (n == 0)
        return VMM_BA_OK;
    HIP_TRY(hipSetDevice(device));
    double* d = nullptr;   // qt | delta | out
    HIP_TRY(hipMalloc((void**)&d, sizeof(double) * 20 * n));
    hipError_t err = hipMemcpy(d, qt, sizeof(double) * 7 * n, hipMemcpyHostToDevice);
    if (err == hipSuccess)
        err = hipMemcpy(d + 7 * n, delta, sizeof(double) * 6 * n, hipMemcpyHostToDevice);
    if (err == hipSuccess) {
        launch_pose_plus(nullptr, n, d, d + 7 * n, d + 13 * n);
        err = hipMemcpy(out, d + 13 * n, sizeof(double) * 7 * n, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d);
    if (err != hipSuccess) {
        set_error(std::string("pose_plus: ") + hipGetErrorString(err));
        return VMM_BA_ERR_HIP;
    }
    return VMM_BA_OK;
}

int vmm_ba_eval_blocks(vmm_ba_handle h, int robustify, double huber_a, double* cost, double* V, double* U,
                       double* W, double* g_cam, double* g_tag)
{
    if (!h) {
        set_error("null handle");
        return VMM_BA_ERR_ARGUMENT;
    }
    Engine& e = *reinterpret_cast<Engine*>(h);
    if (e.points && (U || W || g_tag)) {
        set_error("eval_blocks: with point landmarks only cost, V and g_cam are in the caller's index space");
        return VMM_BA_ERR_STATE;
    }
    HIP_TRY(hipSetDevice(e.device));
    {
        int frc;
        if ((frc = flush_state(e))) return frc;
    }
    launch_eval_passes(e, robustify, huber_a, false);
    HIP_TRY(hipGetLastError());
    if (cost) HIP_TRY(hipMemcpyAsync(cost, e.cost_slot, sizeof(double), hipMemcpyDeviceToHost, e.stream));
    if (V) HIP_TRY(hipMemcpyAsync(V, e.H_cam, sizeof(double) * 36 * e.n_cams, hipMemcpyDeviceToHost, e.stream));
    if (U) HIP_TRY(hipMemcpyAsync(U, e.H_tag, sizeof(double) * 36 * e.n_tags, hipMemcpyDeviceToHost, e.stream));
    if (g_cam) HIP_TRY(hipMemcpyAsync(g_cam, e.g_cam, sizeof(double) * 6 * e.n_cams, hipMemcpyDeviceToHost, e.stream));
    if (g_tag) HIP_TRY(hipMemcpyAsync(g_tag, e.g_tag, sizeof(double) * 6 * e.n_tags, hipMemcpyDeviceToHost, e.stream));
    HIP_TRY(hipStreamSynchronize(e.stream));
    if (W && e.n_obs > 0) {
        std::vector<double> w((size_t)36 * e.ordE.n_pad);
        std::vector<int32_t> caller((size_t)e.n_obs);
        if (e.f32_accum) {
            std::vector<float> wf(w.size());
            HIP_TRY(hipMemcpy(wf.data(), e.Wf, sizeof(float) * wf.size(), hipMemcpyDeviceToHost));
            for (size_t q = 0; q < w.size(); ++q)
                w[q] = (double)wf[q];
        } else {
            HIP_TRY(hipMemcpy(w.data(), e.W, sizeof(double) * w.size(), hipMemcpyDeviceToHost));
        }
        HIP_TRY(hipMemcpy(caller.data(), e.ordE.caller, sizeof(int32_t) * caller.size(), hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < e.n_obs; ++i)
            for (int a = 0; a < 6; ++a)
                for (int b = 0; b < 6; ++b) {
                    // device: rows = eliminated family's tangent, columns = kept family's
                    const double v = w[(size_t)(6 * a + b) * e.ordE.n_pad + i];
                    const int ca = e.elim_cams ? a : b, tb = e.elim_cams ? b : a;
                    W[(size_t)36 * caller[i] + 6 * ca + tb] = v;
                }
    }
    return VMM_BA_OK;
}

// Minimal engine for the dense test entry points: stream + panel buffer + a control block.
static int make_scratch(Engine& e, int device, int ld)
{
    e.device = device;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&e.stream, hipStreamNonBlocking));
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            e.n_cu = prop.multiProcessorCount;
    }
    int rc;
    if ((rc = dev_alloc(e, &e.P, (size_t)4 * kNB * ld))) return rc;
    if ((rc = setup_lookahead(e, ld / kNB, ld))) return rc;
    if ((rc = dev_alloc(e, &e.dinv, (size_t)ld + kNB))) return rc;
    if ((rc = dev_alloc(e, &e.Ldiag, (size_t)(ld / kNB + 1) * 4096))) return rc;
    if ((rc = dev_alloc(e, &e.Linv, (size_t)(ld / kNB + 1) * 4096))) return rc;
    if ((rc = dev_alloc(e, &e.flags, 264))) return rc;
    if ((rc = dev_alloc(e, &e.gran, (size_t)2 * ld))) return rc;
    { const char* nc = getenv("VMM_BA_NO_CHAIN"); e.no_chain = nc && nc[0] == '1'; }
    { const char* nd = getenv("VMM_BA_NO_DATAFLOW"); e.no_dataflow = nd && nd[0] == '1'; }
    read_spin_debug_env(e);
    {
        const int nb = ld / kNB - 1;   // ld = n_pad + 64
        const size_t nd = e.no_dataflow ? 0 : (size_t)dataflow_blocks(nb, e.n_cu);
        if (nd > 0 && (rc = dev_alloc(e, &e.df_gran, nd * (nd + 1) / 2 * 8 * 1024)))
            return rc;
        if (nd > 0 && (rc = dev_alloc(e, &e.df_compact, nd * (nd + 1) / 2 * 4096, false)))
            return rc;
        if (nd > 0 && (rc = dev_alloc(e, &e.df_done, nd * (nd + 1) / 2)))
            return rc;
    }
    if ((rc = dev_alloc(e, &e.ctl, 1))) return rc;
    return VMM_BA_OK;
}

static void free_scratch(Engine& e)
{
    if (e.stream)
        (void)hipStreamSynchronize(e.stream);
    for (void* p : e.allocs)
        (void)hipFree(p);
    if (e.stream)
        (void)hipStreamDestroy(e.stream);
    e.allocs.clear();
    e.stream = nullptr;
}

int vmm_ba_dense_spd_solve(int device, int n, const double* A, const double* b, double* x, int* info)
{
    if (n <= 0 || !A || !b || !x) {
        set_error("bad argument");
        return VMM_BA_ERR_ARGUMENT;
    }
    const int n_pad = round_up(n, kNB), ld = n_pad + kNB;
    Engine e;
    int rc = make_scratch(e, device, ld);
    double *S = nullptr, *y = nullptr;
    if (!rc) rc = dev_alloc(e, &S, (size_t)ld * ld);
    if (!rc) rc = dev_alloc(e, &y, (size_t)ld);
    if (rc) {
        free_scratch(e);
        return rc;
    }
    std::vector<double> hs((size_t)ld * ld, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j)
            hs[(size_t)i * ld + j] = A[(size_t)i * n + j];
    for (int i = n; i < n_pad; ++i)
        hs[(size_t)i * ld + i] = 1.0;
    for (int j = 0; j < n; ++j)
        hs[(size_t)n_pad * ld + j] = b[j];
    LmCtl c;
    hipError_t err = hipSuccess;
    // second attempt only after a spin give-up of the one-launch kernels: the same system on the fallback path
    for (int attempt = 0; attempt < 2 && err == hipSuccess; ++attempt) {
        memset(&c, 0, sizeof(c));
        if (attempt == 0) {
            c.spin_limit_df = e.dbg_spin_df;
            c.spin_limit_chain = e.dbg_spin_chain;
            c.spin_wg = e.dbg_spin_wg;
        }
        err = hipMemcpyAsync(e.ctl, &c, sizeof(LmCtl), hipMemcpyHostToDevice, e.stream);
        if (err == hipSuccess)
            err = hipMemcpyAsync(S, hs.data(), sizeof(double) * hs.size(), hipMemcpyHostToDevice, e.stream);
        if (err == hipSuccess) {
            launch_cholesky_solve(e, S, n_pad, ld, y, e.ctl, attempt > 0);
            err = hipGetLastError();
        }
        if (err == hipSuccess) err = hipMemcpyAsync(x, y, sizeof(double) * n, hipMemcpyDeviceToHost, e.stream);
        if (err == hipSuccess) err = hipMemcpyAsync(&c, e.ctl, sizeof(LmCtl), hipMemcpyDeviceToHost, e.stream);
        if (err == hipSuccess) err = hipStreamSynchronize(e.stream);
        if (c.done != 2)
            break;
    }
    free_scratch(e);
    if (err != hipSuccess) {
        set_error(std::string("dense_spd_solve: ") + hipGetErrorString(err));
        return VMM_BA_ERR_HIP;
    }
    if (info)
        *info = c.lin_fail;
    return VMM_BA_OK;
}

int vmm_ba_dense_syrk(int device, int k, int n, const double* Zh, double* C)
{
    if (k <= 0 || n <= 0 || !Zh || !C) {
        set_error("bad argument");
        return VMM_BA_ERR_ARGUMENT;
    }
    const int n_pad = round_up(n, kST), ld = n_pad;
    const int n_blk = n_pad / kST;
    const int k_pad = round_up(k, kKT);
    Engine e;
    int rc = make_scratch(e, device, ld);
    double *Z = nullptr, *S = nullptr;
    SyrkPlan plan;
    if (!rc) rc = dev_alloc(e, &Z, (size_t)k_pad * ld);
    if (!rc) rc = dev_alloc(e, &S, (size_t)ld * ld);
    if (!rc) rc = make_syrk_plan(e, plan, n_blk, n_blk, k_pad);
    if (rc) {
        free_scratch(e);
        return rc;
    }
    std::vector<double> hz((size_t)k_pad * ld, 0.0);
    for (int r = 0; r < k; ++r)
        for (int c = 0; c < n; ++c)
            hz[(size_t)r * ld + c] = Zh[(size_t)r * n + c];
    hipError_t err = hipMemcpyAsync(Z, hz.data(), sizeof(double) * hz.size(), hipMemcpyHostToDevice, e.stream);
    std::vector<double> hs((size_t)ld * ld, 0.0);
    if (err == hipSuccess) {
        launch_syrk_plan(e.stream, nullptr, Z, ld, plan);
        launch_reduce_plan(e.stream, nullptr, plan, ld, n_pad, S);
        err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipMemcpyAsync(hs.data(), S, sizeof(double) * hs.size(), hipMemcpyDeviceToHost, e.stream);
    if (err == hipSuccess) err = hipStreamSynchronize(e.stream);
    free_scratch(e);
    if (err != hipSuccess) {
        set_error(std::string("dense_syrk: ") + hipGetErrorString(err));
        return VMM_BA_ERR_HIP;
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            C[(size_t)i * n + j] = (j <= i) ? -hs[(size_t)i * ld + j] : -hs[(size_t)j * ld + i];
    return VMM_BA_OK;
}

int vmm_ba_time_kernels(vmm_ba_handle h, const vmm_ba_options* opt, int reps, vmm_ba_kernel_times* out)
{
    if (!h || !out || reps <= 0) {
        set_error("bad argument");
        return VMM_BA_ERR_ARGUMENT;
    }
    Engine& e = *reinterpret_cast<Engine*>(h);
    if (e.multi) {
        set_error("time_kernels is a single-GPU diagnostic");
        return VMM_BA_ERR_STATE;
    }
    vmm_ba_options o;
    if (opt)
        o = *opt;
    else
        vmm_ba_default_options(&o);
    HIP_TRY(hipSetDevice(e.device));
    {
        int frc;
        if ((frc = flush_state(e))) return frc;
    }
    memset(out, 0, sizeof(*out));
    out->n_obs = e.n_obs;
    out->reduced_dim = e.n_red;
    out->elim_dim = e.k_dim;
    out->schur_sparse = e.sparse_schur ? 1 : 0;
    out->syrk_wide = (!e.sparse_schur && e.syrk.wide) ? 1 : 0;
    out->schur_flops = e.schur_flops;
    out->chol_flops = e.chol_nz_on ? e.chol_flops : 0.0;
    // keep the caller's state: timing runs real iterations.  The RAW device buffers are saved and restored: through
    // vmm_ba_get_state / vmm_ba_set_state a point-landmark handle would get its tags back as exact rectangles rebuilt
    // from re-orthogonalised poses, not the optimised free corners it held.
    std::vector<double> cam0((size_t)7 * e.n_cams), tag0((size_t)7 * e.n_tags);
    int rc;
    HIP_TRY(hipMemcpyAsync(cam0.data(), e.cam_qt, sizeof(double) * cam0.size(), hipMemcpyDeviceToHost, e.stream));
    HIP_TRY(hipMemcpyAsync(tag0.data(), e.tag_qt, sizeof(double) * tag0.size(), hipMemcpyDeviceToHost, e.stream));
    HIP_TRY(hipStreamSynchronize(e.stream));
    auto restore_raw = [&]() -> int {
        HIP_TRY(hipMemcpyAsync(e.cam_qt, cam0.data(), sizeof(double) * cam0.size(), hipMemcpyHostToDevice, e.stream));
        HIP_TRY(hipMemcpyAsync(e.tag_qt, tag0.data(), sizeof(double) * tag0.size(), hipMemcpyHostToDevice, e.stream));
        HIP_TRY(hipStreamSynchronize(e.stream));
        e.dirty_cam = e.dirty_tag = false;
        return VMM_BA_OK;
    };
    if (e.trace_capacity < 1) {
        if ((rc = dev_alloc(e, &e.trace, 1, false))) return rc;
        e.trace_capacity = 1;
        drop_graphs(e);
    }
    vmm_ba_options ot = o;
    ot.max_num_iterations = 1 << 30;
    ot.function_tolerance = 0.0;
    ot.parameter_tolerance = 0.0;
    ot.gradient_tolerance = 0.0;
    if ((rc = begin_lm_loop(e, ot, 0))) return rc;
    if ((rc = enqueue_iteration(e, ot))) return rc;   // populates every buffer of an iteration
    HIP_TRY(hipStreamSynchronize(e.stream));
    // the priming step was accepted and moved x; go back so that the W recomputed by the timed
    // evaluation passes stays consistent with the H blocks of the priming evaluation
    if ((rc = restore_raw())) return rc;   // the timed pieces read the poses on the device
    hipEvent_t ev0, ev1;
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
    // each timed piece is captured into a hipGraph once and replayed, so that the gaps between its
    // launches are the ones the production iteration graph sees
    // `inner` > 1 (idempotent pieces only): the piece is captured that many times back to back and the
    // graph time divided by it, which takes the ~9 us of graph-launch overhead out of a 5-30 us kernel
    // (rocprofv3's per-kernel durations are the reference the bench line must agree with).
    auto timed = [&](auto&& fn, auto&& prep, double* ms_out, int inner = 1) -> int {
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        HIP_TRY(hipStreamBeginCapture(e.stream, hipStreamCaptureModeThreadLocal));
        for (int q = 0; q < inner; ++q)
            fn();
        HIP_TRY(hipStreamEndCapture(e.stream, &g));
        HIP_TRY(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        (void)hipGraphDestroy(g);
        double total = 0.0;
        for (int r = 0; r < reps + 1; ++r) {   // replay 0 is untimed
            prep();
            HIP_TRY(hipEventRecord(ev0, e.stream));
            HIP_TRY(hipGraphLaunch(ge, e.stream));
            HIP_TRY(hipEventRecord(ev1, e.stream));
            HIP_TRY(hipEventSynchronize(ev1));
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
            if (r > 0) total += ms;
        }
        (void)hipGraphExecDestroy(ge);
        *ms_out = total / reps / inner;
        return VMM_BA_OK;
    };
    auto nop = [] {};
    // the guards read ctl->done / lin_fail only; both are 0 after the priming iteration unless it failed
    HIP_TRY(hipMemcpy(e.ctl_host, e.ctl, sizeof(LmCtl), hipMemcpyDeviceToHost));
    e.ctl_host->done = 0;
    e.ctl_host->lin_fail = 0;
    HIP_TRY(hipMemcpy(e.ctl, e.ctl_host, sizeof(LmCtl), hipMemcpyHostToDevice));
    if ((rc = timed([&] { launch_eval_passes(e, o.robustify, o.huber_a, false); }, nop, &out->eval_elim_ms, 8))) return rc;
    out->eval_keep_ms = 0.0;
    if ((rc = timed([&] { launch_cost_kernel(e, e.cam_qt, e.tag_qt, false, o.robustify, o.huber_a); }, nop, &out->cost_ms, 8))) return rc;
    if ((rc = timed([&] { launch_elim(e); }, nop, &out->form_z_ms, 8))) return rc;
    if ((rc = timed([&] { launch_syrk_only(e); }, nop, &out->syrk_ms, 4))) return rc;
    if ((rc = timed([&] { launch_cholesky_solve(e, e.S, e.n_pad, e.ldz, e.yf, e.ctl); },
                    [&] {
                        launch_syrk_reduced(e);
                        launch_pack_lower(e, false);   // world > 1: this rank's share alone (no all-reduce here)
                        launch_pack_lower(e, true);
                    },
                    &out->cholesky_ms)))
        return rc;
    if ((rc = timed([&] { launch_backsub(e); }, nop, &out->backsub_ms, 8))) return rc;
    if (getenv("VMM_BA_DEBUG")) {
        HIP_TRY(hipMemcpy(e.ctl_host, e.ctl, sizeof(LmCtl), hipMemcpyDeviceToHost));
        fprintf(stderr, "[vmm_ba debug] after kernel timing: done=%d lin_fail=%d termination=%d iteration=%d\n",
                e.ctl_host->done, e.ctl_host->lin_fail, e.ctl_host->termination, e.ctl_host->iteration);
    }
    // whole iterations from the caller's state
    if ((rc = restore_raw())) return rc;
    if ((rc = begin_lm_loop(e, ot, 0))) return rc;
    HIP_TRY(hipEventRecord(ev0, e.stream));
    int passes = 0;
    for (int r = 0; r < reps; ++r) {
        if ((rc = run_iteration(e, ot))) return rc;
        passes += e.last_passes;
    }
    HIP_TRY(hipEventRecord(ev1, e.stream));
    HIP_TRY(hipEventSynchronize(ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    out->lm_iteration_ms = ms / passes;
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    return restore_raw();
}

// Diagnostic behind DESIGN.md's question "can the reduced-system assembly hide behind the factorisation?": times, with
// HIP events, (0) rank-k update + partial-tile sum alone, (1) factorisation + triangular solves alone, (2) both back to
// back on one stream (today's order), (3) both at once on two streams -- the factorisation on a valid S, the rank-k
// update of the same Z writing its sum into a scratch matrix, so that only the sharing of the chip is measured, not a
// dependency; (4) the factorisation's own duration inside (3).  Dense elimination, one GPU.  ms[5] averages over reps.
int vmm_ba_debug_chol_schedule(int n_blk, int n_df, int n_cu, int32_t* launches, int cap, int* n_launches, int* n_df_used)
{
    if (n_blk < 1 || n_cu < 1 || cap < 0 || (cap > 0 && !launches) || !n_launches || (n_df > 0 && n_df > n_blk - 2)) {
        set_error("bad argument");
        return VMM_BA_ERR_ARGUMENT;
    }
    if (n_df < 0) {
        n_df = dataflow_blocks(n_blk, n_cu);
        if (n_df == n_blk)
            n_df = 0;   // the one-launch kernel takes the whole system: what is returned is the fallback schedule
    }
    const std::vector<CholLaunch> sched = chol_step_schedule(n_blk, n_df);
    for (size_t i = 0; i < sched.size() && (int)i < cap; ++i) {
        const CholLaunch& L = sched[i];
        const int32_t row[8] = { L.k, L.lazy[0], L.lazy[1], L.upd[0], L.upd[1], L.c0, L.t0, L.t1 };
        memcpy(launches + 8 * i, row, sizeof(row));
    }
    *n_launches = (int)sched.size();
    if (n_df_used)
        *n_df_used = n_df;
    return VMM_BA_OK;
}

int vmm_ba_debug_chol_tile(int n_blk, const int32_t* launch, int t, int* bi, int* bj)
{
    if (!launch || !bi || !bj || t < 0 || t >= launch[7]) {
        set_error("bad argument");
        return VMM_BA_ERR_ARGUMENT;
    }
    const CholLaunch L = { launch[0], { launch[1], launch[2] }, { launch[3], launch[4] }, launch[5], launch[6], launch[7] };
    chol_schedule_tile(n_blk, L, t, bi, bj);
    return VMM_BA_OK;
}

int vmm_ba_debug_overlap(vmm_ba_handle h, int reps, double* ms)
{
    if (!h || !ms || reps <= 0) {
        set_error("bad argument");
        return VMM_BA_ERR_ARGUMENT;
    }
    Engine& e = *reinterpret_cast<Engine*>(h);
    if (e.multi || e.sparse_schur || !e.Z) {
        set_error("debug_overlap needs a single-GPU handle on the dense elimination path");
        return VMM_BA_ERR_STATE;
    }
    HIP_TRY(hipSetDevice(e.device));
    int rc;
    if ((rc = flush_state(e))) return rc;
    vmm_ba_options o;
    vmm_ba_default_options(&o);
    o.max_num_iterations = 1 << 30;
    o.function_tolerance = o.parameter_tolerance = o.gradient_tolerance = 0.0;
    if (e.trace_capacity < 1) {
        if ((rc = dev_alloc(e, &e.trace, 1, false))) return rc;
        e.trace_capacity = 1;
        drop_graphs(e);
    }
    if ((rc = begin_lm_loop(e, o, 0))) return rc;
    if ((rc = enqueue_iteration(e, o))) return rc;   // populates Z, the small blocks and the control block
    HIP_TRY(hipStreamSynchronize(e.stream));
    double* S2 = nullptr;
    hipStream_t sb = nullptr;
    hipEvent_t ev[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };
    hipError_t err = hipMalloc((void**)&S2, sizeof(double) * (size_t)e.ldz * e.ldz);
    if (err == hipSuccess) err = hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    for (int i = 0; i < 5 && err == hipSuccess; ++i)
        err = hipEventCreate(&ev[i]);
    auto reset_flags = [&]() -> hipError_t {
        hipError_t r = hipMemcpy(e.ctl_host, e.ctl, sizeof(LmCtl), hipMemcpyDeviceToHost);
        if (r != hipSuccess) return r;
        e.ctl_host->done = 0;
        e.ctl_host->lin_fail = 0;
        e.ctl_host->sync_timeout = 0;
        return hipMemcpy(e.ctl, e.ctl_host, sizeof(LmCtl), hipMemcpyHostToDevice);
    };
    double acc[8] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
    for (int r = 0; r < reps + 1 && err == hipSuccess; ++r) {   // repetition 0 is untimed
        float t[8] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
        // (0) rank-k update + sum (leaves a valid S), (1) factorisation + solves
        if ((err = reset_flags()) != hipSuccess) break;
        (void)hipEventRecord(ev[0], e.stream);
        launch_syrk_reduced(e);
        (void)hipEventRecord(ev[1], e.stream);
        launch_cholesky_solve(e, e.S, e.n_pad, e.ldz, e.yf, e.ctl);
        (void)hipEventRecord(ev[2], e.stream);
        if ((err = hipEventSynchronize(ev[2])) != hipSuccess) break;
        (void)hipEventElapsedTime(&t[0], ev[0], ev[1]);
        (void)hipEventElapsedTime(&t[1], ev[1], ev[2]);
        (void)hipEventElapsedTime(&t[2], ev[0], ev[2]);
        // (3) both at once: S rebuilt first (untimed), then the factorisation beside a second rank-k update into S2
        if ((err = reset_flags()) != hipSuccess) break;
        launch_syrk_reduced(e);
        if ((err = hipStreamSynchronize(e.stream)) != hipSuccess) break;
        // the factorisation is enqueued first: its workgroups (84 KB of LDS, one per CU) take their CUs, the rank-k
        // update's (72 KB) fill what is left beside them
        (void)hipEventRecord(ev[0], e.stream);
        (void)hipStreamWaitEvent(sb, ev[0], 0);
        launch_cholesky_solve(e, e.S, e.n_pad, e.ldz, e.yf, e.ctl);
        (void)hipEventRecord(ev[3], e.stream);
        launch_syrk_plan(sb, e.ctl, e.Z, e.ldz, e.syrk);
        launch_reduce_plan(sb, e.ctl, e.syrk, e.ldz, e.n_pad + 1, S2);
        (void)hipEventRecord(ev[1], sb);
        (void)hipStreamWaitEvent(e.stream, ev[1], 0);
        (void)hipEventRecord(ev[2], e.stream);
        if ((err = hipEventSynchronize(ev[2])) != hipSuccess) break;
        (void)hipEventElapsedTime(&t[3], ev[0], ev[2]);
        (void)hipEventElapsedTime(&t[4], ev[0], ev[3]);
        (void)hipEventElapsedTime(&t[5], ev[0], ev[1]);
        // (6) the same with the rank-k update enqueued FIRST (its 495 workgroups take their slots, the factorisation's
        // workgroups follow as slots fall free)
        if ((err = reset_flags()) != hipSuccess) break;
        launch_syrk_reduced(e);
        if ((err = hipStreamSynchronize(e.stream)) != hipSuccess) break;
        (void)hipEventRecord(ev[0], e.stream);
        (void)hipStreamWaitEvent(sb, ev[0], 0);
        launch_syrk_plan(sb, e.ctl, e.Z, e.ldz, e.syrk);
        launch_reduce_plan(sb, e.ctl, e.syrk, e.ldz, e.n_pad + 1, S2);
        (void)hipEventRecord(ev[1], sb);
        (void)hipEventRecord(ev[4], e.stream);
        launch_cholesky_solve(e, e.S, e.n_pad, e.ldz, e.yf, e.ctl);
        (void)hipEventRecord(ev[3], e.stream);
        (void)hipStreamWaitEvent(e.stream, ev[1], 0);
        (void)hipEventRecord(ev[2], e.stream);
        if ((err = hipEventSynchronize(ev[2])) != hipSuccess) break;
        (void)hipEventElapsedTime(&t[6], ev[0], ev[2]);
        (void)hipEventElapsedTime(&t[7], ev[4], ev[3]);
        if (r > 0)
            for (int i = 0; i < 8; ++i)
                acc[i] += t[i];
    }
    if (err == hipSuccess)
        err = hipGetLastError();
    for (auto& x : ev)
        if (x) (void)hipEventDestroy(x);
    if (sb) (void)hipStreamDestroy(sb);
    (void)hipFree(S2);
    if (err != hipSuccess) {
        set_error(std::string("debug_overlap: ") + hipGetErrorString(err));
        return VMM_BA_ERR_HIP;
    }
    for (int i = 0; i < 8; ++i)
        ms[i] = acc[i] / reps;
    return VMM_BA_OK;
}

} // extern "C"
