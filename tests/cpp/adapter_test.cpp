// Drives include/vmm_ba_adapter.hpp with POD stand-ins that spell their members like the reference's
// Eigen-based types (q(i), t(i), corners[k].x(), distortionCoefficients(i,0)); this image has no Eigen.
// Reads a flat scene from stdin, runs doBundleAdjustment + statistics on the GPU, prints the result.
// Usage: adapter_test < scene.txt   (see tests/test_gpu_cpp_adapter.py)
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <iterator>
#include <iostream>
#include <map>
#include <vector>

#include "vmm_ba_adapter.hpp"

struct Vec4 { double v[4]; double& operator()(int i) { return v[i]; } double operator()(int i) const { return v[i]; } };
struct Vec3 { double v[3]; double& operator()(int i) { return v[i]; } double operator()(int i) const { return v[i]; } };
struct Vec2 { double v[2]; double x() const { return v[0]; } double y() const { return v[1]; } };
struct Vec5 { double v[5]; double operator()(int i, int) const { return v[i]; } };
struct Camera { int cameraId = -1; Vec4 q{{1, 0, 0, 0}}; Vec3 t{{0, 0, 0}}; };
struct ReconstructedTag { int id = -1; std::string tagType; Vec4 q{{1, 0, 0, 0}}; Vec3 t{{0, 0, 0}}; double tagWidth = 0, tagHeight = 0; };
struct TagObservation { int imageId = -1, tagId = -1; std::vector<Vec2> corners; };
struct TagImg { int imageId = -1; };
struct Tag { int tagId = -1; double width = 0, height = 0; };
struct DetectionResult { std::vector<TagImg> images; std::vector<Tag> tags; std::vector<TagObservation> tagObservations; };
struct CameraModel { double fx, fy, cx, cy; Vec5 distortionCoefficients; int verticalResolution, horizontalResolution; };

int main()
{
    CameraModel cm{};
    int nc, nt, no, origin;
    if (scanf("%lf %lf %lf %lf", &cm.fx, &cm.fy, &cm.cx, &cm.cy) != 4) return 2;
    for (double& d : cm.distortionCoefficients.v) if (scanf("%lf", &d) != 1) return 2;
    if (scanf("%d %d %d %d", &nc, &nt, &no, &origin) != 4) return 2;
    std::map<int, Camera> cams;
    std::map<int, ReconstructedTag> tags;
    for (int i = 0; i < nc; ++i) {
        Camera c; if (scanf("%d", &c.cameraId) != 1) return 2;
        for (double& d : c.q.v) if (scanf("%lf", &d) != 1) return 2;
        for (double& d : c.t.v) if (scanf("%lf", &d) != 1) return 2;
        cams[c.cameraId] = c;
    }
    for (int i = 0; i < nt; ++i) {
        ReconstructedTag t; if (scanf("%d", &t.id) != 1) return 2;
        for (double& d : t.q.v) if (scanf("%lf", &d) != 1) return 2;
        for (double& d : t.t.v) if (scanf("%lf", &d) != 1) return 2;
        if (scanf("%lf %lf", &t.tagWidth, &t.tagHeight) != 2) return 2;
        tags[t.id] = t;
    }
    DetectionResult det;
    for (int i = 0; i < no; ++i) {
        TagObservation ob; ob.corners.resize(4);
        if (scanf("%d %d", &ob.imageId, &ob.tagId) != 2) return 2;
        for (auto& c : ob.corners) if (scanf("%lf %lf", &c.v[0], &c.v[1]) != 2) return 2;
        det.tagObservations.push_back(ob);
    }
    for (const auto& kv : cams) { TagImg im; im.imageId = kv.first; det.images.push_back(im); }
    for (const auto& kv : tags) { Tag t; t.tagId = kv.first; t.width = kv.second.tagWidth; t.height = kv.second.tagHeight; det.tags.push_back(t); }
    const std::map<int, Camera> cams0 = cams;
    const std::map<int, ReconstructedTag> tags0 = tags;
    try {
        {
            // device-resident handle (one vmm_ba_create for the whole detection set, masks per step) against the
            // free functions on a sub-problem: the last camera and the last tag are not reconstructed yet
            std::map<int, Camera> ca = cams0, cb = cams0;
            std::map<int, ReconstructedTag> ta = tags0, tb = tags0;
            ca.erase(std::prev(ca.end())); cb.erase(std::prev(cb.end()));
            ta.erase(std::prev(ta.end())); tb.erase(std::prev(tb.end()));
            vmm_ba_adapter::Resident<DetectionResult, CameraModel> res(det, cm);
            res.doBundleAdjustment(ta, ca, origin, 400, 1, true);
            vmm_ba_adapter::doBundleAdjustment(tb, cb, det, cm, origin, 400, 1, true, false);
            double md = 0.0;
            for (const auto& kv : ca) { for (int i = 0; i < 4; ++i) md = std::max(md, std::fabs(kv.second.q(i) - cb.at(kv.first).q(i))); for (int i = 0; i < 3; ++i) md = std::max(md, std::fabs(kv.second.t(i) - cb.at(kv.first).t(i))); }
            for (const auto& kv : ta) { for (int i = 0; i < 4; ++i) md = std::max(md, std::fabs(kv.second.q(i) - tb.at(kv.first).q(i))); for (int i = 0; i < 3; ++i) md = std::max(md, std::fabs(kv.second.t(i) - tb.at(kv.first).t(i))); }
            const auto sa = res.reprojectionStatistics(ta, ca, origin, true);
            const auto sb = vmm_ba_adapter::reprojectionStatistics(tb, cb, det, cm, true);
            double sd = std::fabs(sa.avg - sb.avg);
            for (const auto& kv : sb.per_img) sd = std::max(sd, std::fabs(kv.second - sa.per_img.at(kv.first)));
            for (const auto& kv : sb.per_tag) sd = std::max(sd, std::fabs(kv.second - sa.per_tag.at(kv.first)));
            // grow by the camera and tag left out, like the next step of the driver: same handle, new mask
            ca = cams0; ta = tags0;
            res.doBundleAdjustment(ta, ca, origin, 400, 1, true);
            printf("RESIDENT_MAXDIFF %.3e STATSDIFF %.3e NCORNER %zu %zu\n", md, sd, sa.per_corner.size(), sb.per_corner.size());
            for (const auto& kv : ta) { printf("RTAG %d", kv.first); for (double d : kv.second.q.v) printf(" %.17g", d); for (double d : kv.second.t.v) printf(" %.17g", d); printf("\n"); }
        }
        const int term = vmm_ba_adapter::doBundleAdjustment(tags, cams, det, cm, origin, 400, 1, true, true);
        const auto st = vmm_ba_adapter::reprojectionStatistics(tags, cams, det, cm, true);
        const auto uv = vmm_ba_adapter::projectPoint(cm, 0.3, -0.2, 2.5);
        printf("TERM %d\n", term);
        for (const auto& kv : cams) { printf("CAM %d", kv.first); for (double d : kv.second.q.v) printf(" %.17g", d); for (double d : kv.second.t.v) printf(" %.17g", d); printf("\n"); }
        for (const auto& kv : tags) { printf("TAG %d", kv.first); for (double d : kv.second.q.v) printf(" %.17g", d); for (double d : kv.second.t.v) printf(" %.17g", d); printf("\n"); }
        printf("AVG %.17g NCORNER %zu UV %.17g %.17g\n", st.avg, st.per_corner.size(), uv[0], uv[1]);
        for (const auto& kv : st.per_img) printf("IMG %d %.17g\n", kv.first, kv.second);
    } catch (const std::exception& e) {
        // the reference's main() catches and prints (src/main_mapping.cpp:90-93)
        std::cerr << "error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
