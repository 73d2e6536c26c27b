// The call pattern of TagReconstructor::startReconstruction (src/TagReconstructor.cpp:233,236,271-277: one bundle
// adjustment + removeBadMarkers per added image, then BA(1500, robust), both prunings, BA(1500, plain, summary)) driven
// through the members integration/visual_marker_mapping.patch adds to the class: a `mutable std::unique_ptr` to a
// nested struct derived from vmm_ba_adapter::Resident, built on first use, dropped by setCameraModel.  The same
// sequence then runs through the free functions (one vmm_ba_create per call) and the two must agree.
// POD stand-ins as in adapter_test.cpp (this image has no Eigen); initial poses come with the scene (they stand in
// for the OpenCV PnP initialisation of :156,167-230).
// Usage: incremental_test [--time] < scene.txt
// --time: wall time of the N + 2 solves + prunings with fixed (precomputed) initial poses, resident handle against one
// vmm_ba_create per call, after an untimed pass that pays the one-off costs (code objects, allocator); best of three.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <iostream>
#include <map>
#include <memory>
#include <set>
#include <string>
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

// Member for member the part of class TagReconstructor the patch touches (include/visual_marker_mapping/
// TagReconstructor.h:134-145 + the guarded block of the patch), with the patched method bodies.
class MiniReconstructor {
public:
    MiniReconstructor(DetectionResult d, bool resident) : originTagId(-1), detectionResults_(std::move(d)), resident_(resident) {}
    void setCameraModel(const CameraModel& cameraModel)
    {
        camModel = cameraModel;
        vmmBaResident_.reset();
    }
    void setOriginTagId(int id) { originTagId = id; }

    // declared here, defined behind the nested struct like the patched src/TagReconstructor.cpp
    std::map<int, double> computeReprojectionErrorPerImg() const;
    std::map<int, double> computeReprojectionErrorPerTag(double& avg) const;
    void doBundleAdjustment(int maxNumIterations, size_t ceresThreads, bool robustify = true, bool printSummary = false);
    void removeBadMarkers(double threshold)   // src/TagReconstructor.cpp:786-802
    {
        double avg = 0.0;
        const auto err = computeReprojectionErrorPerTag(avg);
        for (const auto& kv : err)
            if (kv.second > threshold && kv.first != originTagId)
                reconstructedTags.erase(kv.first);
    }
    void removeBadCameras(double threshold)   // :804-816
    {
        const auto err = computeReprojectionErrorPerImg();
        for (const auto& kv : err)
            if (kv.second > threshold || kv.second < 0)
                reconstructedCameras.erase(kv.first);
    }

    int originTagId;
    DetectionResult detectionResults_;
    std::map<int, ReconstructedTag> reconstructedTags;
    std::map<int, Camera> reconstructedCameras;
    CameraModel camModel;

private:
    // ---- what the patch adds to the class ----
    struct VmmBaResident;
    struct VmmBaResidentDeleter
    {
        void operator()(VmmBaResident* p) const;
    };
    mutable std::unique_ptr<VmmBaResident, VmmBaResidentDeleter> vmmBaResident_;
    VmmBaResident& vmmBa() const;
    bool resident_;
};

struct MiniReconstructor::VmmBaResident : vmm_ba_adapter::Resident<DetectionResult, CameraModel>
{
    using vmm_ba_adapter::Resident<DetectionResult, CameraModel>::Resident;
};
void MiniReconstructor::VmmBaResidentDeleter::operator()(VmmBaResident* p) const { delete p; }
MiniReconstructor::VmmBaResident& MiniReconstructor::vmmBa() const
{
    if (!vmmBaResident_)
        vmmBaResident_.reset(new VmmBaResident(detectionResults_, camModel));
    return *vmmBaResident_;
}

std::map<int, double> MiniReconstructor::computeReprojectionErrorPerImg() const
{
    if (!resident_)
        return vmm_ba_adapter::reprojectionStatistics(reconstructedTags, reconstructedCameras, detectionResults_, camModel, false).per_img;
    return vmmBa().reprojectionStatistics(reconstructedTags, reconstructedCameras, originTagId, false).per_img;
}
std::map<int, double> MiniReconstructor::computeReprojectionErrorPerTag(double& avg) const
{
    const auto st = resident_ ? vmmBa().reprojectionStatistics(reconstructedTags, reconstructedCameras, originTagId, false)
                              : vmm_ba_adapter::reprojectionStatistics(reconstructedTags, reconstructedCameras, detectionResults_, camModel, false);
    avg = st.avg;
    return st.per_tag;
}
void MiniReconstructor::doBundleAdjustment(int maxNumIterations, size_t ceresThreads, bool robustify, bool printSummary)
{
    if (!resident_)
        vmm_ba_adapter::doBundleAdjustment(reconstructedTags, reconstructedCameras, detectionResults_, camModel, originTagId,
                                           maxNumIterations, ceresThreads, robustify, printSummary);
    else
        vmmBa().doBundleAdjustment(reconstructedTags, reconstructedCameras, originTagId, maxNumIterations, ceresThreads,
                                   robustify, printSummary);
}

// the N + 2 pattern; returns the number of bundle adjustments
static int run(MiniReconstructor& r, const std::map<int, Camera>& cam_init, const std::map<int, ReconstructedTag>& tag_init, int origin)
{
    std::map<int, int> seen;   // images per tag: a tag seen once is never reconstructed (:189-194)
    for (const auto& ob : r.detectionResults_.tagObservations)
        seen[ob.tagId]++;
    r.setOriginTagId(origin);
    // image order: the first image that sees the origin tag (:115-124), then the others by id
    std::vector<int> order;
    for (const auto& ob : r.detectionResults_.tagObservations)
        if (ob.tagId == origin && cam_init.count(ob.imageId)) {
            order.push_back(ob.imageId);
            break;
        }
    for (const auto& kv : cam_init)
        if (order.empty() || kv.first != order[0])
            order.push_back(kv.first);
    int n_ba = 0;
    for (const int img : order) {
        r.reconstructedCameras[img] = cam_init.at(img);
        for (const auto& ob : r.detectionResults_.tagObservations)
            if (ob.imageId == img && seen[ob.tagId] >= 2 && !r.reconstructedTags.count(ob.tagId))
                r.reconstructedTags[ob.tagId] = tag_init.at(ob.tagId);
        r.doBundleAdjustment(400, 1, true);        // :233
        r.removeBadMarkers(2.0);                   // :236
        ++n_ba;
    }
    r.doBundleAdjustment(1500, 1, true, false);    // :271
    r.removeBadMarkers(2.0);                       // :273
    r.removeBadCameras(2.0);                       // :275
    r.doBundleAdjustment(1500, 1, false, true);    // :277
    return n_ba + 2;
}

int main(int argc, char** argv)
{
    const bool timing = argc > 1 && std::string(argv[1]) == "--time";
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
    try {
        double secs[2] = { 0.0, 0.0 };
        int n_ba[2] = { 0, 0 };
        std::map<int, Camera> rc[2];
        std::map<int, ReconstructedTag> rt[2];
        for (int mode = 0; mode < 2; ++mode) {   // 0: the patched class (resident handle), 1: one vmm_ba_create per call
            MiniReconstructor r(det, mode == 0);
            r.setCameraModel(cm);
            const auto t0 = std::chrono::steady_clock::now();
            n_ba[mode] = run(r, cams, tags, origin);
            secs[mode] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            rc[mode] = r.reconstructedCameras;
            rt[mode] = r.reconstructedTags;
        }
        double md = 0.0;
        bool same_keys = rc[0].size() == rc[1].size() && rt[0].size() == rt[1].size();
        for (const auto& kv : rc[0]) {
            if (!rc[1].count(kv.first)) { same_keys = false; continue; }
            for (int i = 0; i < 4; ++i) md = std::max(md, std::fabs(kv.second.q(i) - rc[1].at(kv.first).q(i)));
            for (int i = 0; i < 3; ++i) md = std::max(md, std::fabs(kv.second.t(i) - rc[1].at(kv.first).t(i)));
        }
        for (const auto& kv : rt[0]) {
            if (!rt[1].count(kv.first)) { same_keys = false; continue; }
            for (int i = 0; i < 4; ++i) md = std::max(md, std::fabs(kv.second.q(i) - rt[1].at(kv.first).q(i)));
            for (int i = 0; i < 3; ++i) md = std::max(md, std::fabs(kv.second.t(i) - rt[1].at(kv.first).t(i)));
        }
        printf("INCREMENTAL n_ba %d %d cams %zu tags %zu same_keys %d maxdiff %.3e resident_s %.4f per_call_s %.4f\n", n_ba[0], n_ba[1],
               rc[0].size(), rt[0].size(), same_keys ? 1 : 0, md, secs[0], secs[1]);
        if (timing) {
            // the pass above was the warm-up; stdout of the solves (the reference prints a line per solve) goes to /dev/null
            double best[2] = { 1e30, 1e30 };
            FILE* keep = stdout;
            FILE* nul = fopen("/dev/null", "w");
            for (int rep = 0; rep < 3; ++rep)
                for (int mode = 0; mode < 2; ++mode) {
                    MiniReconstructor r(det, mode == 0);
                    r.setCameraModel(cm);
                    std::cout.flush();
                    fflush(stdout);
                    std::streambuf* cb = std::cout.rdbuf();
                    std::cout.rdbuf(nullptr);
                    if (nul) stdout = nul;
                    const auto t0 = std::chrono::steady_clock::now();
                    run(r, cams, tags, origin);
                    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    stdout = keep;
                    std::cout.rdbuf(cb);
                    std::cout.clear();
                    best[mode] = std::min(best[mode], dt);
                }
            if (nul) fclose(nul);
            printf("TIMING images %d tags %d observations %d bundle_adjustments %d resident_s %.4f per_call_s %.4f "
                   "resident_ms_per_ba %.3f per_call_ms_per_ba %.3f\n", nc, nt, no, n_ba[0], best[0], best[1],
                   1e3 * best[0] / n_ba[0], 1e3 * best[1] / n_ba[1]);
        }
        for (const auto& kv : rt[0]) { printf("ITAG %d", kv.first); for (double d : kv.second.q.v) printf(" %.17g", d); for (double d : kv.second.t.v) printf(" %.17g", d); printf("\n"); }
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
