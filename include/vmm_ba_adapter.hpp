// vmm_ba_adapter.hpp -- header-only C++ adapter between visual_marker_mapping's own types and the
// C-ABI of libvmm_ba.so (include/vmm_ba.h).
//
// It lets the reference keep its headers and class unchanged: the three hot-path member functions of
// TagReconstructor (src/TagReconstructor.cpp:340-455, 646-743) and CameraModel::projectPoint
// (src/CameraModel.cpp:6-26) become one-line calls into the function templates below
// (INTEGRATION.md shows the patch).  The templates only assume what the reference's types offer:
//   pose types     .q(i) i=0..3 (w,x,y,z), .t(i) i=0..2            Camera.h:13-14, TagReconstructor.h:19-20
//   tags           .id, .tagWidth, .tagHeight                      TagReconstructor.h:17,21-22
//   camera model   .fx .fy .cx .cy, .distortionCoefficients(i, 0)  CameraModel.h:14-19
//   detections     .tagObservations[k].imageId/.tagId/.corners[c].x()/.y()   DetectionResults.h:10-18
// so they work with Eigen types when Eigen is present and with any look-alike otherwise
// (tests/cpp/adapter_test.cpp uses 30-line POD stand-ins, this image has no Eigen).
#ifndef VMM_BA_ADAPTER_HPP_
#define VMM_BA_ADAPTER_HPP_

#include <array>
#include <cmath>
#include <cstdint>
#include <iostream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "vmm_ba.h"

namespace vmm_ba_adapter {

struct Packed {
    std::vector<int> tag_ids, cam_ids;
    std::vector<double> cam_qt, tag_qt, tag_wh, obs_px;
    std::vector<int32_t> obs_cam, obs_tag;
    vmm_ba_problem problem;
};

inline void check(int status, const char* what)
{
    if (status != VMM_BA_OK)
        throw std::runtime_error(std::string(what) + ": " + vmm_ba_last_error());
}

// The problem doBundleAdjustment assembles (src/TagReconstructor.cpp:663-724): tags in map order,
// cameras with >= 1 reconstructed tag (for_ba) or all reconstructed cameras (statistics),
// observations whose camera and tag are both reconstructed, in file order.
template <class TagMap, class CamMap, class Detection, class CamModel>
Packed pack(const TagMap& tags, const CamMap& cams, const Detection& det, const CamModel& cm, int originTagId,
            bool for_ba)
{
    Packed p;
    std::map<int, int> tag_index, cam_index, tags_in_image;
    for (const auto& kv : tags) {
        tag_index[kv.first] = (int)p.tag_ids.size();
        p.tag_ids.push_back(kv.first);
        for (int i = 0; i < 4; ++i) p.tag_qt.push_back(kv.second.q(i));
        for (int i = 0; i < 3; ++i) p.tag_qt.push_back(kv.second.t(i));
        p.tag_wh.push_back(kv.second.tagWidth);
        p.tag_wh.push_back(kv.second.tagHeight);
    }
    for (const auto& ob : det.tagObservations)
        if (tag_index.count(ob.tagId))
            tags_in_image[ob.imageId]++;                              // :679-684
    for (const auto& kv : cams) {
        if (for_ba && !tags_in_image[kv.first])                       // :689-690
            continue;
        cam_index[kv.first] = (int)p.cam_ids.size();
        p.cam_ids.push_back(kv.first);
        for (int i = 0; i < 4; ++i) p.cam_qt.push_back(kv.second.q(i));
        for (int i = 0; i < 3; ++i) p.cam_qt.push_back(kv.second.t(i));
    }
    for (const auto& ob : det.tagObservations) {                      // :699-708
        const auto c = cam_index.find(ob.imageId);
        const auto t = tag_index.find(ob.tagId);
        if (c == cam_index.end() || t == tag_index.end())
            continue;
        p.obs_cam.push_back(c->second);
        p.obs_tag.push_back(t->second);
        for (int k = 0; k < 4; ++k) {
            p.obs_px.push_back(ob.corners[k].x());
            p.obs_px.push_back(ob.corners[k].y());
        }
    }
    vmm_ba_problem& q = p.problem;
    q.intr[0] = cm.fx; q.intr[1] = cm.fy; q.intr[2] = cm.cx; q.intr[3] = cm.cy;
    for (int i = 0; i < 5; ++i) q.dist[i] = cm.distortionCoefficients(i, 0);
    q.n_cams = (int32_t)p.cam_ids.size();
    q.n_tags = (int32_t)p.tag_ids.size();
    q.cam_qt = p.cam_qt.data();
    q.tag_qt = p.tag_qt.data();
    q.tag_wh = p.tag_wh.data();
    const auto o = tag_index.find(originTagId);                       // :669-673
    q.fixed_tag = (o == tag_index.end()) ? -1 : o->second;
    q.n_obs = (int64_t)p.obs_cam.size();
    q.obs_cam = p.obs_cam.data();
    q.obs_tag = p.obs_tag.data();
    q.obs_px = p.obs_px.data();
    return p;
}

// What the reference prints with summary.FullReport() (src/TagReconstructor.cpp:741-742), as far as it exists
// here: counts, costs, and where the device time went (measured on the device, vmm_ba_summary.time_*_s).
inline void printFullReport(const vmm_ba_summary& s)
{
    const double accounted = s.time_eval_s + s.time_eliminate_s + s.time_factor_solve_s + s.time_step_s + s.time_control_s;
    std::cout << "vmm_ba: iterations " << s.iterations << " (successful " << s.num_successful_steps << ", unsuccessful "
              << s.num_unsuccessful_steps << "), initial cost " << s.initial_cost << ", final cost " << s.final_cost
              << ", solver time " << s.time_solve_s << " s" << std::endl
              << "vmm_ba: time (s): residual + Jacobian evaluation " << s.time_eval_s << ", elimination + rank-k update "
              << s.time_eliminate_s << ", linear solver (Cholesky + triangular solves) " << s.time_factor_solve_s
              << ", step + candidate cost " << s.time_step_s << ", trust-region control " << s.time_control_s
              << ", host / idle " << (s.time_solve_s > accounted ? s.time_solve_s - accounted : 0.0) << std::endl;
}

// The covariance report of the final bundle adjustment (src/TagReconstructor.cpp:761-782): per-tag standard deviations
// of the translation and the "Marker Position RMS" line.  cov = 9 doubles per entry of tag_ids (row-major 3x3).
inline void printCovarianceReport(const std::vector<int>& tag_ids, const std::vector<double>& cov)
{
    double avg[3] = { 0.0, 0.0, 0.0 };
    for (size_t k = 0; k < tag_ids.size(); ++k) {
        const double* c = &cov[9 * k];
        const double sd[3] = { std::sqrt(c[0]), std::sqrt(c[4]), std::sqrt(c[8]) };
        std::cout << "StdDev of tag " << tag_ids[k] << ": " << sd[0] << " " << sd[1] << " " << sd[2]
                  << " | StdDevNorm: " << std::sqrt(std::sqrt(sd[0] * sd[0] + sd[1] * sd[1] + sd[2] * sd[2]))
                  << std::endl;
        avg[0] += c[0];
        avg[1] += c[4];
        avg[2] += c[8];
    }
    const double n = (double)tag_ids.size();
    std::cout << "Marker Position RMS = " << std::sqrt(avg[0] / n + avg[1] / n + avg[2] / n) << std::endl;
}

// Body of TagReconstructor::doBundleAdjustment (src/TagReconstructor.cpp:646-743).  Updates the map
// nodes in place, prints "Solution <termination_type>" like :740.  Returns the termination type.
template <class TagMap, class CamMap, class Detection, class CamModel>
int doBundleAdjustment(TagMap& tags, CamMap& cams, const Detection& det, const CamModel& cm, int originTagId,
                       int maxNumIterations, size_t ceresThreads, bool robustify = true, bool printSummary = false,
                       int device = 0)
{
    Packed p = pack(tags, cams, det, cm, originTagId, true);
    if (p.problem.n_cams == 0 || p.problem.n_tags == 0 || p.problem.n_obs == 0) {
        std::cout << "Solution " << VMM_BA_CONVERGENCE << std::endl;
        return VMM_BA_CONVERGENCE;
    }
    vmm_ba_create_options co;
    vmm_ba_default_create_options(&co);
    co.device = device;
    vmm_ba_handle h = nullptr;
    check(vmm_ba_create(&p.problem, &co, &h), "vmm_ba_create");
    vmm_ba_options o;
    vmm_ba_default_options(&o);
    o.max_num_iterations = maxNumIterations;
    o.num_threads = (int32_t)ceresThreads;
    o.robustify = robustify ? 1 : 0;
    vmm_ba_summary s;
    s.trace = nullptr;
    s.trace_capacity = 0;
    int rc = vmm_ba_solve(h, &o, &s);
    if (rc == VMM_BA_OK)
        rc = vmm_ba_get_state(h, p.cam_qt.data(), p.tag_qt.data());
    std::vector<double> cov;                                          // :744-760
    if (rc == VMM_BA_OK && printSummary) {
        cov.resize(9 * p.tag_ids.size());
        rc = vmm_ba_tag_translation_covariance(h, o.robustify, o.huber_a, cov.data());
    }
    vmm_ba_destroy(h);
    check(rc, "vmm_ba_solve");
    for (size_t k = 0; k < p.cam_ids.size(); ++k) {
        auto& c = cams.at(p.cam_ids[k]);
        for (int i = 0; i < 4; ++i) c.q(i) = p.cam_qt[7 * k + i];
        for (int i = 0; i < 3; ++i) c.t(i) = p.cam_qt[7 * k + 4 + i];
    }
    for (size_t k = 0; k < p.tag_ids.size(); ++k) {
        auto& t = tags.at(p.tag_ids[k]);
        for (int i = 0; i < 4; ++i) t.q(i) = p.tag_qt[7 * k + i];
        for (int i = 0; i < 3; ++i) t.t(i) = p.tag_qt[7 * k + 4 + i];
    }
    std::cout << "Solution " << s.termination_type << std::endl;     // :740
    if (printSummary)                                                 // :741-742 (FullReport stand-in)
        printFullReport(s);
    if (printSummary)                                                 // :761-782
        printCovarianceReport(p.tag_ids, cov);
    return s.termination_type;
}

struct Stats {
    std::map<int, double> per_img, per_tag;
    double avg = 0.0;
    std::vector<std::array<double, 2>> per_corner;
};

// Bodies of computeReprojectionErrorPerImg / PerTag / PerCorner (src/TagReconstructor.cpp:340-455).
template <class TagMap, class CamMap, class Detection, class CamModel>
Stats reprojectionStatistics(const TagMap& tags, const CamMap& cams, const Detection& det, const CamModel& cm,
                             bool want_corners, int device = 0)
{
    Stats out;
    Packed p = pack(tags, cams, det, cm, -1, false);
    if (p.problem.n_cams == 0 || p.problem.n_tags == 0)
        return out;
    vmm_ba_create_options co;
    vmm_ba_default_create_options(&co);
    co.device = device;
    vmm_ba_handle h = nullptr;
    check(vmm_ba_create(&p.problem, &co, &h), "vmm_ba_create");
    std::vector<double> pc(p.cam_ids.size()), pt(p.tag_ids.size()), corner(want_corners ? p.obs_px.size() : 0);
    const int rc = vmm_ba_reprojection_stats(h, pc.data(), pt.data(), &out.avg, want_corners ? corner.data() : nullptr);
    vmm_ba_destroy(h);
    check(rc, "vmm_ba_reprojection_stats");
    for (size_t k = 0; k < p.cam_ids.size(); ++k)
        out.per_img[p.cam_ids[k]] = pc[k];                            // -1.0 for cameras without observations
    for (size_t k = 0; k < p.tag_ids.size(); ++k)
        if (!std::isnan(pt[k]))
            out.per_tag[p.tag_ids[k]] = pt[k];
    for (size_t i = 0; i + 1 < corner.size(); i += 2)
        out.per_corner.push_back({ corner[i], corner[i + 1] });
    return out;
}

// Body of CameraModel::projectPoint (src/CameraModel.cpp:6-26) for one camera-frame point.
template <class CamModel>
std::array<double, 2> projectPoint(const CamModel& cm, double X, double Y, double Z, int device = 0)
{
    const double intr[4] = { cm.fx, cm.fy, cm.cx, cm.cy };
    double dist[5];
    for (int i = 0; i < 5; ++i) dist[i] = cm.distortionCoefficients(i, 0);
    const double pc[3] = { X, Y, Z };
    double uv[2];
    check(vmm_ba_project_points(intr, dist, 1, pc, uv, device), "vmm_ba_project_points");
    return { uv[0], uv[1] };
}

// Device-resident variant for the incremental driver (startReconstruction, src/TagReconstructor.cpp:86-278): ONE
// handle is built from the whole detection result and kept as a member of the reconstructor; every bundle
// adjustment and every statistics call of the run only sends the poses and a mask over the observations
// (camera and tag both reconstructed, :699-708) -- vmm_ba_set_observation_mask.  Same results as the functions
// above, without a vmm_ba_create per call.  Detection needs .images[].imageId, .tags[].{tagId,width,height},
// .tagObservations[].{imageId,tagId,corners}.
template <class Detection, class CamModel>
class Resident {
public:
    Resident(const Detection& det, const CamModel& cm, int device = 0) : device_(device)
    {
        std::map<int, int> ci, ti;
        for (const auto& im : det.images) ci[im.imageId] = 0;
        for (const auto& ob : det.tagObservations) ci[ob.imageId] = 0;
        for (const auto& t : det.tags) ti[t.tagId] = 0;
        for (const auto& ob : det.tagObservations) ti[ob.tagId] = 0;
        for (auto& kv : ci) { kv.second = (int)cam_ids_.size(); cam_ids_.push_back(kv.first); }
        for (auto& kv : ti) { kv.second = (int)tag_ids_.size(); tag_ids_.push_back(kv.first); }
        cam_index_ = ci;
        tag_index_ = ti;
        tag_wh_.assign(2 * tag_ids_.size(), 1.0);
        for (const auto& t : det.tags) {
            tag_wh_[2 * ti[t.tagId]] = t.width;
            tag_wh_[2 * ti[t.tagId] + 1] = t.height;
        }
        for (const auto& ob : det.tagObservations) {
            obs_img_.push_back(ob.imageId);
            obs_tagid_.push_back(ob.tagId);
            obs_cam_.push_back(ci[ob.imageId]);
            obs_tag_.push_back(ti[ob.tagId]);
            for (int k = 0; k < 4; ++k) {
                obs_px_.push_back(ob.corners[k].x());
                obs_px_.push_back(ob.corners[k].y());
            }
        }
        intr_[0] = cm.fx; intr_[1] = cm.fy; intr_[2] = cm.cx; intr_[3] = cm.cy;
        for (int i = 0; i < 5; ++i) dist_[i] = cm.distortionCoefficients(i, 0);
    }
    ~Resident() { if (h_) vmm_ba_destroy(h_); }
    Resident(const Resident&) = delete;
    Resident& operator=(const Resident&) = delete;

    // doBundleAdjustment on the current maps; returns the termination type and prints like the free function
    template <class TagMap, class CamMap>
    int doBundleAdjustment(TagMap& tags, CamMap& cams, int originTagId, int maxNumIterations, size_t ceresThreads,
                           bool robustify = true, bool printSummary = false)
    {
        if (!prepare(tags, cams, originTagId, true)) {
            std::cout << "Solution " << VMM_BA_CONVERGENCE << std::endl;
            return VMM_BA_CONVERGENCE;
        }
        vmm_ba_options o;
        vmm_ba_default_options(&o);
        o.max_num_iterations = maxNumIterations;
        o.num_threads = (int32_t)ceresThreads;
        o.robustify = robustify ? 1 : 0;
        vmm_ba_summary s;
        s.trace = nullptr;
        s.trace_capacity = 0;
        check(vmm_ba_solve(h_, &o, &s), "vmm_ba_solve");
        check(vmm_ba_get_state(h_, cam_qt_.data(), tag_qt_.data()), "vmm_ba_get_state");
        std::vector<double> cov_all;                                  // :744-760, on the handle that just solved
        if (printSummary) {
            cov_all.resize(9 * tag_ids_.size());
            check(vmm_ba_tag_translation_covariance(h_, o.robustify, o.huber_a, cov_all.data()),
                  "vmm_ba_tag_translation_covariance");
        }
        for (auto& kv : cams) {
            if (!cam_active_[cam_index_[kv.first]]) continue;         // :689-690 cameras without reconstructed tags
            const double* q = &cam_qt_[7 * cam_index_[kv.first]];
            for (int i = 0; i < 4; ++i) kv.second.q(i) = q[i];
            for (int i = 0; i < 3; ++i) kv.second.t(i) = q[4 + i];
        }
        for (auto& kv : tags) {
            const double* q = &tag_qt_[7 * tag_index_[kv.first]];
            for (int i = 0; i < 4; ++i) kv.second.q(i) = q[i];
            for (int i = 0; i < 3; ++i) kv.second.t(i) = q[4 + i];
        }
        std::cout << "Solution " << s.termination_type << std::endl;  // :740
        if (printSummary) {                                           // :741-742, :761-782 over the reconstructed tags
            printFullReport(s);
            std::vector<int> ids;
            std::vector<double> cov;
            for (const auto& kv : tags) {
                ids.push_back(kv.first);
                const double* c = &cov_all[9 * tag_index_[kv.first]];
                cov.insert(cov.end(), c, c + 9);
            }
            printCovarianceReport(ids, cov);
        }
        return s.termination_type;
    }

    template <class TagMap, class CamMap>
    Stats reprojectionStatistics(const TagMap& tags, const CamMap& cams, int originTagId, bool want_corners)
    {
        Stats out;
        if (tags.empty() || cams.empty())
            return out;
        prepare(tags, cams, originTagId, false);
        std::vector<double> pc(cam_ids_.size()), pt(tag_ids_.size()), corner(want_corners ? obs_px_.size() : 0);
        check(vmm_ba_reprojection_stats(h_, pc.data(), pt.data(), &out.avg, want_corners ? corner.data() : nullptr),
              "vmm_ba_reprojection_stats");
        for (const auto& kv : cams)
            out.per_img[kv.first] = pc[cam_index_[kv.first]];         // -1.0 for cameras without observations
        for (const auto& kv : tags)
            if (!std::isnan(pt[tag_index_[kv.first]]))
                out.per_tag[kv.first] = pt[tag_index_[kv.first]];
        for (size_t i = 0; i < mask_.size() && want_corners; ++i)
            if (mask_[i])
                for (int k = 0; k < 4; ++k)
                    out.per_corner.push_back({ corner[8 * i + 2 * k], corner[8 * i + 2 * k + 1] });
        return out;
    }

private:
    // state + mask of this step; (re)creates the handle when the constant block changes.  false: nothing to solve
    template <class TagMap, class CamMap>
    bool prepare(const TagMap& tags, const CamMap& cams, int originTagId, bool for_ba)
    {
        cam_qt_.assign(7 * cam_ids_.size(), 0.0);
        tag_qt_.assign(7 * tag_ids_.size(), 0.0);
        for (size_t c = 0; c < cam_ids_.size(); ++c) { cam_qt_[7 * c] = 1.0; cam_qt_[7 * c + 6] = 1.0; }   // finite defaults
        for (size_t t = 0; t < tag_ids_.size(); ++t) tag_qt_[7 * t] = 1.0;
        for (const auto& kv : cams) {
            double* q = &cam_qt_[7 * cam_index_.at(kv.first)];
            for (int i = 0; i < 4; ++i) q[i] = kv.second.q(i);
            for (int i = 0; i < 3; ++i) q[4 + i] = kv.second.t(i);
        }
        for (const auto& kv : tags) {
            double* q = &tag_qt_[7 * tag_index_.at(kv.first)];
            for (int i = 0; i < 4; ++i) q[i] = kv.second.q(i);
            for (int i = 0; i < 3; ++i) q[4 + i] = kv.second.t(i);
        }
        mask_.assign(obs_cam_.size(), 0);
        cam_active_.assign(cam_ids_.size(), 0);
        size_t n_on = 0;
        for (size_t i = 0; i < obs_cam_.size(); ++i)
            if (cams.count(obs_img_[i]) && tags.count(obs_tagid_[i])) {
                mask_[i] = 1;
                cam_active_[obs_cam_[i]] = 1;
                ++n_on;
            }
        const auto o = tags.count(originTagId) ? tag_index_.find(originTagId) : tag_index_.end();
        const int fixed = (o == tag_index_.end()) ? -1 : o->second;
        if (for_ba && n_on == 0)
            return false;
        if (!h_ || fixed != fixed_) {
            if (h_) vmm_ba_destroy(h_);
            h_ = nullptr;
            vmm_ba_problem q;
            for (int i = 0; i < 4; ++i) q.intr[i] = intr_[i];
            for (int i = 0; i < 5; ++i) q.dist[i] = dist_[i];
            q.n_cams = (int32_t)cam_ids_.size();
            q.n_tags = (int32_t)tag_ids_.size();
            q.cam_qt = cam_qt_.data();
            q.tag_qt = tag_qt_.data();
            q.tag_wh = tag_wh_.data();
            q.fixed_tag = fixed;
            q.n_obs = (int64_t)obs_cam_.size();
            q.obs_cam = obs_cam_.data();
            q.obs_tag = obs_tag_.data();
            q.obs_px = obs_px_.data();
            vmm_ba_create_options co;
            vmm_ba_default_create_options(&co);
            co.device = device_;
            check(vmm_ba_create(&q, &co, &h_), "vmm_ba_create");
            fixed_ = fixed;
        } else {
            check(vmm_ba_set_state(h_, cam_qt_.data(), tag_qt_.data()), "vmm_ba_set_state");
        }
        check(vmm_ba_set_observation_mask(h_, mask_.data()), "vmm_ba_set_observation_mask");
        return true;
    }

    int device_;
    vmm_ba_handle h_ = nullptr;
    int fixed_ = -2;
    std::vector<int> cam_ids_, tag_ids_, obs_img_, obs_tagid_;
    std::map<int, int> cam_index_, tag_index_;
    std::vector<double> tag_wh_, obs_px_, cam_qt_, tag_qt_;
    std::vector<int32_t> obs_cam_, obs_tag_;
    std::vector<uint8_t> mask_, cam_active_;
    double intr_[4], dist_[5];
};

} // namespace vmm_ba_adapter
#endif
