"""GPU tests of the TagReconstructor mirror (reference API names) against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _make(s, tag_ids, cam_ids):
    from visual_marker_mapping_amd import tag_reconstructor as tr
    det = tr.DetectionResult(
        [tr.TagImg(cid, "i%d.jpg" % cid) for cid in cam_ids],
        [tr.Tag(tid, "apriltag_36h11", *s.tag_wh[k]) for k, tid in enumerate(tag_ids)],
        [tr.TagObservation(cam_ids[c], tag_ids[t], px.reshape(4, 2)) for c, t, px in zip(s.obs_cam, s.obs_tag, s.obs_px)])
    rec = tr.TagReconstructor(det)
    rec.setCameraModel(tr.CameraModel(*s.intr, s.dist, 4000, 6000))
    rec.setReconstructedTags({tid: tr.ReconstructedTag(tid, "apriltag_36h11", s.tag_init[k, :4], s.tag_init[k, 4:],
                                                        *s.tag_wh[k]) for k, tid in enumerate(tag_ids)})
    rec.setReconstructedCameras({cid: tr.Camera(cid, s.cam_init[k, :4], s.cam_init[k, 4:])
                                 for k, cid in enumerate(cam_ids)})
    rec.setOriginTagId(tag_ids[0])
    return rec


def test_do_bundle_adjustment_matches_oracle_with_sparse_ids(oracle, capsys):
    from visual_marker_mapping_amd.synthetic import make_scene
    s = make_scene(1)
    tag_ids = [230 + 3 * k for k in range(len(s.tag_init))]
    cam_ids = [1000 - 7 * k for k in range(len(s.cam_init))]      # descending: map order != array order
    rec = _make(s, tag_ids, cam_ids)
    rec.doBundleAdjustment(400, 4, True, False)
    assert "Solution 0" in capsys.readouterr().out                 # src/TagReconstructor.cpp:740
    sc = oracle.Scene(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, 0, s.obs_cam, s.obs_tag, s.obs_px)
    summ, _ = oracle.solve(sc, oracle.default_options(robustify=1, linear_solver=oracle.DENSE_NORMAL))
    assert rec.lastSummary["iterations"] == summ["iterations"]
    for k, cid in enumerate(cam_ids):
        np.testing.assert_allclose(np.r_[rec.reconstructedCameras[cid].q, rec.reconstructedCameras[cid].t],
                                   sc.cam_qt[k], rtol=0, atol=1e-6 * np.abs(sc.cam_qt).max())
    for k, tid in enumerate(tag_ids):
        np.testing.assert_allclose(np.r_[rec.reconstructedTags[tid].q, rec.reconstructedTags[tid].t],
                                   sc.tag_qt[k], rtol=0, atol=1e-6 * np.abs(sc.tag_qt).max())
    # the origin tag is constant (src/TagReconstructor.cpp:669-673)
    np.testing.assert_array_equal(rec.reconstructedTags[tag_ids[0]].t, s.tag_init[0, 4:])


def test_reprojection_statistics_and_pruning(oracle, capsys):
    from visual_marker_mapping_amd import tag_reconstructor as tr
    from visual_marker_mapping_amd.synthetic import make_scene
    s = make_scene(1)
    tag_ids = list(range(10, 10 + len(s.tag_init)))
    cam_ids = list(range(len(s.cam_init)))
    rec = _make(s, tag_ids, cam_ids)
    rec.doBundleAdjustment(400, 1, True)
    sc = oracle.Scene(s.intr, s.dist, [np.r_[rec.reconstructedCameras[c].q, rec.reconstructedCameras[c].t] for c in cam_ids],
                      [np.r_[rec.reconstructedTags[t].q, rec.reconstructedTags[t].t] for t in tag_ids],
                      s.tag_wh, 0, s.obs_cam, s.obs_tag, s.obs_px)
    pc, pt, avg, corner = oracle.reprojection_stats(sc)
    per_img = rec.computeReprojectionErrorPerImg()
    per_tag, gavg = rec.computeReprojectionErrorPerTag()
    per_corner = rec.computeReprojectionErrorPerCorner()
    np.testing.assert_allclose([per_img[c] for c in cam_ids], pc, rtol=1e-11)
    np.testing.assert_allclose([per_tag[t] for t in tag_ids], pt, rtol=1e-11)
    assert abs(gavg - avg) < 1e-11 * avg
    np.testing.assert_allclose(np.array(per_corner), corner.reshape(-1, 2), rtol=0, atol=1e-9)
    assert max(per_tag.values()) < 2.0
    # nothing to prune at 0.3 px noise ...
    rec.removeBadMarkers(2.0)
    rec.removeBadCameras(2.0)
    assert len(rec.reconstructedTags) == len(tag_ids) and len(rec.reconstructedCameras) == len(cam_ids)
    # ... a displaced tag and a displaced camera are pruned, the origin tag never is (:792-799)
    rec.reconstructedTags[tag_ids[3]].t += 0.05
    rec.reconstructedTags[tag_ids[0]].t += 0.05
    rec.removeBadMarkers(2.0)
    assert tag_ids[3] not in rec.reconstructedTags and tag_ids[0] in rec.reconstructedTags
    rec.reconstructedTags[tag_ids[0]].t -= 0.05
    rec.reconstructedCameras[cam_ids[5]].t += 0.2
    # a reconstructed camera without any observation of a reconstructed tag gets -1 and is removed (:379-383, :809)
    rec.reconstructedCameras[999] = tr.Camera(999)
    assert rec.computeReprojectionErrorPerImg()[999] == -1.0
    rec.removeBadCameras(2.0)
    assert cam_ids[5] not in rec.reconstructedCameras and 999 not in rec.reconstructedCameras
    out = capsys.readouterr().out
    assert "Removing bad marker with id %d" % tag_ids[3] in out and "Removing bad camera with id 999" in out
    # BA still runs on the pruned problem
    rec.doBundleAdjustment(400, 1, False)
    assert rec.lastSummary["termination_type"] == 0


def test_project_point_is_camera_model_project_point(oracle, kats):
    # through vmm_ba_project_points; CameraModel.cpp:20-23 (aliased y term) pinned by the mpmath KATs at 1e-9 px
    from visual_marker_mapping_amd import tag_reconstructor as tr
    from visual_marker_mapping_amd import engine as eng
    for case in kats["project_point"]:
        cm = tr.CameraModel(*case["intr"], case["dist"])
        np.testing.assert_allclose(cm.projectPoint(np.array(case["point_cam"])), case["uv"], rtol=0, atol=1e-9)
    c0 = kats["project_point"][0]
    pts = np.array([c["point_cam"] for c in kats["project_point"] if c["dist"] == c0["dist"]])
    ref = np.array([c["uv"] for c in kats["project_point"] if c["dist"] == c0["dist"]])
    np.testing.assert_allclose(eng.project_points(c0["intr"], c0["dist"], pts), ref, rtol=0, atol=1e-9)
    np.testing.assert_allclose(eng.project_points(c0["intr"], c0["dist"], pts[0]),
                               [oracle.project_point(c0["intr"], c0["dist"], pts[0])], rtol=0, atol=1e-9)


def test_reprojection_stats_match_camera_model_kats(kats):
    # through vmm_ba_reprojection_stats: src/TagReconstructor.cpp:353-363 = Eigen rotations + projectPoint
    from visual_marker_mapping_amd import engine as eng
    for case in kats["obs"]:
        with eng.BundleAdjuster(case["intr"], case["dist"], [case["cam_qt"]], [case["tag_qt"]], [case["wh"]], -1,
                                [0], [0], [case["px"]]) as ba:
            pc, pt, avg, corner = ba.reprojection_stats()
        ref = np.array(case["reprojection_error_camera_model"])
        np.testing.assert_allclose(corner[0], ref, rtol=0, atol=2e-9)
        mean = np.sqrt((ref.reshape(4, 2) ** 2).sum(axis=1)).sum() / 4
        np.testing.assert_allclose([pc[0], pt[0], avg], mean, rtol=1e-12)


def test_do_bundle_adjustment_points_matches_oracle(oracle, capsys):
    """TagReconstructor.doBundleAdjustment_points (src/TagReconstructor.cpp:457-644) with sparse ids."""
    from visual_marker_mapping_amd.synthetic import make_scene
    s = make_scene(1)
    tag_ids = [230 + 3 * k for k in range(len(s.tag_init))]
    cam_ids = [1000 - 7 * k for k in range(len(s.cam_init))]
    rec = _make(s, tag_ids, cam_ids)
    rec.doBundleAdjustment_points(400, 4, False)
    assert "Solution 0" in capsys.readouterr().out
    sc, pts0 = oracle.point_scene(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag,
                                  s.obs_px)
    summ, _ = oracle.solve(sc, oracle.default_options(robustify=0, linear_solver=oracle.DENSE_NORMAL))
    assert rec.lastSummary["iterations"] == summ["iterations"]
    np.testing.assert_allclose(rec.lastSummary["final_cost"], summ["final_cost"], rtol=1e-8)
    ref = oracle.scene_points(sc)
    for k, tid in enumerate(tag_ids):
        np.testing.assert_allclose(rec.lastPoints[tid], ref[k], rtol=0, atol=1e-6 * np.abs(ref).max())
        # the rebuilt pose puts the tag centre at the mean of its corners (:633)
        np.testing.assert_allclose(rec.reconstructedTags[tid].t, ref[k].mean(axis=0), rtol=0, atol=1e-9)
    for k, cid in enumerate(cam_ids):
        np.testing.assert_allclose(np.r_[rec.reconstructedCameras[cid].q, rec.reconstructedCameras[cid].t], sc.cam_qt[k],
                                   rtol=0, atol=1e-6 * np.abs(sc.cam_qt).max())
    np.testing.assert_array_equal(rec.lastPoints[tag_ids[0]], pts0[0])          # origin tag constant
    # the tag-pose bundle adjustment still runs on the same object afterwards
    rec.doBundleAdjustment(400, 1, False)
    assert rec.lastSummary["termination_type"] == 0
