"""File formats, PnP initialisation and the incremental driver's host logic -- CPU only.

Format fixtures: tests/golden/readme_*.json are the example documents of the reference's README.md
(:130-263, with the "[...]" ellipses removed) -- the only files in the reference tree that show the
formats.  The driver test replaces the GPU bundle adjustment by the CPU oracle (test infrastructure) to
exercise startReconstruction's control flow without a GPU.
"""
import json
import os

import numpy as np
import pytest

from visual_marker_mapping_amd import io as vio
from visual_marker_mapping_amd import pnp, synthetic
from visual_marker_mapping_amd.tag_reconstructor import (Camera, CameraModel, ReconstructedTag, TagReconstructor,
                                                         _quat_to_R)

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_detection_file_layout_matches_readme_example(tmp_path):
    src = os.path.join(GOLD, "readme_marker_detections.json")
    det = vio.readDetectionResult(src)
    assert [i.filename for i in det.images] == ["DSC05028.JPG", "DSC05076.JPG"]
    assert det.tags[1].width == 0.11650000000000001 and det.tags[0].tagType == "apriltag_36h11"
    ob = det.tagObservations[0]
    assert (ob.imageId, ob.tagId) == (0, 137)
    assert ob.corners[3].tolist() == [4744.53369140625, 205.41494750976562]      # UL is last (LL, LR, UR, UL)
    out = tmp_path / "marker_detections.json"
    vio.writeDetectionResult(det, str(out))
    assert out.read_text() == open(src).read()     # byte-identical: quoting, %.17g, indentation, key order


def test_camera_intrinsics_readme_example(tmp_path):
    m = vio.readCameraModel(os.path.join(GOLD, "readme_camera_intrinsics.json"))
    assert m.fx == 8.0752937867635346e+03 and m.cy == 1.9962896554785455e+03
    assert m.distortionCoefficients.tolist() == [-1.8618183262669760e-01, 3.7018092365577054e-01,
                                                 -2.9390604003594177e-04, 4.1533180829908799e-04,
                                                 5.7043887874185996e-02]
    assert (m.verticalResolution, m.horizontalResolution) == (4000, 6000)
    out = tmp_path / "camera_intrinsics.json"
    vio.writeCameraModel(m, str(out))
    m2 = vio.readCameraModel(str(out))
    assert m2.fx == m.fx and m2.distortionCoefficients.tolist() == m.distortionCoefficients.tolist()
    # every scalar is a quoted string, as Boost's write_json emits them
    tree = json.loads(out.read_text())
    assert all(isinstance(tree[k], str) for k in ("fx", "fy", "cx", "cy", "vertical_resolution"))
    assert all(isinstance(v, str) for v in tree["distortion_coefficients"])


def test_reconstruction_readme_example_and_roundtrip(tmp_path):
    src = os.path.join(GOLD, "readme_reconstruction_excerpt.json")
    tree = vio.read_json(src)
    tree["camera_model"] = vio.read_json(os.path.join(GOLD, "readme_camera_intrinsics.json"))
    p = tmp_path / "in.json"
    p.write_text(json.dumps(tree))
    tags, cams, model = vio.parseReconstructions(str(p))
    assert tags[0].q.tolist() == [0.99998768285276463, -0.0026651883409995361, -1.5875056612292212e-05,
                                  0.0041869633189374018]
    assert cams[0].t.tolist() == [1.6566629838776454, -1.0628296493529241, -0.5984995791803972]
    out = tmp_path / "reconstruction.json"
    vio.exportReconstructions(str(out), tags, cams, model)
    text = out.read_text()
    # the tag and camera entries are laid out exactly like the README excerpt
    ref = open(src).read()
    tag_block = ref[ref.index('        {\n            "id": "0",\n            "type"'):ref.index("    ],\n    \"reconstructed_cameras\"")]
    assert tag_block in text
    tags2, cams2, model2 = vio.parseReconstructions(str(out))
    assert tags2[0].t.tolist() == tags[0].t.tolist() and cams2[0].q.tolist() == cams[0].q.tolist()
    assert cams2[0].cameraId == 0 and model2.fy == model.fy
    # reconstructed_marker_corners: world corners LL, LR, UR, UL of every tag (ReconstructionIO.cpp:67-86)
    t = json.loads(text)
    corners = t["reconstructed_marker_corners"]
    assert [c["corner_index"] for c in corners] == ["0", "1", "2", "3"]
    want = tags[0].computeMarkerCorners3D()
    for c, w in zip(corners, want):
        assert [float(v) for v in c["coords"]] == w.tolist()
    assert list(t.keys()) == ["reconstructed_tags", "reconstructed_marker_corners", "reconstructed_cameras",
                              "camera_model"]


def test_reader_errors_and_bare_numbers(tmp_path):
    p = tmp_path / "d.json"
    doc = {"images": [{"filename": "a.jpg", "id": 3}], "tags": [{"id": 1, "tag_type": "t", "width": 0.1, "height": 0.2}],
           "tag_observations": [{"image_id": 3, "tag_id": 1, "observations": [[1, 2], [3, 4], [5, 6], [7, 8.5]]}]}
    p.write_text(json.dumps(doc))                     # bare JSON numbers parse like quoted ones
    det = vio.readDetectionResult(str(p))
    assert det.images[0].imageId == 3 and det.tagObservations[0].corners[3, 1] == 8.5
    doc["tag_observations"][0]["observations"].pop()
    p.write_text(json.dumps(doc))
    with pytest.raises(RuntimeError, match="Unexpected number of values"):   # DetectionIO.cpp:50-51
        vio.readDetectionResult(str(p))
    doc["tag_observations"][0]["observations"] = [[1, 2, 3]] * 4
    p.write_text(json.dumps(doc))
    with pytest.raises(RuntimeError, match="Unexpected number of values"):   # :45-46
        vio.readDetectionResult(str(p))
    cm = {"fx": 1, "fy": 1, "cx": 0, "cy": 0, "distortion_coefficients": [0, 0, 0, 0],
          "vertical_resolution": 1, "horizontal_resolution": 1}
    p.write_text(json.dumps(cm))
    with pytest.raises(RuntimeError, match="Not enough many parameters in vector. Expected 5, got 4"):
        vio.readCameraModel(str(p))
    cm["distortion_coefficients"] = [0] * 6
    p.write_text(json.dumps(cm))
    with pytest.raises(RuntimeError, match="Too many parameters in vector. Expected 5"):
        vio.readCameraModel(str(p))
    # an empty array is written as "" and read back as empty
    vio.writeDetectionResult(vio.DetectionResult(), str(p))
    assert json.loads(p.read_text()) == {"images": "", "tags": "", "tag_observations": ""}
    assert vio.readDetectionResult(str(p)).tagObservations == []


def test_project_directory_roundtrip_is_exact(tmp_path):
    s = synthetic.make_scene(1)
    model, det = synthetic.write_project(s, str(tmp_path))
    det2 = vio.readDetectionResult(str(tmp_path / "marker_detections.json"))
    assert len(det2.tagObservations) == s.n_obs
    for a, b in zip(det.tagObservations, det2.tagObservations):
        assert (a.imageId, a.tagId) == (b.imageId, b.tagId) and np.array_equal(a.corners, b.corners)
    tags, cams, m2 = vio.parseReconstructions(str(tmp_path / "ground_truth.json"))
    assert np.array_equal(np.array([np.r_[tags[t].q, tags[t].t] for t in sorted(tags)]), s.tag_gt)
    assert np.array_equal(np.array([np.r_[cams[c].q, cams[c].t] for c in sorted(cams)]), s.cam_gt)
    assert m2.fx == model.fx


# ---- PnP ---------------------------------------------------------------------------------------------------

INTR = (8075.29, 8083.17, 3016.39, 1996.29)
DIST = (-0.18618, 0.37018, -2.939e-4, 4.153e-4, 0.05704)


@pytest.mark.parametrize("dist", [(0.0,) * 5, DIST])
def test_pnp_recovers_exact_poses(dist):
    rng = np.random.default_rng(7)
    w = 0.1285
    quad = np.array([[-w / 2, -w / 2, 0], [w / 2, -w / 2, 0], [w / 2, w / 2, 0], [-w / 2, w / 2, 0]])
    for _ in range(10):
        R = pnp.rodrigues(rng.normal(size=3) * 0.4)
        t = np.array([rng.normal() * 0.2, rng.normal() * 0.2, 3.0 + 3.0 * rng.random()])
        R2, t2 = pnp.solvePnP(quad, pnp.project(quad, R, t, INTR, dist), INTR, dist)          # one tag (planar, 4 points)
        assert np.abs(R2 - R).max() < 1e-9 and np.abs(t2 - t).max() < 1e-9
        wall = np.c_[rng.uniform(-3, 3, 40), rng.uniform(-1.5, 1.5, 40), rng.normal(0, 0.03, 40)]   # near-planar wall
        tw = t + np.array([0, 0, 4.0])
        R3, t3 = pnp.solvePnPRansac(wall, pnp.project(wall, R, tw, INTR, dist), INTR, dist)
        assert np.abs(R3 - R).max() < 1e-9 and np.abs(t3 - tw).max() < 1e-8
        cloud = rng.uniform(-1, 1, (30, 3))                                                    # general position
        px = pnp.project(cloud, R, t, INTR, dist)
        px[:4] += rng.uniform(50, 200, (4, 2))                                                 # gross outliers
        R4, t4 = pnp.solvePnPRansac(cloud, px, INTR, dist)
        assert np.abs(R4 - R).max() < 1e-9 and np.abs(t4 - t).max() < 1e-9


def test_rotation_helpers():
    rng = np.random.default_rng(3)
    for _ in range(50):
        r = rng.normal(size=3) * rng.choice([1e-9, 0.3, 1.5, 3.1])
        R = pnp.rodrigues(r)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-14) and np.allclose(pnp.rodrigues(pnp.rodrigues_inv(R)), R, atol=1e-9)
        q = pnp.quat_from_R(R)
        assert np.allclose(_quat_to_R(q), R, atol=1e-12)
    with pytest.raises(RuntimeError, match="same number of objectPoints"):
        pnp.solvePnPRansac(np.zeros((5, 3)), np.zeros((4, 2)), INTR, DIST)


# ---- incremental driver (control flow; BA replaced by the oracle) ---------------------------------------


from oracle_reconstructor import OracleReconstructor as _OracleReconstructor  # noqa: E402


def _pose_err(q_a, t_a, q_b, t_b):
    Ra, Rb = _quat_to_R(q_a / np.linalg.norm(q_a)), _quat_to_R(q_b / np.linalg.norm(q_b))
    return float(np.abs(Ra - Rb).max()), float(np.abs(t_a - t_b).max())


def test_start_reconstruction_driver_on_cpu(tmp_path, capsys):
    s = synthetic.make_scene(1, n_cams=6, n_tags=8, visibility=0.7)
    model, det = synthetic.write_project(s, str(tmp_path))
    rec = _OracleReconstructor(vio.readDetectionResult(str(tmp_path / "marker_detections.json")))
    rec.ba_calls = []
    rec.setCameraModel(vio.readCameraModel(str(tmp_path / "camera_intrinsics.json")))
    rec.startReconstruction(2)
    out = capsys.readouterr().out
    assert rec.originTagId == 0                                   # -1 -> lowest tag id (:89-92)
    n_cams = len(rec.reconstructedCameras)
    assert n_cams == 6 and len(rec.reconstructedTags) >= 7
    # cadence: one BA(400, robust) per image, then BA(1500, robust), BA(1500, plain, summary) (:233, :271-277)
    assert [c[:3] for c in rec.ba_calls[:-2]] == [(400, True, False)] * n_cams
    assert [c[:3] for c in rec.ba_calls[-2:]] == [(1500, True, False), (1500, False, True)]
    assert [c[3] for c in rec.ba_calls[:n_cams]] == list(range(1, n_cams + 1))     # one more camera each time
    assert "Starting final bundle adjustment" in out and "Reconstructing image 0/6" in out
    # the first image is the one with the most tags among those that see the origin tag
    first = int(out.split("with id: ")[1].split()[0])
    tags_in = {}
    for ob in det.tagObservations:
        tags_in.setdefault(ob.imageId, set()).add(ob.tagId)
    cands = [i for i in sorted(tags_in) if 0 in tags_in[i]]
    assert first == max(cands, key=lambda i: (len(tags_in[i]), -i))
    # result = the optimum a one-shot BA reaches from the perturbed ground truth (same observations)
    one = _OracleReconstructor(det)
    one.ba_calls = []
    one.setCameraModel(model)
    one.setOriginTagId(0)
    one.setReconstructedTags({t: ReconstructedTag(t, "apriltag_36h11", s.tag_init[t, :4], s.tag_init[t, 4:],
                                                  s.tag_wh[t, 0], s.tag_wh[t, 1]) for t in rec.reconstructedTags})
    one.setReconstructedCameras({c: Camera(c, s.cam_init[c, :4], s.cam_init[c, 4:]) for c in rec.reconstructedCameras})
    one.doBundleAdjustment(1500, 1, True)
    one.doBundleAdjustment(1500, 1, False)
    for t in rec.reconstructedTags:
        er, et = _pose_err(rec.reconstructedTags[t].q, rec.reconstructedTags[t].t, one.reconstructedTags[t].q,
                           one.reconstructedTags[t].t)
        assert er < 1e-5 and et < 1e-5, (t, er, et)
    for c in rec.reconstructedCameras:
        er, et = _pose_err(rec.reconstructedCameras[c].q, rec.reconstructedCameras[c].t, one.reconstructedCameras[c].q,
                           one.reconstructedCameras[c].t)
        assert er < 1e-5 and et < 1e-4, (c, er, et)
    # and it is near the ground truth (0.3 px noise)
    for t in rec.reconstructedTags:
        er, et = _pose_err(rec.reconstructedTags[t].q, rec.reconstructedTags[t].t, s.tag_gt[t, :4], s.tag_gt[t, 4:])
        assert er < 5e-3 and et < 5e-3


def test_start_reconstruction_errors():
    det = vio.DetectionResult([vio.TagImg(0, "a")], [vio.Tag(5, "t", 0.1, 0.1)], [])
    rec = _OracleReconstructor(det)
    rec.ba_calls = []
    rec.setOriginTagId(9)
    with pytest.raises(RuntimeError, match="Could not use tag with id 9 as origin tag, because it was not detected."):
        rec.startReconstruction()
    rec = _OracleReconstructor(det)
    rec.ba_calls = []
    with pytest.raises(RuntimeError, match="No reconstructed tags in image found"):
        rec.startReconstruction()      # origin tag 5 exists but no image observes it
