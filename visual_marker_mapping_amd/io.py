"""On-disk formats at the boundary of the mapping step.

Mirrors, with the reference's function names,
  readDetectionResult / writeDetectionResult   /root/reference/src/DetectionIO.cpp:12-59, 61-118
  readCameraModel                              /root/reference/src/CameraUtilities.cpp:59-66 (fields :45-58)
  exportReconstructions / parseReconstructions /root/reference/src/ReconstructionIO.cpp:57-97, 129-142
so that one project directory (camera_intrinsics.json, marker_detections.json -> reconstruction.json) can
drive the reference binary and this package alike.

The reference writes through Boost property_tree's write_json, which has no number type: EVERY scalar is
emitted as a quoted string, doubles with 17 significant digits ("%.17g"), and an empty array or object
as "".  The writers here produce that text layout (4-space indentation, ": " separators, keys in the
reference's insertion order); the readers accept quoted and bare numbers alike, as ptree's
get<int>/get<double> would after parsing either.  The layout is restated from the format description
(README.md:130-271 of the reference) -- Boost is not installed here, so byte identity with a Boost-written
file is not pinned by a fixture.
"""
import json
import os

import numpy as np

from .tag_reconstructor import (Camera, CameraModel, DetectionResult, ReconstructedTag, Tag, TagImg,
                                TagObservation)


# ---- property_tree-style JSON text ----------------------------------------------------------------------


def _scalar(v):
    if isinstance(v, str):
        return json.dumps(v)
    if isinstance(v, (bool, np.bool_)):
        return '"true"' if v else '"false"'
    if isinstance(v, (int, np.integer)):
        return '"%d"' % int(v)
    return '"%s"' % ("%.17g" % float(v))


def _is_container(v):
    return isinstance(v, (dict, list, tuple)) and len(v) > 0


def _emit(node, indent, out):
    """Pretty layout of the reference's output files as README.md:155-263 shows them: 4-space indentation,
    `"key": "scalar"` on one line, and a child array/object opening on its own line below `"key":`."""
    pad = "    " * indent
    if isinstance(node, dict) and node:
        out.append("{\n")
        items = list(node.items())
        for i, (k, v) in enumerate(items):
            out.append(pad + "    " + json.dumps(k) + ":")
            out.append("\n" + pad + "    " if _is_container(v) else " ")
            _emit(v, indent + 1, out)
            out.append(",\n" if i + 1 < len(items) else "\n")
        out.append(pad + "}")
    elif isinstance(node, (list, tuple)) and node:
        out.append("[\n")
        for i, v in enumerate(node):
            out.append(pad + "    ")
            _emit(v, indent + 1, out)
            out.append(",\n" if i + 1 < len(node) else "\n")
        out.append(pad + "]")
    elif isinstance(node, (dict, list, tuple)):
        out.append('""')   # property_tree has no empty array/object: an empty node is an empty string
    else:
        out.append(_scalar(node))


def write_json(path, tree):
    """boost::property_tree::json_parser::write_json layout (all scalars quoted)."""
    out = []
    _emit(tree, 0, out)
    out.append("\n")
    with open(path, "w") as f:
        f.write("".join(out))


def read_json(path):
    with open(path) as f:
        return json.load(f)


def _children(node, key):
    """get_child(key) as a list; write_json stores an empty array as ""."""
    if key not in node:
        raise RuntimeError("No such node (%s)" % key)
    v = node[key]
    if v == "" or v is None:
        return []
    return v


def _get(node, key, typ):
    if key not in node:
        raise RuntimeError("No such node (%s)" % key)
    v = node[key]
    try:
        if typ is int:
            return int(v) if not isinstance(v, str) else int(v.strip())
        if typ is float:
            return float(v)
        return str(v)
    except (TypeError, ValueError):
        raise RuntimeError('conversion of data to type "%s" failed' % typ.__name__)


def _vector(node, n):
    """propertyTree2EigenMatrix for a static vector (PropertyTreeUtilities.h:101-123)."""
    vals = [float(v) for v in (node if node != "" else [])]
    if len(vals) > n:
        raise RuntimeError("Too many parameters in vector. Expected %d!" % n)
    if len(vals) < n:
        raise RuntimeError("Not enough many parameters in vector. Expected %d, got %d!" % (n, len(vals)))
    return np.array(vals, np.float64)


# ---- camera_intrinsics.json ----------------------------------------------------------------------------


def cameraModelToPropertyTree(m):
    return {"fx": float(m.fx), "fy": float(m.fy), "cx": float(m.cx), "cy": float(m.cy),
            "distortion_coefficients": [float(v) for v in m.distortionCoefficients],
            "vertical_resolution": int(m.verticalResolution),
            "horizontal_resolution": int(m.horizontalResolution)}


def propertyTreeToCameraModel(t):
    return CameraModel(fx=_get(t, "fx", float), fy=_get(t, "fy", float), cx=_get(t, "cx", float),
                       cy=_get(t, "cy", float),
                       distortionCoefficients=_vector(_children(t, "distortion_coefficients"), 5),
                       horizontalResolution=_get(t, "horizontal_resolution", int),
                       verticalResolution=_get(t, "vertical_resolution", int))


def readCameraModel(path):
    return propertyTreeToCameraModel(read_json(path))


def writeCameraModel(model, path):
    write_json(path, cameraModelToPropertyTree(model))


# ---- marker_detections.json ----------------------------------------------------------------------------


def readDetectionResult(path):
    t = read_json(path)
    res = DetectionResult()
    for pt in _children(t, "images"):
        res.images.append(TagImg(imageId=_get(pt, "id", int), filename=_get(pt, "filename", str)))
    for pt in _children(t, "tags"):
        res.tags.append(Tag(_get(pt, "id", int), _get(pt, "tag_type", str), _get(pt, "width", float),
                            _get(pt, "height", float)))
    for pt in _children(t, "tag_observations"):
        corners = []
        for corner in _children(pt, "observations"):
            vals = [float(v) for v in corner]
            if len(vals) != 2:
                raise RuntimeError("Unexpected number of values")
            corners.append(vals)
        if len(corners) != 4:
            raise RuntimeError("Unexpected number of values")
        res.tagObservations.append(TagObservation(_get(pt, "image_id", int), _get(pt, "tag_id", int), corners))
    return res


def writeDetectionResult(result, path):
    base = os.path.dirname(os.path.abspath(path))
    images = []
    for img in result.images:
        fn = img.filename
        if os.path.isabs(fn):
            fn = os.path.relpath(fn, base)   # filenames are stored relative to the project directory
        images.append({"filename": fn, "id": img.imageId})
    tags = [{"id": t.tagId, "tag_type": t.tagType, "width": t.width, "height": t.height} for t in result.tags]
    obs = [{"image_id": o.imageId, "tag_id": o.tagId,
            "observations": [[float(o.corners[i][0]), float(o.corners[i][1])] for i in range(4)]}
           for o in result.tagObservations]
    write_json(path, {"images": images, "tags": tags, "tag_observations": obs})
    return True


# ---- reconstruction.json ---------------------------------------------------------------------------------


def exportReconstructions(path, reconstructedTags, reconstructedCameras, camModel):
    tags, corners, cams = [], [], []
    for _, tag in sorted(reconstructedTags.items()):
        tags.append({"id": tag.id, "type": tag.tagType, "width": tag.tagWidth, "height": tag.tagHeight,
                     "rotation": [float(v) for v in tag.q], "translation": [float(v) for v in tag.t]})
    for _, tag in sorted(reconstructedTags.items()):
        for i, c in enumerate(tag.computeMarkerCorners3D()):
            corners.append({"marker_id": tag.id, "corner_index": i, "coords": [float(v) for v in c]})
    for _, cam in sorted(reconstructedCameras.items()):
        cams.append({"id": cam.cameraId, "rotation": [float(v) for v in cam.q],
                     "translation": [float(v) for v in cam.t]})
    write_json(path, {"reconstructed_tags": tags, "reconstructed_marker_corners": corners,
                      "reconstructed_cameras": cams, "camera_model": cameraModelToPropertyTree(camModel)})


def parseReconstructions(path):
    """Returns (reconstructedTags, reconstructedCameras, camModel).

    Unlike the reference's importReconstructedCameras (which never restores cameraId and so collapses all
    cameras onto key -1), the camera ids written by exportReconstructions are read
    back."""
    t = read_json(path)
    tags, cams = {}, {}
    for pt in _children(t, "reconstructed_tags"):
        i = _get(pt, "id", int)
        tags[i] = ReconstructedTag(id=i, tagType=_get(pt, "type", str), q=_vector(_children(pt, "rotation"), 4),
                                   t=_vector(_children(pt, "translation"), 3),
                                   tagWidth=_get(pt, "width", float), tagHeight=_get(pt, "height", float))
    for pt in _children(t, "reconstructed_cameras"):
        i = _get(pt, "id", int)
        cams[i] = Camera(cameraId=i, q=_vector(_children(pt, "rotation"), 4),
                         t=_vector(_children(pt, "translation"), 3))
    return tags, cams, propertyTreeToCameraModel(_children(t, "camera_model"))
