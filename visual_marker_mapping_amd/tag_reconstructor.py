"""Host-side mirror of the reference's TagReconstructor / CameraModel interface for the hot path.

Same names, argument meaning and error behaviour as
/root/reference/include/visual_marker_mapping/{TagReconstructor,CameraModel,Camera,DetectionResults}.h,
so that the parity tests read like tests of the reference class.  The bodies of the hot-path methods
pack the std::map-style state into flat arrays and call libvmm_ba.so (MI355X); nothing here computes
residuals or solves on the CPU.

startReconstruction (the incremental driver) is host logic around the hot
path; its two PnP initialisations (OpenCV in the reference) live in pnp.py.  Callers may also provide
initial poses through setReconstructedTags / setReconstructedCameras and call doBundleAdjustment directly.
"""
import math
import os

import numpy as np

from . import engine as _engine


# ---- plain data types (DetectionResults.h:10-37, Camera.h:9-18, TagReconstructor.h:15-53) -----------


class TagObservation:
    def __init__(self, imageId=-1, tagId=-1, corners=None):
        self.imageId = int(imageId)
        self.tagId = int(tagId)
        # observed tag corners LL, LR, UR, UL as (u, v)
        self.corners = np.zeros((4, 2)) if corners is None else np.asarray(corners, np.float64).reshape(4, 2)


class TagImg:
    def __init__(self, imageId=-1, filename=""):
        self.imageId = int(imageId)
        self.filename = filename


class Tag:
    def __init__(self, tagId, tagType, width, height):
        self.tagId = int(tagId)
        self.tagType = tagType
        self.width = float(width)
        self.height = float(height)


class DetectionResult:
    def __init__(self, images=None, tags=None, tagObservations=None):
        self.images = list(images or [])
        self.tags = list(tags or [])
        self.tagObservations = list(tagObservations or [])


def _quat_to_R(q):
    """Eigen::Quaterniond::toRotationMatrix for q = (w, x, y, z) (no normalisation)."""
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _quat_mul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                     a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
                     a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


class Camera:
    """World->camera pose; q = (w, x, y, z), default identity (Camera.h:13-14)."""

    def __init__(self, cameraId=-1, q=None, t=None):
        self.cameraId = int(cameraId)
        self.q = np.array([1.0, 0.0, 0.0, 0.0]) if q is None else np.asarray(q, np.float64).reshape(4).copy()
        self.t = np.zeros(3) if t is None else np.asarray(t, np.float64).reshape(3).copy()

    def quat(self):
        return self.q.copy()

    def setQuat(self, quat):
        self.q = np.asarray(quat, np.float64).reshape(4).copy()


class ReconstructedTag:
    """Tag->world pose plus the tag's metric size (TagReconstructor.h:15-53)."""

    def __init__(self, id=-1, tagType="", q=None, t=None, tagWidth=0.0, tagHeight=0.0):
        self.id = int(id)
        self.tagType = tagType
        self.q = np.array([1.0, 0.0, 0.0, 0.0]) if q is None else np.asarray(q, np.float64).reshape(4).copy()
        self.t = np.zeros(3) if t is None else np.asarray(t, np.float64).reshape(3).copy()
        self.tagWidth = float(tagWidth)
        self.tagHeight = float(tagHeight)

    def quat(self):
        return self.q.copy()

    def setQuat(self, quat):
        self.q = np.asarray(quat, np.float64).reshape(4).copy()

    def computeLocalMarkerCorners3D(self):
        w, h = self.tagWidth, self.tagHeight   # TagReconstructor.h:47-50: LL, LR, UR, UL
        return [np.array([-w / 2.0, -h / 2.0, 0.0]), np.array([w / 2.0, -h / 2.0, 0.0]),
                np.array([w / 2.0, h / 2.0, 0.0]), np.array([-w / 2.0, h / 2.0, 0.0])]

    def computeMarkerCorners3D(self):
        R = _quat_to_R(self.q)
        return [R @ p + self.t for p in self.computeLocalMarkerCorners3D()]


class CameraModel:
    """OpenCV pinhole + 5-coefficient distortion (CameraModel.h:12-34)."""

    def __init__(self, fx=0.0, fy=0.0, cx=0.0, cy=0.0, distortionCoefficients=None, verticalResolution=0,
                 horizontalResolution=0):
        self.fx, self.fy, self.cx, self.cy = float(fx), float(fy), float(cx), float(cy)
        self.distortionCoefficients = (np.zeros(5) if distortionCoefficients is None
                                       else np.asarray(distortionCoefficients, np.float64).reshape(5).copy())
        self.verticalResolution = int(verticalResolution)
        self.horizontalResolution = int(horizontalResolution)

    def getK(self):
        """CameraModel::getK, src/CameraModel.cpp:28-36."""
        K = np.eye(3)
        K[0, 0], K[1, 1], K[0, 2], K[1, 2] = self.fx, self.fy, self.cx, self.cy
        return K

    def projectPoint(self, point3D, device=0):
        """CameraModel::projectPoint, src/CameraModel.cpp:6-26: camera-frame point(s) -> pixel(s), including the
        reference's aliasing at :20-23 (the y tangential term is evaluated with the already distorted x).

        Accepts one point (3,) or a batch (n, 3); evaluated by the device kernel.
        """
        pts = np.asarray(point3D, np.float64)
        single = pts.ndim == 1
        uv = _engine.project_points([self.fx, self.fy, self.cx, self.cy], self.distortionCoefficients,
                                    pts.reshape(-1, 3), device=device)
        return uv[0] if single else uv


# ---- the class whose hot path this package replaces --------------------------------------------------


class TagReconstructor:
    """Mirror of visual_marker_mapping::TagReconstructor (TagReconstructor.h:58-146)."""

    def __init__(self, detection_result, device=0):
        self.originTagId = -1                                  # src/TagReconstructor.cpp:70
        self.detectionResults_ = detection_result
        self.reconstructedTags = {}                            # tag id -> ReconstructedTag
        self.reconstructedCameras = {}                         # image id -> Camera
        self.camModel = CameraModel()
        self.device = int(device)
        self.lastSummary = None                                # summary of the last doBundleAdjustment
        self.lastCovariances = None                            # tag id -> 3x3 (last printSummary call)
        self._cached = None                                    # (structure key, BundleAdjuster) of the last call
        self._obs_cache = None                                 # observation list as arrays (see _obs_arrays)
        self._resident = False                                 # device-resident packing (startReconstruction)
        self._full = None                                      # the whole detection set as problem arrays

    # -- trivial accessors (src/TagReconstructor.cpp:75-84, 818-842) --
    def getLowestTag(self):
        tags = self.detectionResults_.tags
        m = tags[0].tagId     # the reference dereferences tags[0] unguarded (:77)
        for t in tags:
            if t.tagId < m:
                m = t.tagId
        return m

    def getReconstructedTags(self):
        return dict(self.reconstructedTags)

    def getReconstructedCameras(self):
        return dict(self.reconstructedCameras)

    def setReconstructedTags(self, tags):
        """Not in the reference API: stands in for the PnP initialisation of startReconstruction."""
        self.reconstructedTags = {int(k): v for k, v in dict(tags).items()}

    def setReconstructedCameras(self, cams):
        """Not in the reference API: stands in for the PnP initialisation of startReconstruction."""
        self.reconstructedCameras = {int(k): v for k, v in dict(cams).items()}

    def getCameraModel(self):
        return self.camModel

    def setCameraModel(self, cameraModel):
        self.camModel = cameraModel

    def setOriginTagId(self, originTagId):
        self.originTagId = int(originTagId)

    def startReconstruction(self, numThreads=1, deviceResident=True):
        """deviceResident: ONE device handle holds every image, tag and observation of the detection result for
        the whole run and each of the N+2 bundle adjustments / prunings only sends an observation mask and the
        poses (vmm_ba_set_observation_mask); False rebuilds a handle whenever the problem structure changes.
        Same results either way (tests/test_gpu_driver.py).

        The incremental driver, src/TagReconstructor.cpp:86-278: origin tag at identity; the image that
        sees the origin tag and the most tags first; then per image: camera pose from the already
        reconstructed tags (PnP+RANSAC), every new tag seen in >= 2 images from its own four corners (PnP),
        robust bundle adjustment (400 iterations), pruning of tags above 2 px; next = the image with the most
        reconstructed tags.  Ends with BA(1500, robust), both prunings, BA(1500, plain, summary).

        The bundle adjustments and the reprojection statistics run on the MI355X through libvmm_ba; the two
        PnP initialisations are host code in pnp.py (OpenCV's role in the reference)."""
        from . import pnp as _pnp
        self._resident = bool(deviceResident)
        try:
            self._start_reconstruction(numThreads, _pnp)
        finally:
            self._resident = False
            self._drop_cached()
            self._full = None

    def _start_reconstruction(self, numThreads, _pnp):
        if self.originTagId == -1:
            self.originTagId = self.getLowestTag()
        intr = (self.camModel.fx, self.camModel.fy, self.camModel.cx, self.camModel.cy)
        dist = tuple(float(v) for v in self.camModel.distortionCoefficients)
        whichImagesObserveTag, tagsInImage, obsOfImage = {}, {}, {}
        for ob in self.detectionResults_.tagObservations:
            whichImagesObserveTag.setdefault(ob.tagId, set()).add(ob.imageId)
            tagsInImage.setdefault(ob.imageId, set()).add(ob.tagId)
            obsOfImage.setdefault(ob.imageId, []).append(ob)
        imageFilenames = {img.imageId: img.filename for img in self.detectionResults_.images}
        tagById = {t.tagId: t for t in self.detectionResults_.tags}

        # the image with the most markers among those that see the origin tag (:117-127)
        maxObservations, curImageId = 0, -1
        for imageId in sorted(whichImagesObserveTag.get(self.originTagId, ())):
            if len(tagsInImage[imageId]) > maxObservations:
                curImageId, maxObservations = imageId, len(tagsInImage[imageId])
        if self.originTagId not in tagById:
            raise RuntimeError("Could not use tag with id %d as origin tag, because it was not detected."
                               % self.originTagId)
        o = tagById[self.originTagId]
        self.reconstructedTags.setdefault(self.originTagId, ReconstructedTag(
            id=self.originTagId, tagType=o.tagType, tagWidth=o.width, tagHeight=o.height))

        while True:
            print("Reconstructing image %d/%d | %s with id: %d" % (
                len(self.reconstructedCameras), len(self.detectionResults_.images),
                imageFilenames.get(curImageId, ""), curImageId))
            pose = self.computeRelativeCameraPoseFromImg(curImageId, intr, dist, obsOfImage.get(curImageId, []))
            if pose is None:
                raise RuntimeError("No reconstructed tags in image found. To reconstruct the image pose "
                                   "already reconstructed markers are needed. This should NOT happen.")
            newCamera = Camera(cameraId=curImageId, q=pose[0], t=pose[1])
            print("   Initialized camera with id %d" % curImageId)
            self.reconstructedCameras.setdefault(curImageId, newCamera)
            cam = self.reconstructedCameras[curImageId]
            Rc, tc = _quat_to_R(cam.q), cam.t
            for ob in obsOfImage.get(curImageId, []):
                if ob.tagId in self.reconstructedTags:
                    continue
                if len(whichImagesObserveTag[ob.tagId]) < 2:
                    print("   Skipping reconstruction of tag %d: Only observed once!" % ob.tagId)
                    continue
                d = tagById[ob.tagId]
                recTag = ReconstructedTag(id=ob.tagId, tagType=d.tagType, tagWidth=d.width, tagHeight=d.height)
                R, t = _pnp.solvePnP(recTag.computeLocalMarkerCorners3D(), ob.corners, intr, dist)   # tag -> camera
                # TMarker2World = extrinsic^-1 * T  (:213-221)
                recTag.setQuat(_pnp.quat_from_R(Rc.T @ R))
                recTag.t = Rc.T @ (t - tc)
                self.reconstructedTags[ob.tagId] = recTag
                print("   Initialized tag with id %d" % ob.tagId)

            self.doBundleAdjustment(400, numThreads, True)
            self.removeBadMarkers(2.0)

            # next: the unreconstructed image with the most reconstructed tags (:236-259)
            maxPairs = 0
            for img in self.detectionResults_.images:
                if img.imageId in self.reconstructedCameras:
                    continue
                n = sum(1 for tid in tagsInImage.get(img.imageId, ()) if tid in self.reconstructedTags)
                if n > maxPairs:
                    maxPairs, curImageId = n, img.imageId
            if maxPairs == 0:
                break
            print("----------------------------------------------------------")

        print("Starting final bundle adjustment")
        self.doBundleAdjustment(1500, numThreads, True, False)
        self.removeBadMarkers(2.0)
        self.removeBadCameras(2.0)
        self.doBundleAdjustment(1500, numThreads, False, True)

    def computeRelativeCameraPoseFromImg(self, imageId, intr, dist, observations=None):
        """src/TagReconstructor.cpp:280-312: (q, t) of the camera from every correspondence between this
        image's detected corners and the corners of already reconstructed tags, or None without any."""
        from . import pnp as _pnp
        if observations is None:
            observations = [ob for ob in self.detectionResults_.tagObservations if ob.imageId == imageId]
        X, px = [], []
        for ob in observations:
            tag = self.reconstructedTags.get(ob.tagId)
            if tag is None:
                continue
            X.extend(tag.computeMarkerCorners3D())
            px.extend(np.asarray(ob.corners, np.float64).reshape(4, 2))
        print("   Reconstructing camera pose from %d 2d/3d correspondences" % len(px))
        if not px:
            return None
        R, t = _pnp.solvePnPRansac(np.asarray(X), np.asarray(px), intr, dist, seed=int(imageId) & 0x7FFFFFFF)
        return _pnp.quat_from_R(R), t

    def moveTagIntoOrigin(self, tagId):
        """src/TagReconstructor.cpp:314-338 (applies the same map to tags AND cameras, as the reference does)."""
        if tagId not in self.reconstructedTags:
            raise RuntimeError("Tag with id %d is not reconstructed." % tagId)
        print("Transforming tag with id %d into origin." % tagId)
        q0 = self.reconstructedTags[tagId].quat()
        qinv = np.array([q0[0], -q0[1], -q0[2], -q0[3]]) / float(q0 @ q0)
        t0 = self.reconstructedTags[tagId].t.copy()
        R = _quat_to_R(qinv)
        for tag in self.reconstructedTags.values():
            tag.setQuat(_quat_mul(qinv, tag.quat()))
            tag.t = R @ (tag.t - t0)
        for cam in self.reconstructedCameras.values():
            cam.setQuat(_quat_mul(qinv, cam.quat()))
            cam.t = R @ (cam.t - t0)
        print("Finished transforming Tags")

    # -- packing of the map state into the flat arrays of the C-ABI --
    def _obs_arrays(self):
        """The observation list as arrays (image id, tag id, 8 pixel coordinates), rebuilt only when the list object
        or its length changes: the incremental driver packs the problem three times per image."""
        obs = self.detectionResults_.tagObservations
        key = (id(obs), len(obs))
        if self._obs_cache is None or self._obs_cache[0] != key:
            img = np.fromiter((ob.imageId for ob in obs), np.int64, len(obs))
            tag = np.fromiter((ob.tagId for ob in obs), np.int64, len(obs))
            px = (np.stack([np.asarray(ob.corners, np.float64).reshape(8) for ob in obs]) if len(obs)
                  else np.zeros((0, 8)))
            self._obs_cache = (key, img, tag, px)
        return self._obs_cache[1:]

    def _pack(self, for_ba):
        """Dense problem arrays exactly as doBundleAdjustment assembles the ceres::Problem
        (src/TagReconstructor.cpp:663-724): tags in map (id) order; cameras with >= 1 reconstructed tag
        in map order (for_ba) or all cameras (statistics); observations whose camera and tag are both
        reconstructed, in file order."""
        if self._resident:
            return self._pack_resident(for_ba)
        tag_ids = sorted(self.reconstructedTags)
        ob_img, ob_tag, ob_px = self._obs_arrays()
        tag_arr = np.asarray(tag_ids, np.int64)
        tag_ok = np.isin(ob_tag, tag_arr)
        images_with_tags = set(np.unique(ob_img[tag_ok]).tolist())                            # :679-684
        cam_ids = [cid for cid in sorted(self.reconstructedCameras)
                   if (cid in images_with_tags or not for_ba)]                               # :689-690
        cam_arr = np.asarray(cam_ids, np.int64)
        keep = tag_ok & np.isin(ob_img, cam_arr)                                              # :699-708
        # ids -> dense indices (both id lists are sorted)
        obs_cam = np.searchsorted(cam_arr, ob_img[keep]).astype(np.int32)
        obs_tag = np.searchsorted(tag_arr, ob_tag[keep]).astype(np.int32)
        obs_px = ob_px[keep]
        cam_qt = np.array([np.r_[self.reconstructedCameras[c].q, self.reconstructedCameras[c].t] for c in cam_ids],
                          np.float64).reshape(-1, 7)
        tag_qt = np.array([np.r_[self.reconstructedTags[t].q, self.reconstructedTags[t].t] for t in tag_ids],
                          np.float64).reshape(-1, 7)
        # the cost functor receives the DETECTION tag's width/height but builds the quad from the
        # reconstructed tag's (:713 vs :718); only the quad enters the arithmetic
        tag_wh = np.array([[self.reconstructedTags[t].tagWidth, self.reconstructedTags[t].tagHeight] for t in tag_ids],
                          np.float64).reshape(-1, 2)
        fixed = tag_ids.index(self.originTagId) if self.originTagId in self.reconstructedTags else -1   # :669-673
        intr = [self.camModel.fx, self.camModel.fy, self.camModel.cx, self.camModel.cy]
        return dict(tag_ids=tag_ids, cam_ids=cam_ids, intr=intr, dist=self.camModel.distortionCoefficients,
                    cam_qt=cam_qt, tag_qt=tag_qt, tag_wh=tag_wh, fixed=fixed,
                    obs_cam=obs_cam, obs_tag=obs_tag, obs_px=np.ascontiguousarray(obs_px, np.float64).reshape(-1, 8),
                    cam_rows=list(range(len(cam_ids))), tag_rows=list(range(len(tag_ids))), mask=None, key=None,
                    n_active=int(len(obs_cam)))

    def _pack_resident(self, for_ba):
        """Device-resident variant used by startReconstruction: the arrays describe EVERY image, tag and
        observation of the detection result (built once), and the step's problem is a mask over the observations
        (camera and tag both reconstructed, :699-708) -- vmm_ba_set_observation_mask.  cam_ids / tag_ids list the
        poses of this step like the cold packing does; cam_rows / tag_rows are their rows in the full arrays.
        Poses that are not reconstructed yet sit at harmless finite defaults; they have no active observation."""
        ob_img, ob_tag, ob_px = self._obs_arrays()
        full = self._full
        if full is None or full["src"] is not self._obs_cache:
            all_cams = np.unique(np.concatenate([ob_img, np.asarray([i.imageId for i in self.detectionResults_.images],
                                                                    np.int64)]))
            det_tags = {t.tagId: t for t in self.detectionResults_.tags}
            all_tags = np.unique(np.concatenate([ob_tag, np.asarray(sorted(det_tags), np.int64)]))
            wh = np.array([[det_tags[t].width, det_tags[t].height] if t in det_tags else [1.0, 1.0]
                           for t in all_tags.tolist()], np.float64).reshape(-1, 2)
            full = dict(src=self._obs_cache, cams=all_cams, tags=all_tags, tag_wh=wh,
                        obs_cam=np.searchsorted(all_cams, ob_img).astype(np.int32),
                        obs_tag=np.searchsorted(all_tags, ob_tag).astype(np.int32),
                        obs_px=np.ascontiguousarray(ob_px, np.float64).reshape(-1, 8))
            self._full = full
        tag_ids = sorted(self.reconstructedTags)
        tag_arr = np.asarray(tag_ids, np.int64)
        rec_cams = np.asarray(sorted(self.reconstructedCameras), np.int64)
        tag_ok = np.isin(ob_tag, tag_arr)
        mask = tag_ok & np.isin(ob_img, rec_cams)
        images_with_tags = set(np.unique(ob_img[mask]).tolist())
        cam_ids = [c for c in rec_cams.tolist() if (c in images_with_tags or not for_ba)]
        cam_rows = np.searchsorted(full["cams"], np.asarray(cam_ids, np.int64)).tolist()
        tag_rows = np.searchsorted(full["tags"], tag_arr).tolist()
        cam_qt = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 1.0]), (len(full["cams"]), 1))   # identity, 1 m in front
        tag_qt = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0.0]), (len(full["tags"]), 1))
        for c, r in zip(cam_ids, cam_rows):
            cam_qt[r] = np.r_[self.reconstructedCameras[c].q, self.reconstructedCameras[c].t]
        tag_wh = full["tag_wh"].copy()
        for t, r in zip(tag_ids, tag_rows):
            tag_qt[r] = np.r_[self.reconstructedTags[t].q, self.reconstructedTags[t].t]
            tag_wh[r] = (self.reconstructedTags[t].tagWidth, self.reconstructedTags[t].tagHeight)
        fixed = (int(np.searchsorted(full["tags"], self.originTagId))
                 if self.originTagId in self.reconstructedTags else -1)
        intr = [self.camModel.fx, self.camModel.fy, self.camModel.cx, self.camModel.cy]
        key = ("resident", id(full), fixed, tuple(float(v) for v in intr),
               tuple(float(v) for v in self.camModel.distortionCoefficients), tag_wh.tobytes())
        return dict(tag_ids=tag_ids, cam_ids=cam_ids, cam_rows=cam_rows, tag_rows=tag_rows, intr=intr,
                    dist=self.camModel.distortionCoefficients, cam_qt=cam_qt, tag_qt=tag_qt, tag_wh=tag_wh,
                    fixed=fixed, obs_cam=full["obs_cam"], obs_tag=full["obs_tag"], obs_px=full["obs_px"],
                    mask=mask, key=key, n_active=int(mask.sum()))

    def _engine_for(self, p, **kw):
        """A device handle for the packed problem.  The handle of the previous call is kept and reused when the
        problem structure (ids, observations, constants) is unchanged -- the usual case for the statistics that
        follow every bundle adjustment of the incremental driver -- so only the 7 doubles per pose travel."""
        if p.get("key") is not None:
            key = (p["key"], tuple(sorted(kw.items())))
        else:
            import hashlib
            h = hashlib.blake2b(digest_size=16)
            for a in (np.asarray(p["cam_ids"], np.int64), np.asarray(p["tag_ids"], np.int64), p["obs_cam"],
                      p["obs_tag"], p["obs_px"], p["tag_wh"], np.asarray(p["intr"], np.float64),
                      np.asarray(p["dist"], np.float64)):
                h.update(np.ascontiguousarray(a).tobytes())
                h.update(b"|")
            key = (h.hexdigest(), int(p["fixed"]), tuple(sorted(kw.items())))
        if self._cached is not None and self._cached[0] == key and not os.environ.get("VMM_BA_NO_HANDLE_CACHE"):
            ba = self._cached[1]
            ba.set_state(p["cam_qt"], p["tag_qt"])
            ba.set_observation_mask(p.get("mask"))
            return ba
        self._drop_cached()
        ba = _engine.BundleAdjuster(p["intr"], p["dist"], p["cam_qt"], p["tag_qt"], p["tag_wh"], p["fixed"],
                                    p["obs_cam"], p["obs_tag"], p["obs_px"], device=self.device, **kw)
        if p.get("mask") is not None:
            ba.set_observation_mask(p["mask"])
        self._cached = (key, ba)
        return ba

    def _drop_cached(self):
        if self._cached is not None:
            self._cached[1].close()
            self._cached = None

    def close(self):
        """Releases the cached device handle (also done on garbage collection)."""
        self._drop_cached()

    def __del__(self):
        try:
            self._drop_cached()
        except Exception:   # noqa: BLE001 -- interpreter shutdown
            pass

    # -- the hot path --
    def doBundleAdjustment(self, maxNumIterations, ceresThreads=1, robustify=True, printSummary=False,
                           elimination=_engine.ELIM_AUTO):
        """src/TagReconstructor.cpp:646-743.  Poses are updated in place like the reference's map nodes."""
        p = self._pack(for_ba=True)
        if len(p["cam_ids"]) == 0 or len(p["tag_ids"]) == 0 or p["n_active"] == 0:
            # Ceres solves an empty problem trivially: CONVERGENCE, nothing changes
            print("Solution %d" % _engine.CONVERGENCE)
            self.lastSummary = {"termination_type": _engine.CONVERGENCE, "iterations": 1}
            return
        ba = self._engine_for(p, elimination=elimination)
        try:
            opts = _engine.default_options(max_num_iterations=int(maxNumIterations), robustify=int(bool(robustify)),
                                           num_threads=int(ceresThreads))
            summary = ba.solve(opts, trace_capacity=int(maxNumIterations) + 2 if printSummary else 0)
            cam, tag = ba.get_state()
            cov = ba.tag_translation_covariance(robustify, opts.huber_a) if printSummary else None   # :744-760
        except Exception:
            self._drop_cached()
            raise
        for k, cid in zip(p["cam_rows"], p["cam_ids"]):
            self.reconstructedCameras[cid].q = cam[k, :4].copy()
            self.reconstructedCameras[cid].t = cam[k, 4:].copy()
        for k, tid in zip(p["tag_rows"], p["tag_ids"]):
            self.reconstructedTags[tid].q = tag[k, :4].copy()
            self.reconstructedTags[tid].t = tag[k, 4:].copy()
        self.lastSummary = summary
        print("Solution %d" % summary["termination_type"])                                   # :740
        if printSummary:                                                                     # :741-742
            print("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius")
            for it in summary["trace"]:
                print("%4d  %.6e  %9.2e  %9.2e  %9.2e  %9.2e  %9.2e" % (
                    it["iteration"], it["cost"], it["cost_change"], it["gradient_max_norm"], it["step_norm"],
                    it["relative_decrease"], it["trust_region_radius"]))
            print("Cost: initial %.6e final %.6e; iterations %d; time in solver %.4f s" % (
                summary["initial_cost"], summary["final_cost"], summary["iterations"], summary["time_solve_s"]))
            print("Time (s): residual + Jacobian evaluation %.6f, elimination + rank-k update %.6f, linear solver %.6f, "
                  "step + candidate cost %.6f, trust-region control %.6f" % (
                      summary["time_eval_s"], summary["time_eliminate_s"], summary["time_factor_solve_s"],
                      summary["time_step_s"], summary["time_control_s"]))
            # covariance report, :761-782 (Eigen prints a row vector with single spaces, 6 significant digits)
            self.lastCovariances = {}
            avg_diag = np.zeros(3)
            for k, tid in zip(p["tag_rows"], p["tag_ids"]):
                self.lastCovariances[tid] = cov[k].copy()
                diag = np.diag(cov[k])
                std = np.sqrt(diag)
                print("StdDev of tag %d: %s | StdDevNorm: %s" % (
                    tid, " ".join("%.6g" % v for v in std), "%.6g" % math.sqrt(np.linalg.norm(std))))
                avg_diag += diag
            avg_diag /= len(p["tag_ids"])
            print("Marker Position RMS = %.6g" % np.linalg.norm(np.sqrt(avg_diag)))

    def doBundleAdjustment_points(self, maxNumIterations, ceresThreads=1, printSummary=False,
                                  elimination=_engine.ELIM_AUTO):
        """src/TagReconstructor.cpp:457-644 (`#if 0` in the reference): the same bundle adjustment with every
        reconstructed tag replaced by its four world corners as free 3-D points (OpenCVReprojectionError,
        TagReconstructionCostFunction.h:9-84; 3x3 landmark blocks, SPARSE_SCHUR ordering :492,534-535), the origin
        tag's corners constant (:494-497), no loss (:556).  Cameras are written back as at :583-605, tag poses are
        rebuilt from the optimised corners as at :608-639 (with today's corner order, see vmm_ba.h); the corners
        themselves are kept in self.lastPoints (tag id -> 4 x 3)."""
        p = self._pack(for_ba=True)
        if len(p["cam_ids"]) == 0 or len(p["tag_ids"]) == 0 or p["n_active"] == 0:
            print("Solution %d" % _engine.CONVERGENCE)
            self.lastSummary = {"termination_type": _engine.CONVERGENCE, "iterations": 1}
            return
        ba = self._engine_for(p, elimination=elimination, landmarks=_engine.LANDMARK_POINTS)
        try:
            opts = _engine.default_options(max_num_iterations=int(maxNumIterations), robustify=0,
                                           num_threads=int(ceresThreads))
            summary = ba.solve(opts, trace_capacity=int(maxNumIterations) + 2 if printSummary else 0)
            cam, tag = ba.get_state()
            pts = ba.get_points()
        except Exception:
            self._drop_cached()
            raise
        for k, cid in zip(p["cam_rows"], p["cam_ids"]):
            self.reconstructedCameras[cid].q = cam[k, :4].copy()
            self.reconstructedCameras[cid].t = cam[k, 4:].copy()
        self.lastPoints = {}
        for k, tid in zip(p["tag_rows"], p["tag_ids"]):
            self.reconstructedTags[tid].q = tag[k, :4].copy()
            self.reconstructedTags[tid].t = tag[k, 4:].copy()
            self.lastPoints[tid] = pts[k].copy()
        self.lastSummary = summary
        print("Solution %d" % summary["termination_type"])                                   # :579
        print("Cost: initial %.6e final %.6e; iterations %d; time in solver %.4f s" % (     # :580-581 (FullReport)
            summary["initial_cost"], summary["final_cost"], summary["iterations"], summary["time_solve_s"]))

    # -- reprojection statistics + pruning (src/TagReconstructor.cpp:340-455, 786-816) --
    def _stats(self, per_corner):
        p = self._pack(for_ba=False)
        if len(p["cam_ids"]) == 0 or len(p["tag_ids"]) == 0:
            return p, np.zeros(0), np.zeros(0), 0.0, np.zeros((0, 8))
        ba = self._engine_for(p, elimination=_engine.ELIM_AUTO)
        try:
            pc, pt, avg, corner = ba.reprojection_stats(per_corner=per_corner)
        except Exception:
            self._drop_cached()
            raise
        return p, pc, pt, avg, corner

    def computeReprojectionErrorPerImg(self):
        """:340-385 -- image id -> mean corner reprojection error; -1.0 for a camera without observations."""
        p, pc, _, _, _ = self._stats(False)
        return {cid: float(pc[k]) for k, cid in zip(p["cam_rows"], p["cam_ids"])}

    def computeReprojectionErrorPerTag(self):
        """:387-428 -- returns (tag id -> mean error, avg).  The reference returns avg through a reference
        parameter; tags without observations do not appear in the map."""
        p, _, pt, avg, _ = self._stats(False)
        return {tid: float(pt[k]) for k, tid in zip(p["tag_rows"], p["tag_ids"]) if not math.isnan(pt[k])}, float(avg)

    def computeReprojectionErrorPerCorner(self):
        """:430-455 -- list of signed (du, dv) per detected corner, observation order."""
        p, _, _, _, corner = self._stats(True)
        if p.get("mask") is not None:
            corner = corner[p["mask"]]   # resident packing: only the observations of this step
        return [corner.reshape(-1, 2)[i].copy() for i in range(corner.size // 2)]

    def removeBadMarkers(self, threshold):
        """:786-802."""
        reperrors, _avg = self.computeReprojectionErrorPerTag()
        for tid in sorted(reperrors):
            if reperrors[tid] > threshold and tid != self.originTagId:
                print("Removing bad marker with id %d and reprojection error %g" % (tid, reperrors[tid]))
                del self.reconstructedTags[tid]

    def removeBadCameras(self, threshold):
        """:804-816."""
        rep = self.computeReprojectionErrorPerImg()
        for cid in sorted(rep):
            if rep[cid] > threshold or rep[cid] < 0:
                print("Removing bad camera with id %d and reprojection error %g" % (cid, rep[cid]))
                del self.reconstructedCameras[cid]


def detection_result_from_arrays(obs_cam, obs_tag, obs_px, tag_wh, n_cams, tag_type="apriltag_36h11"):
    """Builds a DetectionResult (DetectionResults.h:32-37) from flat arrays; image id = camera index."""
    images = [TagImg(i, "img_%05d.jpg" % i) for i in range(n_cams)]
    tags = [Tag(t, tag_type, tag_wh[t][0], tag_wh[t][1]) for t in range(len(tag_wh))]
    obs = [TagObservation(int(c), int(t), np.asarray(px, np.float64).reshape(4, 2))
           for c, t, px in zip(obs_cam, obs_tag, obs_px)]
    return DetectionResult(images, tags, obs)
