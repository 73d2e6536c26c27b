"""TagReconstructor whose three libvmm_ba calls (solve, state, statistics) are served by the CPU oracle.

TEST INFRASTRUCTURE ONLY: lets the CPU suite exercise startReconstruction's control flow without a GPU and
gives the GPU suite an independent run of the whole incremental driver to compare against.
"""
from visual_marker_mapping_amd.tag_reconstructor import TagReconstructor


class OracleReconstructor(TagReconstructor):
    def __init__(self, detection_result, device=0):
        super().__init__(detection_result, device)
        self.ba_calls = []

    def startReconstruction(self, numThreads=1, deviceResident=False):
        # the oracle solves the problem of each step as packed from scratch (no device handle, no mask)
        return super().startReconstruction(numThreads, deviceResident=False)

    def doBundleAdjustment(self, maxNumIterations, ceresThreads=1, robustify=True, printSummary=False, **kw):
        from oracle import oracle as O
        p = self._pack(for_ba=True)
        self.ba_calls.append((maxNumIterations, bool(robustify), bool(printSummary), len(p["cam_ids"]), len(p["tag_ids"])))
        sc = O.Scene(p["intr"], p["dist"], p["cam_qt"], p["tag_qt"], p["tag_wh"], p["fixed"], p["obs_cam"],
                     p["obs_tag"], p["obs_px"])
        summ, _trace = O.solve(sc, O.default_options(max_num_iterations=maxNumIterations, robustify=int(robustify)))
        cam, tag = sc.cam_qt, sc.tag_qt      # solved in place
        for k, cid in enumerate(p["cam_ids"]):
            self.reconstructedCameras[cid].q, self.reconstructedCameras[cid].t = cam[k, :4].copy(), cam[k, 4:].copy()
        for k, tid in enumerate(p["tag_ids"]):
            self.reconstructedTags[tid].q, self.reconstructedTags[tid].t = tag[k, :4].copy(), tag[k, 4:].copy()
        self.lastSummary = summ

    def _stats(self, per_corner):
        from oracle import oracle as O
        p = self._pack(for_ba=False)
        sc = O.Scene(p["intr"], p["dist"], p["cam_qt"], p["tag_qt"], p["tag_wh"], p["fixed"], p["obs_cam"],
                     p["obs_tag"], p["obs_px"])
        pc, pt, avg, corner = O.reprojection_stats(sc)
        return p, pc, pt, avg, corner
