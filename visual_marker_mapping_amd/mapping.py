"""Command-line mapping step: <project>/camera_intrinsics.json + marker_detections.json -> reconstruction.json.

Mirrors the reference's `visual_marker_mapping` executable (/root/reference/src/main_mapping.cpp:15-96):
same options (--project_path, --start_tag_id), same file names, the same overwrite question, and -- like
the reference -- exit code 0 even after an exception (:90-95), which is printed the same way.

    python -m visual_marker_mapping_amd.mapping --project_path DIR [--start_tag_id N] [--yes]
"""
import argparse
import os
import sys

from . import io as _io
from .tag_reconstructor import TagReconstructor


def main(argv=None):
    ap = argparse.ArgumentParser(description="Allowed options")
    ap.add_argument("--project_path", required=True, help="Path to project to be processed")
    ap.add_argument("--start_tag_id", type=int, default=-1,
                    help="Id of the marker which will be in the origin of the model.")
    ap.add_argument("--yes", action="store_true", help="overwrite an existing reconstruction.json without asking")
    ap.add_argument("--device", type=int, default=0, help="HIP device ordinal")
    a = ap.parse_args(argv)
    try:
        detections = os.path.join(a.project_path, "marker_detections.json")
        intrinsics = os.path.join(a.project_path, "camera_intrinsics.json")
        out = os.path.join(a.project_path, "reconstruction.json")
        max_threads = os.cpu_count() or 4
        if os.path.exists(out) and not a.yes:
            while True:
                sys.stderr.write("Output file '%s' already exists. Overwrite? (y/n) " % out)
                sys.stderr.flush()
                yn = sys.stdin.readline().strip()[:1]
                if yn == "n" or yn == "":
                    print("Exiting!")
                    return 1
                if yn == "y":
                    break
        camera_model = _io.readCameraModel(intrinsics)
        reconstructor = TagReconstructor(_io.readDetectionResult(detections), device=a.device)
        reconstructor.setCameraModel(camera_model)
        reconstructor.setOriginTagId(a.start_tag_id)
        reconstructor.startReconstruction(max_threads)
        _io.exportReconstructions(out, reconstructor.getReconstructedTags(), reconstructor.getReconstructedCameras(),
                                  camera_model)
        print("Wrote %s!" % out)
    except Exception as ex:   # noqa: BLE001 -- the reference catches std::exception and still returns 0
        print("An exception occurred: %s" % ex)
    return 0


if __name__ == "__main__":
    sys.exit(main())
