#!/usr/bin/env python3
"""Writes integration/visual_marker_mapping.patch: the binding a maintainer of visual_marker_mapping applies so that
TagReconstructor / CameraModel run on libvmm_ba (MI355X).

The patch is purely additive: the five method bodies it serves --
    TagReconstructor::computeReprojectionErrorPerImg / PerTag / PerCorner   src/TagReconstructor.cpp:340-455
    TagReconstructor::doBundleAdjustment (+ the covariance report)          src/TagReconstructor.cpp:646-784
    CameraModel::projectPoint                                               src/CameraModel.cpp:6-26
-- stay in the files under `#else`; with -DVMM_BA_ROOT=<this repository> CMake defines
VISUAL_MARKER_MAPPING_WITH_VMM_BA and the bodies become calls into include/vmm_ba_adapter.hpp.

The TagReconstructor owns ONE device-resident handle for its whole life (vmm_ba_adapter::Resident, a guarded
`mutable std::unique_ptr` next to its other members, include/visual_marker_mapping/TagReconstructor.h:134-145): it is
built on first use from detectionResults_ and camModel, dropped by setCameraModel, and every one of the N+2 bundle
adjustments and of the statistics calls of startReconstruction (src/TagReconstructor.cpp:233,236,271-277) only sends
poses and an observation mask -- no vmm_ba_create per call.

Usage:  python integration/make_patch.py [/root/reference]
The reference tree is only read; the edits are made on a temporary copy and `diff -u` writes the patch.
"""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
GUARD = "VISUAL_MARKER_MAPPING_WITH_VMM_BA"

BODIES = {
    "const std::map<int, double> TagReconstructor::computeReprojectionErrorPerImg() const": """\
    return vmmBa().reprojectionStatistics(reconstructedTags, reconstructedCameras, originTagId, false).per_img;
""",
    "const std::map<int, double> TagReconstructor::computeReprojectionErrorPerTag(double& avg) const": """\
    const auto st = vmmBa().reprojectionStatistics(reconstructedTags, reconstructedCameras, originTagId, false);
    avg = st.avg;
    return st.per_tag;
""",
    "const std::vector<Eigen::Vector2d> TagReconstructor::computeReprojectionErrorPerCorner() const": """\
    const auto st = vmmBa().reprojectionStatistics(reconstructedTags, reconstructedCameras, originTagId, true);
    std::vector<Eigen::Vector2d> out;
    for (const auto& e : st.per_corner)
        out.emplace_back(e[0], e[1]);
    return out;
""",
    "void TagReconstructor::doBundleAdjustment(": """\
    vmmBa().doBundleAdjustment(reconstructedTags, reconstructedCameras, originTagId, maxNumIterations, ceresThreads,
        robustify, printSummary);
""",
    "Eigen::Vector2d CameraModel::projectPoint(const Eigen::Vector3d& point3D) const": """\
    const auto uv = vmm_ba_adapter::projectPoint(*this, point3D.x(), point3D.y(), point3D.z());
    return Eigen::Vector2d(uv[0], uv[1]);
""",
}

# PUBLIC: the guard adds a member to class TagReconstructor, so every target that includes its header must see it
CMAKE_BLOCK = """
# MI355X bundle adjustment: cmake -DVMM_BA_ROOT=<checkout of the libvmm_ba repository>
if(VMM_BA_ROOT)
target_compile_definitions(visual_marker_mapping_lib PUBLIC %s)
target_include_directories(visual_marker_mapping_lib PUBLIC ${VMM_BA_ROOT}/include)
target_link_libraries(visual_marker_mapping_lib ${VMM_BA_ROOT}/visual_marker_mapping_amd/libvmm_ba.so)
endif(VMM_BA_ROOT)
""" % GUARD

# include/visual_marker_mapping/TagReconstructor.h: the resident handle as a member (behind `CameraModel camModel;`)
HEADER_MEMBERS = """\
#ifdef %s
    // The MI355X engine's device-resident problem (vmm_ba_adapter::Resident<DetectionResult, CameraModel>, defined in
    // src/TagReconstructor.cpp): built on first use from detectionResults_ and camModel, dropped by setCameraModel.
    struct VmmBaResident;
    struct VmmBaResidentDeleter
    {
        void operator()(VmmBaResident* p) const;
    };
    mutable std::unique_ptr<VmmBaResident, VmmBaResidentDeleter> vmmBaResident_;
    VmmBaResident& vmmBa() const;
#endif
""" % GUARD

# src/TagReconstructor.cpp: the member's definition, in front of the first method that uses it
SOURCE_RESIDENT = """\
#ifdef %s
struct TagReconstructor::VmmBaResident : vmm_ba_adapter::Resident<DetectionResult, CameraModel>
{
    using vmm_ba_adapter::Resident<DetectionResult, CameraModel>::Resident;
};
void TagReconstructor::VmmBaResidentDeleter::operator()(VmmBaResident* p) const { delete p; }
TagReconstructor::VmmBaResident& TagReconstructor::vmmBa() const
{
    if (!vmmBaResident_)
        vmmBaResident_.reset(new VmmBaResident(detectionResults_, camModel));
    return *vmmBaResident_;
}
#endif
//-----------------------------------------------------------------------------
""" % GUARD


def wrap_body(lines, signature, new_body):
    """Inserts `#ifdef GUARD <new_body> #else` behind the opening brace of the function whose signature starts
    with `signature`, and `#endif` in front of its closing brace."""
    start = next(i for i, l in enumerate(lines) if l.startswith(signature))
    open_i = next(i for i in range(start, len(lines)) if lines[i].strip() == "{")
    depth, close_i = 0, None
    for i in range(open_i, len(lines)):
        depth += lines[i].count("{") - lines[i].count("}")
        if depth == 0:
            close_i = i
            break
    assert close_i is not None and lines[close_i].strip() == "}", (signature, close_i)
    lines.insert(close_i, "#endif\n")
    lines[open_i + 1:open_i + 1] = ["#ifdef %s\n" % GUARD] + new_body.splitlines(True) + ["#else\n"]


def include_adapter(lines, after_prefix):
    last = max(i for i, l in enumerate(lines) if l.startswith(after_prefix))
    lines[last + 1:last + 1] = ["#ifdef %s\n" % GUARD, '#include "vmm_ba_adapter.hpp"\n', "#endif\n"]


FILES = ("CMakeLists.txt", "include/visual_marker_mapping/TagReconstructor.h", "src/CameraModel.cpp",
         "src/TagReconstructor.cpp")


def patched_tree(ref, dst):
    for rel in FILES:
        os.makedirs(os.path.dirname(os.path.join(dst, rel)), exist_ok=True)
        shutil.copy(os.path.join(ref, rel), os.path.join(dst, rel))
    p = os.path.join(dst, "src/TagReconstructor.cpp")
    lines = open(p).readlines()
    for sig, body in BODIES.items():
        if "TagReconstructor::" in sig:
            wrap_body(lines, sig, body)
    include_adapter(lines, "#include <")
    # the resident member's definition in front of the first statistics method
    k = next(i for i, l in enumerate(lines)
             if l.startswith("const std::map<int, double> TagReconstructor::computeReprojectionErrorPerImg() const"))
    lines[k:k] = SOURCE_RESIDENT.splitlines(True)
    # setCameraModel: another camera model is another device problem
    k = next(i for i, l in enumerate(lines) if l.startswith("void TagReconstructor::setCameraModel("))
    k = next(i for i in range(k, len(lines)) if lines[i].strip() == "camModel = cameraModel;")
    lines[k + 1:k + 1] = ["#ifdef %s\n" % GUARD, "    vmmBaResident_.reset();\n", "#endif\n"]
    open(p, "w").writelines(lines)
    p = os.path.join(dst, "include/visual_marker_mapping/TagReconstructor.h")
    lines = open(p).readlines()
    k = next(i for i, l in enumerate(lines) if l.strip() == "CameraModel camModel;")
    lines[k + 1:k + 1] = HEADER_MEMBERS.splitlines(True)
    k = max(i for i, l in enumerate(lines) if l.startswith("#include <"))
    lines[k + 1:k + 1] = ["#ifdef %s\n" % GUARD, "#include <memory>\n", "#endif\n"]
    open(p, "w").writelines(lines)
    p = os.path.join(dst, "src/CameraModel.cpp")
    lines = open(p).readlines()
    wrap_body(lines, "Eigen::Vector2d CameraModel::projectPoint", BODIES[
        "Eigen::Vector2d CameraModel::projectPoint(const Eigen::Vector3d& point3D) const"])
    include_adapter(lines, "#include ")
    open(p, "w").writelines(lines)
    p = os.path.join(dst, "CMakeLists.txt")
    lines = open(p).readlines()
    i = next(k for k, l in enumerate(lines) if l.startswith("target_link_libraries(visual_marker_mapping_lib umich_apriltags)"))
    j = next(k for k in range(i, len(lines)) if lines[k].startswith("endif(BUILD_UMICH)"))
    lines[j + 1:j + 1] = CMAKE_BLOCK.splitlines(True)
    open(p, "w").writelines(lines)


def make_patch(ref):
    with tempfile.TemporaryDirectory() as tmp:
        a, b = os.path.join(tmp, "a"), os.path.join(tmp, "b")
        for rel in FILES:
            os.makedirs(os.path.dirname(os.path.join(a, rel)), exist_ok=True)
            shutil.copy(os.path.join(ref, rel), os.path.join(a, rel))
        patched_tree(ref, b)
        text = []
        for rel in FILES:
            out = subprocess.run(["diff", "-U2", "--label", "a/" + rel, "--label", "b/" + rel, os.path.join("a", rel),
                                  os.path.join("b", rel)], cwd=tmp, stdout=subprocess.PIPE, universal_newlines=True)
            assert out.returncode == 1, (rel, out.returncode)
            text.append("diff --git a/%s b/%s\n" % (rel, rel))
            text.append(out.stdout)
        return "".join(text)


if __name__ == "__main__":
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    text = make_patch(ref)
    with open(os.path.join(HERE, "visual_marker_mapping.patch"), "w") as f:
        f.write(text)
    print("wrote integration/visual_marker_mapping.patch: %d lines, %d added, %d removed" % (
        text.count("\n"), sum(l.startswith("+") and not l.startswith("+++") for l in text.splitlines()),
        sum(l.startswith("-") and not l.startswith("---") for l in text.splitlines())))
