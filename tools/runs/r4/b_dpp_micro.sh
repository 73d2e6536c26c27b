# DPP pivot-block primitives: correctness against the host and cycles on one wave (tools/micro/dpp_chol.hip)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -I visual_marker_mapping_amd/csrc tools/micro/dpp_chol.hip -o /tmp/dpp_chol 2>/dev/null || exit 1
timeout -k 5 60 /tmp/dpp_chol | tee gpurun_out/r4b/dpp_chol.txt
