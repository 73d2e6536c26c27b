cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r4b_a; mkdir -p $O
hipcc -O2 --offload-arch=gfx950 -o /tmp/share_simd tools/micro/share_simd.hip && timeout -k 10 60 /tmp/share_simd > $O/share_simd.txt 2>&1; cat $O/share_simd.txt
timeout -k 10 300 python - > $O/overlap.txt 2>&1 <<'PY'
from visual_marker_mapping_amd import engine as eng
from visual_marker_mapping_amd.synthetic import make_scene
s = make_scene(2)
ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
ba.solve(eng.default_options(robustify=0))
print({k: round(v * 1000, 1) for k, v in ba.debug_overlap(20).items()})
PY
cat $O/overlap.txt
