# Round 3: sparse pair kernel with split term lists (variants), overlap diagnostic with the factorisation first.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
export TMPDIR=/tmp
export VMM_BA_EVAL=twopass
timeout -k 10 600 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_kernels.py -q -m gpu -x > gpurun_out/r3f/tests.txt 2>&1; tail -5 gpurun_out/r3f/tests.txt
b() { name=$1; shift; "$@" > gpurun_out/r3f/$name.json 2> gpurun_out/r3f/$name.err; python -c "
import json; d=json.load(open('gpurun_out/r3f/$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],4)); print('   ', {k:round(v['ms']*1000,1) for k,v in d['kernels'].items()})"; }
for t in N4T8 N4T4 N2T8 N8T4 N1T4; do
lib=$PWD/visual_marker_mapping_amd/libvmm_ba_$t.so; [ $t = N4T8 ] && lib=$PWD/visual_marker_mapping_amd/libvmm_ba.so
VMM_BA_LIB=$lib VMM_BA_SCHUR=sparse b sparse_v0.25_$t timeout -k 10 300 python bench.py --no-cpu-baseline --visibility 0.25 --steps 70
done
VMM_BA_SCHUR=sparse b sparse_v0.5 timeout -k 10 300 python bench.py --no-cpu-baseline --visibility 0.5 --steps 70
b closeup_sparse timeout -k 10 300 python bench.py --no-cpu-baseline --neighbors 6 10 --steps 70
timeout -k 10 300 python - <<'PY'
from visual_marker_mapping_amd import engine as eng
from visual_marker_mapping_amd.synthetic import make_scene
s = make_scene(2)
ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
for _ in range(2):
    print("overlap 500x200:", {k: round(v * 1000, 1) for k, v in ba.debug_overlap(20).items()})
ba.close()
PY
