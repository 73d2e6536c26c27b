# Round 3: sparse pair kernel variants (terms per batch), close-up scene, overlap diagnostic, per-launch k_chol_step.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e gpurun_out/prof4
export TMPDIR=/tmp
export VMM_BA_EVAL=twopass
b() { name=$1; shift; "$@" > gpurun_out/r3e/$name.json 2> gpurun_out/r3e/$name.err; python -c "
import json; d=json.load(open('gpurun_out/r3e/$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],4)); print('   ', {k:round(v['ms']*1000,1) for k,v in d['kernels'].items()})"; }
for t in T8 T4 T2; do
lib=$PWD/visual_marker_mapping_amd/libvmm_ba_$t.so; [ $t = T8 ] && lib=$PWD/visual_marker_mapping_amd/libvmm_ba.so
VMM_BA_LIB=$lib VMM_BA_SCHUR=sparse b sparse_v0.25_$t timeout -k 10 300 python bench.py --no-cpu-baseline --visibility 0.25 --steps 70
done
b closeup_sparse timeout -k 10 300 python bench.py --no-cpu-baseline --neighbors 6 10 --steps 70
VMM_BA_SCHUR=dense b closeup_dense timeout -k 10 300 python bench.py --no-cpu-baseline --neighbors 6 10 --steps 70
timeout -k 10 300 python - <<'PY'
from visual_marker_mapping_amd import engine as eng
from visual_marker_mapping_amd.synthetic import make_scene
s = make_scene(2)
ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
for _ in range(2):
    print("overlap 500x200:", {k: round(v * 1000, 1) for k, v in ba.debug_overlap(20).items()})
ba.close()
PY
bash tools/gpu_prof4.sh > gpurun_out/r3e/prof4.txt 2>&1; tail -40 gpurun_out/r3e/prof4.txt
