cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4l
for m in r4 r3; do for c in "--neighbors 6 10 --steps 70" "--config 4 --neighbors 6 10 --steps 30 --warmup 10 --precision f64"; do VMM_BA_ORDER=nd VMM_BA_TREE_MODEL=$m timeout -k 10 400 python bench.py --no-cpu-baseline $c 2> gpurun_out/r4l/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('model $m: $c', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v['ms']*1000,1) for k,v in d['kernels'].items() if k in ('cholesky_solve','schur_syrk')})"; done; done
bash tools/gpu_df_stamps_tree.sh > gpurun_out/r4l/tree_stamps.txt 2>&1; grep "^j=" gpurun_out/r4l/tree_stamps.txt
