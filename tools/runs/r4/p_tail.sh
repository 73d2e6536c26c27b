# configs[3] (2000 x 1000 dense, f32 accumulation): how many trailing block columns go to the one-launch kernel now that it is faster
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
for t in 34 40 46 52 58; do VMM_BA_DF_MAX_WG=4000 VMM_BA_CHOL_TAIL=$t timeout -k 10 400 python bench.py --config 4 --steps 12 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('tail $t', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v['ms']*1000,1) for k,v in d['kernels'].items() if k in ('cholesky_solve','schur_syrk')})"; done
