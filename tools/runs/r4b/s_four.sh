cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r4b_s; mkdir -p $O
VMM_BA_DF_FOUR=1 timeout -k 10 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_sync_timeout.py -m gpu -x -q > $O/pytest_tree4.txt 2>&1; rc=$?; tail -4 $O/pytest_tree4.txt; [ $rc -eq 0 ] || exit 1
b() { name=$1; shift; "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$name', round(d['value'],1), round(d.get('ms_per_step', 0),4), {k:round(v['ms']*1000,1) for k,v in d.get('kernels', {}).items() if k in ('cholesky_solve',)})"; }
for four in 0 1; do
export VMM_BA_DF_FOUR=$four
b "closeup four=$four" timeout -k 10 300 python bench.py --no-cpu-baseline --neighbors 6 10 --steps 70
b "corridor four=$four" timeout -k 10 300 python bench.py --no-cpu-baseline --neighbors 6 10 --wall-rows 2 --steps 70
b "closeup2000 four=$four" timeout -k 10 400 python bench.py --no-cpu-baseline --config 4 --neighbors 6 10 --precision f64 --steps 30 --warmup 10
done
