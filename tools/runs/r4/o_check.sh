cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r4o; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sync_timeout.py tests/test_gpu_sparse.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O -o prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 35 --warmup 7 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$O/prof_err.log
cd $GRAFT_REPO_ROOT
python tools/kernel_trace_stats.py $(find $O -name 'prof_kernel_trace.csv' | head -1) | head -4
for c in "" "--neighbors 6 10 --steps 70"; do timeout -k 10 400 python bench.py --no-cpu-baseline $c 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$c', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v['ms']*1000,1) for k,v in d['kernels'].items() if k in ('cholesky_solve',)})"; done
