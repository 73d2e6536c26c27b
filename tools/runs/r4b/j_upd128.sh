cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r4b_j; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "fallback_paths" > $O/pytest_fallback.txt 2>&1; rc=$?; tail -5 $O/pytest_fallback.txt; [ $rc -eq 0 ] || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "chol or cholesky or spd" > $O/pytest_chol.txt 2>&1; rc=$?; tail -5 $O/pytest_chol.txt; [ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py --config 4 --steps 24 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('cfg4', round(d['value'],2), round(d['ms_per_step'],3), {k:(round(v['ms']*1000,1), round(v.get('frac') or 0,3)) for k,v in d['kernels'].items()})"
