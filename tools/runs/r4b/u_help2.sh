cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r4b_u; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_sync_timeout.py -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?; tail -4 $O/pytest.txt; [ $rc -eq 0 ] || exit 1
VMM_BA_DF_HELP=1 timeout -k 10 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_sync_timeout.py -m gpu -x -q > $O/pytest_forced.txt 2>&1; rc=$?; tail -2 $O/pytest_forced.txt; [ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py --no-cpu-baseline --config 4 --neighbors 6 10 --precision f64 --steps 30 --warmup 10 > $O/bench_closeup_2000x1000.json 2>/dev/null; python -c "
import json; d=json.loads([l for l in open('$O/bench_closeup_2000x1000.json') if l.startswith('{')][-1]); print(round(d['value'],1), d['ms_per_step'], {k:round(v['ms']*1000,1) for k,v in d['kernels'].items()})"
