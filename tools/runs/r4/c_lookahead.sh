# Dataflow Cholesky with the pivot waves forming the next pivot block themselves: kernel tests, time line, bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solve.py tests/test_gpu_sync_timeout.py -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?; tail -5 $O/pytest.txt
[ $rc -eq 0 ] || exit 1
bash tools/gpu_df_stamps.sh > $O/df_stamps.txt 2>&1; tail -23 $O/df_stamps.txt
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python -c "
import json; d=json.load(open('$O/bench.json')); print(round(d['value'],1), d['ms_per_step'], {k:round(v['ms']*1000,1) for k,v in d['kernels'].items()})"
