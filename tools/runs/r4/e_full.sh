# whole gpu suite + headline bench + kernel trace -> gpurun_out/r4e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r4e; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; tail -4 $O/pytest_gpu.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python -c "
import json; d=json.load(open('$O/bench.json')); print(round(d['value'],1), d['ms_per_step'], {k:round(v['ms']*1000,1) for k,v in d['kernels'].items()})"
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O -o prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 35 --warmup 7 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$O/prof_err.log
cd $GRAFT_REPO_ROOT
python tools/kernel_trace_stats.py $(find $O -name 'prof_kernel_trace.csv' | head -1) | head -12
for c in "--config 4 --steps 24 --warmup 8" "--config 5" "--visibility 0.25 --steps 70" "--neighbors 6 10 --steps 70"; do timeout -k 10 400 python bench.py --no-cpu-baseline $c 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$c', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v['ms']*1000,1) for k,v in d['kernels'].items()})"; done
