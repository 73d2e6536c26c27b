cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r4b_d; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; tail -3 $O/pytest_gpu.txt
[ $rc -eq 0 ] || exit 1
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O -o prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 35 --warmup 7 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$O/prof_err.log
cd $GRAFT_REPO_ROOT
python tools/kernel_trace_stats.py $(find $O -name 'prof_kernel_trace.csv' | head -1) --out $O/kernel_stats_working.csv | head -14
find $O -name '*.csv' -size +6M -delete
