set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r01 -- python bench.py --steps 35 --warmup 7 --no-cpu-baseline > gpurun_out/prof/bench_under_rocprof.json 2> gpurun_out/prof/err.log
tail -3 gpurun_out/prof/err.log
find gpurun_out/prof -type f | head -20
f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1); cat $f
# keep only the small summaries (the per-dispatch trace is large)
find gpurun_out/prof -name '*kernel_trace.csv' -size +4M -delete
