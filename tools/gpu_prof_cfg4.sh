# rocprofv3 kernel statistics of the 2000 x 1000 problem (BASELINE configs[3], f32 accumulation)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/cfg4
export TMPDIR=/tmp
timeout -k 10 800 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cfg4 -o c4 -- python bench.py --config 4 --steps 8 --warmup 0 --no-cpu-baseline > gpurun_out/cfg4/bench.json 2> gpurun_out/cfg4/err.log || exit 1
find gpurun_out/cfg4 -name '*.csv' -size +4M -delete
cut -d, -f1-5 gpurun_out/cfg4/c4_kernel_stats.csv | head -8
