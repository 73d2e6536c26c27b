cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 800 python bench.py --config 4 --steps 8 --warmup 0 --no-cpu-baseline > gpurun_out/bench_cfg4.json 2> gpurun_out/bench_cfg4.err; tail -3 gpurun_out/bench_cfg4.err; cat gpurun_out/bench_cfg4.json
timeout -k 10 600 python bench.py --config 5 --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/bench_cfg5.json 2> gpurun_out/bench_cfg5.err; tail -3 gpurun_out/bench_cfg5.err; cat gpurun_out/bench_cfg5.json
timeout -k 10 600 python bench.py --config 2 --visibility 0.25 --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/bench_v025.json 2> gpurun_out/bench_v025.err; tail -3 gpurun_out/bench_v025.err; cat gpurun_out/bench_v025.json
