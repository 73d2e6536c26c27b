set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/pytest_gpu.log; cat gpurun_out/pytest_gpu.log
grep -q passed gpurun_out/pytest_gpu.log && ! grep -q failed gpurun_out/pytest_gpu.log && bash tools/gpu_stamps.sh && timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -3 gpurun_out/bench.err; cat gpurun_out/bench.json
