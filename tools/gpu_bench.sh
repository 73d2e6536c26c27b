set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -5 gpurun_out/bench.err; cat gpurun_out/bench.json
