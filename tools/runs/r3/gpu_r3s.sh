# full gpu suite + smoke + short bench lines (headline, configs[3]) at HEAD
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3s
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r3s/pytest_gpu.txt 2>&1; rc=$?; tail -6 gpurun_out/r3s/pytest_gpu.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r3s/bench.json 2> gpurun_out/r3s/bench.err || exit 1
timeout -k 10 400 python bench.py --config 4 --steps 24 --warmup 8 --no-cpu-baseline > gpurun_out/r3s/bench_cfg4.json 2> gpurun_out/r3s/bench_cfg4.err || exit 1
python - <<'PY'
import json
for n in ("bench", "bench_cfg4"):
    d = json.load(open("gpurun_out/r3s/%s.json" % n))
    print(n, round(d["value"], 1), "it/s", round(d["ms_per_step"], 4), {k: round(v["ms"] * 1000, 1) for k, v in d["kernels"].items()})
PY
