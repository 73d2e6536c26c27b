# Bench lines of the other BASELINE.json configurations: configs[3] (2000 x 1000, f32 accumulation), configs[4]
# (distortion + outliers + Huber), 25 % visibility, and the point-landmark variant is covered by tests only.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/cfg
timeout -k 10 300 python bench.py --config 5 --no-cpu-baseline > gpurun_out/cfg/bench_config5_robust.json 2> gpurun_out/cfg/err5.log || exit 1
timeout -k 10 300 python bench.py --visibility 0.25 --no-cpu-baseline > gpurun_out/cfg/bench_visibility025.json 2> gpurun_out/cfg/err025.log || exit 1
timeout -k 10 400 python bench.py --config 4 --steps 24 --warmup 8 --no-cpu-baseline > gpurun_out/cfg/bench_config4_f32accum.json 2> gpurun_out/cfg/err4.log || exit 1
python - <<'PY'
import json
for n in ("config5_robust", "visibility025", "config4_f32accum"):
    d = json.load(open("gpurun_out/cfg/bench_%s.json" % n))
    print(n, round(d["value"], 1), "it/s", {k: round(v["ms"] * 1000, 1) for k, v in d["kernels"].items()})
PY
