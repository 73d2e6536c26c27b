# Quick loop: kernel tests, bench line, rocprofv3 per-kernel stats of a short bench run.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/q
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solve.py -x -q 2>&1 | tail -3 || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/q/bench.json 2> gpurun_out/q/bench.err || exit 1
python -c "
import json; d=json.load(open('gpurun_out/q/bench.json')); print(d['value'], d['ms_per_step']); print({k:round(v['ms']*1000,1) for k,v in d['kernels'].items()})"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q -o prof -- python bench.py --steps 35 --warmup 7 --no-cpu-baseline > gpurun_out/q/bench_prof.json 2> gpurun_out/q/prof_err.log || exit 1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/q/**/prof_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:16]:
    print("%-70s calls %6s avg %9.1f ns  %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), float(r["Percentage"])))
PY
find gpurun_out/q -name '*.csv' -size +6M -delete
