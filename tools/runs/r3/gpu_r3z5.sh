cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3z5
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/r3z5
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; echo "== whole suite: $(tail -1 $O/tests.txt)"; grep -E "^FAILED" $O/tests.txt | head
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench.json')); print('headline %.1f it/s %.4f ms' % (d['value'], d['ms_per_step']), {k: round(v['ms']*1e3,1) for k,v in d['kernels'].items()})"
timeout -k 10 300 python bench.py --no-cpu-baseline --neighbors 6 10 --steps 70 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('close-up %.1f it/s' % d['value'], {k: round(v['ms']*1e3,1) for k,v in d['kernels'].items()})"
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O -o prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 35 --warmup 7 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r3z5/**/prof_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0].replace("void ", "")
    if any(k in n for k in ("chol_dataflow", "backsolve_chain")):
        print("   %-40s calls %4s avg %.1f us" % (n[:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
find $O -name '*.csv' -size +6M -delete
