# Round-4 (second session) bundle: gpu suite, smoke, headline bench (+ CPU baseline), kernel trace -> working-launch means, secondary lines,
# mid-size systems, PMC passes (HBM traffic, MFMA busy).  Outputs under gpurun_out/r4bfinal (copied to profiles/r04_* by hand).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r4bfinal; mkdir -p $O gpurun_out/pmc gpurun_out/mfma
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; rc=$?; tail -3 $O/pytest_gpu.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err || exit 1
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O -o prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 35 --warmup 7 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$O/prof_err.log
cd $GRAFT_REPO_ROOT
python tools/kernel_trace_stats.py $(find $O -name 'prof_kernel_trace.csv' | head -1) --out $O/kernel_stats_working.csv | head -12
b() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; exit 1; }; python -c "
import json; d=json.loads([l for l in open('$O/$name.json') if l.startswith('{')][-1]); print('$name', round(d['value'],1), d['unit'], round(d.get('ms_per_step', 0),4), {k:round(v['ms']*1000,1) for k,v in d.get('kernels', {}).items()})"; }
b bench_config5_robust timeout -k 10 300 python bench.py --no-cpu-baseline --config 5
b bench_config4_f32accum timeout -k 10 400 python bench.py --config 4 --steps 24 --warmup 8 --no-cpu-baseline
b bench_visibility025 timeout -k 10 300 python bench.py --no-cpu-baseline --visibility 0.25 --steps 70
b bench_closeup timeout -k 10 300 python bench.py --no-cpu-baseline --neighbors 6 10 --steps 70
VMM_BA_ORDER=natural b bench_closeup_natural_order timeout -k 10 300 python bench.py --no-cpu-baseline --neighbors 6 10 --steps 70
b bench_corridor timeout -k 10 300 python bench.py --no-cpu-baseline --neighbors 6 10 --wall-rows 2 --steps 70
b bench_closeup_2000x1000 timeout -k 10 400 python bench.py --no-cpu-baseline --config 4 --neighbors 6 10 --precision f64 --steps 30 --warmup 10
VMM_BA_ORDER=natural b bench_closeup_2000x1000_natural_order timeout -k 10 400 python bench.py --no-cpu-baseline --config 4 --neighbors 6 10 --precision f64 --steps 12 --warmup 6
b bench_incremental timeout -k 10 600 python bench.py --no-cpu-baseline --workload incremental
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O -o prof_closeup -- python3 $GRAFT_REPO_ROOT/bench.py --steps 35 --warmup 7 --no-cpu-baseline --neighbors 6 10 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/kernel_trace_stats.py $(find $O -name 'prof_closeup_kernel_trace.csv' | head -1) --out $O/kernel_stats_working_closeup.csv | head -5
cat > /tmp/mid.py <<'PY'
import json, sys
from visual_marker_mapping_amd import engine as eng
from visual_marker_mapping_amd.synthetic import make_scene
out = {}
for (nc, nt) in ((400, 250), (600, 320), (800, 400), (1000, 500)):
    s = make_scene(2, n_cams=nc, n_tags=nt)
    ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
    o = ba.solve(eng.default_options(robustify=0))
    kt = ba.time_kernels(eng.default_options(robustify=0), reps=5)
    r = dict(blocks=(6 * nt + 63) // 64, iterations=o["num_lm_iterations"], final_cost=o["final_cost"], sync_timeouts=o["num_sync_timeouts"],
             cholesky_us=kt["cholesky_ms"] * 1e3, syrk_us=kt["syrk_ms"] * 1e3, iteration_us=kt["lm_iteration_ms"] * 1e3)
    out["%dx%d" % (nc, nt)] = r
    print(nc, nt, r, flush=True)
    ba.close()
json.dump(out, open(sys.argv[1], "w"), indent=1)
PY
timeout -k 10 400 python /tmp/mid.py $O/mid_sizes.json || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 14 --warmup 7 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc/bench_$c.json 2> $GRAFT_REPO_ROOT/gpurun_out/pmc/err_$c.log || exit 1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, json, collections
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmc/**/pmc_%s_counter_collection.csv" % c, recursive=True)
    agg = collections.defaultdict(lambda: [0.0, 0])
    with open(f[0]) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") != c: continue
            k = row["Kernel_Name"].split("(")[0]
            agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    out[c] = {k: {"sum": v[0], "dispatches": v[1]} for k, v in agg.items()}
json.dump(out, open("gpurun_out/pmc/pmc_summary.json", "w"), indent=1)
PY
python tools/pmc_to_traffic.py gpurun_out/pmc/pmc_summary.json $O/pmc_traffic.json | tail -12
cp gpurun_out/pmc/pmc_summary.json $O/pmc_summary.json
cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CU_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/mfma -o c2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 14 --warmup 7 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/mfma/err_c2.log
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, json, collections
f = glob.glob("gpurun_out/mfma/**/c2_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
with open(f) as fh:
    for row in csv.DictReader(fh):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("vmm::", "")
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
res = {}
for k, c in agg.items():
    if c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0: continue
    gui = c.get("GRBM_GUI_ACTIVE", 0)
    res[k] = {"dispatches": n[k], "mfma_busy_cycles": c["SQ_VALU_MFMA_BUSY_CYCLES"], "gui_active_cycles": gui,
              "mfma_f64_instructions": c.get("SQ_INSTS_VALU_MFMA_F64", 0),
              "mfma_util_percent": 100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8.0 * 1024.0) if gui else None}
json.dump({"500x200": res, "note": "rocprofv3 MfmaUtil expression with SIMD_NUM = 1024; GRBM_GUI_ACTIVE is summed over the 8 XCDs"}, open("gpurun_out/r4bfinal/pmc_mfma.json", "w"), indent=1)
for k, v in res.items(): print(k, round(v["mfma_util_percent"], 1))
PY
find gpurun_out -name '*.csv' -size +6M -delete
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4bfinal/bench.json"))
print("headline", round(d["value"], 1), d["ms_per_step"], "cpu", d["cpu_baseline"]["value"], "roofline", d["roofline"]["frac"])
PY
