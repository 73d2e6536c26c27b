# Round 3 profiling: per-launch k_chol_step durations at 2000 x 1000 (paired updates), k_schur_pairs under rocprofv3 + PMC.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d gpurun_out/prof4
export TMPDIR=/tmp
export VMM_BA_EVAL=twopass
bash tools/gpu_prof4.sh > gpurun_out/r3d/prof4.txt 2>&1; tail -30 gpurun_out/r3d/prof4.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3d -o sp -- python bench.py --visibility 0.25 --steps 35 --warmup 7 --no-cpu-baseline > gpurun_out/r3d/bench_sp.json 2> gpurun_out/r3d/sp_err.log
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r3d/**/sp_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:14]:
    print("%-70s calls %6s avg %9.1f ns  %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), float(r["Percentage"])))
PY
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/r3d -o pmc_$n -- python bench.py --visibility 0.25 --steps 14 --warmup 7 --no-cpu-baseline > /dev/null 2> gpurun_out/r3d/pmc_err_$n.log
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/r3d/**/pmc_*_counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0]
            if "k_schur_pairs" not in k and "k_syrk" not in k and "k_form_z" not in k: continue
            a = agg[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
for k, d in agg.items():
    print(k[:60])
    for c, v in sorted(d.items()):
        print("   %-26s per launch %14.0f  (%d launches)" % (c, v[0] / max(v[1], 1), v[1]))
PY
find gpurun_out/r3d -name '*.csv' -size +2M -delete
