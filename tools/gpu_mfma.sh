# MFMA utilisation per kernel from PMC counters (own pass, kernel-trace only): 500 x 200 and 2000 x 1000
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/mfma
export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CU_CYCLES --output-format csv -d gpurun_out/mfma -o c2 -- python bench.py --steps 14 --warmup 7 --no-cpu-baseline > gpurun_out/mfma/bench_c2.json 2> gpurun_out/mfma/err_c2.log || exit 1
timeout -k 10 700 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CU_CYCLES --output-format csv -d gpurun_out/mfma -o c4 -- python bench.py --config 4 --steps 4 --warmup 0 --no-cpu-baseline > gpurun_out/mfma/bench_c4.json 2> gpurun_out/mfma/err_c4.log || exit 1
python - <<'PY'
import csv, glob, json, collections
out = {}
for tag in ("c2", "c4"):
    f = glob.glob("gpurun_out/mfma/%s_counter_collection.csv" % tag)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(int)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("vmm::", "")
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                n[k] += 1
    res = {}
    for k, c in agg.items():
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0:
            continue
        gui = c.get("GRBM_GUI_ACTIVE", 0)
        res[k] = {"dispatches": n[k], "mfma_busy_cycles": c["SQ_VALU_MFMA_BUSY_CYCLES"], "gui_active_cycles": gui,
                  "mfma_f64_instructions": c.get("SQ_INSTS_VALU_MFMA_F64", 0),
                  "busy_cu_cycles": c.get("SQ_BUSY_CU_CYCLES", 0),
                  # rocprofv3's MfmaUtil expression with SIMD_NUM = 1024 (256 CUs x 4)
                  # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs: the kernel's active time is gui / 8
                  "mfma_util_percent": 100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8.0 * 1024.0) if gui else None,
                  "cycles_per_mfma": c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_INSTS_VALU_MFMA_F64"] if c.get("SQ_INSTS_VALU_MFMA_F64") else None}
    out[tag] = res
json.dump({"source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CU_CYCLES "
                     "on bench.py (c2: 500x200 f64, c4: 2000x1000 f32 accumulation); mfma_util_percent = "
                     "sum(MFMA_BUSY) / (sum(GUI_ACTIVE) / 8 * 1024 SIMDs) per kernel, the MfmaUtil expression of rocprofv3 -L with "
                     "GRBM_GUI_ACTIVE (reported summed over the 8 XCDs) divided by 8",
           "kernels": out}, open("gpurun_out/mfma/mfma_util.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
find gpurun_out/mfma -name '*.csv' -size +4M -delete
