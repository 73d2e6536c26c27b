# HBM read traffic of the rank-k update at 2000 x 1000 with and without the XCD-aware tile rounds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc4
export TMPDIR=/tmp
for mode in xcd noxcd; do
  if [ $mode = noxcd ]; then export VMM_BA_SYRK_NO_XCD=1; else unset VMM_BA_SYRK_NO_XCD; fi
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc4 -o f_$mode -- python bench.py --config 4 --steps 3 --warmup 0 --no-cpu-baseline > gpurun_out/pmc4/bench_$mode.json 2> gpurun_out/pmc4/err_$mode.log || exit 1
done
python - <<'PY'
import csv, glob, json, collections
out = {}
for mode in ("xcd", "noxcd"):
    f = glob.glob("gpurun_out/pmc4/f_%s_counter_collection.csv" % mode)[0]
    agg = collections.defaultdict(lambda: [0.0, 0])
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] != "FETCH_SIZE": continue
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("vmm::", "")
            agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    out[mode] = {k: {"dispatches": v[1], "fetch_bytes_per_launch": 2.0 * 1024.0 * v[0] / v[1]} for k, v in agg.items()
                 if k.startswith("k_syrk") or k.startswith("k_reduce_partials") or k.startswith("k_chol_step")}
json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE on bench.py --config 4 --steps 3; bytes = 2 * FETCH_SIZE * 1024 "
                     "(gfx950 correction); noxcd = VMM_BA_SYRK_NO_XCD=1 (plain stream-K order)", "modes": out},
          open("gpurun_out/pmc4/traffic_cfg4.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find gpurun_out/pmc4 -name '*.csv' -size +4M -delete
