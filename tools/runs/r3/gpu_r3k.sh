# PMC look at k_schur_pairs (25 % visibility): where do the cycles go?
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCP|TCC|TA|TD|SQ|SQC)_[A-Z0-9_]+(_sum|_avr)?\b" | sort -u > gpurun_out/r3k/counters.txt; wc -l gpurun_out/r3k/counters.txt
for c in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  n=$(echo $c | cut -d' ' -f1)
  VMM_BA_SCHUR=sparse timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/r3k -o pmc_$n -- python bench.py --visibility 0.25 --steps 14 --warmup 7 --no-cpu-baseline > /dev/null 2> gpurun_out/r3k/err_$n.log || echo "FAILED $c"
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/r3k/**/pmc_*_counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if "k_schur_pairs" not in row["Kernel_Name"]: continue
            a = agg[row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
for c, v in sorted(agg.items()):
    print("   %-34s per launch %16.0f  (%d launches)" % (c, v[0] / max(v[1], 1), v[1]))
PY
find gpurun_out/r3k -name '*.csv' -size +2M -delete
