cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r4m; mkdir -p $O
cd /tmp && VMM_BA_ORDER=nd timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O -o prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 35 --warmup 7 --no-cpu-baseline --neighbors 6 10 > /dev/null 2> $GRAFT_REPO_ROOT/$O/prof_err.log
cd $GRAFT_REPO_ROOT
python tools/kernel_trace_stats.py $(find $O -name 'prof_kernel_trace.csv' | head -1) | head -8
VMM_BA_ORDER=nd bash tools/gpu_df_stamps_tree.sh > $O/tree_stamps.txt 2>&1; grep "^j=\|tree nodes" $O/tree_stamps.txt
