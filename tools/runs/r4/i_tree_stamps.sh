cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4i
bash tools/gpu_df_stamps_tree.sh > gpurun_out/r4i/tree_stamps.txt 2>&1; cat gpurun_out/r4i/tree_stamps.txt | tail -70
