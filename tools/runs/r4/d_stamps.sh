cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
bash tools/gpu_df_stamps.sh > gpurun_out/r4d/df_stamps.txt 2>&1; tail -34 gpurun_out/r4d/df_stamps.txt
