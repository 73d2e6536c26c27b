cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
timeout -k 10 600 python -m pytest tests/test_gpu_sync_timeout.py tests/test_gpu_sparse.py -m gpu -x -q > gpurun_out/r4f/pytest.txt 2>&1; tail -15 gpurun_out/r4f/pytest.txt
