cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
timeout -k 10 900 python -m pytest tests/test_cpp_adapter.py tests/test_gpu_distributed.py -m gpu -x -q -s > gpurun_out/r4g/pytest.txt 2>&1; tail -12 gpurun_out/r4g/pytest.txt; grep TIMING gpurun_out/r4g/pytest.txt
