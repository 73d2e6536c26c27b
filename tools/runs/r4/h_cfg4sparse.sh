cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
timeout -k 10 1100 python -m pytest tests/test_gpu_full_size.py -m gpu -x -q --durations=8 -k "config4" > gpurun_out/r4h/pytest.txt 2>&1; tail -25 gpurun_out/r4h/pytest.txt
