cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4n
timeout -k 10 1100 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_distributed.py -m gpu -x -q --durations=5 -k "2000 or tree" > gpurun_out/r4n/pytest.txt 2>&1; tail -14 gpurun_out/r4n/pytest.txt | grep -v "^Hostname\|^Librccl\|libdrm\|RCCL version\|HIP version\|ROCm version\|socket.cpp\|Gloo"
