cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4j
timeout -k 10 900 python -m pytest tests/test_gpu_distributed.py -m gpu -x -q -k "tree_ordering or bench_harness or two_ranks_match" > gpurun_out/r4j/pytest.txt 2>&1; tail -25 gpurun_out/r4j/pytest.txt | grep -v "^Hostname\|^Librccl\|libdrm\|RCCL version\|HIP version\|ROCm version\|socket.cpp\|Gloo"
