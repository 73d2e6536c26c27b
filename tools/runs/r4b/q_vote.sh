cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r4b_q; mkdir -p $O
VMM_BA_DEBUG=1 timeout -k 10 600 python -m pytest tests/test_gpu_distributed.py -m gpu -x -q -k "native_rccl_path_single_rank and not full" -s > $O/pytest.txt 2>&1; rc=$?; tail -3 $O/pytest.txt; grep -c "RCCL not recorded" $O/pytest.txt; exit $rc
