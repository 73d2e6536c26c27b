cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3z
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
VMM_BA_ORDER=nd VMM_BA_SCHUR=sparse timeout -k 10 600 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_solve.py tests/test_gpu_driver.py tests/test_gpu_sync_timeout.py -x -q -m gpu > gpurun_out/r3z/tests_c.txt 2>&1; echo "== sparse/solve/driver/sync, sparse + nd: $(tail -1 gpurun_out/r3z/tests_c.txt)"; grep -E "^FAILED" gpurun_out/r3z/tests_c.txt | head
for leaf in 21 42 64; do
  VMM_BA_DEBUG=1 VMM_BA_ND_LEAF=$leaf VMM_BA_ORDER=nd timeout -k 10 300 python bench.py --no-cpu-baseline --steps 70 --neighbors 6 10 2> gpurun_out/r3z/err_nd_$leaf.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('leaf %-3s %.1f it/s  %s' % ('$leaf', d['value'], {k: round(v['ms']*1e3,1) for k,v in d['kernels'].items()}))"
  grep -E "tree ordering|longest chain" gpurun_out/r3z/err_nd_$leaf.txt | head -3
done
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 70 --neighbors 6 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('natural  %.1f it/s  %s' % (d['value'], {k: round(v['ms']*1e3,1) for k,v in d['kernels'].items()}))"
