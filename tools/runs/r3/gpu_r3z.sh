# tree ordering of the kept family (VMM_BA_ORDER=nd), dense factorisation still: correctness + what the ordering looks like
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3z
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
VMM_BA_ORDER=nd timeout -k 10 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_driver.py tests/test_gpu_solve.py -x -q -m gpu > gpurun_out/r3z/tests_nd.txt 2>&1; echo "== VMM_BA_ORDER=nd: $(tail -1 gpurun_out/r3z/tests_nd.txt)"; grep -E "^FAILED|Error" gpurun_out/r3z/tests_nd.txt | head -5
VMM_BA_ORDER=nd VMM_BA_SCHUR=sparse timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3z/tests_all_nd.txt 2>&1; echo "== whole suite, sparse + nd everywhere: $(tail -1 gpurun_out/r3z/tests_all_nd.txt)"; grep -E "^FAILED" gpurun_out/r3z/tests_all_nd.txt | head
for ov in natural nd; do
  VMM_BA_DEBUG=1 VMM_BA_ORDER=$ov timeout -k 10 300 python bench.py --no-cpu-baseline --steps 70 --neighbors 6 10 2> gpurun_out/r3z/err_$ov.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-8s %.1f it/s  %s' % ('$ov', d['value'], {k: round(v['ms']*1e3,1) for k,v in d['kernels'].items()}))"
  grep "tree ordering" gpurun_out/r3z/err_$ov.txt | head -2
done
