# explicit pair lists (only co-observed pairs, S zero-filled first): sparse tests with both forms, bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3y
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
for pv in explicit implicit; do
  VMM_BA_PAIRS=$pv timeout -k 10 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_distributed.py tests/test_gpu_driver.py -x -q -m gpu > gpurun_out/r3y/tests_$pv.txt 2>&1; echo "== VMM_BA_PAIRS=$pv: $(tail -1 gpurun_out/r3y/tests_$pv.txt)"; grep -E "^FAILED|Error" gpurun_out/r3y/tests_$pv.txt | head -5
done
VMM_BA_PAIRS=explicit VMM_BA_SCHUR=sparse timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3y/tests_all_explicit.txt 2>&1; echo "== whole suite, sparse + explicit everywhere: $(tail -1 gpurun_out/r3y/tests_all_explicit.txt)"; grep -E "^FAILED" gpurun_out/r3y/tests_all_explicit.txt | head
for sc in "--visibility 0.25" "--neighbors 6 10"; do
  for pv in implicit explicit; do
    VMM_BA_PAIRS=$pv timeout -k 10 300 python bench.py --no-cpu-baseline --steps 70 $sc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-9s %-18s %.1f it/s  pairs %.1f us form_z %.1f us' % ('$pv', '$sc', d['value'], d['kernels']['schur_syrk']['ms']*1e3, d['kernels']['form_z']['ms']*1e3))"
  done
done
