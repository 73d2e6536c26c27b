# less-travelled paths: per-block back-substitution kernels, tags eliminated at full size, eager (no graph) iterations
cd $GRAFT_REPO_ROOT
VMM_BA_NO_CHAIN=1 timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solve.py -m gpu -x -q 2>&1 | tail -2
VMM_BA_NO_GRAPH=1 timeout -k 10 300 python -m pytest tests/test_gpu_solve.py -m gpu -x -q 2>&1 | tail -2
for e in tags cams; do
timeout -k 10 300 python bench.py --steps 28 --warmup 7 --no-cpu-baseline --elimination $e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('elimination=$e', round(d['value'],1), 'it/s, reduced order', d['config']['reduced_system_order'], {k:round(v['ms'],3) for k,v in d['kernels'].items() if k in ('schur_syrk','cholesky_solve')})"
done
