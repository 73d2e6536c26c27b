cd $GRAFT_REPO_ROOT
for g in 0 1; do
  VMM_BA_NO_GRAPH=$g timeout -k 10 300 python bench.py --steps 70 --warmup 14 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NO_GRAPH=$g', d['value'], d['ms_per_step'], d['kernels']['cholesky_solve']['ms'])"
done
