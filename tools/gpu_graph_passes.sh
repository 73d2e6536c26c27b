cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for gp in 1 2 3 4; do
  VMM_BA_GRAPH_PASSES=$gp timeout -k 10 300 python bench.py --steps 140 --warmup 14 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('passes per graph $gp:', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms  enqueued', round(d['kernels']['lm_iteration_enqueued']['ms'],4))"
done
done
