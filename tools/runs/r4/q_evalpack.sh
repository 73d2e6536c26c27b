cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r4q; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solve.py tests/test_gpu_sparse.py tests/test_gpu_points.py tests/test_gpu_edge_cases.py tests/test_gpu_fused_eval.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
for c in "" "--neighbors 6 10 --steps 70" "--visibility 0.25 --steps 70" "--config 4 --steps 12 --warmup 6"; do timeout -k 10 400 python bench.py --no-cpu-baseline $c 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$c', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v['ms']*1000,1) for k,v in d['kernels'].items() if k in ('eval_jacobian','eval_cost','cholesky_solve')})"; done
