cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4k
timeout -k 10 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_sync_timeout.py tests/test_gpu_kernels.py -m gpu -x -q > gpurun_out/r4k/pytest.txt 2>&1; tail -12 gpurun_out/r4k/pytest.txt
for c in "--neighbors 6 10 --steps 70" "--neighbors 6 10 --wall-rows 2 --steps 70" "--config 4 --neighbors 6 10 --steps 30 --warmup 10 --precision f64"; do VMM_BA_DEBUG=1 timeout -k 10 400 python bench.py --no-cpu-baseline $c 2> gpurun_out/r4k/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$c', round(d['value'],1), round(d['ms_per_step'],4), d['config'].get('kept_family_order'), {k:round(v['ms']*1000,1) for k,v in d['kernels'].items()})"; grep "tree ordering\|factor:" gpurun_out/r4k/err.txt | head -4; done
