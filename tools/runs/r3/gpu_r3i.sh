# Round 3: k_chol_step with a tile counter (panel workgroups join the trailing update).
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3i gpurun_out/prof4
export TMPDIR=/tmp
timeout -k 10 800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_full_size.py tests/test_gpu_solve.py tests/test_gpu_sync_timeout.py -q -m gpu -x > gpurun_out/r3i/tests.txt 2>&1; tail -5 gpurun_out/r3i/tests.txt
b() { name=$1; shift; "$@" > gpurun_out/r3i/$name.json 2> gpurun_out/r3i/$name.err; python -c "
import json; d=json.load(open('gpurun_out/r3i/$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],4)); print('   ', {k:round(v['ms']*1000,1) for k,v in d['kernels'].items()})"; }
b cfg4 timeout -k 10 300 python bench.py --no-cpu-baseline --config 4 --steps 20 --warmup 5
bash tools/gpu_prof4.sh > gpurun_out/r3i/prof4.txt 2>&1; tail -22 gpurun_out/r3i/prof4.txt
