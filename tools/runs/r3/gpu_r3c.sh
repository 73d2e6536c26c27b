# Round 3 inner loop: batched sparse pair kernel, paired trailing updates of the large Cholesky.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_kernels.py tests/test_gpu_full_size.py -q -m gpu -x > gpurun_out/r3c/new_tests.txt 2>&1
tail -30 gpurun_out/r3c/new_tests.txt
b() { name=$1; shift; "$@" > gpurun_out/r3c/$name.json 2> gpurun_out/r3c/$name.err; python -c "
import json; d=json.load(open('gpurun_out/r3c/$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],4)); print('   ', {k:round(v['ms']*1000,1) for k,v in d['kernels'].items()})"; }
export VMM_BA_EVAL=twopass
for v in 0.25 0.5; do
VMM_BA_SCHUR=sparse b sparse_v$v timeout -k 10 300 python bench.py --no-cpu-baseline --visibility $v --steps 70
done
VMM_BA_SCHUR=dense b dense_v0.25 timeout -k 10 300 python bench.py --no-cpu-baseline --visibility 0.25 --steps 70
b cfg4 timeout -k 10 300 python bench.py --no-cpu-baseline --config 4 --steps 20 --warmup 5
b cfg2 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 70
