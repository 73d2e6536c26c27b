# Round 3 inner loop: sparse pair kernel + prefetching fused evaluation.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_fused_eval.py -q -m gpu -x > gpurun_out/r3b/new_tests.txt 2>&1
tail -30 gpurun_out/r3b/new_tests.txt
run() { # name, env..., args
  name=$1; shift
  env "$@" > /dev/null 2>&1
}
b() { name=$1; shift; "$@" > gpurun_out/r3b/$name.json 2> gpurun_out/r3b/$name.err; python -c "
import json; d=json.load(open('gpurun_out/r3b/$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],4)); print('   ', {k:round(v['ms']*1000,1) for k,v in d['kernels'].items()})"; }
for g in 1 2 4; do
VMM_BA_FUSED_GROUP=$g b fused_g$g timeout -k 10 300 python bench.py --no-cpu-baseline --steps 70
done
VMM_BA_EVAL=twopass b twopass timeout -k 10 300 python bench.py --no-cpu-baseline --steps 70
for v in 0.25 0.5; do
VMM_BA_SCHUR=sparse b sparse_v$v timeout -k 10 300 python bench.py --no-cpu-baseline --visibility $v --steps 70
VMM_BA_SCHUR=dense b dense_v$v timeout -k 10 300 python bench.py --no-cpu-baseline --visibility $v --steps 70
done
for g in 8 16; do
VMM_BA_FUSED_GROUP=$g b cfg4_g$g timeout -k 10 300 python bench.py --no-cpu-baseline --config 4 --steps 20 --warmup 5
done
VMM_BA_EVAL=twopass b cfg4_twopass timeout -k 10 300 python bench.py --no-cpu-baseline --config 4 --steps 20 --warmup 5
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3b/pytest_gpu.txt 2>&1; tail -15 gpurun_out/r3b/pytest_gpu.txt
