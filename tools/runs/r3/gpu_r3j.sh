set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3j
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_sparse.py -q -m gpu -x > gpurun_out/r3j/tests.txt 2>&1; tail -3 gpurun_out/r3j/tests.txt
b() { name=$1; shift; "$@" > gpurun_out/r3j/$name.json 2> gpurun_out/r3j/$name.err; python -c "
import json; d=json.load(open('gpurun_out/r3j/$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],4)); print('   ', {k:round(v['ms']*1000,1) for k,v in d['kernels'].items()})"; }
VMM_BA_SCHUR=sparse b sparse_v0.25 timeout -k 10 300 python bench.py --no-cpu-baseline --visibility 0.25 --steps 70
VMM_BA_SCHUR=sparse b sparse_v0.5 timeout -k 10 300 python bench.py --no-cpu-baseline --visibility 0.5 --steps 70
b closeup timeout -k 10 300 python bench.py --no-cpu-baseline --neighbors 6 10 --steps 70
