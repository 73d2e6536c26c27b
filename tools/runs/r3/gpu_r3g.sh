# Round 3: column-major compressed Z in the sparse pair kernel; the overlapped assembly (opt-in) against the sequential order.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3g
export TMPDIR=/tmp
export VMM_BA_EVAL=twopass
timeout -k 10 600 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_kernels.py -q -m gpu -x > gpurun_out/r3g/tests.txt 2>&1; tail -5 gpurun_out/r3g/tests.txt
VMM_BA_OVERLAP=1 timeout -k 10 600 python -m pytest tests/test_gpu_solve.py tests/test_gpu_sync_timeout.py -q -m gpu -x > gpurun_out/r3g/tests_overlap.txt 2>&1; tail -5 gpurun_out/r3g/tests_overlap.txt
b() { name=$1; shift; "$@" > gpurun_out/r3g/$name.json 2> gpurun_out/r3g/$name.err; python -c "
import json; d=json.load(open('gpurun_out/r3g/$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],4)); print('   ', {k:round(v['ms']*1000,1) for k,v in d['kernels'].items()})"; }
VMM_BA_SCHUR=sparse b sparse_v0.25 timeout -k 10 300 python bench.py --no-cpu-baseline --visibility 0.25 --steps 70
VMM_BA_SCHUR=sparse b sparse_v0.5 timeout -k 10 300 python bench.py --no-cpu-baseline --visibility 0.5 --steps 70
b closeup timeout -k 10 300 python bench.py --no-cpu-baseline --neighbors 6 10 --steps 70
b seq timeout -k 10 300 python bench.py --no-cpu-baseline --steps 140
VMM_BA_OVERLAP=1 b overlap_default timeout -k 10 300 python bench.py --no-cpu-baseline --steps 140
VMM_BA_OVERLAP=1 VMM_BA_OVERLAP_GROUPS=0,2 b overlap_g02 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 140
VMM_BA_OVERLAP=1 VMM_BA_OVERLAP_GROUPS=0,1,3,6 VMM_BA_OVERLAP_SLOTS=512 b overlap_g0136_s512 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 140
VMM_BA_OVERLAP=1 VMM_BA_OVERLAP_GROUPS=0 VMM_BA_OVERLAP_SLOTS=512 b overlap_onegroup timeout -k 10 300 python bench.py --no-cpu-baseline --steps 140
