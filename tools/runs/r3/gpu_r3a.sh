# Round 3 inner loop: the new tests first (all failures reported), then the whole gpu suite, a bench line and kernel stats.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_sync_timeout.py tests/test_gpu_fused_eval.py tests/test_cpp_adapter.py tests/test_gpu_distributed.py -q -m gpu > gpurun_out/r3a/new_tests.txt 2>&1
tail -60 gpurun_out/r3a/new_tests.txt
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
python -c "
import json; d=json.load(open('gpurun_out/r3a/bench.json')); print(d['value'], d['ms_per_step']); print({k:round(v['ms']*1000,1) for k,v in d['kernels'].items()})"
for v in 0.25 0.5; do
timeout -k 10 300 python bench.py --no-cpu-baseline --visibility $v > gpurun_out/r3a/bench_v$v.json 2> gpurun_out/r3a/bench_v$v.err
python -c "
import json; d=json.load(open('gpurun_out/r3a/bench_v$v.json')); print('v=$v', d['value'], d['ms_per_step']); print({k:round(v['ms']*1000,1) for k,v in d['kernels'].items()})"
VMM_BA_SCHUR=dense timeout -k 10 300 python bench.py --no-cpu-baseline --visibility $v > gpurun_out/r3a/bench_v${v}_dense.json 2> gpurun_out/r3a/bench_v${v}_dense.err
python -c "
import json; d=json.load(open('gpurun_out/r3a/bench_v${v}_dense.json')); print('v=$v dense', d['value'], d['ms_per_step']); print({k:round(v['ms']*1000,1) for k,v in d['kernels'].items()})"
done
VMM_BA_EVAL=twopass timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r3a/bench_twopass.json 2> gpurun_out/r3a/bench_twopass.err
python -c "
import json; d=json.load(open('gpurun_out/r3a/bench_twopass.json')); print('twopass', d['value'], d['ms_per_step']); print({k:round(v['ms']*1000,1) for k,v in d['kernels'].items()})"
timeout -k 10 300 python bench.py --no-cpu-baseline --config 4 --steps 20 --warmup 5 > gpurun_out/r3a/bench_cfg4.json 2> gpurun_out/r3a/bench_cfg4.err
python -c "
import json; d=json.load(open('gpurun_out/r3a/bench_cfg4.json')); print('cfg4', d['value'], d['ms_per_step']); print({k:round(v['ms']*1000,1) for k,v in d['kernels'].items()})"
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3a/pytest_gpu.txt 2>&1; tail -15 gpurun_out/r3a/pytest_gpu.txt
