# Multi-rank plumbing after the fusions (pack in the reduction, unpack + diagonal in one kernel, cross terms in the step
# all-reduce): distributed tests, then the one-rank forced-collectives bench line (native RCCL, graph).
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3n
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_sync_timeout.py tests/test_gpu_sparse.py -x -q -m gpu > gpurun_out/r3n/tests.txt 2>&1; rc=$?; tail -15 gpurun_out/r3n/tests.txt
[ $rc -eq 0 ] || exit 1
for i in 1 2; do
VMM_BA_FORCE_COLLECTIVES=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 70 --warmup 7 --no-cpu-baseline > gpurun_out/r3n/forced_$i.json 2> gpurun_out/r3n/forced_$i.err || exit 1
python -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/r3n/forced_$i.json') if l.startswith('{')][-1])
print('forced collectives: %.1f it/s %.4f ms' % (d['value'], d['ms_per_step']), d.get('collectives'))"
done
timeout -k 10 300 python bench.py --steps 70 --warmup 7 --no-cpu-baseline > gpurun_out/r3n/plain.json 2>/dev/null
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r3n/plain.json') if l.startswith('{')][-1])
print('plain: %.1f it/s %.4f ms' % (d['value'], d['ms_per_step']))"
