# Diagnosis of the round-1 capture failure: 2000 x 1000 in fresh processes WITHOUT the kernel preload of vmm_ba_create
# (VMM_BA_NO_PRELOAD=1) and without the eager first iteration, so kernels are launched for the first time under capture.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/capdiag
for i in 1 2 3 4; do
  VMM_BA_NO_PRELOAD=1 VMM_BA_DEBUG=1 timeout -k 10 250 python bench.py --config 4 --steps 8 --warmup 8 --no-cpu-baseline > gpurun_out/capdiag/out_$i.json 2> gpurun_out/capdiag/err_$i.log
  echo "no-preload run $i: rc=$? $(cut -c1-120 gpurun_out/capdiag/out_$i.json) $(grep -i -m2 'capture\|error' gpurun_out/capdiag/err_$i.log)"
done
# one rank with the collectives forced on: native RCCL recorded in the graph / RCCL between five graphs / host callback
for mode in "rccl 1" "rccl 0" "callback 1"; do
  set -- $mode
  VMM_BA_RCCL_GRAPH=$2 VMM_BA_FORCE_COLLECTIVES=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --collective $1 --steps 70 --warmup 14 --no-cpu-baseline > gpurun_out/capdiag/dist_$1_$2.json 2> gpurun_out/capdiag/dist_$1_$2.err
  echo "collective=$1 graph=$2: $(python -c "import json,sys; d=json.loads([l for l in open('gpurun_out/capdiag/dist_$1_$2.json') if l.startswith('{')][-1]); print(round(d['value'],1),'it/s', round(d['ms_per_step'],4),'ms')")"
done
