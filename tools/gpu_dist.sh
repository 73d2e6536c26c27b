cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# bench.py under torch.distributed.run with one rank (the driver's launch line with N=1)
VMM_BA_FORCE_COLLECTIVES=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 35 --warmup 7 --no-cpu-baseline 2>&1 | tail -6 | cut -c1-700
