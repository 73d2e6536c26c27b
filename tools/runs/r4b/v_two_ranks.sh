# bench.py launched as the driver launches it, with two ranks on the one GPU of this box (--backend gloo: host-callback collectives)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT MASTER_ADDR=127.0.0.1
O=gpurun_out/r4b_v; mkdir -p $O
for n in 2 4; do
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $n --backend gloo --steps 70 --warmup 14 --no-cpu-baseline > $O/bench_${n}ranks.json 2> $O/bench_${n}ranks.err || { tail -5 $O/bench_${n}ranks.err; exit 1; }
python -c "
import json; d=json.loads([l for l in open('$O/bench_${n}ranks.json') if l.startswith('{')][-1]); print(d['n_gpus'], round(d['value'],1), d['ms_per_step'], d.get('collective'), d['scaling'])"
done
