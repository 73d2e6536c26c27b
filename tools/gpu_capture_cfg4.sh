# Does the 2000 x 1000 problem capture its first iteration cleanly now that every kernel is touched at create?
# (round 1 ran the first iteration of a handle eagerly after an intermittent capture failure at this size)
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  VMM_BA_DEBUG=1 timeout -k 10 250 python bench.py --config 4 --steps 16 --warmup 8 --no-cpu-baseline 2> gpurun_out/cfg4_err_$i.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config 4 run $i:', round(d['value'],2), 'it/s', {k:round(v['ms'],3) for k,v in d['kernels'].items()})" || { echo FAILED run $i; tail -5 gpurun_out/cfg4_err_$i.log; exit 1; }
done
