# the tree-ordered bench lines again: their factorisation priced with the flops of the factor's non-zero blocks (ABI 5)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r4b_g; mkdir -p $O
b() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; exit 1; }; python -c "
import json; d=json.loads([l for l in open('$O/$name.json') if l.startswith('{')][-1]); print('$name', round(d['value'],1), d['unit'], round(d.get('ms_per_step', 0),4), {k:(round(v['ms']*1000,1), round(v.get('frac') or 0, 3)) for k,v in d.get('kernels', {}).items()})"; }
b bench_closeup timeout -k 10 300 python bench.py --no-cpu-baseline --neighbors 6 10 --steps 70
b bench_corridor timeout -k 10 300 python bench.py --no-cpu-baseline --neighbors 6 10 --wall-rows 2 --steps 70
b bench_closeup_2000x1000 timeout -k 10 400 python bench.py --no-cpu-baseline --config 4 --neighbors 6 10 --precision f64 --steps 30 --warmup 10
b bench timeout -k 10 300 python bench.py --no-cpu-baseline --steps 140
