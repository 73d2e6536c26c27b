cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r4b_r; mkdir -p $O
b() { name=$1; shift; "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$name', round(d['value'],1), round(d.get('ms_per_step', 0),4), d['config'].get('kept_family_order'), {k:round(v['ms']*1000,1) for k,v in d.get('kernels', {}).items() if k in ('cholesky_solve',)})"; }
for leaf in 21 32 42 53 64 85; do
export VMM_BA_ND_LEAF=$leaf
b "closeup leaf=$leaf" timeout -k 10 300 python bench.py --no-cpu-baseline --neighbors 6 10 --steps 70
b "closeup2000 leaf=$leaf" timeout -k 10 400 python bench.py --no-cpu-baseline --config 4 --neighbors 6 10 --precision f64 --steps 30 --warmup 10
done > $O/leaf.txt 2>&1
cat $O/leaf.txt
