# Second final bundle of round 3 (after the tree ordering): gpu suite (default and with the sparse / tree-ordered paths
# forced everywhere), smoke, headline bench with the CPU baseline, rocprofv3 kernel statistics, close-up lines.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final2
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/final2
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; rc=$?; tail -4 $O/pytest_gpu.txt
[ $rc -eq 0 ] || exit 1
VMM_BA_ORDER=nd VMM_BA_SCHUR=sparse timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/pytest_gpu_sparse_nd.txt 2>&1; tail -2 $O/pytest_gpu_sparse_nd.txt; grep -E "^FAILED" $O/pytest_gpu_sparse_nd.txt | head
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o prof -- python bench.py --steps 35 --warmup 7 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/prof_err.log || exit 1
b() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; exit 1; }; python -c "
import json; d=json.loads([l for l in open('$O/$name.json') if l.startswith('{')][-1]); print('$name', round(d['value'],1), d['unit'], round(d.get('ms_per_step', 0),4), {k:round(v['ms']*1000,1) for k,v in d.get('kernels', {}).items()})"; }
b closeup_tree timeout -k 10 300 python bench.py --no-cpu-baseline --neighbors 6 10 --steps 70
VMM_BA_ORDER=natural b closeup_natural timeout -k 10 300 python bench.py --no-cpu-baseline --neighbors 6 10 --steps 70
VMM_BA_SCHUR=dense b closeup_dense timeout -k 10 300 python bench.py --no-cpu-baseline --neighbors 6 10 --steps 70
b v025 timeout -k 10 300 python bench.py --no-cpu-baseline --visibility 0.25 --steps 70
b cfg5 timeout -k 10 300 python bench.py --no-cpu-baseline --config 5
b cfg4 timeout -k 10 400 python bench.py --config 4 --steps 24 --warmup 8 --no-cpu-baseline
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O -o prof_closeup -- python3 $GRAFT_REPO_ROOT/bench.py --steps 35 --warmup 7 --no-cpu-baseline --neighbors 6 10 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find $O -name '*.csv' -size +6M -delete
python - <<'PY'
import json
d = json.load(open("gpurun_out/final2/bench.json"))
print("headline", round(d["value"], 1), d["ms_per_step"], "cpu", d["cpu_baseline"]["value"])
PY
