cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3z
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3z/tests_default.txt 2>&1; echo "== whole suite (default): $(tail -1 gpurun_out/r3z/tests_default.txt)"; grep -E "^FAILED|^E  " gpurun_out/r3z/tests_default.txt | head -12
VMM_BA_DEBUG=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 70 --neighbors 6 10 2> gpurun_out/r3z/err_auto.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('auto  %.1f it/s  %s' % (d['value'], {k: round(v['ms']*1e3,1) for k,v in d['kernels'].items()}))"
grep -E "tree ordering|longest chain" gpurun_out/r3z/err_auto.txt | head -4
VMM_BA_DEBUG=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 70 --visibility 0.25 2> gpurun_out/r3z/err_v025.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('v0.25 %.1f it/s  %s' % (d['value'], {k: round(v['ms']*1e3,1) for k,v in d['kernels'].items()}))"
grep -E "tree ordering|longest chain" gpurun_out/r3z/err_v025.txt | head -4
