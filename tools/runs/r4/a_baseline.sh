# Round-4 starting point: latency micro-benchmarks, the dataflow Cholesky's in-kernel time line, headline bench + kernel trace.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r4a; mkdir -p $O
for m in lat lat2; do /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/micro/$m.hip -o /tmp/$m 2>/dev/null && timeout -k 5 60 /tmp/$m; done > $O/micro.txt 2>&1
cat $O/micro.txt
bash tools/gpu_df_stamps.sh > $O/df_stamps.txt 2>&1; tail -25 $O/df_stamps.txt
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python -c "
import json; d=json.load(open('$O/bench.json')); print(round(d['value'],1), d['ms_per_step'], {k:round(v['ms']*1000,1) for k,v in d['kernels'].items()})"
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O -o prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 35 --warmup 7 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$O/prof_err.log
cd $GRAFT_REPO_ROOT
python tools/kernel_trace_stats.py $(find $O -name 'prof_kernel_trace.csv' | head -1) | head -14
