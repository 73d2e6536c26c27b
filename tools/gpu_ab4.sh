# A/B of two library builds on config 4 (2000 x 1000); also appended to gpurun_out/ab.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lib in libvmm_ba.so libvmm_ba_B.so; do
  VMM_BA_LIB=$GRAFT_REPO_ROOT/visual_marker_mapping_amd/$lib timeout -k 10 500 python bench.py --config 4 --steps 8 --warmup 0 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg4 $lib', round(d['value'],2), round(d['ms_per_step'],3), {k:(round(v['ms'],3), round(v.get('achieved',0),1)) for k,v in d['kernels'].items() if k in ('schur_syrk','cholesky_solve')})" | tee -a gpurun_out/ab.txt
done
