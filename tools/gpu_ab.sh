# A/B timing of two builds of the library on the same box: libvmm_ba.so vs visual_marker_mapping_amd/libvmm_ba_B.so
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
for lib in libvmm_ba.so libvmm_ba_B.so; do
  VMM_BA_LIB=$GRAFT_REPO_ROOT/visual_marker_mapping_amd/$lib timeout -k 10 300 python bench.py --steps 140 --warmup 14 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('cfg2 $lib', round(d['value'],1), round(d['ms_per_step'],4), 'chol', round(k['cholesky_solve']['ms'],4), 'syrk', round(k['schur_syrk']['ms'],4), 'eval', round(k['eval_jacobian']['ms'],4), 'cost', round(k['eval_cost']['ms'],4))" | tee -a gpurun_out/ab.txt
done
done
