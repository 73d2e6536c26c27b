# A/B on one box: HEAD against the library built from an earlier commit (libvmm_ba_OLD.so)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
for i in 1 2 3; do
  for lib in HEAD OLD; do
    if [ $lib = OLD ]; then export VMM_BA_LIB=$GRAFT_REPO_ROOT/visual_marker_mapping_amd/libvmm_ba_OLD.so; else unset VMM_BA_LIB; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-4s %.1f it/s %.4f ms  chol %.1f syrk %.1f' % ('$lib', d['value'], d['ms_per_step'], d['kernels']['cholesky_solve']['ms']*1e3, d['kernels']['schur_syrk']['ms']*1e3))"
  done
done
