# corridor scene (tags in two rows, 6-10 nearest tags per image): tree ordering against the natural order; bench config fields
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
mkdir -p gpurun_out/r3z6
for rows in 1 2 4; do
for ov in auto natural; do
  if [ $ov = natural ]; then export VMM_BA_ORDER=natural; else unset VMM_BA_ORDER; fi
  VMM_BA_DEBUG=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 70 --neighbors 6 10 --wall-rows $rows > gpurun_out/r3z6/corridor_${rows}_$ov.json 2> gpurun_out/r3z6/err_${rows}_$ov.txt
  python -c "
import json; d=json.load(open('gpurun_out/r3z6/corridor_${rows}_$ov.json')); print('rows $rows %-8s %.1f it/s %.4f ms  %s | %s' % ('$ov', d['value'], d['ms_per_step'], {k: round(v['ms']*1e3,1) for k,v in d['kernels'].items() if k in ('schur_syrk','cholesky_solve')}, d['config']['kept_family_order']))"
  grep -E "tree ordering candidate|tree ordering:" gpurun_out/r3z6/err_${rows}_$ov.txt | head -2
done
done
