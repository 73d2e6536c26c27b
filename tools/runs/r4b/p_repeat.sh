# the gpu suite twice more at HEAD (stability of the round's new synchronisation: completion words, staggered halves) and
# the headline line with the corrected kernel label
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r4b_p; mkdir -p $O
for i in 1 2; do
  timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/pytest_$i.txt 2>&1; rc=$?; tail -1 $O/pytest_$i.txt; [ $rc -eq 0 ] || exit 1
done
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err || exit 1
python -c "
import json; d=json.load(open('$O/bench.json')); print(round(d['value'],1), d['ms_per_step'], d['roofline']['frac'], d['kernels']['schur_syrk'])"
