# the whole gpu suite with the less-travelled paths forced on every test: block-sparse elimination everywhere, tree ordering
# + explicit pair lists everywhere, launch-per-column factorisation everywhere, eager iterations, the two-workgroups-per-CU
# rank-k kernel -> profiles/r04_forced_paths_pytest.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b_m; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
: > $O/summary.txt
for v in "VMM_BA_SCHUR=sparse" "VMM_BA_ORDER=nd VMM_BA_SCHUR=sparse VMM_BA_PAIRS=explicit" "VMM_BA_NO_DATAFLOW=1" "VMM_BA_NO_GRAPH=1" "VMM_BA_SYRK_WIDE=0"; do
  n=$(echo $v | tr ' =' '__')
  env $v timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/$n.txt 2>&1
  echo "== $v: $(tail -1 $O/$n.txt)" | tee -a $O/summary.txt
  grep -E "^FAILED|^ERROR" $O/$n.txt | cut -c1-200 | tee -a $O/summary.txt
done
