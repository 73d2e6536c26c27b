# the whole gpu suite with the less-travelled paths forced: block-sparse elimination everywhere, launch-per-column
# factorisation everywhere, eager iterations
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3w
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
for v in "VMM_BA_SCHUR=sparse" "VMM_BA_NO_DATAFLOW=1" "VMM_BA_NO_GRAPH=1"; do
  n=$(echo $v | cut -d= -f1)
  env $v timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r3w/$n.txt 2>&1
  echo "== $v: $(tail -1 gpurun_out/r3w/$n.txt)"
  grep -E "^FAILED|^ERROR" gpurun_out/r3w/$n.txt | cut -c1-200
done
