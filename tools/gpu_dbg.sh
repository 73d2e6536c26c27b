cd $GRAFT_REPO_ROOT
for chain in 0 1; do
VMM_BA_NO_CHAIN=$chain timeout -k 10 500 python - <<'PY'
import sys, os
sys.path.insert(0, '.')
import numpy as np
from visual_marker_mapping_amd import engine as eng
rng = np.random.default_rng(0)
n = 1200
B = rng.standard_normal((n, n)); A = B @ B.T + n * np.eye(n); b = rng.standard_normal(n)
ref = np.linalg.solve(A, b)
bad = 0; worst = 0
for rep in range(300):
    x, info = eng.dense_spd_solve(A, b)
    err = np.abs(x - ref).max() / np.abs(ref).max()
    worst = max(worst, err)
    if not (err < 1e-10) or info: bad += 1
print("NO_CHAIN=%s: %d bad of 300, worst rel err %.3e" % (os.environ.get("VMM_BA_NO_CHAIN"), bad, worst))
PY
done
