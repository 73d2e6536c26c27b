# Round-3 bench lines beside the headline: low visibility and close-up scenes on the block-sparse and on the dense
# elimination, the incremental workload (resident handle vs a handle per bundle adjustment), the assembly/factorisation
# overlap diagnostic, mid-size reduced systems (24 to 47 block columns) on both factorisation paths.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3lines
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r3lines
b() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; exit 1; }; python -c "
import json; d=json.loads([l for l in open('$O/$name.json') if l.startswith('{')][-1]); print('$name', round(d['value'],1), d['unit'], round(d.get('ms_per_step', 0),4)); print('   ', {k:round(v['ms']*1000,1) for k,v in d.get('kernels', {}).items()})"; }
b v025_auto timeout -k 10 300 python bench.py --no-cpu-baseline --visibility 0.25 --steps 70
VMM_BA_SCHUR=dense b v025_dense timeout -k 10 300 python bench.py --no-cpu-baseline --visibility 0.25 --steps 70
b v050_auto timeout -k 10 300 python bench.py --no-cpu-baseline --visibility 0.5 --steps 70
VMM_BA_SCHUR=sparse b v050_sparse timeout -k 10 300 python bench.py --no-cpu-baseline --visibility 0.5 --steps 70
b closeup_auto timeout -k 10 300 python bench.py --no-cpu-baseline --neighbors 6 10 --steps 70
VMM_BA_SCHUR=dense b closeup_dense timeout -k 10 300 python bench.py --no-cpu-baseline --neighbors 6 10 --steps 70
b incremental timeout -k 10 600 python bench.py --no-cpu-baseline --workload incremental
cut -c1-1500 $O/incremental.json
cat > /tmp/mid.py <<'PY'
import json, numpy as np
from visual_marker_mapping_amd import engine as eng
from visual_marker_mapping_amd.synthetic import make_scene
out = {}
s = make_scene(2)
ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
ba.solve(eng.default_options(robustify=0))
out["overlap_500x200_ms"] = ba.debug_overlap(20)
ba.close()
print(out["overlap_500x200_ms"], flush=True)
for (nc, nt) in ((400, 250), (600, 320), (800, 400), (1000, 500)):
    s = make_scene(2, n_cams=nc, n_tags=nt)
    ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
    o = ba.solve(eng.default_options(robustify=0))
    kt = ba.time_kernels(eng.default_options(robustify=0), reps=5)
    r = dict(blocks=(6 * nt + 63) // 64, iterations=o["num_lm_iterations"], final_cost=o["final_cost"], sync_timeouts=o["num_sync_timeouts"],
             cholesky_us=kt["cholesky_ms"] * 1e3, syrk_us=kt["syrk_ms"] * 1e3, iteration_us=kt["lm_iteration_ms"] * 1e3)
    out["%dx%d" % (nc, nt)] = r
    print(nc, nt, r, flush=True)
    ba.close()
import os, sys
json.dump(out, open(sys.argv[1], "w"), indent=1)
PY
timeout -k 10 400 python /tmp/mid.py $O/mid_dataflow.json || exit 1
VMM_BA_NO_DATAFLOW=1 timeout -k 10 400 python /tmp/mid.py $O/mid_steps.json || exit 1
