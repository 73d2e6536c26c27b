"""bench.py's secondary lines say what was done: no roofline fraction above 1 (a fraction above 1 means the label is not the
work -- round 3 priced the block-sparse elimination with the dense flop count, and before vmm_ba_kernel_times.chol_flops
a tree-ordered factorisation read 0.91 of the MFMA peak with the dense n^3 / 3), the workload string names the scene, the
kernel names follow the path taken."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(*args):
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "14", "--warmup", "7"] + list(args)
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.parametrize("args,sparse,tree", [((), False, False), (("--neighbors", "6", "10"), True, True),
                                              (("--visibility", "0.25"), True, False)])
def test_secondary_lines_price_the_work_that_is_done(args, sparse, tree):
    d = _line(*args)
    assert d["metric"] == "lm_iterations_per_sec" and d["n_gpus"] == 1 and d["steps"] == 14 and d["value"] > 0
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 1.0) < 1e-6
    r = d["roofline"]
    assert 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    for name, k in d["kernels"].items():
        if "frac" in k and k["frac"] is not None:
            assert 0.0 <= k["frac"] < 1.0, (name, k)
    cfg = d["config"]
    assert ("block-sparse" in cfg["elimination_form"]) == sparse
    assert cfg["kept_family_order"].startswith("nested-dissection") == tree
    if args and args[0] == "--neighbors":
        assert "6" in cfg["workload"] and "10" in cfg["workload"] and "visibility 1.00" not in cfg["workload"]
    if not sparse:
        assert d["kernels"]["schur_syrk"]["traffic"] is None or d["kernels"]["schur_syrk"]["traffic"] > 0
    else:
        assert d["kernels"]["schur_syrk"]["traffic"] is None   # the committed counters are of the dense headline command only
