#!/usr/bin/env python3
"""Turns the per-kernel PMC sums of tools/gpu_pmc.sh (gpurun_out/pmc/pmc_summary.json) into HBM bytes
per launch for bench.py's roofline.traffic.

Corrections per /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of wide coalesced streaming reads, so it is
doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.  Other access widths are
uncalibrated (the guide says so) -- the small kernels' numbers are indicative only.
"""
import json
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc/pmc_summary.json"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r03_pmc_traffic.json"
d = json.load(open(src))


def per_launch(counter, name):
    for k, v in d.get(counter, {}).items():
        if k.strip().endswith(name) or name in k:
            # skipped launches (ctl->done) dilute the mean only slightly: ~1 in 8 dispatches
            return v["sum"] / max(v["dispatches"], 1)
    return 0.0


def traffic(names):
    tot = 0.0
    for n in names:
        tot += 2.0 * per_launch("FETCH_SIZE", n) * 1024.0 + per_launch("WRITE_SIZE", n) * 1024.0
    return tot


groups = {
    "eval_jacobian": ["k_eval_both"],
    "eval_cost": ["k_cost<true, false>", "k_cost<true>"],
    "schur_syrk": ["k_syrk_streamk", "k_syrk_wide"],
    "syrk_reduce": ["k_reduce_partials<true>"],
    "form_z": ["k_form_z"],
    "chol_step": ["k_chol_step"],
    "chol_dataflow": ["k_chol_dataflow"],
    "backsolve_chain": ["k_backsolve_chain"],
    "backsub": ["k_backsub"],
    "reduce_pose": ["k_reduce_pose"],
}
bpl = {g: traffic(n) for g, n in groups.items()}
# Calibration of the x2 read correction in THIS code's access patterns (the guide states it for 16-B-per-lane streams
# and calls other widths uncalibrated): k_cost reads 8 B per lane, coalesced, and nothing else of size -- 72 B per tag
# observation = 7.2 MB at 500 x 200 -- and writes a few KB.  corrected / algorithmic should be ~1 (L2 line granularity
# and the 39 KB of poses on top); ~0.5 would mean the correction does not apply to 8-B-per-lane loads.
n_obs = 100000
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --steps 14 --warmup 7`, "
                 "500x200; bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 per launch (gfx950 correction)",
       "calibration": {"kernel": "k_cost (8-B-per-lane coalesced SoA reads)", "algorithmic_bytes": 72.0 * n_obs,
                       "corrected_bytes": bpl["eval_cost"],
                       "ratio": bpl["eval_cost"] / (72.0 * n_obs) if bpl["eval_cost"] else None},
       "bytes_per_launch": bpl}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
