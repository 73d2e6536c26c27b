#!/usr/bin/env python3
"""bench.py -- LM iterations/s and residual-evaluations/s of the bundle-adjustment hot path.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): synthetic scene of
500 images x 200 tags, every image sees every tag (100 000 tag observations = 400 000 corner
residual blocks), f64 residuals/Jacobians, non-robust unless --config 5.  One "step" = one
Levenberg-Marquardt iteration (linear solve + candidate cost evaluation, + a Jacobian evaluation
after every accepted step) following Ceres' trust-region policy.  Steps are executed as back-to-back
solves from the perturbed initial state (each solve runs its natural ~7 iterations); the last solve
is capped so that exactly --steps iterations are timed.  Inputs are resident in HBM before the timed
region; only 700 poses (39 KB) are re-uploaded per solve.

N > 1: launched by torch.distributed.run, one rank per GPU, observations sharded by camera, the
reduced system all-reduced over RCCL -> "scaling": "strong" (the scene is fixed).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F64_MFMA_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix = FP64 vector datasheet rate;
                              # MI355X_MICROARCH.md has no f64 row, tools/mfma_f64_peak.hip measures it
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=140)
    ap.add_argument("--warmup", type=int, default=14)
    ap.add_argument("--config", type=int, default=2, help="BASELINE.json configs index (1-based)")
    ap.add_argument("--visibility", type=float, default=1.0)
    ap.add_argument("--neighbors", type=int, nargs=2, default=None, metavar=("MIN", "MAX"),
                    help="close-up scene: every image sees the MIN..MAX tags nearest to the wall point it looks at")
    ap.add_argument("--wall-rows", type=int, default=0,
                    help="with --neighbors: the tags hang in this many rows (1 or 2: a corridor) instead of a wall")
    ap.add_argument("--poll", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--collective", choices=["rccl", "callback"], default="rccl",
                    help="N > 1: 'rccl' = the library's own ncclAllReduce recorded into the iteration graph "
                         "(vmm_ba_enable_rccl); 'callback' = host callback into torch.distributed per collective")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="process-group backend for N > 1.  'nccl' (= RCCL over xGMI) is what the driver's multi-GPU run "
                         "uses.  'gloo' rehearses the same N-rank harness (sharding, fences, MAX-reduced elapsed time) on ONE "
                         "GPU: every rank uses device 0 and the collectives go through the host callback "
                         "(--collective callback is implied); its value prices the harness, not the links")
    ap.add_argument("--elimination", choices=["auto", "cams", "tags"], default="auto")
    ap.add_argument("--workload", choices=["ba", "incremental"], default="ba",
                    help="'ba' (default): the LM loop of one full-size bundle adjustment -- the metric of BASELINE.json. "
                         "'incremental': a SECONDARY line, never the headline value -- wall time of startReconstruction's "
                         "N + 2 growing solves (src/TagReconstructor.cpp:233-277) on a 100 x 60, visibility 0.3 project, "
                         "device-resident handle against one vmm_ba_create per problem")
    ap.add_argument("--precision", choices=["f64", "f32"], default=None,
                    help="f32 = J^T J blocks accumulated/stored in f32, everything else f64 (default for --config 4, "
                         "as BASELINE.json configs[3] words it); f64 otherwise")
    return ap.parse_args()


def run_steps(ba, eng, s, opts_kw, n_steps):
    """Runs exactly n_steps LM iterations as consecutive solves from the initial state."""
    done = 0
    evals = 0
    solves = 0
    last = None
    while done < n_steps:
        ba.set_state(s.cam_init, s.tag_init)
        out = ba.solve(eng.default_options(max_num_iterations=n_steps - done, **opts_kw))
        if out["num_lm_iterations"] <= 0:
            raise RuntimeError("solve made no progress: %r" % (out,))
        done += out["num_lm_iterations"]
        # passes over the observations actually executed: one combined residual + Jacobian evaluation at the
        # candidate per LM iteration (it serves as Ceres' cost evaluation and, when accepted, as its Jacobian
        # evaluation) plus the one of iteration zero
        evals += out["num_cost_evals"] + 1
        solves += 1
        last = out
    return done, evals, solves, last


def incremental_workload(a):
    """Wall time of the incremental driver (N + 2 bundle adjustments of growing size
    interleaved with prunings) with the device-resident handle and with a handle per problem."""
    import contextlib
    import io
    import torch
    from visual_marker_mapping_amd.synthetic import make_scene
    from visual_marker_mapping_amd.tag_reconstructor import CameraModel, TagReconstructor, detection_result_from_arrays
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible")
    n_cams, n_tags, vis = 100, 60, 0.3
    s = make_scene(2, n_cams=n_cams, n_tags=n_tags, visibility=vis)
    res = {}
    for mode in ("resident", "per_problem", "resident"):     # the first run also pays the one-off code-object load
        det = detection_result_from_arrays(s.obs_cam, s.obs_tag, s.obs_px, s.tag_wh, n_cams)
        rec = TagReconstructor(det)
        rec.setCameraModel(CameraModel(*[float(v) for v in s.intr], s.dist, 4000, 6000))
        buf = io.StringIO()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(buf):
            rec.startReconstruction(1, deviceResident=(mode == "resident"))
        dt = time.perf_counter() - t0
        res[mode] = dict(seconds=dt, bundle_adjustments=buf.getvalue().count("Solution "),
                         cameras=len(rec.getReconstructedCameras()), tags=len(rec.getReconstructedTags()))
        rec.close()
    n_ba = res["resident"]["bundle_adjustments"]
    line = {"metric": "incremental_reconstruction_wall_time", "value": res["resident"]["seconds"], "unit": "s",
            "n_gpus": 1, "steps": n_ba, "warmup": 0, "ms_per_step": 1e3 * res["resident"]["seconds"] / max(n_ba, 1),
            "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "secondary": True,
            "config": {"workload": "startReconstruction call pattern (src/TagReconstructor.cpp:233,236,271-277): %d images x "
                                   "%d tags, visibility %.2f, %d tag observations; N + 2 = %d bundle adjustments + prunings; "
                                   "PnP initialisation on the host (pnp.py) included" % (n_cams, n_tags, vis, s.n_obs, n_ba)},
            "resident_handle_s": res["resident"]["seconds"], "handle_per_problem_s": res["per_problem"]["seconds"],
            "reconstructed": {k: res["resident"][k] for k in ("cameras", "tags")}}
    print(json.dumps(line))


def main():
    a = parse()
    if a.workload == "incremental":
        return incremental_workload(a)
    import torch
    import torch.distributed as dist
    from visual_marker_mapping_amd import engine as eng
    from visual_marker_mapping_amd import distributed as vdist
    from visual_marker_mapping_amd.synthetic import make_scene

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (a.gpus, a.gpus))
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, a.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible")
    gloo = a.backend == "gloo"
    if gloo:
        local_rank = 0          # all ranks share the one GPU of the box
        a.collective = "callback"
    torch.cuda.set_device(local_rank)
    # VMM_BA_FORCE_COLLECTIVES=1 (test hook) runs the multi-rank code path with a single rank
    use_dist = world > 1 or os.environ.get("VMM_BA_FORCE_COLLECTIVES") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    overrides = {}
    if a.visibility < 1.0:
        overrides["visibility"] = a.visibility
    if a.neighbors:
        overrides["neighbors_min"], overrides["neighbors_max"] = a.neighbors
    if a.wall_rows > 0:
        overrides["wall_rows"] = a.wall_rows
    s = make_scene(a.config, **overrides)
    n_cams, n_tags = len(s.cam_init), len(s.tag_init)
    elim = {"auto": eng.ELIM_AUTO, "cams": eng.ELIM_CAMERAS, "tags": eng.ELIM_TAGS}[a.elimination]
    elim_cams = None if a.elimination == "auto" else (a.elimination == "cams")
    idx, elim_cams = vdist.shard_observations(s.obs_cam, s.obs_tag, n_cams, n_tags, rank, world, elim_cams)
    precision = a.precision or ("f32" if a.config == 4 else "f64")
    t0 = time.time()
    ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam[idx],
                            s.obs_tag[idx], s.obs_px[idx], device=local_rank, elimination=elim, rank=rank,
                            world_size=world,
                            precision=eng.PRECISION_F32_ACCUM if precision == "f32" else eng.PRECISION_F64,
                            # every rank generates the whole scene: the structure of all ranks' observations lets them
                            # order the kept family alike (tree ordering on close-up scenes; no effect at visibility 1.0)
                            structure_obs=(s.obs_cam, s.obs_tag) if world > 1 else None)
    setup_s = time.time() - t0
    collective_used = None
    if use_dist:
        collective_used = a.collective
        if a.collective == "rccl":
            # ncclAllReduce inside the iteration's hipGraph.  If the library's own communicator cannot be set up on
            # some rank (librccl.so not loadable, ncclCommInitRank error), ALL ranks fall back to the host-callback
            # path together -- the decision is all-reduced so that nobody is left in a collective alone.
            err = None
            try:
                vdist.enable_native_rccl(ba, rank)
            except Exception as exc:   # noqa: BLE001 -- reported in the JSON line
                err = exc
            bad = torch.tensor([1.0 if err is not None else 0.0], device="cuda")
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if float(bad.item()) > 0.0:
                if err is None:
                    # this rank has a communicator the others do not: start over on the callback path
                    ba.close()
                    ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag,
                                            s.obs_cam[idx], s.obs_tag[idx], s.obs_px[idx], device=local_rank,
                                            elimination=elim, rank=rank, world_size=world,
                                            precision=eng.PRECISION_F32_ACCUM if precision == "f32" else eng.PRECISION_F64,
                                            structure_obs=(s.obs_cam, s.obs_tag))
                ba.set_allreduce(vdist.make_allreduce(local_rank))
                collective_used = "callback (native RCCL set-up failed%s)" % (": %s" % err if err is not None else " on another rank")
        else:
            ba.set_allreduce(vdist.make_allreduce(local_rank))   # host callback -> torch.distributed
    robust = 1 if s.robustify else 0
    opts_kw = dict(robustify=robust, poll_interval=a.poll)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if a.warmup > 0:
        run_steps(ba, eng, s, opts_kw, a.warmup)
    fence()
    t0 = time.perf_counter()
    done, evals, solves, last = run_steps(ba, eng, s, opts_kw, a.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if gloo else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert done == a.steps, (done, a.steps)

    line = None
    # what the scene is: BASELINE's configurations are visibility-1.0 walls; --visibility / --neighbors / --wall-rows
    # give secondary scenes and are named as such
    standard_scene = a.visibility == 1.0 and not a.neighbors and a.wall_rows == 0
    if a.neighbors:
        scene_desc = "close-up scene: every image sees its %d..%d nearest tags%s" % (
            a.neighbors[0], a.neighbors[1], (", tags in %d row(s)" % a.wall_rows) if a.wall_rows > 0 else "")
    else:
        scene_desc = "visibility %.2f" % a.visibility
    if rank == 0:
        n_obs_total = s.n_obs
        it_per_s = a.steps / elapsed
        res_evals_per_s = 4.0 * n_obs_total * evals / elapsed
        line = {
            "metric": "lm_iterations_per_sec", "value": it_per_s, "unit": "LM iterations/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64" if precision == "f64" else "f32 J^T J accumulation, f64 residuals/gradient/reduced system/solve",
            "data": "synthetic",
            "config": {"workload": "configs[%d]%s: %d images x %d tags, %s, %d tag observations "
                                   "(%d corner residual blocks), %s, perturbed initial guess"
                                   % (a.config - 1, "" if standard_scene else " (modified scene)", n_cams, n_tags, scene_desc,
                                      n_obs_total, 4 * n_obs_total, "Huber(1.0)" if robust else "no loss"),
                       "eliminated_family": "cameras" if elim_cams else "tags",
                       "reduced_system_order": 6 * (n_tags if elim_cams else n_cams),
                       "elimination_form": "block-sparse (k_schur_pairs)" if last.get("block_sparse") else "dense Z + MFMA rank-k",
                       "kept_family_order": ("nested-dissection tree, %d nodes" % last["tree_ordering"]) if last.get("tree_ordering")
                                            else "natural",
                       "solves_timed": solves, "lm_iterations_per_solve": last["num_lm_iterations"],
                       "sharding": "observations by %s" % ("camera" if elim_cams else "tag")},
            "residual_evals_per_sec": res_evals_per_s,
            "residual_evals_per_sec_is": "whole-loop average: corner residual blocks evaluated (with Jacobians) / wall "
                                         "time of the timed LM iterations, linear solve included; the evaluation "
                                         "kernels alone are kernels.eval_cost / kernels.eval_jacobian "
                                         ".residual_evals_per_sec (SURVEY.md 8(d): K1 and K2 reported separately)",
            "setup_s": setup_s,
        }
        if collective_used is not None:
            line["collective"] = collective_used
    if world == 1 and not use_dist:
        ba.set_state(s.cam_init, s.tag_init)
        kt = ba.time_kernels(eng.default_options(**opts_kw), reps=10)
        n_obs = kt["n_obs"]
        n_red, k_dim = kt["reduced_dim"], kt["elim_dim"]
        n_aug = n_red + 1
        kern = {
            # algorithmic bytes per tag observation: SURVEY.md 8(d) -- 360 B for the fused
            # residual+Jacobian+accumulate evaluation (72 B in + one f64 6x6 W block out), 72 B cost-only.
            # Timed as the iteration runs it: k_eval_both (both family passes) + k_reduce_pose, two launches; since
            # round 2 this is the ONLY evaluation of an LM iteration (at the candidate: its cost decides the step,
            # its blocks are the next iteration's when the step is accepted).
            "eval_jacobian": {"ms": kt["eval_elim_ms"] + kt["eval_keep_ms"], "bound": "hbm",
                              "alg": (360.0 if precision == "f64" else 216.0) * n_obs, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s"},
            # cost-only pass (vmm_ba_cost): no longer part of an LM iteration, reported for its HBM roofline
            "eval_cost": {"ms": kt["cost_ms"], "bound": "hbm", "alg": 72.0 * n_obs, "peak": HBM_PEAK_GBS,
                          "unit": "GB/s"},
            # dense form: S = Z^T Z, lower triangle incl. the rhs row: (n+1)(n+2)/2 * K multiply-adds.  Block-sparse
            # form (k_schur_pairs): the flops of the terms that exist (2 * 36 * 6 per term of a pair, counted by the
            # library when it builds the plan) -- pricing it with the dense count would credit work that is not done
            "schur_syrk": {"ms": kt["syrk_ms"], "bound": "mfma",
                           "alg": float(kt["schur_flops"]) if kt.get("schur_sparse") else 1.0 * n_aug * (n_aug + 1) * k_dim,
                           "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s"},
            # Cholesky n^3/3 + forward/backward substitution 2 n^2; a tree-ordered factor is priced with the flops of
            # its non-zero blocks (counted by the library from the symbolic factor), not with the dense count
            "cholesky_solve": {"ms": kt["cholesky_ms"], "bound": "mfma",
                               "alg": float(kt["chol_flops"]) if kt.get("chol_flops") else n_red ** 3 / 3.0 + 2.0 * n_red ** 2,
                               "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s"},
        }
        for k, v in kern.items():
            scale = 1e-9 if v["unit"] == "GB/s" else 1e-12
            v["achieved"] = v["alg"] * scale / (v["ms"] * 1e-3) if v["ms"] > 0 else 0.0
            v["frac"] = v["achieved"] / v["peak"]
        # HBM bytes per launch: NOT measured in this run -- read from the PMC passes committed under profiles/
        # (tools/gpu_final.sh + tools/pmc_to_traffic.py: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this
        # same command, gfx950 x2 read correction calibrated on k_cost's known 7.2 MB); traffic_source says so
        traffic = {}
        tname = "r04_pmc_traffic.json"
        tpath = os.path.join(ROOT, "profiles", tname)
        # the committed counters are of the headline command only: dense elimination of the standard 500 x 200 scene
        if (a.config == 2 and standard_scene and not kt.get("schur_sparse") and not last.get("tree_ordering")
                and precision == "f64" and os.path.exists(tpath)):
            with open(tpath) as f:
                bpl = json.load(f)["bytes_per_launch"]
            traffic = {"eval_jacobian": bpl.get("eval_jacobian"), "eval_cost": bpl.get("eval_cost"),
                       "schur_syrk": bpl.get("schur_syrk"),
                       "cholesky_solve": (bpl.get("chol_dataflow") or 0) + (bpl.get("backsolve_chain") or 0)}
        n_blk = (n_red + 63) // 64
        dataflow = n_blk <= 48          # dataflow_max_workgroups (csrc/kernels_chol.hip)
        dom = max(kern, key=lambda k: kern[k]["ms"])
        d = kern[dom]
        tree = int(last.get("tree_ordering", 0)) if isinstance(last, dict) else 0
        names = {"cholesky_solve": ("k_chol_dataflow_tree + k_backsolve_chain_tree (kept family ordered by a nested-dissection "
                                    "tree of %d nodes; natural order: %d block columns)" % (tree, n_blk)) if tree else
                                   ("k_chol_dataflow (one launch: %d workgroups, %d block columns) + k_backsolve_chain"
                                    % (n_blk * (n_blk + 1) // 2 + n_blk, n_blk)) if dataflow else
                                   ("cholesky_solve = %d x k_chol_step + k_chol_dataflow on the last block columns + "
                                    "k_backsolve_chain" % max(n_blk - 34, 0)),
                 "schur_syrk": "k_form_z<SPARSE> + k_schur_pairs" if kt.get("schur_sparse") else
                               ("k_syrk_wide" if kt.get("syrk_wide") else "k_syrk_streamk")
                               + " (the rank-k kernel alone: its partial-tile sum k_reduce_partials is a launch of its own, "
                                 "9 us at 500 x 200, profiles/r04_kernel_stats_working.csv)",
                 "eval_jacobian": "k_eval_both + k_reduce_pose", "eval_cost": "k_cost"}
        line["roofline"] = {"kernel": names.get(dom, dom), "bound": d["bound"], "achieved": d["achieved"],
                            "peak": d["peak"], "unit": d["unit"], "frac": d["frac"], "traffic": traffic.get(dom),
                            "traffic_source": ("profiles/%s (committed rocprofv3 --pmc passes of this command; not "
                                               "collected in this run)" % tname) if traffic.get(dom) else None,
                            "avg_launch_ms": d["ms"]}
        if dom == "cholesky_solve":
            line["roofline"]["launches"] = ({"k_chol_dataflow": 1, "k_backsolve_chain": 1} if dataflow
                                            else {"k_chol_step": n_blk, "k_backsolve_chain": 1})
            if dataflow:
                # informational: what actually bounds this group (measured with the stamps build) --
                # 8 rounds of 8 dependent pivots per block column at ~1.0 us, one granule hand-off (~1.3 us) between
                # block columns, then n_blk hops of ~1.1 us in the back-substitution
                floor_ms = 1e-3 * (8 * n_blk * 1.0 + (n_blk - 1) * 1.3 + 4.4 + (n_blk - 1) * 1.1)
                line["roofline"]["latency_model"] = {"floor_ms": floor_ms, "frac_of_floor": floor_ms / d["ms"],
                                                     "source": "DESIGN.md sections 4.1 and 4.4 (in-kernel time stamps)"}
            line["roofline"]["note"] = ("avg_launch_ms, achieved and traffic are for one whole factorisation + solve (the "
                                        "launches above, back to back); algorithmic flops %s; bound by the %d "
                                        "dependent pivots (a chain of 8-column rounds at ~1 us each), not by the matrix "
                                        "cores (DESIGN.md section 4)"
                                        % ("of the non-zero blocks of the tree-ordered factor (vmm_ba_kernel_times.chol_flops)"
                                           if kt.get("chol_flops") else "n^3/3 + 2 n^2", n_red))
        line["kernels"] = {k: {"ms": v["ms"], "bound": v["bound"], "achieved": v["achieved"],
                               "unit": v["unit"], "frac": v["frac"], "traffic": traffic.get(k)}
                           for k, v in kern.items()}
        # residual evaluations per second of K1 (cost only) and K2 (residual + Jacobian + accumulation)
        # by themselves: 4 corner blocks per tag observation / the kernel group's own time
        for k in ("eval_cost", "eval_jacobian"):
            if kern[k]["ms"] > 0:
                line["kernels"][k]["residual_evals_per_sec"] = 4.0 * n_obs / (kern[k]["ms"] * 1e-3)
        line["kernels"]["schur_syrk"]["kernel"] = names["schur_syrk"]
        line["kernels"]["form_z"] = {"ms": kt["form_z_ms"]}
        line["kernels"]["backsub"] = {"ms": kt["backsub_ms"]}
        line["kernels"]["lm_iteration_enqueued"] = {"ms": kt["lm_iteration_ms"]}
        # where the device time of the timed solves went, measured on the device (vmm_ba_summary.time_*_s of the last one)
        line["phase_report_last_solve_s"] = {k: last[k] for k in ("time_eval_s", "time_eliminate_s", "time_factor_solve_s",
                                                                    "time_step_s", "time_control_s", "time_solve_s")}
        if not a.no_cpu_baseline:
            from oracle import oracle as O
            # Thread counts tried: every visible core (capped at 64) and the CPU share of a one-GPU box (16: with a
            # quota smaller than the visible core count, 64 threads oversubscribe and run at half the speed); the
            # faster one is reported, `cores` = its thread count.
            n_vis = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 64)
            best = None
            for threads in sorted({n_vis, min(n_vis, 16)}, reverse=True):
                # untimed: one LM iteration on a copy starts the OpenMP team and touches the work arrays (the first
                # parallel region of a process costs ~0.5 s on a 64-thread host, half of a 7-iteration solve)
                scw = O.Scene(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag,
                              s.obs_px)
                O.solve(scw, O.default_options(robustify=robust, num_threads=threads, max_num_iterations=1,
                                               linear_solver=O.SCHUR_AUTO))
                sc = O.Scene(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag,
                             s.obs_px)
                t0 = time.perf_counter()
                summ, _ = O.solve(sc, O.default_options(robustify=robust, num_threads=threads,
                                                        linear_solver=O.SCHUR_AUTO))
                dt_t = time.perf_counter() - t0
                if best is None or summ["num_cost_evals"] / dt_t > best[0]:
                    best = (summ["num_cost_evals"] / dt_t, threads, summ["num_cost_evals"], dt_t)
            _, threads, cpu_iters, dt = best   # one cost evaluation per LM iteration
            # the same oracle on ONE thread, bounded to two LM iterations (single-threaded and
            # all-cores timings)
            sc1 = O.Scene(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag,
                          s.obs_px)
            t0 = time.perf_counter()
            summ1, _ = O.solve(sc1, O.default_options(robustify=robust, num_threads=1, max_num_iterations=2,
                                                      linear_solver=O.SCHUR_AUTO))
            dt1 = time.perf_counter() - t0
            line["cpu_baseline"] = {"value": cpu_iters / dt, "unit": "LM iterations/s", "cores": threads,
                                    "kind": "port",
                                    "sample": "one complete solve of the same workload (%d LM iterations, %.2f s) "
                                              "by oracle/liboracle.so (C + OpenMP, Schur elimination of the same "
                                              "family); Ceres itself is not installable here" % (cpu_iters, dt),
                                    "single_thread_value": max(summ1["num_cost_evals"], 1) / dt1,
                                    "single_thread_sample": "%d LM iterations, %.2f s on one thread"
                                                            % (summ1["num_cost_evals"], dt1)}
    ba.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
