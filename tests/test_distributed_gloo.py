"""world_size-2 gloo test (CPU) of the multi-rank exchange protocol of SURVEY.md 8(e):
each rank accumulates the normal-equation blocks of ITS shard of observations (sharded by camera),
one sum-all-reduce makes them global.  The per-observation arithmetic here is the oracle's -- this
checks the sharding + reduction logic of visual_marker_mapping_amd.distributed, not the HIP kernels
(their 2-rank run is tests/test_gpu_distributed.py)."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _blocks(O, s, idx):
    n_c, n_t = len(s.cam_init), len(s.tag_init)
    buf = np.zeros(42 * (n_c + n_t) + 1)
    V = buf[:36 * n_c].reshape(n_c, 6, 6)
    U = buf[36 * n_c:36 * (n_c + n_t)].reshape(n_t, 6, 6)
    g = buf[36 * (n_c + n_t):-1].reshape(n_c + n_t, 6)
    for i in idx:
        c, t = s.obs_cam[i], s.obs_tag[i]
        r, Jc, Jt = O.obs_eval(s.intr, s.dist, s.cam_init[c], s.tag_init[t], s.tag_wh[t], s.obs_px[i])
        if t == s.fixed_tag:
            Jt[:] = 0
        V[c] += Jc.T @ Jc
        U[t] += Jt.T @ Jt
        g[c] += Jc.T @ r
        g[n_c + t] += Jt.T @ r
        buf[-1] += 0.5 * r @ r
    return buf


def _worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    from visual_marker_mapping_amd import distributed as vd
    from visual_marker_mapping_amd.synthetic import make_scene
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    s = make_scene(1, visibility=0.6)
    idx, elim_cams = vd.shard_observations(s.obs_cam, s.obs_tag, len(s.cam_init), len(s.tag_init), rank, world)
    assert elim_cams
    t = torch.from_numpy(_blocks(O, s, idx))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.save(out, np.concatenate([t.numpy(), [len(idx)]]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_block_reduction_equals_single_rank(tmp_path, oracle):
    mp = pytest.importorskip("torch.multiprocessing")
    from visual_marker_mapping_amd.synthetic import make_scene
    out = str(tmp_path / "r0.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    s = make_scene(1, visibility=0.6)
    ref = _blocks(oracle, s, range(s.n_obs))
    np.testing.assert_allclose(got[:-1], ref, rtol=1e-12, atol=1e-9)
    assert 0 < got[-1] < s.n_obs          # rank 0 really held only a shard
