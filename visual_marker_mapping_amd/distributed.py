"""One process per GPU: observation sharding and the all-reduce hook of the engine.

The reference is single-process ("Parallelism strategies: none"); this is the
MI355X side of BASELINE.json configs[2].  Observations are sharded by the ELIMINATED pose family
(cameras by default: "observations shard naturally by camera"), every rank keeps all poses, and per
LM iteration the ranks exchange by sum-all-reduce: the small normal-equation blocks, gradient and cost of
the evaluation at the candidate, the reduced system, and the eliminated family's step.
"""
import numpy as np


def shard_bounds(counts, world_size):
    """Splits poses 0..n-1 into world_size contiguous groups with balanced observation counts.

    Returns an int array b of length world_size+1; rank r owns poses b[r] .. b[r+1]-1.
    """
    counts = np.asarray(counts, np.int64)
    n = len(counts)
    total = int(counts.sum())
    bounds = [0]
    acc, p = 0, 0
    for r in range(1, world_size):
        target = total * r / world_size
        while p < n and acc + counts[p] / 2.0 <= target:
            acc += int(counts[p])
            p += 1
        # leave at least one pose for every remaining rank when possible
        p = min(p, max(bounds[-1], n - (world_size - r)))
        p = max(p, min(bounds[-1] + 1, n))
        bounds.append(p)
    bounds.append(n)
    return np.asarray(bounds, np.int64)


def shard_observations(obs_cam, obs_tag, n_cams, n_tags, rank, world_size, eliminate_cameras=None):
    """Index array of the observations rank `rank` owns (caller order preserved)."""
    obs_cam = np.asarray(obs_cam)
    obs_tag = np.asarray(obs_tag)
    if eliminate_cameras is None:
        eliminate_cameras = n_cams >= n_tags   # VMM_BA_ELIM_AUTO
    key = obs_cam if eliminate_cameras else obs_tag
    n = n_cams if eliminate_cameras else n_tags
    counts = np.bincount(key, minlength=n)
    b = shard_bounds(counts, world_size)
    return np.nonzero((key >= b[rank]) & (key < b[rank + 1]))[0], bool(eliminate_cameras)


class _DeviceBuffer:
    """Zero-copy view of `count` doubles at a raw device pointer for torch.as_tensor."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False),
                                         "version": 2}


def make_allreduce(device_index, group=None):
    """Returns fn(ptr, count, stream) for BundleAdjuster.set_allreduce using torch.distributed.

    Backend nccl (= RCCL over xGMI): reduces the device buffer in place on the engine's stream.
    Backend gloo (CPU tests, or several ranks sharing one GPU): stages through host memory.
    """
    import torch
    import torch.distributed as dist

    backend = dist.get_backend(group)
    dev = torch.device("cuda", device_index)

    def fn(ptr, count, stream):
        ext = torch.cuda.ExternalStream(stream, device=dev) if stream else torch.cuda.current_stream(dev)
        with torch.cuda.stream(ext):
            t = torch.as_tensor(_DeviceBuffer(ptr, count), device=dev)
            if backend == "nccl":
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            else:
                h = t.cpu()          # synchronises `ext`
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                t.copy_(h)
    return fn


def enable_native_rccl(ba, rank, group=None):
    """Attaches an RCCL communicator to the handle (BundleAdjuster.enable_rccl): rank 0 draws the id, the process
    group that launched the ranks (torch.distributed, any backend) carries its 128 bytes to the others.  From then
    on the library issues ncclAllReduce itself, inside the iteration's hipGraph: no host callback per collective."""
    import torch
    import torch.distributed as dist
    from . import engine as eng

    payload = [None]
    if rank == 0:
        try:
            payload[0] = eng.rccl_unique_id()
        except Exception as exc:   # noqa: BLE001 -- carried to every rank so that all of them raise together
            payload[0] = "rank 0 could not draw an RCCL id: %s" % exc
    dist.broadcast_object_list(payload, src=0, group=group)
    # Every rank says whether it can enter ncclCommInitRank (library resolved, id received) and the ranks agree on
    # the minimum over the launcher's process group BEFORE anyone calls vmm_ba_enable_rccl: a rank that raised here on
    # its own would leave the others blocked inside ncclCommInitRank waiting for it.
    mine_ok = isinstance(payload[0], (bytes, bytearray)) and eng.rccl_available()
    ready = torch.tensor([1 if mine_ok else 0], dtype=torch.int32,
                         device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(ready, op=dist.ReduceOp.MIN, group=group)
    if int(ready.item()) != 1:
        if not isinstance(payload[0], (bytes, bytearray)):
            raise RuntimeError(str(payload[0]))
        raise RuntimeError("librccl.so is not loadable on %s" % ("this rank" if not mine_ok else "another rank"))
    ba.enable_rccl(payload[0])
