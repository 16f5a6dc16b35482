"""Frame sharding across the GPUs of one node.

Frames are independent given fixed cameras (every cross-frame state of the reference -- Kalman,
low-pass -- is downstream of the hot path), so the path shards with NO data-path collective:
one process per GPU, contiguous frame blocks, camera tables replicated (a few KB).  The only
exchange is the final gather of fixed-stride track records to rank 0 -- one collective call on
one packed byte tensor (RCCL over xGMI on the GPU box: 7 point-to-point links into the root;
gloo in the CPU tests).

Record layout per frame (bytes, little endian), K = K_max:
    n_out  int32   [1]  (+4 pad)
    xyz    float64 [K][3]
    err    float64 [K]
    corr   int16   [K][C]   (padded to 8 bytes)
"""
import os

import numpy as np


def shard_bounds(n_items, rank, world_size):
    """Contiguous block of `n_items` owned by `rank` (first n_items % world_size ranks get one more)."""
    base, rem = divmod(int(n_items), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def record_bytes(C, K_max):
    corr = (2 * K_max * C + 7) // 8 * 8
    return 8 + 24 * K_max + 8 * K_max + corr


def env_rank():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_process_group(backend=None):
    """One process per GPU, launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" IS RCCL on ROCm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def pack_records(n_out, xyz, err, corr):
    """torch tensors of one shard -> uint8 [F][record_bytes] (device-side views/copies, no host trip)."""
    import torch
    F, K = err.shape
    C = corr.shape[2]
    rb = record_bytes(C, K)
    rec = torch.zeros((F, rb), dtype=torch.uint8, device=err.device)
    rec[:, 0:4] = n_out.contiguous().view(torch.uint8).reshape(F, 4)
    o = 8
    rec[:, o:o + 24 * K] = xyz.contiguous().view(torch.uint8).reshape(F, 24 * K)
    o += 24 * K
    rec[:, o:o + 8 * K] = err.contiguous().view(torch.uint8).reshape(F, 8 * K)
    o += 8 * K
    rec[:, o:o + 2 * K * C] = corr.contiguous().view(torch.uint8).reshape(F, 2 * K * C)
    return rec


def unpack_records(rec, C, K_max):
    """uint8 [F][record_bytes] (any device) -> dict of numpy arrays."""
    a = rec.detach().cpu().numpy()
    F = a.shape[0]
    K = K_max
    n_out = a[:, 0:4].copy().view(np.int32).reshape(F)
    o = 8
    xyz = a[:, o:o + 24 * K].copy().view(np.float64).reshape(F, K, 3)
    o += 24 * K
    err = a[:, o:o + 8 * K].copy().view(np.float64).reshape(F, K)
    o += 8 * K
    corr = a[:, o:o + 2 * K * C].copy().view(np.int16).reshape(F, K, C)
    return {"n_out": n_out, "xyz": xyz, "err": err, "corr": corr}


def gather_records(rec, dst=0):
    """The single exchange step: gather every rank's packed records on `dst`.
    All shards must have the same number of frames (pad the last shard).  Returns the
    concatenated [world*F][rb] tensor on dst, None elsewhere."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rec
    world, rank = dist.get_world_size(), dist.get_rank()
    if rank == dst:
        out = torch.empty((world,) + tuple(rec.shape), dtype=rec.dtype, device=rec.device)
        dist.gather(rec, gather_list=list(out.unbind(0)), dst=dst)
        return out.reshape(world * rec.shape[0], rec.shape[1])
    dist.gather(rec, gather_list=None, dst=dst)
    return None


class PendingGather:
    """Handle of a gather that is still in flight (gather_records_async).  result() completes it:
    on `dst` the concatenated [world*F][rb] tensor, None elsewhere."""

    def __init__(self, work, out, rec, is_dst):
        self._work, self._out, self._rec, self._is_dst = work, out, rec, is_dst

    def result(self):
        if self._work is not None:
            self._work.wait()      # RCCL: orders the current stream after the collective, no host block
            self._work = None
        if self._out is None:
            return self._rec if self._is_dst else None
        return self._out.reshape(self._out.shape[0] * self._out.shape[1], self._out.shape[2])


def gather_records_async(rec, dst=0):
    """gather_records without waiting: the collective is queued behind the work that produced `rec`
    and runs on the process group's own stream, so the next batch's kernels can be enqueued (and run)
    while the records travel.  One step's exchange then hides behind the next step's compute; only the
    last one of a run is exposed.  Keep the returned handle and call .result()."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return PendingGather(None, None, rec, True)
    world, rank = dist.get_world_size(), dist.get_rank()
    if rank == dst:
        out = torch.empty((world,) + tuple(rec.shape), dtype=rec.dtype, device=rec.device)
        work = dist.gather(rec, gather_list=list(out.unbind(0)), dst=dst, async_op=True)
        return PendingGather(work, out, rec, True)
    work = dist.gather(rec, gather_list=None, dst=dst, async_op=True)
    return PendingGather(work, None, rec, False)


def gather_tracks_async(tensors, dst=0):
    """The exchange without a packing pass: each output array of a shard (n_out, xyz, err, corr -- contiguous
    slices along the frame axis) is gathered on `dst` by its own asynchronous collective.  Packing them into
    one record tensor first costs a read and a write of the whole payload on every rank (about 10 % of a step
    at 8 x 16); four collectives on four contiguous tensors cost four launches.
    The arrays travel as bytes (uint8 views, no copy): RCCL/NCCL has no 16-bit integer type for `corr`.
    Returns a list of PendingGather handles in the order of `tensors`; result() on `dst` is the uint8
    [world * F][bytes per frame] tensor (reinterpret with .view(dtype)), None elsewhere."""
    import torch
    out = []
    for t in tensors:
        flat = t.contiguous().reshape(t.shape[0], -1)
        out.append(gather_records_async(flat.view(torch.uint8), dst=dst))
    return out
