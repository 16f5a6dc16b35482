"""Frame sharding across the GPUs of one node.

Frames are independent given fixed cameras (every cross-frame state of the reference -- Kalman,
low-pass -- is downstream of the hot path), so the path shards with NO data-path collective:
one process per GPU, contiguous frame blocks, camera tables replicated (a few KB).  The only
exchange is the final gather of fixed-stride track records to rank 0 -- one collective call on
one packed byte tensor (RCCL over xGMI on the GPU box: 7 point-to-point links into the root;
gloo in the CPU tests).

Two payload formats:
  * fixed capacity (pack_records / gather_tracks_async): every frame ships K_max slots --
        n_out int32 [1] (+4 pad) | xyz float64 [K][3] | err float64 [K] | corr int16 [K][C] (padded to 8 bytes)
    2.3 KB per 8 x 16 frame at K_max = 48, of which 23 slots are valid on average;
  * compact (compact_tracks / gather_compact_async, what bench.py uses at N > 1): only the valid points travel,
    as records of track_record_bytes(C) = 32 + 2 C bytes (padded to 8) in frame order, plus n_out int32 per frame:
    1.1 KB per frame.  The records are produced on the device by mocap_compact_tracks_dev (prefix sum over n_out
    + scatter); shard sizes differ, so the exchange is count-first with the count riding in the first point-to-point
    message ({record count | n_out}, fixed size), then exactly the valid bytes (7 xGMI links into the root, no ring,
    no collective, no host synchronisation on the senders).
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
        elif torch.cuda.is_available():
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
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


# ----------------------------------------------------------------------------- compact exchange
def track_record_bytes(C):
    """= mocap_track_record_bytes(C): xyz 24 | err 8 | corr 2 C | pad to a multiple of 8."""
    return (32 + 2 * int(C) + 7) // 8 * 8


def compact_tracks_reference(n_out, xyz, err, corr):
    """Host restatement of mocap_compact_tracks_dev for tests and CPU-only runs (gloo): numpy arrays in,
    (records uint8 [P][stride], offsets int64 [F + 1]) out."""
    n_out = np.asarray(n_out)
    F, K = err.shape
    C = corr.shape[2]
    stride = track_record_bytes(C)
    n = np.where((n_out < 0) | (n_out > K), 0, n_out).astype(np.int64)   # > K: a frame that needs more slots than it was given (nothing was written)
    offsets = np.zeros(F + 1, dtype=np.int64)
    np.cumsum(n, out=offsets[1:])
    valid = np.arange(K)[None, :] < n[:, None]
    P = int(offsets[-1])
    rec = np.zeros((P, stride), dtype=np.uint8)
    rec[:, 0:24] = np.ascontiguousarray(xyz[valid]).view(np.uint8).reshape(P, 24)
    rec[:, 24:32] = np.ascontiguousarray(err[valid]).view(np.uint8).reshape(P, 8)
    rec[:, 32:32 + 2 * C] = np.ascontiguousarray(corr[valid]).view(np.uint8).reshape(P, 2 * C)
    return rec, offsets


def unpack_compact(n_out, records, C, K_max, fill=np.nan):
    """Inverse of the compaction on the gathering side: n_out [F], records uint8 [P][stride] -> the dense arrays of
    the fixed-capacity layout (unused slots = fill / -1)."""
    n_out = np.asarray(n_out).astype(np.int64)
    F = n_out.shape[0]
    rec = np.ascontiguousarray(np.asarray(records)).reshape(-1, track_record_bytes(C))
    n = np.where((n_out < 0) | (n_out > K_max), 0, n_out)
    valid = np.arange(K_max)[None, :] < n[:, None]
    assert int(n.sum()) == rec.shape[0], (int(n.sum()), rec.shape)
    xyz = np.full((F, K_max, 3), fill)
    err = np.full((F, K_max), fill)
    corr = np.full((F, K_max, C), -1, dtype=np.int16)
    xyz[valid] = rec[:, 0:24].copy().view(np.float64).reshape(-1, 3)
    err[valid] = rec[:, 24:32].copy().view(np.float64).reshape(-1)
    corr[valid] = rec[:, 32:32 + 2 * C].copy().view(np.int16).reshape(-1, C)
    return {"n_out": n_out.astype(np.int32), "xyz": xyz, "err": err, "corr": corr}


class TrackCompactor:
    """Device buffers + launch of mocap_compact_tracks_dev for one shard (reused from step to step).
    compact() enqueues on the core's stream and returns immediately; count() is the number of records of the LAST
    compact() and waits for it (the kernel drops the count into pinned host memory).

    Buffer reuse: an exchange that is still in flight reads records[i] / n_out[i] (the sender's isend, the root's lazy
    view of its own shard), so the handle returned by gather_compact_async is attached to its buffer (attach()) and
    compact() completes that handle before it lets the kernels overwrite the buffer.  With n_buffers >= the number of
    exchanges the caller keeps in flight + 1 that wait never blocks."""

    def __init__(self, core, F, K_max, C, device, n_buffers=3):
        import torch
        self.core, self.F, self.K, self.C = core, int(F), int(K_max), int(C)
        self.stride = track_record_bytes(C)
        self.records = [torch.empty((self.F * self.K, self.stride), dtype=torch.uint8, device=device) for _ in range(n_buffers)]
        self.offsets = [torch.empty(self.F + 1, dtype=torch.int64, device=device) for _ in range(n_buffers)]
        self.n_out = [torch.empty(self.F, dtype=torch.int32, device=device) for _ in range(n_buffers)]
        self.totals = [torch.zeros(1, dtype=torch.int64).pin_memory() for _ in range(n_buffers)]
        self.events = [torch.cuda.Event() for _ in range(n_buffers)]
        self._pending = [None] * n_buffers
        self._i = -1

    def attach(self, i, handle):
        """The exchange `handle` reads buffer i: compact() will not reuse the buffer before the handle is complete."""
        self._pending[i] = handle
        return handle

    def compact(self, n_out, xyz, err, corr, stream):
        self._i = (self._i + 1) % len(self.records)
        i = self._i
        if self._pending[i] is not None:
            self._pending[i].result()      # idempotent: the caller's own result() later returns the same tensors
            self._pending[i] = None
        self.core.compact_tracks_dev(self.F, self.K, n_out.data_ptr(), xyz.data_ptr(), err.data_ptr(), corr.data_ptr(),
                                     self.offsets[i].data_ptr(), self.records[i].data_ptr(), self.F * self.K,
                                     self.totals[i].data_ptr())
        self.n_out[i].copy_(n_out, non_blocking=True)     # the next step's kernel rewrites n_out while this one travels
        self.events[i].record(stream)
        return i

    def count(self, i):
        self.events[i].synchronize()
        return int(self.totals[i][0])


class PendingCompactGather:
    """In-flight compact exchange.  result() on `dst`: (n_out int32 [sum F_r] tensor, records uint8 [sum P_r][stride]
    tensor) in rank order; None elsewhere.  result() may be called more than once (the first call completes the
    transfers and materialises the concatenation, later calls return it)."""

    def __init__(self, works, n_parts, r_parts, is_dst, alloc_stream=None, r_whole=None):
        self._works, self._n, self._r, self._is_dst = works, n_parts, r_parts, is_dst
        self._done, self._out = False, None
        # r_whole: the root's ONE receive buffer -- r_parts are row ranges of it, in rank order, already in place when the
        # transfers are complete (no concatenation of the records: at 8 ranks that copy was 0.8 GB per 8 x 16 step)
        self._whole = r_whole
        # the stream gather_compact_async was called on: its caching-allocator pool owns the receive buffers
        self._alloc_stream = alloc_stream

    def result(self):
        import torch
        if not self._done:
            for w in self._works:
                w.wait()
            self._works = []
            if self._is_dst:
                # torch.cat copies: the result no longer aliases the compactor's buffers
                self._out = (torch.cat(self._n), self._whole if self._whole is not None else torch.cat(self._r))
                # The parts were allocated on the stream the exchange was posted on (bench.py: `comm`), the cat above
                # runs on the CALLER's current stream (bench.py: the compute stream, queued behind a step's kernels).
                # Dropping the parts returns their blocks to the allocation stream's pool at once, where the next
                # exchange's torch.empty + irecv could overwrite them while the cat is still queued: tell the allocator
                # which stream still reads them.
                if self._out[1].is_cuda:
                    cur = torch.cuda.current_stream(self._out[1].device)
                    if self._alloc_stream is None or cur != self._alloc_stream:
                        for part in list(self._n) + list(self._r) + ([self._whole] if self._whole is not None else []):
                            if part.is_cuda and part.numel():
                                part.record_stream(cur)
            self._n = self._r = None
            self._done = True
        return self._out


COMPACT_HEADER_BYTES = 8   # int64 record count in front of the shard's n_out array


def gather_compact_async(n_out, records, n_records, frames_per_rank, dst=0):
    """The one exchange of the path in its compact form.  n_out: this shard's int32 [F_r]; records: uint8
    [>= n_records][stride] (only the first n_records rows travel); frames_per_rank: F_r of every rank (static,
    from shard_bounds).

    Count-first WITHOUT a collective: every sender's first message has a fixed size -- {int64 record count | n_out
    int32 [F_r]} -- and its second message carries exactly the valid records.  Senders post both and return at once
    (no host synchronisation at all).  The root posts the header receives, waits for those small messages only (on
    the stream it was called on: the compute stream keeps running the kernels that were queued before), reads the
    counts and posts the record receives of the right sizes.  This replaces an all_gather + .item() per step, which
    synchronised every rank's host with every other rank's GPU inside the pipeline."""
    import torch
    import torch.distributed as dist
    rec = records[:n_records]
    alloc_stream = torch.cuda.current_stream(n_out.device) if n_out.is_cuda else None
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return PendingCompactGather([], [n_out], [rec], True, alloc_stream)
    world, rank = dist.get_world_size(), dist.get_rank()
    stride = records.shape[1]
    dev = n_out.device
    if rank != dst:
        head = torch.empty(COMPACT_HEADER_BYTES + 4 * n_out.numel(), dtype=torch.uint8, device=dev)
        head[:COMPACT_HEADER_BYTES] = torch.tensor([int(n_records)], dtype=torch.int64).view(torch.uint8).to(dev, non_blocking=True)
        head[COMPACT_HEADER_BYTES:] = n_out.contiguous().view(torch.uint8)
        # two groups, mirroring the root's two receive groups (header first, then the records)
        works = list(dist.batch_isend_irecv([dist.P2POp(dist.isend, head, dst)]))
        if n_records:
            works += list(dist.batch_isend_irecv([dist.P2POp(dist.isend, rec.contiguous(), dst)]))
        return PendingCompactGather(works, [head], [rec], False, alloc_stream)   # keeps the staged header alive until the sends are done
    heads, ops = {}, []
    for r in range(world):
        if r == dst:
            continue
        heads[r] = torch.empty(COMPACT_HEADER_BYTES + 4 * int(frames_per_rank[r]), dtype=torch.uint8, device=dev)
        ops.append(dist.P2POp(dist.irecv, heads[r], r))
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    if heads:
        # one small device -> host copy of the counts (synchronises the CALLING stream only)
        cnt = torch.stack([heads[r][:COMPACT_HEADER_BYTES].view(torch.int64)[0] for r in sorted(heads)]).cpu().tolist()
        counts = dict(zip(sorted(heads), (int(c) for c in cnt)))
    else:
        counts = {}
    # one receive buffer for all ranks' records, in rank order: every sender's rows land where the result wants them, and
    # the root's own shard is copied in on the stream this was called on (behind its compaction)
    counts[dst] = int(n_records)
    whole = torch.empty((sum(counts[r] for r in range(world)), stride), dtype=torch.uint8, device=dev)
    ops, n_parts, r_parts, row = [], [], [], 0
    for r in range(world):
        rb = whole[row:row + counts[r]]
        row += counts[r]
        r_parts.append(rb)
        if r == dst:
            n_parts.append(n_out)
            if counts[r]:
                rb.copy_(rec, non_blocking=True)
            continue
        n_parts.append(heads[r][COMPACT_HEADER_BYTES:].view(torch.int32))
        if counts[r]:
            ops.append(dist.P2POp(dist.irecv, rb, r))
    works = list(dist.batch_isend_irecv(ops)) if ops else []
    return PendingCompactGather(works, n_parts, r_parts, True, alloc_stream, r_whole=whole)
