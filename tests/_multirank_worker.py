"""Worker of tests/test_gpu_multirank.py: one rank of a frame-sharded run whose ranks share ONE GPU (gloo; RCCL refuses
two ranks on a device).  Every rank runs the frame kernel on its contiguous shard, compacts its tracks on the device
and takes part in the count-first point-to-point exchange; rank 0 also runs the whole batch alone and compares the
gathered payload with it bit for bit."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "low-cost-mocap_amd")]
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from mocap_core import capi, dist as mdist, synth  # noqa: E402


def run_shard(core, dev, blobs, counts, K, gate=5.0):
    F, C, M, _ = blobs.shape
    d_b, d_c = torch.from_numpy(blobs).to(dev), torch.from_numpy(counts).to(dev)
    out = dict(xyz=torch.empty((F, K, 3), dtype=torch.float64, device=dev), err=torch.empty((F, K), dtype=torch.float64, device=dev),
               corr=torch.empty((F, K, C), dtype=torch.int16, device=dev), n_out=torch.zeros(F, dtype=torch.int32, device=dev),
               status=torch.zeros(F, dtype=torch.int32, device=dev))
    core.match_triangulate_dev_auto(F, M, d_b.data_ptr(), d_c.data_ptr(), gate, K, 1 << 20, out["xyz"].data_ptr(), out["err"].data_ptr(),
                               out["corr"].data_ptr(), out["n_out"].data_ptr(), out["status"].data_ptr())
    return out


def main():
    rank, _, world = mdist.init_process_group(backend="gloo")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if os.environ.get("MULTIRANK_WORKLOAD") == "64x256":  # BASELINE configs[4]'s shape: the wide variant, 160-byte records
        C, M, K, F = 64, 256, 384, 37
        rig = synth.stress_rig(C)
        blobs, counts, _ = synth.make_stress_stream(rig, F, M, seed=78)
        gate = synth.STRESS_GATE_PX
    else:
        C, M, K, F = 8, 16, 48, 1003                      # 1003 frames over the ranks: uneven shards
        rig = synth.ring_rig(C)
        blobs, counts, _ = synth.make_blob_stream(rig, F, M, seed=77)
        gate = 5.0
    core = capi.MocapCore(0)
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    stream = torch.cuda.current_stream(dev)
    core.set_stream(stream.cuda_stream)
    lo, hi = mdist.shard_bounds(F, rank, world)
    frames_per_rank = [b - a for a, b in (mdist.shard_bounds(F, r, world) for r in range(world))]
    comp = mdist.TrackCompactor(core, hi - lo, K, C, dev)
    handles = []
    # MULTIRANK_COMM_STREAM=1: bench.py's arrangement -- the exchange is posted under a side stream (whose allocator pool
    # then owns the root's receive buffers) and completed on the compute stream while that stream is busy, with further
    # exchanges posted in between (PendingCompactGather.result() has to record_stream the parts before releasing them)
    comm = torch.cuda.Stream(dev) if os.environ.get("MULTIRANK_COMM_STREAM") == "1" else None
    results = []
    for step in range(5 if comm is not None else 3):       # several exchanges, two of them in flight: buffers are reused
        mine = run_shard(core, dev, blobs[lo:hi], counts[lo:hi], K, gate)
        i = comp.compact(mine["n_out"], mine["xyz"], mine["err"], mine["corr"], stream)
        n = comp.count(i)
        if comm is None:
            handles.append(comp.attach(i, mdist.gather_compact_async(comp.n_out[i], comp.records[i], n, frames_per_rank, dst=0)))
            continue
        with torch.cuda.stream(comm):
            comm.wait_event(comp.events[i])
            handles.append(comp.attach(i, mdist.gather_compact_async(comp.n_out[i], comp.records[i], n, frames_per_rank, dst=0)))
        if len(handles) > 2:
            run_shard(core, dev, blobs[lo:hi], counts[lo:hi], K, gate)       # keep the compute stream busy in front of the cat
            results.append(handles.pop(0).result())
    results += [h.result() for h in handles]
    if rank == 0:
        whole = run_shard(core, dev, blobs, counts, K, gate)
        torch.cuda.synchronize(dev)
        assert core.last_frame_kernel().startswith("frame_bb_kernel" if C == 8 else "frame_kernel<")
        base = {k: v.cpu().numpy() for k, v in whole.items()}
        assert C == 64 or not base["status"].any()
        base["n_out"] = np.where(base["status"] != 0, 0, base["n_out"])
        valid = np.arange(K)[None, :] < base["n_out"][:, None]
        for n_all, r_all in results:
            got = mdist.unpack_compact(n_all.cpu().numpy(), r_all.cpu().numpy(), C, K)
            assert np.array_equal(got["n_out"], base["n_out"])
            for key in ("xyz", "err", "corr"):
                assert np.array_equal(got[key][valid], base[key][valid]), key
        print(f"MULTIRANK OK world {world} frames {F} shards {frames_per_rank} records {int(base['n_out'].sum())}", flush=True)
    else:
        assert all(r is None for r in results)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
