"""Timing probe (GPU box): does the ORDER of the frames matter to the headline kernel?  The same 100 k frames of the 8 x 16 bench
stream in stream order, sorted by candidate count descending (longest first: the best case for a dynamic queue) and ascending
(the worst: the heaviest frames start last) -- the spread is what the end-of-launch tail costs."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "low-cost-mocap_amd"))
from mocap_core import capi, synth
F = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
C, M, K = 8, 16, 48
rig = synth.ring_rig(C)
blobs, counts, _ = synth.make_blob_stream(rig, F, M, seed=1)
dev = torch.device("cuda:0")
core = capi.MocapCore(0)
core.set_cameras(rig["K"], rig["R"], rig["t"])
core.set_stream(torch.cuda.current_stream(dev).cuda_stream)
def bufs():
    return (torch.empty((F, K, 3), dtype=torch.float64, device=dev), torch.empty((F, K), dtype=torch.float64, device=dev),
            torch.empty((F, K, C), dtype=torch.int16, device=dev), torch.zeros(F, dtype=torch.int32, device=dev),
            torch.zeros(F, dtype=torch.int32, device=dev), torch.zeros(F, dtype=torch.int32, device=dev))
def timed(b, c, reps=7):
    d_b = torch.from_numpy(np.ascontiguousarray(b)).to(dev); d_c = torch.from_numpy(np.ascontiguousarray(c)).to(dev)
    x, e, r, n, s, g = bufs()
    run = lambda: core.match_triangulate_dev(F, M, d_b.data_ptr(), d_c.data_ptr(), 5.0, K, 1 << 22, x.data_ptr(), e.data_ptr(), r.data_ptr(), n.data_ptr(), s.data_ptr(), g.data_ptr())
    for _ in range(2):
        run(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, bb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); bb.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(bb))
    return sorted(ts)[len(ts) // 2], g.cpu().numpy()
t0, g = timed(blobs, counts)
order = np.argsort(-g.astype(np.int64), kind="stable")
t1, _ = timed(blobs[order], counts[order])
t2, _ = timed(blobs[order[::-1]], counts[order[::-1]])
rng = np.random.default_rng(5); perm = rng.permutation(F)
t3, _ = timed(blobs[perm], counts[perm])
print("frames", F, core.last_frame_kernel(), "stream order", round(t0, 4), "ms; heaviest first", round(t1, 4), "; heaviest last", round(t2, 4), "; shuffled", round(t3, 4),
      "; candidates max", int(g.max()), "p99.9", int(np.percentile(g, 99.9)))
