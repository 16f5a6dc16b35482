"""Timing probe (GPU box): ms per pass of the wide frame variant over F resident 64 x 256 stress frames."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "low-cost-mocap_amd"))
from mocap_core import capi, synth
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
C, M, K = 64, 256, int(os.environ.get("TW_K", "384"))
rig = synth.stress_rig(C)
SEED = int(os.environ.get("TW_SEED", "1"))
cache = f"/tmp/stress_{F}_{SEED}.npz"
if os.path.exists(cache):
    z = np.load(cache); blobs, counts = z["b"], z["c"]
else:
    blobs, counts, _ = synth.make_stress_stream_chunked(rig, F, M, seed=SEED)   # (bench.py's stream since round 6)
    np.savez(cache, b=blobs, c=counts)
dev = torch.device("cuda:0")
core = capi.MocapCore(0)
core.set_cameras(rig["K"], rig["R"], rig["t"])
if os.environ.get("TW_HIT_CAP"):
    core.set_frame_limits(hit_cap=int(os.environ["TW_HIT_CAP"]))
core.set_stream(torch.cuda.current_stream(dev).cuda_stream)
d_b = torch.from_numpy(blobs).to(dev); d_c = torch.from_numpy(counts).to(dev)
d_xyz = torch.empty((F, K, 3), dtype=torch.float64, device=dev); d_err = torch.empty((F, K), dtype=torch.float64, device=dev)
d_corr = torch.empty((F, K, C), dtype=torch.int16, device=dev)
d_n = torch.zeros(F, dtype=torch.int32, device=dev); d_s = torch.zeros(F, dtype=torch.int32, device=dev); d_g = torch.zeros(F, dtype=torch.int32, device=dev)
def run():
    core.match_triangulate_dev(F, M, d_b.data_ptr(), d_c.data_ptr(), synth.STRESS_GATE_PX, K, 1 << 20, d_xyz.data_ptr(), d_err.data_ptr(),
                               d_corr.data_ptr(), d_n.data_ptr(), d_s.data_ptr(), d_g.data_ptr())
run(); torch.cuda.synchronize()
ts = []
for _ in range(reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
print("frames", F, "ms", round(sorted(ts)[len(ts) // 2], 4), "frames/s", round(F / sorted(ts)[len(ts) // 2] * 1e3), "roots/frame", float(d_n.double().mean()),
      "cands/frame", float(d_g.double().mean()), "overflow", int((d_s != 0).sum()), core.last_frame_kernel(), {k: v for k, v in os.environ.items() if k.startswith(("MOCAP_", "TW_"))})
