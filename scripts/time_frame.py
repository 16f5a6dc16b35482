"""Timing probe (GPU box): ms per pass of the frame path over F resident frames of the 8 x 16 bench stream.
   python scripts/time_frame.py [frames] [reps] [K_max]   (honours the MOCAP_* environment knobs)"""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "low-cost-mocap_amd"))
from mocap_core import capi, synth
F = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
C, M, K = 8, 16, (int(sys.argv[3]) if len(sys.argv) > 3 else 48)
rig = synth.ring_rig(C)
cache = f"/tmp/stream_{F}.npz"
if os.path.exists(cache):
    z = np.load(cache); blobs, counts = z["b"], z["c"]
else:
    blobs, counts, _ = synth.make_blob_stream(rig, F, M, seed=1)
    np.savez(cache, b=blobs, c=counts)
dev = torch.device("cuda:0")
core = capi.MocapCore(0)
core.set_cameras(rig["K"], rig["R"], rig["t"])
core.set_stream(torch.cuda.current_stream(dev).cuda_stream)
d_b = torch.from_numpy(blobs).to(dev); d_c = torch.from_numpy(counts).to(dev)
d_xyz = torch.empty((F, K, 3), dtype=torch.float64, device=dev); d_err = torch.empty((F, K), dtype=torch.float64, device=dev)
d_corr = torch.empty((F, K, C), dtype=torch.int16, device=dev)
d_n = torch.zeros(F, dtype=torch.int32, device=dev); d_s = torch.zeros(F, dtype=torch.int32, device=dev); d_g = torch.zeros(F, dtype=torch.int32, device=dev)
AUTO = os.environ.get("TF_AUTO") == "1"   # the product's device-buffer path: + device-side re-submit (three more enqueues)
def run():
    (core.match_triangulate_dev_auto if AUTO else core.match_triangulate_dev)(F, M, d_b.data_ptr(), d_c.data_ptr(), 5.0, K, 1 << 22, d_xyz.data_ptr(), d_err.data_ptr(),
                               d_corr.data_ptr(), d_n.data_ptr(), d_s.data_ptr(), d_g.data_ptr())
for _ in range(2):
    run(); torch.cuda.synchronize()
ts = []
for _ in range(reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
valid = (torch.arange(K, device=dev)[None, :] < d_n[:, None])
print("auto" if AUTO else "plain", "frames", F, "K_max", K, core.last_frame_kernel(), "ms", round(sorted(ts)[len(ts) // 2], 4), "per100k", round(sorted(ts)[len(ts) // 2] * 1e5 / F, 3),
      "cands/frame", float(d_g.double().mean()), "errsum", float(d_err[valid].nan_to_num(posinf=0).sum()), {k: v for k, v in os.environ.items() if k.startswith("MOCAP_")})
