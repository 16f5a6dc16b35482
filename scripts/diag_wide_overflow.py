import os, sys, numpy as np, torch
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "low-cost-mocap_amd"))
from mocap_core import capi, synth
F, C, M, K = 4096, 64, 256, 384
rig = synth.stress_rig(C)
blobs, counts, _ = synth.make_stress_stream(rig, F, M, seed=1)
core = capi.MocapCore(0); core.set_cameras(rig["K"], rig["R"], rig["t"])
res = core.match_triangulate(blobs, counts, gate_px=synth.STRESS_GATE_PX, K_max=K, G_cap=1 << 20)
st = res["status"]; print("frames", F, "status counts", {int(b): int((st & b != 0).sum()) for b in (1, 2, 4)}, "any", int((st != 0).sum()))
bad = np.nonzero(st)[0][:5]
res2 = core.match_triangulate(blobs[bad], counts[bad], gate_px=synth.STRESS_GATE_PX, K_max=K, G_cap=1 << 24)
print("with G_cap 2^24:", res2["status"].tolist(), res2["n_cand"].tolist())
