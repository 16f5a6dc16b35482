"""Diagnostic (GPU box): one small batch through the frame path; prints a checksum."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "low-cost-mocap_amd"))
from mocap_core import capi, synth
core = capi.MocapCore(0)
rig = synth.ring_rig(8)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
blobs, counts, _ = synth.make_blob_stream(rig, n, 16, seed=81)
core.set_cameras(rig["K"], rig["R"], rig["t"])
res = core.match_triangulate(blobs, counts, K_max=48)
print("ok n_out", res["n_out"].sum(), "err sum", np.nansum(res["err"][np.arange(48)[None, :] < res["n_out"][:, None]]))
