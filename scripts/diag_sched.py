"""Diagnostic (GPU box): repeat the scheduling-knob comparison and describe any difference."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "low-cost-mocap_amd"))
from mocap_core import capi, synth
core = capi.MocapCore(0)
rig = synth.ring_rig(8)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for seed in (81, 5):
    blobs, counts, _ = synth.make_blob_stream(rig, 3000, 16, seed=seed)
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    core.set_tuning(256, 0, 0)
    base = core.match_triangulate(blobs, counts, K_max=48)
    valid = np.arange(48)[None, :] < base["n_out"][:, None]
    bad = 0
    for rep in range(reps):
        for threads, thr, sl in [(256, 0, 0), (256, 2048, 512), (128, 4096, 1024), (64, 1024, 256), (256, 300, 300), (256, -1, 0), (0, -1, 0)]:
            core.set_tuning(threads, thr, sl)
            res = core.match_triangulate(blobs, counts, K_max=48)
            for key in ("n_out", "status", "n_cand", "xyz", "err", "corr"):
                a, b = (res[key], base[key]) if key in ("n_out", "status", "n_cand") else (res[key][valid], base[key][valid])
                if not np.array_equal(a, b):
                    bad += 1
                    d = np.argwhere(np.asarray(a != b).reshape(a.shape[0], -1).any(1)).ravel()
                    print("DIFF", seed, rep, (threads, thr, sl), key, "rows", d[:5], "count", d.size)
                    if key == "err":
                        fr, slot = np.nonzero(valid)
                        for i in d[:3]:
                            print("   frame", fr[i], "slot", slot[i], "err", a[i], b[i], "n_cand", base["n_cand"][fr[i]],
                                  "corr", res["corr"][fr[i], slot[i]], base["corr"][fr[i], slot[i]])
    print("seed", seed, "mismatching comparisons", bad)
