"""Diagnostic (GPU box): reference-mode bundle adjustment against the solver goldens.
For every tests/golden/ba_*solved / ba_c3_n24 set: (1) GPU residuals at every parameter vector the reference's
optimizer evaluated (ba_eval_xs) vs the C oracle's, float64 and after the float32 cast; (2) scipy mode and
resident mode end to end vs the reference's x_ba / stats.  Prints one JSON line per set."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "low-cost-mocap_amd"))
from mocap_core import capi, helpers  # noqa: E402
from oracle import c_oracle  # noqa: E402  (checker only)

core = capi.MocapCore(0)
helpers.set_core(core)
for name in ("ba_c3_n24", "ba_c4_n60_solved", "ba_c8_n100_solved"):
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    C = g["K"].shape[0]
    obs = g["obs"]
    core.set_cameras(g["K"], g["R_init"], g["t_init"])
    xs = g["ba_eval_xs"]
    r = core.ba_residuals(xs, obs)
    ref = c_oracle.COracle(g["K"], g["R_init"], g["t_init"]).ba_residuals(xs, obs)
    ok = ~np.isnan(ref)
    rel = np.abs(r[ok] - ref[ok]) / np.abs(ref[ok])
    same32 = (r[ok].astype(np.float32) == ref[ok].astype(np.float32))
    out = {"set": name, "evals": int(xs.shape[0]), "res_rel_max": float(rel.max()), "res_rel_median": float(np.median(rel)),
           "f32_equal_frac": float(same32.mean()), "f32_unequal": int((~same32).sum())}
    helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in g["K"]])
    poses0 = [{"R": g["R_init"][i], "t": g["t_init"][i]} for i in range(C)]
    live = np.ones(xs.shape[1], bool)
    live[[0] + [1 + 7 * i for i in range(C - 1)]] = False
    from mocap_core import synth
    for mode in ("scipy", "resident"):
        helpers.set_bundle_adjustment_mode(mode)
        poses, info = helpers.bundle_adjustment(synth.obs_to_reference_array(obs), poses0, None, return_info=True)
        R = np.array([p["R"] for p in poses])
        t = np.array([np.asarray(p["t"], dtype=np.float64).reshape(3) for p in poses])
        out[mode] = {"nfev": int(info["nfev"]), "njev": int(info["njev"]), "status": int(info["status"]),
                     "cost": float(info["cost"]),
                     "dR_max": float(np.abs(R - g["R_ba"]).max()),
                     "dt_rel": float(np.abs(t - g["t_ba"]).max() / np.abs(g["t_ba"]).max())}
    out["reference"] = {"nfev": int(g["ba_stats"][0]), "njev": int(g["ba_stats"][1]), "status": int(g["ba_stats"][2]),
                        "cost": float(g["ba_cost"][0])}
    print(json.dumps(out))
