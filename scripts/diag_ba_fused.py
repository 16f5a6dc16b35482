"""Diagnostic (GPU box): the one-launch linearisation against the five-launch chain it replaces (MOCAP_BA_UNFUSED=1,
separate process) -- bitwise comparison of G, cost, J; iterations/s of both."""
import json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "low-cost-mocap_amd"))
from mocap_core import capi, helpers, synth

def run(tag):
    core = capi.MocapCore(0)
    out = {}
    for C, N in ((8, 1000), (4, 200), (8, 16000), (3, 70)):
        rig = synth.ring_rig(C)
        rng = np.random.default_rng(7)
        obs, _ = synth.make_ba_observations(rig, N, seed=7, dropout=0.1)
        init = synth.perturb_rig(rig, rng)
        core.set_cameras(rig["K"], init["R"], init["t"])
        helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in rig["K"]])
        x0 = helpers._ba_x0([{"R": init["R"][i], "t": init["t"][i]} for i in range(C)])
        ne = core.ba_normal_eq(x0, obs, f32_residuals=True, use_cauchy=True, want_J=True)
        ne2 = core.ba_normal_eq(x0, obs, f32_residuals=True, use_cauchy=True, want_J=False)
        assert np.array_equal(ne["JtJ"], ne2["JtJ"]) and ne["cost"] == ne2["cost"]
        core.ba_solve(x0, obs, ftol=0.0, xtol=0.0, gtol=0.0, max_iter=40)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            x, info = core.ba_solve(x0, obs, ftol=0.0, xtol=0.0, gtol=0.0, max_iter=200)
            ts.append(time.perf_counter() - t0)
        dt = sorted(ts)[2]
        xr, inf2 = core.ba_solve(x0, obs, ftol=1e-2)
        np.savez(os.path.join(ROOT, "gpurun_out", f"fused_{tag}_{C}_{N}.npz"), JtJ=ne["JtJ"], Jtr=ne["Jtr"], J=ne["J"], cost=ne["cost"], x=x, xr=xr)
        out[f"{C}x{N}"] = {"us_per_iter": 1e6 * dt / info["iterations"], "iters": info["iterations"], "cost": info["cost"],
                           "ref_rule": [inf2["nfev"], inf2["njev"], inf2["status"], inf2["cost"]]}
    print(tag, json.dumps(out))

if len(sys.argv) > 1:
    run(sys.argv[1])
else:
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for tag, env in (("fused", {}), ("chain", {"MOCAP_BA_UNFUSED": "1"})):
        subprocess.check_call([sys.executable, __file__, tag], env={**os.environ, **env})
    for C, N in ((8, 1000), (4, 200), (8, 16000), (3, 70)):
        a = np.load(os.path.join(ROOT, "gpurun_out", f"fused_fused_{C}_{N}.npz")); b = np.load(os.path.join(ROOT, "gpurun_out", f"fused_chain_{C}_{N}.npz"))
        print(C, N, "J bitwise", np.array_equal(a["J"], b["J"]), "JtJ rel", float(np.abs(a["JtJ"] - b["JtJ"]).max() / np.abs(b["JtJ"]).max()),
              "Jtr rel", float(np.abs(a["Jtr"] - b["Jtr"]).max() / np.abs(b["Jtr"]).max()), "cost", float(a["cost"]), float(b["cost"]),
              "x rel", float(np.abs(a["x"] - b["x"]).max()), "xr", float(np.abs(a["xr"] - b["xr"]).max()))
