"""Timing probe (GPU box): helpers.bundle_adjustment in its default mode (scipy drives, residuals + batched Jacobian in the core) on
bench.py's 8 cams x 1 000 points; prints wall, time inside the core's calls, and the per-call cost of mocap_ba_residuals."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "low-cost-mocap_amd"))
import torch  # noqa: F401
from mocap_core import capi, helpers, synth
core = capi.MocapCore(0)
rig = synth.ring_rig(8)
rng = np.random.default_rng(7)
obs, _ = synth.make_ba_observations(rig, 1000, seed=7)
init = synth.perturb_rig(rig, rng)
core.set_cameras(rig["K"], init["R"], init["t"])
helpers.set_core(core)
helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in rig["K"]])
x0 = helpers._ba_x0([{"R": init["R"][i], "t": init["t"][i]} for i in range(8)])
for P in (1, 50):
    X = np.repeat(x0[None], P, axis=0)
    core.ba_residuals(X, obs)
    t0 = time.perf_counter()
    for _ in range(50):
        core.ba_residuals(X, obs)
    print("ba_residuals P =", P, "ms per call", 1e3 * (time.perf_counter() - t0) / 50)
ref_obs = synth.obs_to_reference_array(obs)
poses0 = [{"R": init["R"][i].copy(), "t": init["t"][i].copy()} for i in range(8)]
with helpers.bundle_adjustment_mode("scipy"):
    helpers.bundle_adjustment(ref_obs, [dict(p) for p in poses0], None, return_info=True)
    for _ in range(3):
        t0 = time.perf_counter()
        _, info = helpers.bundle_adjustment(ref_obs, [dict(p) for p in poses0], None, return_info=True)
        print("solve wall", round(time.perf_counter() - t0, 4), "core_s", round(info.get("core_s", 0), 4), "njev", info["njev"], "nfev", info["nfev"])
try:
    from threadpoolctl import threadpool_limits, threadpool_info
    print("blas:", [(d.get("internal_api"), d.get("num_threads")) for d in threadpool_info()])
    with threadpool_limits(limits=1), helpers.bundle_adjustment_mode("scipy"):
        for _ in range(3):
            t0 = time.perf_counter()
            _, info = helpers.bundle_adjustment(ref_obs, [dict(p) for p in poses0], None, return_info=True)
            print("1 BLAS thread: solve wall", round(time.perf_counter() - t0, 4), "core_s", round(info.get("core_s", 0), 4), "njev", info["njev"], "nfev", info["nfev"], "cost", info["cost"])
    with helpers.bundle_adjustment_mode("scipy"):
        _, info = helpers.bundle_adjustment(ref_obs, [dict(p) for p in poses0], None, return_info=True)
        print("default threads: cost", info["cost"])
except Exception as e:
    print("threadpoolctl leg failed:", repr(e))
