"""Profile target (GPU box): the resident LM loop at BASELINE.json's BA config (8 cams x 1 000 points) and at
configs[3] (16 000 points); run under rocprofv3 --kernel-trace --stats.  argv[1] = points (default 1000)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "low-cost-mocap_amd"))
from mocap_core import capi, helpers, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
core = capi.MocapCore(0)
rig = synth.ring_rig(8)
rng = np.random.default_rng(7)
obs, _ = synth.make_ba_observations(rig, N, seed=7)
init = synth.perturb_rig(rig, rng)
core.set_cameras(rig["K"], init["R"], init["t"])
helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in rig["K"]])
x0 = helpers._ba_x0([{"R": init["R"][i], "t": init["t"][i]} for i in range(8)])
core.ba_solve(x0, obs, ftol=0.0, xtol=0.0, gtol=0.0, max_iter=40)
t0 = time.perf_counter()
x, info = core.ba_solve(x0, obs, ftol=0.0, xtol=0.0, gtol=0.0, max_iter=400)
dt = time.perf_counter() - t0
print(f"points {N}: {info['iterations']:.0f} iterations, {1e6 * dt / info['iterations']:.1f} us per iteration")
