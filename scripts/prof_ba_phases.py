import os, sys
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
for _ in range(50):
    core.ba_normal_eq(x0, obs, f32_residuals=True, use_cauchy=True)
