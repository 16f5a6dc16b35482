import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "low-cost-mocap_amd"))
import numpy as np
from scipy import optimize
from mocap_core import capi, helpers, synth
from oracle import c_oracle
core = capi.MocapCore(0)
C, N = 8, 1000
rig = synth.ring_rig(C)
rng = np.random.default_rng(65 + C)
obs, _ = synth.make_ba_observations(rig, N, seed=65 + C, noise_px=0.0)
init = synth.perturb_rig(rig, rng, rot_sigma=0.01, trans_sigma=0.02)
core.set_cameras(rig["K"], init["R"], init["t"])
helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in rig["K"]])
x0 = helpers._ba_x0([{"R": init["R"][i], "t": init["t"][i]} for i in range(C)])
core.set_options(f32_rounding=False)
for mi in (5, 20, 100, 400):
    x_gpu, info = core.ba_solve(x0, obs, ftol=1e-12, xtol=1e-12, gtol=1e-12, max_iter=mi, f32_residuals=False)
    print(mi, info)
co = c_oracle.COracle(rig["K"], init["R"], init["t"], f32_rounding=False)
def fun(x):
    r = co.ba_residuals(x, obs)[0]
    return r[~np.isnan(r)]
ref = optimize.least_squares(fun, x0, loss="cauchy", ftol=1e-12, xtol=1e-12, gtol=1e-12, max_nfev=400, verbose=1)
print("scipy", ref.cost, ref.nfev, ref.njev, ref.status, ref.optimality)
truth = helpers._ba_x0([{"R": rig["R"][i], "t": rig["t"][i]} for i in range(C)])
print("cost at truth", 0.5*np.log1p(fun(truth)**2).sum())
