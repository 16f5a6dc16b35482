"""GPU box: is the one-off 60-90 ms inside an early mocap_ba_solve a property of the GPU waking up (idle -> active
power / clock transition some tens of ms after work resumes), i.e. does it come back after every idle gap and stay away
while the GPU is kept busy?  argv[1] = points (default 1000)."""
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "low-cost-mocap_amd"))
from mocap_core import capi, helpers, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
core = capi.MocapCore(0)
rig = synth.ring_rig(8)
rng = np.random.default_rng(9)
obs, _ = synth.make_ba_observations(rig, N, seed=9)
init = synth.perturb_rig(rig, rng)
core.set_cameras(rig["K"], init["R"], init["t"])
helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in rig["K"]])
x0 = helpers._ba_x0([{"R": init["R"][i], "t": init["t"][i]} for i in range(8)])


def clocks():
    out = []
    for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))[:1]:
        try:
            out.append(" ".join(ln.strip() for ln in open(f) if "*" in ln))
        except OSError:
            pass
    return ";".join(out) or "n/a"


def solves(tag, n):
    ms, at = [], []
    t_first = time.perf_counter()
    for i in range(n):
        t0 = time.perf_counter()
        core.ba_solve(x0, obs, ftol=0.0, xtol=0.0, gtol=0.0, max_iter=60)
        ms.append(1e3 * (time.perf_counter() - t0))
        at.append(1e3 * (t0 - t_first))
    med = sorted(ms)[len(ms) // 2]
    print(f"{tag}: ms per solve", " ".join(f"{m:.1f}" for m in ms), "| slow at ms since resume:",
          " ".join(f"{a:.0f}(+{m:.0f})" for a, m in zip(at, ms) if m > 3 * med), "| sclk", clocks(), flush=True)


print("sclk at start", clocks())
solves("1 cold process", 20)
for gap in (3.0, 1.0, 0.3, 0.1, 0.03):
    time.sleep(gap)
    solves(f"after {gap} s idle", 20)
time.sleep(3.0)
t0 = time.perf_counter()
k = 0
while time.perf_counter() - t0 < 0.4:
    core.ba_solve(x0, obs, ftol=0.0, xtol=0.0, gtol=0.0, max_iter=60)
    k += 1
print(f"after 3 s idle: {k} wake-up solves in 0.4 s, then")
solves("timed after the wake-up", 20)
