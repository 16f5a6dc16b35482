"""GPU box: does the one-off ~80 ms hole in kernel dispatch follow device-memory (re)allocation (hipMalloc / hipFree
inside DevBuf::reserve) by some tens of ms, whatever the process's age?"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "low-cost-mocap_amd"))
from mocap_core import capi, helpers, synth  # noqa: E402

core = capi.MocapCore(0)
rig = synth.ring_rig(8)
rng = np.random.default_rng(9)
init = synth.perturb_rig(rig, rng)
core.set_cameras(rig["K"], init["R"], init["t"])
helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in rig["K"]])
x0 = helpers._ba_x0([{"R": init["R"][i], "t": init["t"][i]} for i in range(8)])
OBS = {n: synth.make_ba_observations(rig, n, seed=9)[0] for n in (1000, 4000, 16000)}


def solves(tag, n_pts, n=14):
    ms, at = [], []
    t_first = time.perf_counter()
    for i in range(n):
        t0 = time.perf_counter()
        core.ba_solve(x0, OBS[n_pts], ftol=0.0, xtol=0.0, gtol=0.0, max_iter=60)
        ms.append(1e3 * (time.perf_counter() - t0))
        at.append(1e3 * (t0 - t_first))
    med = sorted(ms)[len(ms) // 2]
    print(f"{tag}: ms per solve", " ".join(f"{m:.1f}" for m in ms), "| slow at ms:",
          " ".join(f"{a:.0f}(+{m:.0f})" for a, m in zip(at, ms) if m > 3 * med), flush=True)


solves("A  cold process, 1k points (allocates)", 1000)
solves("B  1k again (no allocation)", 1000)
solves("C  4k points (workspace regrows: hipFree + hipMalloc)", 4000)
solves("D  4k again (no allocation)", 4000)
solves("E  16k points (regrows)", 16000)
solves("F  16k again", 16000)
solves("G  1k again (no allocation: grow-only buffers)", 1000)
obs_t, _ = synth.make_ba_observations(rig, 200000, seed=3)
core.triangulate(obs_t)            # grows scratch[0] (hipMalloc ~ 35 MB) and copies through pageable memory
solves("H  1k right after an unrelated 35 MB hipMalloc + pageable copies", 1000)
core.triangulate(obs_t)            # same size: no allocation
solves("I  1k right after the same call again (no allocation)", 1000)
core2 = capi.MocapCore(0)          # second context: new stream, fresh buffers
core2.set_cameras(rig["K"], init["R"], init["t"])
t0 = time.perf_counter()
ms = []
for i in range(14):
    t1 = time.perf_counter()
    core2.ba_solve(x0, OBS[1000], ftol=0.0, xtol=0.0, gtol=0.0, max_iter=60)
    ms.append(1e3 * (time.perf_counter() - t1))
print("J  second context, 1k (allocates its own buffers): ms per solve", " ".join(f"{m:.1f}" for m in ms))
