"""GPU box: hunt the 60-90 ms one-off inside mocap_ba_solve.

20 solves of BASELINE configs[3] (8 cams x 16 000 points, 60 iterations) in one process with MOCAP_BA_PROFILE=1
(per-solve breakdown on stderr: setup, first linearisation, longest launch call, longest wait for a completion stamp,
longest gap between two clock reads of the spinning host thread), then the same after a burst of all-core CPU work
(what bench.py's CPU baseline does right before the BA figures) -- a spin-waiting thread inside a CPU-quota'd
container is the prime suspect.  argv[1] = points (default 16000), argv[2] = solves (default 20)."""
import os
import sys
import time

os.environ.setdefault("MOCAP_BA_PROFILE", "1")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "low-cost-mocap_amd"))
from mocap_core import capi, helpers, synth  # noqa: E402


def cg(name):
    for base in ("/sys/fs/cgroup", "/sys/fs/cgroup/cpu"):
        p = os.path.join(base, name)
        if os.path.exists(p):
            return open(p).read().strip().replace("\n", " | ")
    return "n/a"


N = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 20
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "cpu.max", cg("cpu.max"), "cfs_quota", cg("cpu.cfs_quota_us"))
print("cpu.stat before:", cg("cpu.stat"))
core = capi.MocapCore(0)
rig = synth.ring_rig(8)
rng = np.random.default_rng(9)
obs, _ = synth.make_ba_observations(rig, N, seed=9)
init = synth.perturb_rig(rig, rng)
core.set_cameras(rig["K"], init["R"], init["t"])
helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in rig["K"]])
x0 = helpers._ba_x0([{"R": init["R"][i], "t": init["t"][i]} for i in range(8)])


T_PROC = time.perf_counter()


def solves(tag, n):
    ms, at = [], []
    for i in range(n):
        t0 = time.perf_counter()
        _, info = core.ba_solve(x0, obs, ftol=0.0, xtol=0.0, gtol=0.0, max_iter=60)
        ms.append(1e3 * (time.perf_counter() - t0))
        at.append(1e3 * (t0 - T_PROC))
    print(f"{tag}: {n} solves of {info['iterations']:.0f} iterations at {N} points, ms per solve:",
          " ".join(f"{m:.1f}" for m in ms), "| slow ones started at ms since the first solve:",
          " ".join(f"{a - at[0]:.0f}" for a, m in zip(at, ms) if m > 3 * sorted(ms)[len(ms) // 2]),
          {k: v for k, v in os.environ.items() if k.startswith(("MOCAP_", "HSA_", "HIP_", "AMD_"))}, flush=True)
    return ms


solves("cold+warm", S)
print("cpu.stat after plain solves:", cg("cpu.stat"))
if os.environ.get("DIAG_NO_BURST"):
    sys.exit(0)


def burn(sec):
    t0 = time.perf_counter()
    x = 0.0
    while time.perf_counter() - t0 < sec:
        x += 1.0
    return x


from concurrent.futures import ThreadPoolExecutor  # noqa: E402
import multiprocessing as mp  # noqa: E402
nproc = os.cpu_count() or 8
with mp.get_context("fork").Pool(nproc) as pool:
    pool.map(burn, [2.0] * nproc)
print("cpu.stat after an all-core burst:", cg("cpu.stat"))
solves("right after an all-core burst", 6)
print("cpu.stat at the end:", cg("cpu.stat"))
