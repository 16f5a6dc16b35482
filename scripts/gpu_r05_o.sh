#!/bin/bash
# repetition runs of the search kernel against the exhaustive walk (races show as intermittent mismatches): several shapes / layouts
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 800 python - <<'PY' 2>&1 | grep -v amdgpu
import os, sys, numpy as np
sys.path.insert(0, "low-cost-mocap_amd")
from mocap_core import capi, synth
def ctx(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return capi.MocapCore(0)
    finally:
        for k, v in old.items():
            if v is None: del os.environ[k]
            else: os.environ[k] = v
ex = ctx({"MOCAP_EVAL_BB": "0"})
cs = {"default": ctx({}), "runtime_layout": ctx({"MOCAP_BB_FIXED_LAYOUT": "0"}), "pl2": ctx({"MOCAP_BB_PL": "2"})}
for C, M, F, K, seed, reps in [(8, 16, 20000, 48, 3, 30), (8, 16, 8000, 64, 4, 20), (6, 16, 6000, 40, 10, 20), (4, 8, 6000, 24, 8, 20), (12, 12, 1500, 60, 11, 10), (8, 16, 3000, 128, 5, 10)]:
    rig = synth.ring_rig(C)
    blobs, counts, _ = synth.make_blob_stream(rig, F, M, seed=seed, dropout=0.15 if C == 12 else 0.05)
    ex.set_cameras(rig["K"], rig["R"], rig["t"])
    base = ex.match_triangulate(blobs, counts, K_max=K)
    valid = np.arange(K)[None, :] < base["n_out"][:, None]
    for name, c in cs.items():
        c.set_cameras(rig["K"], rig["R"], rig["t"])
        nbad = 0
        for rep in range(reps):
            res = c.match_triangulate(blobs, counts, K_max=K)
            ok = np.array_equal(res["n_out"], base["n_out"]) and all(np.array_equal(res[k][valid], base[k][valid]) for k in ("xyz", "err", "corr"))
            nbad += 0 if ok else 1
        print(C, M, K, name, c.last_frame_kernel(), "bad runs", nbad, "of", reps, "roots/frame", float(base["n_out"].mean()))
PY
