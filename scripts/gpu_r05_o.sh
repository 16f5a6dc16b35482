#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for lib in "" _prev; do export MOCAP_CORE_LIB=$PWD/low-cost-mocap_amd/lib/libmocap_core$lib.so; echo "== lib $lib"; timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu
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
vs = {"pl2": ctx({"MOCAP_BB_PL": "2"}), "pl64": ctx({"MOCAP_BB_PL": "64"}), "pl16f64": ctx({"MOCAP_BB_PL": "16", "MOCAP_BB_FLUSH": "64"}), "rt": ctx({"MOCAP_BB_FIXED_LAYOUT": "0"})}
for C, M, F, K, seed in [(8, 16, 1500, 48, 7)]:
    rig = synth.ring_rig(C)
    blobs, counts, _ = synth.make_blob_stream(rig, F, M, seed=seed)
    ex.set_cameras(rig["K"], rig["R"], rig["t"])
    base = ex.match_triangulate(blobs, counts, K_max=K)
    valid = np.arange(K)[None, :] < base["n_out"][:, None]
    for name, c in vs.items():
        c.set_cameras(rig["K"], rig["R"], rig["t"])
        nbad = 0; info = None
        for rep in range(30):
            res = c.match_triangulate(blobs, counts, K_max=K)
            bad = [key for key in ("xyz", "err", "corr") if not np.array_equal(res[key][valid], base[key][valid])]
            if bad:
                nbad += 1
                fr = np.unique(np.nonzero((res["err"] != base["err"]) & valid)[0])[:6]
                info = (bad, list(fr), [int(base["n_cand"][f]) for f in fr], c.last_frame_kernel())
        print(C, name, "bad runs", nbad, "of 30", info)
PY
done
