#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for lib in ""; do export MOCAP_CORE_LIB=$PWD/low-cost-mocap_amd/lib/libmocap_core$lib.so; echo "== lib $lib"; timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu
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
c = ctx({})
C, M, F, K, seed = 8, 16, 20000, 48, 1
rig = synth.ring_rig(C)
blobs, counts, _ = synth.make_blob_stream(rig, F, M, seed=seed)
ex.set_cameras(rig["K"], rig["R"], rig["t"])
base = ex.match_triangulate(blobs, counts, K_max=K)
valid = np.arange(K)[None, :] < base["n_out"][:, None]
c.set_cameras(rig["K"], rig["R"], rig["t"])
nbad = 0
for rep in range(40):
    res = c.match_triangulate(blobs, counts, K_max=K)
    same_n = np.array_equal(res["n_out"], base["n_out"])
    bad = [key for key in ("xyz", "err", "corr") if not np.array_equal(res[key][valid], base[key][valid])]
    if bad or not same_n:
        nbad += 1
        fr = np.unique(np.nonzero((res["corr"] != base["corr"]).any(-1) & valid)[0])[:4]
        if nbad <= 3:
            for f in fr:
                ks = np.nonzero((res["corr"][f] != base["corr"][f]).any(-1) & valid[f])[0]
                print("rep", rep, "same_n", same_n, "frame", f, "n_out", res["n_out"][f], base["n_out"][f], "cand", base["n_cand"][f], "slots", ks[:6], "got", res["corr"][f, ks[0]], "want", base["corr"][f, ks[0]], "err", res["err"][f, ks[0]], base["err"][f, ks[0]])
print("bad runs", nbad, "of 40")
PY
done
