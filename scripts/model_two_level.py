"""CPU estimate (oracle arithmetic) of what a second, finer level of the block search would save on the 8 x 16 bench
stream: candidates evaluated with coarse blocks (>= PL candidates) only vs. coarse blocks whose survivors are cut into
sub-blocks (>= PL2 candidates) that are tested again.  Bounds against the root's error after its seed block only (what
the kernel has when it tests the first 256 blocks of a frame).  usage: model_two_level.py [frames] [PL] [PL2]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "low-cost-mocap_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mocap_core import synth
from oracle import mocap_oracle as mo
from test_eigcut_bound_cpu import _contribution

NF = int(sys.argv[1]) if len(sys.argv) > 1 else 30
PL = int(sys.argv[2]) if len(sys.argv) > 2 else 16
PL2 = int(sys.argv[3]) if len(sys.argv) > 3 else 4
C, M = 8, 16
rig = synth.ring_rig(C)
blobs, counts, _ = synth.make_blob_stream(rig, NF, M, seed=1)
Ks, R, t = rig["K"], rig["R"], rig["t"]
P = np.stack([mo.projection_matrix(Ks[c], R[c], t[c]) for c in range(C)])
A, b = np.zeros((3, 3)), np.zeros(3)
for c in range(C):
    Pm = np.eye(3) - np.outer(R[c][2], R[c][2]); A += Pm; b += Pm @ (-R[c].T @ t[c])
Mx = np.eye(4); Mx[:3, 3] = np.linalg.solve(A, b)
p3max2 = float(np.max(((P[:, 2] @ Mx) ** 2).sum(1))) * (1 + 1e-5)
Ftab = mo.fundamental_table(Ks, R, t)
tot = ev1 = ev2 = tests2 = 0
t0 = time.time()
for f in range(NF):
    omax = float(np.abs(blobs[f][np.arange(M)[None, :] < counts[f][:, None]]).max())
    o2slack = (1100.0 * 2.0 ** -46) * omax ** 2
    roots, hits = mo.match_frame(blobs[f], counts[f], Ftab)
    for r, root in enumerate(roots):
        groups = list(mo.enumerate_groups(root, hits[r], C))
        if (groups[0] >= 0).sum() < 2:
            continue
        G = len(groups); v = int((groups[0] >= 0).sum()); tot += G
        if G == 1:
            ev1 += 1; ev2 += 1; continue
        def error(g):
            corr = groups[g]; obs = np.full((C, 2), np.nan)
            for c in range(C):
                if corr[c] >= 0: obs[c] = blobs[f, c, corr[c]]
            e = mo.reprojection_error(obs, mo.triangulate_point(obs, Ks, R, t), Ks, R, t)
            return np.inf if e is None or not np.isfinite(e) else float(e)
        active = [c for c in range(root[0] + 1, C) if len(hits[r][c]) >= 2]
        def level(P_):
            pl, nl = 1, 0
            while nl < len(active) and pl < P_:
                pl *= len(hits[r][active[nl]]); nl += 1
            return pl, nl
        pl, nl = level(PL); pl2, nl2 = level(PL2)
        def s1_of(g0, nopen):
            corr = groups[g0]; B = np.zeros((4, 4)); views = 0
            for c in range(C):
                if corr[c] >= 0 and c not in set(active[:nopen]):
                    B += _contribution(P[c], blobs[f, c, corr[c]]); views += 1
            if views < 2: return 0.0, 0.0
            Bs = Mx.T @ B @ Mx
            return float(np.trace(np.linalg.inv(Bs))), float(np.trace(Bs) + 2 * (Mx[:3, 3] @ Mx[:3, 3] + 1) * np.trace(B))
        nblk = G // pl
        s1 = [s1_of(gh * pl, nl) for gh in range(nblk)]
        seed = int(np.argmax([x[0] for x in s1]))
        best = min(error(g) for g in range(seed * pl, (seed + 1) * pl))
        ev1 += pl; ev2 += pl
        limit_adj = 1.002 * best * (2 * v) * (1 + 2.0 ** -40) + (2 * v) * o2slack
        def dropped(s, tr): return s > 0.0 and s * (p3max2 * limit_adj + 2e-12 * tr) < 1.0
        for gh in range(nblk):
            if gh == seed or dropped(*s1[gh]): continue
            ev1 += pl
            for j in range(pl // pl2):
                tests2 += 1
                if not dropped(*s1_of(gh * pl + j * pl2, nl2)): ev2 += pl2
print(f"frames {NF} PL {PL} PL2 {PL2}: candidates {tot}, evaluated coarse-only {ev1} ({100*ev1/tot:.1f} %), with the fine level {ev2} "
      f"({100*ev2/tot:.1f} %), fine tests {tests2}; {time.time()-t0:.0f}s")
