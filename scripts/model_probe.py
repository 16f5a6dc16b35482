"""CPU estimate (oracle arithmetic) on the 8 x 16 bench stream of a probe-first search: candidate 0 of every root (the
closest hit in every camera) is evaluated before anything else and gives the root its first bound; blocks are tested
against it; the candidates of the surviving blocks take their own first-factorisation test (stage 1) and only its
survivors are evaluated in full (stage 2).  Compared with the kernel's seed-block scheme -- and (round 5, the "s1-probe" line) with
the scheme csrc/frame_bb.hip keeps behind -DMOCAP_BB_PROBE: the seed block's candidates take one factorisation each, the one
with the largest s1 is the probe, the others are tested against its error (profiles/r05_bb_eval_cost_decomposition.txt).  usage: model_probe.py [frames] [PL]"""
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
tot = roots_multi = 0
seed_eval = seed_cut1 = 0          # current scheme: full evaluations, and how many of them the candidate-level test would cut
pb_blocks = pb_stage1 = pb_stage2 = 0
probe_is_best = 0
sp_A = sp_probe_best = sp_blocks = sp_stage1 = sp_full = sp_seed_surv = 0
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
            seed_eval += 1; pb_stage2 += 1; continue
        roots_multi += 1
        errs = {}
        def error(g):
            if g not in errs:
                corr = groups[g]; obs = np.full((C, 2), np.nan)
                for c in range(C):
                    if corr[c] >= 0: obs[c] = blobs[f, c, corr[c]]
                e = mo.reprojection_error(obs, mo.triangulate_point(obs, Ks, R, t), Ks, R, t)
                errs[g] = np.inf if e is None or not np.isfinite(e) else float(e)
            return errs[g]
        active = [c for c in range(root[0] + 1, C) if len(hits[r][c]) >= 2]
        pl, nl = 1, 0
        while nl < len(active) and pl < PL:
            pl *= len(hits[r][active[nl]]); nl += 1
        def s1_of(g0, nopen):
            corr = groups[g0]; B = np.zeros((4, 4)); views = 0
            for c in range(C):
                if corr[c] >= 0 and c not in set(active[:nopen]):
                    B += _contribution(P[c], blobs[f, c, corr[c]]); views += 1
            if views < 2: return 0.0, 0.0
            Bs = Mx.T @ B @ Mx
            return float(np.trace(np.linalg.inv(Bs))), float(np.trace(Bs) + 2 * (Mx[:3, 3] @ Mx[:3, 3] + 1) * np.trace(B))
        def dropped(s, tr, best):
            limit_adj = 1.002 * best * (2 * v) * (1 + 2.0 ** -40) + (2 * v) * o2slack
            return s > 0.0 and s * (p3max2 * limit_adj + 2e-12 * tr) < 1.0
        nblk = G // pl
        s1b = [s1_of(gh * pl, nl) for gh in range(nblk)]
        # --- the kernel's scheme: seed block in full, then the blocks that survive its best error, in full
        seed = int(np.argmax([x[0] for x in s1b]))
        best = min(error(g) for g in range(seed * pl, (seed + 1) * pl))
        seed_eval += pl
        ev_blocks = [gh for gh in range(nblk) if gh != seed and not dropped(*s1b[gh], best)]
        seed_eval += pl * len(ev_blocks)
        for gh in ev_blocks:
            for g in range(gh * pl, (gh + 1) * pl):
                if dropped(*s1_of(g, 0), best): seed_cut1 += 1
        # --- probe = the seed block's candidate with the largest candidate-level s1 (round A: s1 of the seed block's candidates)
        cs1 = [s1_of(g, 0) for g in range(seed * pl, (seed + 1) * pl)]
        sp_A += pl
        pg = seed * pl + int(np.argmax([x[0] for x in cs1]))
        sbest = error(pg)
        sp_probe_best += int(sbest <= best)
        sp_full += 1
        for k, g in enumerate(range(seed * pl, (seed + 1) * pl)):
            if g != pg and not dropped(*cs1[k], sbest): sp_full += 1; sp_seed_surv += 1
        for gh in range(nblk):
            if gh == seed or dropped(*s1b[gh], sbest): continue
            sp_blocks += 1
            for g in range(gh * pl, (gh + 1) * pl):
                sp_stage1 += 1
                if not dropped(*s1_of(g, 0), sbest): sp_full += 1
        # --- probe first
        pbest = error(0)
        probe_is_best += int(pbest <= min(error(g) for g in range(seed * pl, (seed + 1) * pl)))
        pb_stage2 += 1
        for gh in range(nblk):
            if dropped(*s1b[gh], pbest): continue
            pb_blocks += 1
            for g in range(gh * pl, (gh + 1) * pl):
                if g == 0: continue
                pb_stage1 += 1
                if not dropped(*s1_of(g, 0), pbest): pb_stage2 += 1
print(f"frames {NF} PL {PL}: candidates {tot} ({tot/NF:.0f} per frame), roots with a choice {roots_multi/NF:.1f} per frame")
print(f"  seed scheme: full evaluations {seed_eval/NF:.0f} per frame ({100*seed_eval/tot:.1f} %); of the non-seed ones the candidate-level test would cut {seed_cut1/NF:.0f}")
print(f"  probe first: probe = best of its seed block in {100*probe_is_best/max(roots_multi,1):.0f} % of the roots; surviving blocks {pb_blocks/NF:.0f}, stage-1 tests {pb_stage1/NF:.0f}, "
      f"full evaluations {pb_stage2/NF:.0f} per frame; {time.time()-t0:.0f}s")
print(f"  s1-probe: round A {sp_A/NF:.0f}, probe is the seed block's best in {100*sp_probe_best/max(roots_multi,1):.0f} %, seed-block survivors {sp_seed_surv/NF:.0f}, surviving non-seed blocks {sp_blocks/NF:.0f}, stage-1 tests {sp_stage1/NF:.0f}, full evaluations {sp_full/NF:.0f} per frame")
