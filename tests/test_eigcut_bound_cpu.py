"""The inequality behind the frame kernel's branch and bound (mocap_device.hpp EigCut), checked on the CPU
against the oracle's restatement of the reference arithmetic (float32 roundings of cv.projectPoints included):

    sum of squared reprojection residuals of ANY completion of a partial group
        >=  lam1(B'_partial) / max_c |P_c[2] M|^2  >=  1 / (trace(B'_partial^-1) * max_c |P_c[2] M|^2),

B' = M^T B M the DLT matrix in a world frame moved to the point closest to all optical axes (M = [[I, c0], [0, 1]]; any
c0 is valid, this one is tight), with the allowances the kernel charges.  A violation here would mean the search could drop a winner."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "low-cost-mocap_amd"))

from mocap_core import synth  # noqa: E402
from oracle import mocap_oracle as mo  # noqa: E402


def _contribution(P, xy):
    """DLT rows of one view (helpers.py:315-316) -> B_c = ra ra^T + rb rb^T."""
    ra = xy[1] * P[2] - P[1]
    rb = P[0] - xy[0] * P[2]
    return np.outer(ra, ra) + np.outer(rb, rb)


def test_partial_group_eigenvalue_bound_never_exceeds_the_reprojection_error():
    C = 8
    rig = synth.ring_rig(C)
    blobs, counts, _ = synth.make_blob_stream(rig, 12, 16, seed=3)
    Ks, R, t = rig["K"], rig["R"], rig["t"]
    P = np.stack([mo.projection_matrix(Ks[c], R[c], t[c]) for c in range(C)])
    # the origin the library picks (capi.hip mocap_set_cameras): least-squares point closest to the optical axes
    A, b = np.zeros((3, 3)), np.zeros(3)
    for c in range(C):
        d = R[c][2]
        Pm = np.eye(3) - np.outer(d, d)
        A += Pm
        b += Pm @ (-R[c].T @ t[c])
    M = np.eye(4)
    M[:3, 3] = np.linalg.solve(A, b)
    p3max2 = float(np.max(((P[:, 2] @ M) ** 2).sum(1)))
    assert p3max2 < 0.5 * float(np.max((P[:, 2] ** 2).sum(1)))  # (this rig's world frame sits in camera 0)
    Ftab = mo.fundamental_table(Ks, R, t)
    rng = np.random.default_rng(0)
    checked = 0
    ratios = []
    for f in range(blobs.shape[0]):
        omax = float(np.abs(blobs[f][np.arange(16)[None, :] < counts[f][:, None]]).max())
        roots, hits = mo.match_frame(blobs[f], counts[f], Ftab)
        for r, root in enumerate(roots):
            groups = list(mo.enumerate_groups(root, hits[r], C))
            if len(groups) < 2:
                continue
            for corr in [groups[i] for i in rng.choice(len(groups), size=min(12, len(groups)), replace=False)]:
                cams = [c for c in range(C) if corr[c] >= 0]
                v = len(cams)
                if v < 2:
                    continue
                obs = np.full((C, 2), np.nan)
                for c in cams:
                    obs[c] = blobs[f, c, corr[c]]
                X = mo.triangulate_point(obs, Ks, R, t)
                S32 = mo.reprojection_error(obs, X, Ks, R, t) * 2 * v  # the reference's own (float32) error, as a sum
                if not np.isfinite(S32):
                    continue
                # the allowance for cv.projectPoints' float32 output, as the kernel charges it (EigCut::o2slack)
                S_adj = 1.002 * S32 + 2 * v * (1100.0 * 2.0 ** -46) * omax ** 2
                Bc = {c: _contribution(P[c], obs[c]) for c in cams}
                full = M.T @ sum(Bc.values()) @ M
                lam = np.linalg.eigvalsh(full)
                assert lam[0] / p3max2 <= S_adj, (f, r, lam[0] / p3max2, S32)
                ratios.append(lam[0] / p3max2 / max(S32, 1e-300))
                # partial groups: the root plus any subset of the other cameras (what a block of the search fixes)
                others = [c for c in cams if c != root[0]]
                for _ in range(4):
                    keep = [c for c in others if rng.random() < 0.6]
                    if not keep:
                        continue
                    part = M.T @ (Bc[root[0]] + sum(Bc[c] for c in keep)) @ M
                    lp = np.linalg.eigvalsh(part)
                    assert lp[0] <= lam[0] * (1 + 1e-9) + 1e-9 * np.trace(full)       # B_partial <= B_full
                    assert lp[0] / p3max2 <= S_adj
                    if lp[0] > 1e-9 * np.trace(part):                                  # s1 = trace(B^-1) >= 1 / lam1
                        s1 = np.trace(np.linalg.inv(part))
                        assert 1.0 / s1 <= lp[0] * (1 + 1e-6)
                        # the test the kernel runs: dropped  <=>  s1 * (p3max2 * limit_adj + 2e-12 tr) < 1
                        assert not (s1 * (p3max2 * S_adj + 2e-12 * np.trace(part)) < 1.0), "a completion's own error dropped it"
                    checked += 1
    assert checked > 500
    print('bound / error: median', np.median(ratios), 'max', np.max(ratios))
    assert np.median(ratios) > 0.4  # the bound is not vacuous (0.03 without the change of origin)


def test_dlt_rows_are_depth_times_residual():
    """x^T B x = sum_c z_c^2 (du_c^2 + dv_c^2) for any point -- the identity the bound rests on (plain K only)."""
    rig = synth.ring_rig(5)
    Ks, R, t = rig["K"], rig["R"], rig["t"]
    rng = np.random.default_rng(1)
    for _ in range(50):
        X = rng.normal(size=3)
        x = np.append(X, 1.0)
        lhs = rhs = 0.0
        for c in range(5):
            P = mo.projection_matrix(Ks[c], R[c], t[c])
            o = rng.uniform(0, 320, size=2)
            lhs += x @ _contribution(P, o) @ x
            p = P @ x
            rhs += p[2] ** 2 * ((o[0] - p[0] / p[2]) ** 2 + (o[1] - p[1] / p[2]) ** 2)
        assert abs(lhs - rhs) <= 1e-9 * max(abs(lhs), 1.0)


def test_block_search_selects_what_the_exhaustive_walk_selects():
    """A CPU model of the search the frame kernel runs (blocks of the Cartesian product, seed = the block
    with the largest s1, drop a block when its bound beats the root's best error by the kernel's allowances, winner =
    lexicographic minimum of (error, candidate index) over what was evaluated) against the reference's selection
    (every candidate evaluated, np.argmin = first minimum), with the oracle's arithmetic for the errors."""
    C, M = 8, 14
    rig = synth.ring_rig(C)
    blobs, counts, _ = synth.make_blob_stream(rig, 5, M, seed=11)
    Ks, R, t = rig["K"], rig["R"], rig["t"]
    P = np.stack([mo.projection_matrix(Ks[c], R[c], t[c]) for c in range(C)])
    A, b = np.zeros((3, 3)), np.zeros(3)
    for c in range(C):
        Pm = np.eye(3) - np.outer(R[c][2], R[c][2])
        A += Pm
        b += Pm @ (-R[c].T @ t[c])
    Mx = np.eye(4)
    Mx[:3, 3] = np.linalg.solve(A, b)
    p3max2 = float(np.max(((P[:, 2] @ Mx) ** 2).sum(1))) * (1 + 1e-5)
    Ftab = mo.fundamental_table(Ks, R, t)
    PL = 8
    roots_checked = evaluated = total = 0
    for f in range(blobs.shape[0]):
        omax = float(np.abs(blobs[f][np.arange(M)[None, :] < counts[f][:, None]]).max())
        o2slack = (1100.0 * 2.0 ** -46) * omax ** 2
        roots, hits = mo.match_frame(blobs[f], counts[f], Ftab)
        for r, root in enumerate(roots):
            groups = list(mo.enumerate_groups(root, hits[r], C))
            if len(groups) < 2 or (groups[0] >= 0).sum() < 2:
                continue
            G = len(groups)
            v = int((groups[0] >= 0).sum())

            def error(g):
                corr = groups[g]
                obs = np.full((C, 2), np.nan)
                for c in range(C):
                    if corr[c] >= 0:
                        obs[c] = blobs[f, c, corr[c]]
                e = mo.reprojection_error(obs, mo.triangulate_point(obs, Ks, R, t), Ks, R, t)
                return np.inf if e is None or not np.isfinite(e) else float(e)

            ref_err = [error(g) for g in range(G)]
            ref_win = int(np.argmin(ref_err))                       # helpers.py:418
            # blocks: the fastest digits (cameras root+1 ... with >= 2 hits) stay open until their product reaches PL
            active = [c for c in range(root[0] + 1, C) if len(hits[r][c]) >= 2]
            pl, nl = 1, 0
            while nl < len(active) and pl < PL:
                pl *= len(hits[r][active[nl]])
                nl += 1
            open_cams = set(active[:nl])
            nblk = G // pl

            def s1_of_block(gh):
                corr = groups[gh * pl]                              # any candidate of the block: fixed cameras agree
                B = np.zeros((4, 4))
                views = 0
                for c in range(C):
                    if corr[c] >= 0 and c not in open_cams:
                        B += _contribution(P[c], blobs[f, c, corr[c]])
                        views += 1
                if views < 2:
                    return 0.0, 0.0
                Bs = Mx.T @ B @ Mx
                return float(np.trace(np.linalg.inv(Bs))), float(np.trace(Bs) + 2 * (Mx[:3, 3] @ Mx[:3, 3] + 1) * np.trace(B))

            s1 = [s1_of_block(gh) for gh in range(nblk)]
            seed = int(np.argmax([x[0] for x in s1]))
            best = (np.inf, -1)
            seen = set()

            def evaluate_block(gh):
                nonlocal best, evaluated
                for g in range(gh * pl, (gh + 1) * pl):
                    e = ref_err[g]
                    evaluated += 1
                    seen.add(g)
                    if (e, g) < best:
                        best = (e, g)

            evaluate_block(seed)
            for gh in range(nblk):
                if gh == seed:
                    continue
                s, tr = s1[gh]
                limit = best[0] * (2 * v) * (1 + 2.0 ** -40)
                limit_adj = 1.002 * limit + (2 * v) * o2slack
                dropped = s > 0.0 and s * (p3max2 * limit_adj + 2e-12 * tr) < 1.0
                if not dropped:
                    evaluate_block(gh)
            assert best[1] == ref_win, (f, r, best, ref_win, ref_err[ref_win])
            roots_checked += 1
            total += G
    assert roots_checked > 60
    print('evaluated', evaluated, 'of', total)
    assert evaluated < 0.5 * total  # and it does skip work
