"""Phase B of the round-3 kernels (csrc/frame_bb.hip match(), csrc/frame_kernel.hip match_wide) no longer walks the cameras
one after the other with every root (helpers.py:359-406): the camera-0 roots are matched against ALL cameras at once, and
only the roots created on the way take part in a chain over the cameras (frame_bb: existing new roots against camera i,
then the unclaimed blobs of i; match_wide: the new roots of camera j against all cameras after j at the moment they are
created).  This file states both schedules in plain Python, with the kernels' own building blocks -- the one-lane-per-pair
"bit mask + repeated minimum" ordering, claims as masks -- and checks them, root list and hit lists, against the oracle's
camera-after-camera restatement of the reference (oracle/mocap_oracle.match_frame), without a GPU."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "low-cost-mocap_amd"))

from mocap_core import synth  # noqa: E402
from oracle import mocap_oracle as mo  # noqa: E402


def _pair(blobs, counts, Ftab, root, i, gate):
    """One (root, camera) pair the way a lane does it: gated blobs as a bit mask, extracted in (distance, index) order by
    repeated minimum (strict <, ascending index: ties keep the smaller index); returns (hit list, claim mask)."""
    rc, rb = root
    a, b, c = mo.epiline(Ftab[rc, i], blobs[rc, rb])
    n = int(counts[i])
    px, py = blobs[i, :n, 0].astype(np.float64), blobs[i, :n, 1].astype(np.float64)
    den = np.sqrt(a ** 2 + b ** 2)
    dist = lambda k: abs(a * px[k] + b * py[k] + c) / den      # helpers.py:373  # noqa: E731
    mask = 0
    for k in range(n):
        if dist(k) < gate:                                      # helpers.py:375,383
            mask |= 1 << k
    order, rem = [], mask
    while rem:
        bd, bk = np.inf, 0
        for k in range(n):
            if rem >> k & 1 and dist(k) < bd:
                bd, bk = dist(k), k
        order.append(bk)
        rem &= ~(1 << bk)
    claim = 0
    if order:
        k0 = order[0]
        for k in range(n):
            if mask >> k & 1 and px[k] == px[k0] and py[k] == py[k0]:   # helpers.py:391: removal BY VALUE
                claim |= 1 << k
    return order, claim


def _schedule(blobs, counts, Ftab, gate, creation_time_matching):
    C = blobs.shape[0]
    n0 = int(counts[0])
    roots = [(0, k) for k in range(n0)]
    hits = [[[] for _ in range(C)] for _ in roots]
    claimed = [0] * C
    for r in range(n0):                                  # B0: camera-0 roots x all cameras, no order between the pairs
        for i in range(1, C):
            hits[r][i], cl = _pair(blobs, counts, Ftab, roots[r], i, gate)
            claimed[i] |= cl
    for i in range(1, C):                                # the chain
        if not creation_time_matching:                   # frame_bb.hip: the roots created so far against camera i
            for r in range(n0, len(roots)):
                hits[r][i], cl = _pair(blobs, counts, Ftab, roots[r], i, gate)
                claimed[i] |= cl
        first_new = len(roots)
        for k in range(int(counts[i])):                  # helpers.py:402-406
            if not claimed[i] >> k & 1:
                roots.append((i, k))
                hits.append([[] for _ in range(C)])
        if creation_time_matching:                       # match_wide: the new roots against every camera after theirs
            for r in range(first_new, len(roots)):
                for j in range(i + 1, C):
                    hits[r][j], cl = _pair(blobs, counts, Ftab, roots[r], j, gate)
                    claimed[j] |= cl
    return roots, hits


@pytest.mark.parametrize("C,M,dropout,gate,seed", [(8, 16, 0.05, 5.0, 1), (8, 16, 0.5, 5.0, 2), (5, 12, 0.2, 8.0, 3),
                                                  (4, 6, 0.0, 3.0, 4), (12, 8, 0.3, 5.0, 5)])
def test_both_schedules_equal_the_camera_after_camera_loop(C, M, dropout, gate, seed):
    rig = synth.ring_rig(C)
    blobs, counts, _ = synth.make_blob_stream(rig, 25, M, seed=seed, dropout=dropout)
    if seed == 2:                                       # duplicate pixels: the by-value claim takes several blobs at once
        ok = counts[:, 3] >= 3
        blobs[ok, 3, 2] = blobs[ok, 3, 0]
    Ftab = mo.fundamental_table(rig["K"], rig["R"], rig["t"])
    new_roots = 0
    for f in range(blobs.shape[0]):
        ref_roots, ref_hits = mo.match_frame(blobs[f], counts[f], Ftab, gate_px=gate)
        for ctm in (False, True):
            roots, hits = _schedule(blobs[f], counts[f], Ftab, gate, ctm)
            assert roots == ref_roots, (f, ctm)
            assert hits == ref_hits, (f, ctm)
        new_roots += len(ref_roots) - int(counts[f, 0])
    assert new_roots > 0                                # the chain had something to do
