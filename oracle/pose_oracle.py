"""TEST INFRASTRUCTURE ONLY (oracle/) -- never imported by the product path.

CPU restatement of the reference's initial pose estimation, the code that feeds
bundle adjustment:  calculate_camera_pose, computer_code/api/index.py:229-270.  The OpenCV calls are
oracle/cv_pose_restate.py (PARITY UNPINNED there); the reference-owned control flow restated here
(pair selection, float32 cast, cheirality vote with its quirks, pose chaining) is pinned by
tests/golden/pose_*.npz, produced by running the reference's own handler through the stub harness.

Quirks kept:
  * essentialFromFundamental always gets the intrinsics of cameras 0 and 1 (index.py:247), and
    triangulate_points takes intrinsics by position in the 2-camera list (helpers.py:305-307);
  * the cheirality vote triangulates under [camera_poses[-1], candidate]: the previous camera's GLOBAL
    pose next to the candidate's RELATIVE pose (index.py:253);
  * first strictly larger vote wins (index.py:258-261).
"""
import numpy as np

from . import cv_pose_restate as cp
from . import mocap_oracle as mo


def initial_poses(obs, Ks, threshold=1.0, confidence=0.99999, return_info=False):
    """obs (N, C, 2) float64 with NaN = unseen; Ks list of C 3x3.  Returns (R [C][3][3], t [C][3])."""
    obs = np.asarray(obs, dtype=np.float64)
    N, C, _ = obs.shape
    R = [np.eye(3)]
    t = [np.zeros(3)]
    infos = []
    for ci in range(C - 1):
        a, b = obs[:, ci], obs[:, ci + 1]
        ok = ~(np.isnan(a).any(axis=1) | np.isnan(b).any(axis=1))
        p1 = a[ok].astype(np.float32)
        p2 = b[ok].astype(np.float32)
        F, _, finfo = cp.find_fundamental_mat(p1, p2, cp.FM_RANSAC, threshold, confidence, return_info=True)
        E = cp.essential_from_fundamental(F, Ks[0], Ks[1])
        Rs, ts = cp.motion_from_essential(E)
        pair = np.stack([p1, p2], axis=1).astype(np.float64)
        best, pick = 0, None
        for i in range(4):
            X = mo.triangulate_points(pair, [Ks[0], Ks[1]], [R[-1], Rs[i]], [t[-1], ts[i].reshape(3)])
            Xc = X @ Rs[i]                      # rows = (R^T x)^T
            front = int((X[:, 2] > 0).sum() + (Xc[:, 2] > 0).sum())
            if front > best:
                best, pick = front, i
        Rn = Rs[pick] @ R[-1]
        tn = t[-1] + R[-1] @ ts[pick].reshape(3)
        R.append(Rn)
        t.append(tn)
        infos.append({"n": int(ok.sum()), "candidate": pick, "F": F, **finfo})
    R, t = np.array(R), np.array(t)
    return (R, t, infos) if return_info else (R, t)
