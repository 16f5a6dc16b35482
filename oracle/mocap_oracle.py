"""TEST INFRASTRUCTURE ONLY (oracle/) -- the checker, never the product.

CPU restatement (NumPy + small Python loops) of the reference hot path
/root/reference/computer_code/api/helpers.py:203-421, re-expressed so that it
carries blob *indices* (the reference carries coordinates) -- needed for the
"bit-exact marker<->camera correspondence" criterion.  Each function cites the
reference lines it follows.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.

PARITY STATUS.  The repo-owned arithmetic (helpers.py) is pinned: tests/golden/*.npz
are produced by running the reference's own functions through oracle/ref_harness.py
(see oracle/make_golden.py) and this restatement is checked against them.
PARITY UNPINNED for the three OpenCV calls (oracle/cv_restate.py): OpenCV is an
un-vendored, un-versioned dependency and the reference has no tests pinning it.

Conventions shared with the C restatement (oracle/c/mocap_oracle.c) and the HIP core:
  * ties in the epipolar distance order are broken by blob index (stable order); the
    reference uses NumPy's default unstable argsort (helpers.py:384), so parity is
    claimed on tie-free inputs; blobs duplicated at the same pixel are harmless.
  * a root's closest hit removes *all* blobs with the same coordinates from the
    "unmatched" set (value comparison, helpers.py:391).
"""
import numpy as np
from scipy import linalg

from . import cv_restate


# ----------------------------------------------------------------------------- cameras
def pose_arrays(camera_poses):
    """list of {"R","t"} (lists / ndarrays, t (3,) or (3,1)) -> R (C,3,3), t (C,3)."""
    R = np.array([np.array(p["R"], dtype=np.float64).reshape(3, 3) for p in camera_poses])
    t = np.array([np.array(p["t"], dtype=np.float64).reshape(3) for p in camera_poses])
    return R, t


def projection_matrix(K, R, t):
    """helpers.py:305-308 / :351-355: P = K @ [R | t]."""
    return np.asarray(K, dtype=np.float64) @ np.c_[R, t]


def fundamental_table(Ks, R, t):
    """F[a][b] = fundamentalFromProjections(P_a, P_b) (helpers.py:362); depends only on the pair."""
    C = len(R)
    Ps = [projection_matrix(Ks[i], R[i], t[i]) for i in range(C)]
    F = np.zeros((C, C, 3, 3))
    for a in range(C):
        for b in range(C):
            if a != b:
                F[a, b] = cv_restate.fundamental_from_projections(Ps[a], Ps[b])
    return F


# ----------------------------------------------------------------------------- DLT + error
def numpy_order_sum(vals, as_float64_array):
    """Sum in the order NumPy uses at helpers.py:241 (`errors.mean()`):
    float64 arrays -> pairwise sum (8-way unrolled for 8 <= n <= 128);
    object arrays (groups holding None) -> plain left-to-right loop."""
    n = len(vals)
    if not as_float64_array or n < 8:
        s = 0.0
        for v in vals:
            s = s + v
        return s
    r = [vals[j] for j in range(8)]
    i = 8
    while i < n - (n % 8):
        for j in range(8):
            r[j] = r[j] + vals[i + j]
        i += 8
    s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
    while i < n:
        s = s + vals[i]
        i += 1
    return s


def triangulate_point(obs, Ks, R, t):
    """helpers.py:293-327.  obs: (C,2) float with NaN = unseen.  Returns (3,) or None.

    Quirk kept: after unseen cameras are dropped, the intrinsic matrix is taken by
    *compacted* position j, the pose by original camera (helpers.py:296-298,305-307)."""
    seen = [c for c in range(len(R)) if not np.isnan(obs[c, 0])]
    if len(seen) <= 1:
        return None
    A = []
    for j, c in enumerate(seen):
        P = projection_matrix(Ks[j], R[c], t[c])
        x, y = float(obs[c, 0]), float(obs[c, 1])
        A.append(y * P[2, :] - P[1, :])
        A.append(P[0, :] - x * P[2, :])
    A = np.array(A).reshape((len(seen) * 2, 4))
    B = A.transpose() @ A
    U, s, Vh = linalg.svd(B, full_matrices=False)
    return Vh[3, 0:3] / Vh[3, 3]


def reprojection_error(obs, X, Ks, R, t, object_dtype=False):
    """helpers.py:214-241: mean over the 2v components of (obs - project(X))^2; None if v <= 1.
    cv.projectPoints gets X rounded to float32 and returns float32 pixels (cv_restate).

    object_dtype: the caller's array is an object ndarray (np.array(cameraPoints) holding None,
    index.py:232) -> NumPy sums it left to right; otherwise a fully seen group is a float64
    array and is summed pairwise (see numpy_order_sum)."""
    C = len(R)
    seen = [c for c in range(C) if not np.isnan(obs[c, 0])]
    if len(seen) <= 1:
        return None
    comps = []
    for j, c in enumerate(seen):
        proj, _ = cv_restate.project_points(np.asarray(X, dtype=np.float64)[None, :], R[c], t[c], Ks[j], [])
        pu, pv = float(proj[0, 0, 0]), float(proj[0, 0, 1])
        du = float(obs[c, 0]) - pu
        dv = float(obs[c, 1]) - pv
        comps.append(du * du)
        comps.append(dv * dv)
    return numpy_order_sum(comps, as_float64_array=(len(seen) == C and not object_dtype)) / len(comps)


def triangulate_points(obs, Ks, R, t):
    """helpers.py:330-336.  obs (N,C,2) NaN-coded -> xyz (N,3) (NaN rows when < 2 views)."""
    out = np.full((len(obs), 3), np.nan)
    for n in range(len(obs)):
        X = triangulate_point(obs[n], Ks, R, t)
        if X is not None:
            out[n] = X
    return out


def reprojection_errors(obs, xyz, Ks, R, t):
    """helpers.py:203-211 but *aligned* with the input (NaN where the reference skips).
    An (N,C,2) capture with any None is an object ndarray in the reference (index.py:232)."""
    object_dtype = bool(np.isnan(obs).any())
    out = np.full(len(obs), np.nan)
    for n in range(len(obs)):
        if np.isnan(xyz[n, 0]):
            continue
        e = reprojection_error(obs[n], xyz[n], Ks, R, t, object_dtype=object_dtype)
        if e is not None:
            out[n] = e
    return out


# ----------------------------------------------------------------------------- frame path
def epiline(F_ab, point_xy):
    """helpers.py:362-364: float32 point in, float32 (a,b,c) out, then `.tolist()` -> Python floats."""
    l = cv_restate.compute_correspond_epilines(np.array([point_xy], dtype=np.float32), 1, F_ab)
    return [float(l[0, 0, 0]), float(l[0, 0, 1]), float(l[0, 0, 2])]


def match_frame(blobs, counts, Ftab, gate_px=5.0):
    """helpers.py:349-406, index-carrying.

    blobs (C,M,2) float, counts (C,).  Returns (roots, hits):
      roots: list of (camera, blob index); hits[r][c] = blob indices of camera c within
      `gate_px` of root r's epipolar line, ascending distance (ties: ascending index);
      [] = no hit (the reference appends [None, None]); cameras <= root camera are []."""
    C = blobs.shape[0]
    roots = [(0, k) for k in range(int(counts[0]))]
    hits = [[[] for _ in range(C)] for _ in roots]
    for i in range(1, C):
        n_i = int(counts[i])
        px = blobs[i, :n_i, 0].astype(np.float64)
        py = blobs[i, :n_i, 1].astype(np.float64)
        claimed = np.zeros(n_i, dtype=bool)
        for r, (rc, rb) in enumerate(roots):
            a, b, c = epiline(Ftab[rc, i], blobs[rc, rb])
            if n_i == 0:
                continue
            # helpers.py:373
            d = np.abs(a * px + b * py + c) / np.sqrt(a ** 2 + b ** 2)
            idx = [k for k in range(n_i) if d[k] < gate_px]          # helpers.py:375,383 (strict)
            idx.sort(key=lambda k: (d[k], k))                       # helpers.py:384 (stable contract)
            hits[r][i] = idx
            if idx:
                k0 = idx[0]                                          # helpers.py:391 (removal by value)
                claimed |= (px == px[k0]) & (py == py[k0])
        for k in range(n_i):                                         # helpers.py:402-406
            if not claimed[k]:
                roots.append((i, k))
                hits.append([[] for _ in range(C)])
    return roots, hits


def enumerate_groups(root, hits_r, C):
    """helpers.py:387-400: Cartesian product; the first processed camera (root cam + 1) is the
    fastest-varying digit.  Yields corr (C,) int arrays, -1 = none."""
    rc, rb = root
    radices = [max(1, len(hits_r[c])) if c > rc else 1 for c in range(C)]
    total = 1
    for x in radices:
        total *= x
    for g in range(total):
        corr = np.full(C, -1, dtype=np.int64)
        corr[rc] = rb
        rem = g
        for c in range(rc + 1, C):
            dgt = rem % radices[c]
            rem //= radices[c]
            if hits_r[c]:
                corr[c] = hits_r[c][dgt]
        yield corr


def find_point_correspondance_and_object_points(blobs, counts, Ks, R, t, gate_px=5.0, Ftab=None):
    """helpers.py:339-421 on the packed layout.  Returns dict with
    errors (K,), object_points (K,3), corr (K,C) int (-1 none), n_candidates."""
    C = blobs.shape[0]
    if Ftab is None:
        Ftab = fundamental_table(Ks, R, t)
    roots, hits = match_frame(blobs, counts, Ftab, gate_px)
    errs, pts, corrs = [], [], []
    n_cand = 0
    for r, root in enumerate(roots):
        best = None
        for corr in enumerate_groups(root, hits[r], C):          # helpers.py:410-411
            obs = np.full((C, 2), np.nan)
            for c in range(C):
                if corr[c] >= 0:
                    obs[c] = blobs[c, corr[c]]
            X = triangulate_point(obs, Ks, R, t)
            if X is None:                                         # helpers.py:413-414
                break
            n_cand += 1                                           # groups actually triangulated
            e = reprojection_error(obs, X, Ks, R, t)              # helpers.py:416
            if best is None or e < best[0]:                       # first minimum (np.argmin)
                best = (e, X, corr)
        if best is not None:
            errs.append(best[0])
            pts.append(best[1])
            corrs.append(best[2])
    return {
        "errors": np.array(errs, dtype=np.float64),
        "object_points": np.array(pts, dtype=np.float64).reshape(-1, 3),
        "corr": np.array(corrs, dtype=np.int64).reshape(-1, C),
        "n_candidates": n_cand,
        "n_roots": len(roots),
    }


# ----------------------------------------------------------------------------- after the path
def world_epilogue(object_points, to_world_coords_matrix):
    """helpers.py:96-103 (inline in Cameras._camera_read): camera-0 coordinates -> world."""
    out = np.array(object_points, dtype=np.float64).reshape(-1, 3).copy()
    for i, object_point in enumerate(out):
        new_object_point = np.array([[-1, 0, 0], [0, -1, 0], [0, 0, 1]]) @ object_point
        new_object_point = np.concatenate((new_object_point, [1]))
        new_object_point = np.array(to_world_coords_matrix) @ new_object_point
        new_object_point = new_object_point[:3] / new_object_point[3]
        new_object_point[1], new_object_point[2] = new_object_point[2], new_object_point[1]
        out[i] = new_object_point
    return out


def locate_objects(object_points, errors):
    """helpers.py:424-480 restated with explicit loops: 3-LED patterns (two points 0.095 m from a lead
    point and 0.15 m from each other, +-0.025), first valid pair in row-major order wins
    (cartesian_product, helpers.py:532-533); only the lead point is checked against
    `already_matched_points` (helpers.py:437).  Returns a list of dicts like the reference plus "lead"."""
    P = np.asarray(object_points, dtype=np.float64).reshape(-1, 3)
    E = np.asarray(errors, dtype=np.float64).reshape(-1)
    K = P.shape[0]
    dist1, dist2 = 0.095, 0.15

    def dist(i, j):
        d = P[i] - P[j]
        return np.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])

    matched, objects = set(), []
    for i in range(K):
        if i in matched:
            continue
        matches = [j for j in range(K) if np.abs(dist(i, j) - dist1) < 0.025]
        if len(matches) < 2:
            continue
        found = False
        for a in matches:
            for b in matches:
                if np.abs(dist(a, b) - dist2) > 0.025:
                    continue
                matched.update((i, a, b))
                location = (P[a] + P[b]) / 2
                error = ((E[i] + E[a]) + E[b]) / 3
                hv = P[a] - P[b]
                hv = hv / np.sqrt((hv[0] * hv[0] + hv[1] * hv[1]) + hv[2] * hv[2])
                heading = np.arctan2(hv[1], hv[0])
                heading = heading - np.pi if heading > np.pi / 2 else heading
                heading = heading + np.pi if heading < -np.pi / 2 else heading
                drone_index = 0 if (P[i] - location)[1] > 0 else 1
                objects.append({"pos": location, "heading": -heading, "error": error,
                                "droneIndex": drone_index, "lead": i})
                found = True
                break
            if found:
                break
    return objects


# ----------------------------------------------------------------------------- bundle adjustment
def rotvec_to_matrix(rv):
    """scipy.spatial.transform.Rotation.from_rotvec(rv).as_matrix() (helpers.py:258), restated:
    rotvec -> unit quaternion (small-angle series below 1e-3 rad) -> matrix."""
    rv = np.asarray(rv, dtype=np.float64)
    angle = np.sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2])
    if angle <= 1e-3:
        a2 = angle * angle
        scale = 0.5 - a2 / 48 + a2 * a2 / 3840
    else:
        scale = np.sin(angle / 2) / angle
    x, y, z = scale * rv
    w = np.cos(angle / 2)
    x2, y2, z2, w2 = x * x, y * y, z * z, w * w
    xy, zw, xz, yw, yz, xw = x * y, z * w, x * z, y * w, y * z, x * w
    return np.array([
        [x2 - y2 - z2 + w2, 2 * (xy - zw), 2 * (xz + yw)],
        [2 * (xy + zw), -x2 + y2 - z2 + w2, 2 * (yz - xw)],
        [2 * (xz - yw), 2 * (yz + xw), -x2 - y2 + z2 + w2],
    ])


def ba_params_to_poses(params):
    """helpers.py:247-262: x = [f0, (f_i, rotvec_i, t_i) for i>=1]; camera 0 = (I, 0)."""
    params = np.asarray(params, dtype=np.float64)
    C = int((params.size - 1) / 7) + 1
    R = np.zeros((C, 3, 3))
    t = np.zeros((C, 3))
    R[0] = np.eye(3)
    for i in range(C - 1):
        R[i + 1] = rotvec_to_matrix(params[i * 7 + 2:i * 7 + 5])
        t[i + 1] = params[i * 7 + 5:i * 7 + 8]
    return R, t


def ba_residuals(params, obs, Ks):
    """helpers.py:264-276 before the float32 cast: per point (>= 2 views) the mean squared
    reprojection error of the point re-triangulated with the current poses.  The focal
    entries of `params` are dead (helpers.py:267-270 writes into a temporary copy).
    Returns r (N,) float64 with NaN where the reference drops the point."""
    R, t = ba_params_to_poses(params)
    xyz = triangulate_points(obs, Ks, R, t)
    return reprojection_errors(obs, xyz, Ks, R, t)
