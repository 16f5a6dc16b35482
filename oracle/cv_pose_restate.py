"""TEST INFRASTRUCTURE ONLY (oracle/) -- never imported by the product path.

NumPy restatement of the OpenCV / opencv_contrib calls of the reference's initial pose estimation
(SURVEY.md 8f row 4), the caller of bundle adjustment:

    calculate_camera_pose   computer_code/api/index.py:229-270
        cv.findFundamentalMat(p1, p2, cv.FM_RANSAC, 1, 0.99999)          index.py:246
        cv.sfm.essentialFromFundamental(F, K0, K1)                       index.py:247
        cv.sfm.motionFromEssential(E)                                    index.py:248

PARITY UNPINNED (OpenCV is absent here): these follow the published OpenCV 4.x (>= 4.5) sources

  findFundamentalMat     calib3d/src/fundam.cpp + ptsetreg.cpp: >= 15 points and FM_RANSAC ->
                         RANSACPointSetRegistrator(modelPoints 7, threshold, confidence, maxIters 1000) with
                         RNG((uint64)-1) (multiply-with-carry, coefficient 4164903690), getSubset's
                         redraw-on-duplicate sampling, FMEstimatorCallback::checkSubset (haveCollinearPoints on
                         the last point of either subset), run7Point with Hartley normalisation, solveCubic,
                         error = max of the two squared point-line distances stored as float, inlier when
                         err <= (float)(thr*thr), RANSACUpdateNumIters.  The result is the best MINIMAL
                         7-point model: cv::findFundamentalMat does not refit on the inliers.
  sfm (libmv)            essentialFromFundamental: E = K2^T F K1;  motionFromEssential: SVD of E, last
                         column of U / last row of Vt flipped to make both determinants positive,
                         R in {U W Vt, U W^T Vt}, t = +-U[:, 2].

Two things in OpenCV depend on an SVD's free choices and are therefore fixed here by a documented
convention instead (the HIP core uses the same one):
  * run7Point takes "the last two right singular vectors" of the 7x9 system as the basis of its null
    space; any basis gives the same set of up to three matrices but in a basis-dependent order.  Here the
    matrices of one sample are ordered by ascending F[0][0] (after the final F[2][2] = 1 scaling).
  * the four (R, t) candidates of motionFromEssential are a set; their order follows this file's SVD.
    The reference keeps the candidate with the most points in front, so the order only matters for exact ties.
"""
import numpy as np

FM_RANSAC = 8


class RNG:
    """cv::RNG: multiply-with-carry."""

    def __init__(self, state=0xFFFFFFFFFFFFFFFF):
        self.state = state if state else 0xFFFFFFFF

    def next(self):
        self.state = ((self.state & 0xFFFFFFFF) * 4164903690 + (self.state >> 32)) & 0xFFFFFFFFFFFFFFFF
        return self.state & 0xFFFFFFFF

    def uniform(self, a, b):
        return a if a == b else int(self.next() % (b - a) + a)


def have_collinear_points(m, count):
    """fundam.cpp haveCollinearPoints: only the last of `count` points is tested."""
    i = count - 1
    for j in range(i):
        dx1 = float(m[j, 0]) - float(m[i, 0])
        dy1 = float(m[j, 1]) - float(m[i, 1])
        for k in range(j):
            dx2 = float(m[k, 0]) - float(m[i, 0])
            dy2 = float(m[k, 1]) - float(m[i, 1])
            if abs(dx2 * dy1 - dy2 * dx1) <= np.finfo(np.float32).eps * (abs(dx1) + abs(dy1) + abs(dx2) + abs(dy2)):
                return True
    return False


def get_subset(m1, m2, rng, max_attempts=10000, model_points=7):
    count = m1.shape[0]
    for _ in range(max_attempts):
        idx = []
        for i in range(model_points):
            idx_i = rng.uniform(0, count)
            while idx_i in idx:
                idx_i = rng.uniform(0, count)
            idx.append(idx_i)
        ms1, ms2 = m1[idx], m2[idx]
        if not have_collinear_points(ms1, model_points) and not have_collinear_points(ms2, model_points):
            return idx
    return None


def real_cubic_roots(c):
    """Real roots of c[0] x^3 + c[1] x^2 + c[2] x + c[3] (cv::solveCubic's root SET)."""
    c = np.asarray(c, dtype=np.float64)
    if c[0] == 0:
        if c[1] == 0:
            return [] if c[2] == 0 else [-c[3] / c[2]]
        d = c[2] * c[2] - 4 * c[1] * c[3]
        if d < 0:
            return []
        d = np.sqrt(d)
        return [(-c[2] + d) / (2 * c[1]), (-c[2] - d) / (2 * c[1])] if d > 0 else [-c[2] / (2 * c[1])]
    a1, a2, a3 = c[1] / c[0], c[2] / c[0], c[3] / c[0]
    Q = (a1 * a1 - 3 * a2) / 9
    R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) / 54
    d = Q * Q * Q - R * R
    if d > 0:
        theta = np.arccos(R / np.sqrt(Q * Q * Q))
        t0 = -2 * np.sqrt(Q)
        return [t0 * np.cos(theta / 3) - a1 / 3, t0 * np.cos((theta + 2 * np.pi) / 3) - a1 / 3,
                t0 * np.cos((theta + 4 * np.pi) / 3) - a1 / 3]
    if d == 0:
        e = -np.cbrt(R)
        return [2 * e - a1 / 3, -e - a1 / 3]
    e = np.cbrt(np.sqrt(-d) + abs(R))
    if R > 0:
        e = -e
    return [(e + Q / e) - a1 / 3]


def run_7point(ms1, ms2):
    """fundam.cpp run7Point -> list of 3x3 matrices (ordered by F[0][0], see the module header)."""
    m1 = np.asarray(ms1, dtype=np.float64)
    m2 = np.asarray(ms2, dtype=np.float64)
    m1c, m2c = m1.mean(axis=0), m2.mean(axis=0)
    s1 = np.sqrt(((m1 - m1c) ** 2).sum(axis=1)).mean()
    s2 = np.sqrt(((m2 - m2c) ** 2).sum(axis=1)).mean()
    if s1 < np.finfo(np.float32).eps or s2 < np.finfo(np.float32).eps:
        return []
    s1, s2 = np.sqrt(2.0) / s1, np.sqrt(2.0) / s2
    x0, y0 = (m1[:, 0] - m1c[0]) * s1, (m1[:, 1] - m1c[1]) * s1
    x1, y1 = (m2[:, 0] - m2c[0]) * s2, (m2[:, 1] - m2c[1]) * s2
    A = np.stack([x1 * x0, x1 * y0, x1, y1 * x0, y1 * y0, y1, x0, y0, np.ones(7)], axis=1)
    _, _, Vt = np.linalg.svd(A, full_matrices=True)
    f1, f2 = Vt[7].copy(), Vt[8].copy()
    f1 -= f2

    def cof(f, g):
        t0 = g[4] * g[8] - g[5] * g[7]
        t1 = g[3] * g[8] - g[5] * g[6]
        t2 = g[3] * g[7] - g[4] * g[6]
        return t0, t1, t2

    c = np.zeros(4)
    t0, t1, t2 = cof(None, f2)
    c[3] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2
    c[2] = (f1[0] * t0 - f1[1] * t1 + f1[2] * t2 - f1[3] * (f2[1] * f2[8] - f2[2] * f2[7])
            + f1[4] * (f2[0] * f2[8] - f2[2] * f2[6]) - f1[5] * (f2[0] * f2[7] - f2[1] * f2[6])
            + f1[6] * (f2[1] * f2[5] - f2[2] * f2[4]) - f1[7] * (f2[0] * f2[5] - f2[2] * f2[3])
            + f1[8] * (f2[0] * f2[4] - f2[1] * f2[3]))
    t0, t1, t2 = cof(None, f1)
    c[0] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2
    c[1] = (f2[0] * t0 - f2[1] * t1 + f2[2] * t2 - f2[3] * (f1[1] * f1[8] - f1[2] * f1[7])
            + f2[4] * (f1[0] * f1[8] - f1[2] * f1[6]) - f2[5] * (f1[0] * f1[7] - f1[1] * f1[6])
            + f2[6] * (f1[1] * f1[5] - f1[2] * f1[4]) - f2[7] * (f1[0] * f1[5] - f1[2] * f1[3])
            + f2[8] * (f1[0] * f1[4] - f1[1] * f1[3]))
    T1 = np.array([[s1, 0, -s1 * m1c[0]], [0, s1, -s1 * m1c[1]], [0, 0, 1]])
    T2 = np.array([[s2, 0, -s2 * m2c[0]], [0, s2, -s2 * m2c[1]], [0, 0, 1]])
    out = []
    for lam in real_cubic_roots(c):
        mu = 1.0
        s = f1[8] * lam + f2[8]
        F = np.empty(9)
        if abs(s) > np.finfo(np.float64).eps:
            mu = 1.0 / s
            lam = lam * mu
            F[8] = 1.0
        else:
            F[8] = 0.0
        F[:8] = f1[:8] * lam + f2[:8] * mu
        F = T2.T @ F.reshape(3, 3) @ T1
        if abs(F[2, 2]) > np.finfo(np.float32).eps:
            F = F * (1.0 / F[2, 2])
        out.append(F)
    out.sort(key=lambda M: M[0, 0])
    return out


def compute_error(m1, m2, F):
    """FMEstimatorCallback::computeError: double arithmetic, float result."""
    f = np.asarray(F, dtype=np.float64).ravel()
    x1, y1 = m1[:, 0].astype(np.float64), m1[:, 1].astype(np.float64)
    x2, y2 = m2[:, 0].astype(np.float64), m2[:, 1].astype(np.float64)
    a = f[0] * x1 + f[1] * y1 + f[2]
    b = f[3] * x1 + f[4] * y1 + f[5]
    c = f[6] * x1 + f[7] * y1 + f[8]
    s2 = 1.0 / (a * a + b * b)
    d2 = x2 * a + y2 * b + c
    a = f[0] * x2 + f[3] * y2 + f[6]
    b = f[1] * x2 + f[4] * y2 + f[7]
    c = f[2] * x2 + f[5] * y2 + f[8]
    s1 = 1.0 / (a * a + b * b)
    d1 = x1 * a + y1 * b + c
    with np.errstate(invalid="ignore", over="ignore"):
        return np.maximum(d1 * d1 * s1, d2 * d2 * s2).astype(np.float32)


def ransac_update_num_iters(p, ep, model_points, max_iters):
    p = min(max(p, 0.0), 1.0)
    ep = min(max(ep, 0.0), 1.0)
    num = max(1.0 - p, np.finfo(np.float64).tiny)
    denom = 1.0 - (1.0 - ep) ** model_points
    if denom < np.finfo(np.float64).tiny:
        return 0
    num, denom = np.log(num), np.log(denom)
    return max_iters if denom >= 0 or -num >= max_iters * (-denom) else int(np.rint(num / denom))


def find_fundamental_mat(points1, points2, method=FM_RANSAC, ransacReprojThreshold=3.0, confidence=0.99,
                         maxIters=1000, return_info=False):
    m1 = np.asarray(points1, dtype=np.float32).reshape(-1, 2)
    m2 = np.asarray(points2, dtype=np.float32).reshape(-1, 2)
    n = m1.shape[0]
    if method != FM_RANSAC or n < 15:
        raise NotImplementedError("the reference calls FM_RANSAC on >= 15 points (fewer would switch OpenCV to LMedS)")
    thr = 3.0 if ransacReprojThreshold <= 0 else ransacReprojThreshold
    eps = np.finfo(np.float64).eps
    conf = 0.99 if (confidence < eps or confidence > 1 - eps) else confidence
    t = np.float32(thr * thr)
    rng = RNG()
    niters = max(maxIters, 1)
    best_F, best_mask, max_good, best_iter = None, None, 0, -1
    it = 0
    while it < niters:
        idx = get_subset(m1, m2, rng)
        if idx is None:
            if it == 0:
                return (None, None) if not return_info else (None, None, {})
            break
        for F in run_7point(m1[idx], m2[idx]):
            err = compute_error(m1, m2, F)
            mask = err <= t
            good = int(mask.sum())
            if good > max(max_good, 6):
                best_F, best_mask, max_good, best_iter = F, mask, good, it
                niters = ransac_update_num_iters(conf, (n - good) / n, 7, niters)
        it += 1
    if best_F is None:
        return (None, None) if not return_info else (None, None, {})
    mask = best_mask.astype(np.uint8).reshape(-1, 1)
    if return_info:
        return best_F, mask, {"inliers": max_good, "iterations": it, "best_iteration": best_iter}
    return best_F, mask


def essential_from_fundamental(F, K1, K2):
    return np.asarray(K2, dtype=np.float64).T @ np.asarray(F, dtype=np.float64) @ np.asarray(K1, dtype=np.float64)


def motion_from_essential(E):
    U, _, Vt = np.linalg.svd(np.asarray(E, dtype=np.float64))
    if np.linalg.det(U) < 0:
        U[:, 2] *= -1
    if np.linalg.det(Vt) < 0:
        Vt[2, :] *= -1
    W = np.array([[0.0, -1, 0], [1, 0, 0], [0, 0, 1]])
    R1, R2 = U @ W @ Vt, U @ W.T @ Vt
    t = U[:, 2].reshape(3, 1)
    return [R1, R1.copy(), R2, R2.copy()], [t.copy(), -t, t.copy(), -t]


def install(cv2_module):
    cv2_module.FM_RANSAC = FM_RANSAC
    cv2_module.findFundamentalMat = find_fundamental_mat
    cv2_module.sfm.essentialFromFundamental = essential_from_fundamental
    cv2_module.sfm.motionFromEssential = motion_from_essential
