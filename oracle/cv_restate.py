"""TEST INFRASTRUCTURE ONLY (oracle/) -- never imported by the product path.

NumPy restatement of the three OpenCV entry points the reference hot path calls.
OpenCV (+ opencv_contrib `sfm`) is an un-vendored, un-pinned dependency of the
reference (README.md:17 of the reference says "build OpenCV from source"), it is
absent from /root/reference and from this image, so these restate the *published*
OpenCV 4.x algorithms.  PARITY UNPINNED at these three calls: the reference has no
tests or golden vectors that pin them.

Call sites in the reference (computer_code/api/helpers.py):
  * cv.sfm.fundamentalFromProjections(P1, P2)            helpers.py:362
  * cv.computeCorrespondEpilines(pts_f32, 1, F)          helpers.py:363
  * cv.projectPoints(X_f32, R(3x3), t, K, dist=[])       helpers.py:231-237

`F32_ROUNDING` toggles the two float32 roundings OpenCV applies because the
reference hands it float32 inputs (epiline coefficients come back as float32;
the projected 3-D point is first rounded to float32 and the projected pixel is
returned as float32).  The GPU core must match the oracle in both modes.
"""
import numpy as np

F32_ROUNDING = True


def cv_determinant(M):
    """opencv/modules/core/src/lapack.cpp `cv::determinant` for an n x n (n > 3) CV_64F
    matrix: hal::LU64f (matrix_decomp `LUImpl`, partial pivoting, eps = 100*DBL_EPSILON),
    then p * prod(diag) multiplied in row order.  Scalar loop, no FMA (baseline x86-64)."""
    A = np.array(M, dtype=np.float64)
    m = A.shape[0]
    p = 1.0
    eps = np.finfo(np.float64).eps * 100
    for i in range(m):
        k = i
        for j in range(i + 1, m):
            if abs(A[j, i]) > abs(A[k, i]):
                k = j
        if abs(A[k, i]) < eps:
            return 0.0
        if k != i:
            A[[i, k], i:] = A[[k, i], i:]
            p = -p
        d = -1.0 / A[i, i]
        for j in range(i + 1, m):
            alpha = A[j, i] * d
            for kk in range(i + 1, m):
                A[j, kk] = A[j, kk] + alpha * A[i, kk]
    result = p
    for i in range(m):
        result = result * A[i, i]
    return float(result)


def fundamental_from_projections(P1, P2):
    """opencv_contrib/modules/sfm/src/fundamental.cpp `fundamentalFromProjections`
    (itself libmv's FundamentalFromProjections): F(i,j) = det([X_j ; Y_i]) with
    X_j = rows (j+1, j+2 mod 3) of P1 and Y_i = rows (i+1, i+2 mod 3) of P2.
    cv::determinant on a 4x4 double is LU with partial pivoting (cv_determinant above)."""
    P1 = np.asarray(P1, dtype=np.float64)
    P2 = np.asarray(P2, dtype=np.float64)
    F = np.empty((3, 3), dtype=np.float64)
    for i in range(3):
        Y = P2[[(i + 1) % 3, (i + 2) % 3], :]
        for j in range(3):
            X = P1[[(j + 1) % 3, (j + 2) % 3], :]
            F[i, j] = cv_determinant(np.vstack([X, Y]))
    return F


def compute_correspond_epilines(points, which_image, F):
    """opencv/modules/calib3d/src/fundam.cpp `computeCorrespondEpilines`:
    l = F [x y 1]^T in double, scaled by 1/sqrt(a^2+b^2) (1 if that is 0),
    stored as float32 (output depth = max(input depth, CV_32F); the reference
    passes float32 points).  Returns shape (n, 1, 3) like cv2."""
    pts = np.asarray(points)
    if pts.dtype != np.float32:
        raise TypeError("reference passes float32 points (helpers.py:363)")
    pts = pts.reshape(-1, 2)
    f = np.asarray(F, dtype=np.float64)
    if which_image == 2:
        f = f.T
    f = f.ravel()
    out = np.empty((pts.shape[0], 1, 3), dtype=np.float32 if F32_ROUNDING else np.float64)
    for n in range(pts.shape[0]):
        x = np.float64(pts[n, 0])
        y = np.float64(pts[n, 1])
        a = f[0] * x + f[1] * y + f[2]
        b = f[3] * x + f[4] * y + f[5]
        c = f[6] * x + f[7] * y + f[8]
        nu = a * a + b * b
        nu = 1.0 / np.sqrt(nu) if nu != 0 else 1.0
        a *= nu
        b *= nu
        c *= nu
        out[n, 0, 0] = a
        out[n, 0, 1] = b
        out[n, 0, 2] = c
    return out


def project_points(object_points, rvec, tvec, camera_matrix, dist_coeffs):
    """opencv/modules/calib3d/src/calibration.cpp `cvProjectPoints2Internal`
    with a 3x3 rotation passed as `rvec` (used verbatim) and empty distortion:
        x = R X + t ; z = z ? 1/z : 1 ; x *= z ; y *= z ; u = x*fx + cx ; v = y*fy + cy
    Input points arrive as float32 (helpers.py:232) and are widened to double;
    the output has the input's depth (float32).  Returns ((n,1,2) array, None)."""
    X = np.asarray(object_points)
    if F32_ROUNDING:
        X = X.astype(np.float32)
    X = X.reshape(-1, 3).astype(np.float64)
    R = np.asarray(rvec, dtype=np.float64)
    if R.shape != (3, 3):
        raise ValueError("reference passes the 3x3 rotation matrix (helpers.py:233)")
    t = np.asarray(tvec, dtype=np.float64).reshape(3)
    if np.asarray(dist_coeffs).size != 0:
        raise ValueError("reference passes empty distortion (helpers.py:236)")
    K = np.asarray(camera_matrix, dtype=np.float64)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    out = np.empty((X.shape[0], 1, 2), dtype=np.float32 if F32_ROUNDING else np.float64)
    for n in range(X.shape[0]):
        Xx, Xy, Xz = X[n]
        x = R[0, 0] * Xx + R[0, 1] * Xy + R[0, 2] * Xz + t[0]
        y = R[1, 0] * Xx + R[1, 1] * Xy + R[1, 2] * Xz + t[1]
        z = R[2, 0] * Xx + R[2, 1] * Xy + R[2, 2] * Xz + t[2]
        z = 1.0 / z if z != 0 else 1.0
        x *= z
        y *= z
        out[n, 0, 0] = x * fx + cx
        out[n, 0, 1] = y * fy + cy
    return out, None
