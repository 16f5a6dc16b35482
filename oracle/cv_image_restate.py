"""TEST INFRASTRUCTURE ONLY (oracle/) -- never imported by the product path.

NumPy / pure-Python restatement of the OpenCV calls of the reference's blob-extraction stage, the step
right BEFORE the hot path:

    Cameras._camera_read   computer_code/api/helpers.py:68-88     rot90, make_square, cv.undistort,
                                                                 cv.GaussianBlur, cv.filter2D, cv.cvtColor
    Cameras._find_dot      computer_code/api/helpers.py:143-163   cv.cvtColor, cv.threshold, cv.findContours,
                                                                 cv.moments  (+ drawing calls: no-ops here)

OpenCV is un-vendored and un-versioned in the reference and absent from this image: PARITY UNPINNED at
every function below.  They restate the published OpenCV 4.x (>= 4.5) algorithms for the exact argument
types the reference passes (8-bit 3-channel images).  All of them are integer / fixed-point pipelines in
OpenCV itself, so the restatement has no rounding freedom apart from the undistortion map (double
arithmetic, quantised to 1/32 px; the SIMD build of OpenCV may differ in the last ulp BEFORE that
quantisation):

  undistort      imgproc/undistort.dispatch.cpp: stripes of (1<<12)/cols rows, initUndistortRectifyMap
                 (CV_16SC2 + CV_16UC1 fixed-point map, INTER_BITS = 5), remap INTER_LINEAR
                 BORDER_CONSTANT with the 15-bit BilinearTab_i weights and FixedPtCast rounding.
  GaussianBlur   imgproc/smooth.dispatch.cpp: 8-bit path = fixed-point kernel of 8 fractional bits
                 (getGaussianKernelFixedPoint_ED, error diffusion), horizontal pass exact in 8.8,
                 vertical pass in 16.16, (+32768) >> 16, BORDER_REFLECT_101.
  filter2D       imgproc/filter.dispatch.cpp: float32 correlation (exact for these small integers),
                 saturate_cast<uchar>, BORDER_REFLECT_101.
  cvtColor       RGB2BGR = channel swap; RGB2GRAY 8-bit = (R*9798 + G*19235 + B*3735 + 16384) >> 15.
  threshold      8-bit THRESH_BINARY: src > floor(thresh) ? maxval : 0.
  findContours   Suzuki-Abe border following (imgproc/contours.cpp: icvFetchContour / cvFindNextContour) on
                 the image padded by one zero pixel; RETR_TREE order = pre-order of the contour tree with
                 siblings in reverse discovery order (cvInsertNodeIntoTree inserts at the head).  Labels
                 are unbounded ints here (OpenCV recycles 7-bit labels and disambiguates with
                 icvTraceContour; the intended result is the same).
  moments        imgproc/moments.cpp contourMoments: Green's-theorem sums over the polygon, exact in
                 double for integer vertices; m00 = a00 * (+-0.5), m10 = a10 * (+-1/6), ...
"""
import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
INTER_REMAP_COEF_BITS = 15
INTER_REMAP_COEF_SCALE = 1 << INTER_REMAP_COEF_BITS

COLOR_RGB2BGR = 4
COLOR_RGB2GRAY = 7
THRESH_BINARY = 0
RETR_TREE = 3
CHAIN_APPROX_SIMPLE = 2
FONT_HERSHEY_SIMPLEX = 0


# ----------------------------------------------------------------------------- undistort
def _invert3(S):
    """core/src/lapack.cpp cv::invert, n == 3, CV_64F, DECOMP_LU: closed form."""
    S = np.asarray(S, dtype=np.float64)
    d = (S[0, 0] * (S[1, 1] * S[2, 2] - S[1, 2] * S[2, 1]) - S[0, 1] * (S[1, 0] * S[2, 2] - S[1, 2] * S[2, 0])
         + S[0, 2] * (S[1, 0] * S[2, 1] - S[1, 1] * S[2, 0]))
    d = 1.0 / d
    t = np.empty(9)
    t[0] = (S[1, 1] * S[2, 2] - S[1, 2] * S[2, 1]) * d
    t[1] = (S[0, 2] * S[2, 1] - S[0, 1] * S[2, 2]) * d
    t[2] = (S[0, 1] * S[1, 2] - S[0, 2] * S[1, 1]) * d
    t[3] = (S[1, 2] * S[2, 0] - S[1, 0] * S[2, 2]) * d
    t[4] = (S[0, 0] * S[2, 2] - S[0, 2] * S[2, 0]) * d
    t[5] = (S[0, 2] * S[1, 0] - S[0, 0] * S[1, 2]) * d
    t[6] = (S[1, 0] * S[2, 1] - S[1, 1] * S[2, 0]) * d
    t[7] = (S[0, 1] * S[2, 0] - S[0, 0] * S[2, 1]) * d
    t[8] = (S[0, 0] * S[1, 1] - S[0, 1] * S[1, 0]) * d
    return t


def undistort_map(K, dist, rows, cols):
    """The fixed-point map cv.undistort builds stripe by stripe (newCameraMatrix = cameraMatrix, R = I):
    returns (sx, sy, fx, fy) int arrays [rows][cols]: integer source pixel and 1/32 fractions."""
    A = np.asarray(K, dtype=np.float64).reshape(3, 3)
    k = np.zeros(14)
    d = np.asarray(dist, dtype=np.float64).ravel()
    k[:d.size] = d
    k1, k2, p1, p2, k3, k4, k5, k6, s1, s2, s3, s4 = k[:12]
    fx, fy, u0, v0 = A[0, 0], A[1, 1], A[0, 2], A[1, 2]
    stripe0 = min(max(1, (1 << 12) // max(cols, 1)), rows)
    iu_all = np.empty((rows, cols), dtype=np.int64)
    iv_all = np.empty((rows, cols), dtype=np.int64)
    for y0 in range(0, rows, stripe0):
        n = min(stripe0, rows - y0)
        Ar = A.copy()
        Ar[1, 2] = v0 - y0
        ir = _invert3(Ar)           # (Ar * I)^-1
        for i in range(n):
            # scalar loop of initUndistortRectifyMapComputer: _x += ir[0] per column (accumulated)
            _x = np.empty(cols)
            _y = np.empty(cols)
            _w = np.empty(cols)
            ax, ay, aw = i * ir[1] + ir[2], i * ir[4] + ir[5], i * ir[7] + ir[8]
            for j in range(cols):
                _x[j], _y[j], _w[j] = ax, ay, aw
                ax += ir[0]
                ay += ir[3]
                aw += ir[6]
            w = 1.0 / _w
            x = _x * w
            y = _y * w
            x2 = x * x
            y2 = y * y
            r2 = x2 + y2
            _2xy = 2 * x * y
            kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2)
            xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + s1 * r2 + s2 * r2 * r2
            yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + s3 * r2 + s4 * r2 * r2
            # matTilt = identity: vecTilt = (xd, yd, 1), invProj = 1
            u = fx * 1.0 * xd + u0
            v = fy * 1.0 * yd + v0
            iu_all[y0 + i] = np.rint(u * INTER_TAB_SIZE).astype(np.int64)   # saturate_cast<int> = cvRound
            iv_all[y0 + i] = np.rint(v * INTER_TAB_SIZE).astype(np.int64)
    sx = (iu_all >> INTER_BITS).astype(np.int16).astype(np.int64)    # stored as short
    sy = (iv_all >> INTER_BITS).astype(np.int16).astype(np.int64)
    return sx, sy, iu_all & (INTER_TAB_SIZE - 1), iv_all & (INTER_TAB_SIZE - 1)


_BILINEAR_TAB = None


def bilinear_tab_i():
    """imgproc/imgwarp.cpp initInterTab2D(INTER_LINEAR, fixpt): [fy*32+fx][4] short weights (w00, w01,
    w10, w11), including the saturation of 32768 to 32767 at (0, 0) and the fix-up that follows it."""
    global _BILINEAR_TAB
    if _BILINEAR_TAB is None:
        n = INTER_TAB_SIZE
        itab = np.zeros(n * n * 4 + 8, dtype=np.int64)
        one_d = [(np.float32(1.0) - np.float32(i) * np.float32(1.0 / n), np.float32(i) * np.float32(1.0 / n)) for i in range(n)]
        for i in range(n):
            for j in range(n):
                base = (i * n + j) * 4
                isum = 0
                for k1 in range(2):
                    for k2 in range(2):
                        v = np.float32(one_d[i][k1] * one_d[j][k2])
                        iv = int(np.clip(np.rint(np.float64(v) * INTER_REMAP_COEF_SCALE), -32768, 32767))
                        itab[base + k1 * 2 + k2] = iv
                        isum += iv
                if isum != INTER_REMAP_COEF_SCALE:
                    diff = isum - INTER_REMAP_COEF_SCALE
                    ks2 = 1
                    Mk = mk = (ks2, ks2)
                    for k1 in range(ks2, ks2 + 2):
                        for k2 in range(ks2, ks2 + 2):
                            if itab[base + k1 * 2 + k2] < itab[base + mk[0] * 2 + mk[1]]:
                                mk = (k1, k2)
                            elif itab[base + k1 * 2 + k2] > itab[base + Mk[0] * 2 + Mk[1]]:
                                Mk = (k1, k2)
                    if diff < 0:
                        itab[base + Mk[0] * 2 + Mk[1]] -= diff
                    else:
                        itab[base + mk[0] * 2 + mk[1]] -= diff
        _BILINEAR_TAB = itab[:n * n * 4].reshape(n * n, 4)
    return _BILINEAR_TAB


def remap_fixed(src, sx, sy, fx, fy):
    """remapBilinear<FixedPtCast<int, uchar, 15>, ..., short>, BORDER_CONSTANT (value 0)."""
    src = np.asarray(src)
    H, W = src.shape[:2]
    tab = bilinear_tab_i()[fy * INTER_TAB_SIZE + fx]              # [rows][cols][4]
    S = np.zeros((H + 2, W + 2) + src.shape[2:], dtype=np.int64)  # zero frame = the constant border
    S[1:-1, 1:-1] = src
    outside = (sx >= W) | (sx + 1 < 0) | (sy >= H) | (sy + 1 < 0)
    cx = np.clip(sx, -1, W - 1) + 1
    cy = np.clip(sy, -1, H - 1) + 1
    w = tab[..., None] if src.ndim == 3 else tab
    acc = (S[cy, cx] * w[:, :, 0] + S[cy, cx + 1] * w[:, :, 1] + S[cy + 1, cx] * w[:, :, 2] + S[cy + 1, cx + 1] * w[:, :, 3])
    out = np.clip((acc + (1 << (INTER_REMAP_COEF_BITS - 1))) >> INTER_REMAP_COEF_BITS, 0, 255)
    out[outside] = 0
    return out.astype(np.uint8)


def undistort(src, cameraMatrix, distCoeffs):
    src = np.asarray(src)
    assert src.dtype == np.uint8
    sx, sy, fx, fy = undistort_map(cameraMatrix, distCoeffs, src.shape[0], src.shape[1])
    return remap_fixed(src, sx, sy, fx, fy)


# ----------------------------------------------------------------------------- GaussianBlur
def gaussian_kernel_fixed(n, sigma):
    """getGaussianKernelBitExact + getGaussianKernelFixedPoint_ED, 8 fractional bits."""
    if sigma <= 0:
        sigma = 0.15 * n + 0.35          # softdouble mulAdd(n, 0.15, 0.35) == ((n-1)*0.5 - 1)*0.3 + 0.8
    scale2x = -0.5 * 0.25 / (sigma * sigma)
    vals = [np.exp((x * x) * scale2x) for x in range(1 - n, 0, 2)]     # x = 2*(i - (n-1)/2)
    total = 2.0 * sum(vals) + 1.0
    kern = [v / total for v in vals] + [1.0 / total]
    out = [0] * n
    err, s = 0.0, 0
    for i in range(n // 2):
        adj = kern[i] * 256.0 + err
        v0 = int(np.rint(adj))
        err = adj - v0
        out[i] = out[n - 1 - i] = v0
        s += v0
    out[n // 2] = 256 - 2 * s
    return out


def _reflect101(idx, n):
    idx = np.where(idx < 0, -idx, idx)
    return np.where(idx >= n, 2 * (n - 1) - idx, idx)


def gaussian_blur(src, ksize, sigma):
    src = np.asarray(src)
    assert src.dtype == np.uint8 and ksize[0] == ksize[1]
    n = ksize[0]
    k = gaussian_kernel_fixed(n, sigma)
    H, W = src.shape[:2]
    r = n // 2
    s = src.astype(np.int64)
    cols = _reflect101(np.arange(-r, W + r), W)
    rows = _reflect101(np.arange(-r, H + r), H)
    sp = s[:, cols]
    h = sum(k[i] * sp[:, i:i + W] for i in range(n))             # ufixedpoint16, exact (<= 255*256)
    hp = h[rows]
    v = sum(k[i] * hp[i:i + H] for i in range(n))                # ufixedpoint32
    return ((v + 32768) >> 16).astype(np.uint8)


# ----------------------------------------------------------------------------- filter2D
def filter2d(src, ddepth, kernel):
    src = np.asarray(src)
    assert src.dtype == np.uint8 and ddepth == -1
    kern = np.asarray(kernel)
    kh, kw = kern.shape
    ay, ax = kh // 2, kw // 2
    H, W = src.shape[:2]
    rows = _reflect101(np.arange(-ay, H + kh - 1 - ay), H)
    cols = _reflect101(np.arange(-ax, W + kw - 1 - ax), W)
    sp = src.astype(np.int64)[rows][:, cols]
    acc = np.zeros(src.shape, dtype=np.int64)
    for i in range(kh):
        for j in range(kw):
            acc += int(kern[i, j]) * sp[i:i + H, j:j + W]          # correlation, anchor at the centre
    return np.clip(acc, 0, 255).astype(np.uint8)


# ----------------------------------------------------------------------------- colour, threshold
def cvt_color(src, code):
    src = np.asarray(src)
    if code == COLOR_RGB2BGR:
        return np.ascontiguousarray(src[:, :, ::-1])
    if code == COLOR_RGB2GRAY:
        s = src.astype(np.int64)
        return ((s[:, :, 0] * 9798 + s[:, :, 1] * 19235 + s[:, :, 2] * 3735 + (1 << 14)) >> 15).astype(np.uint8)
    raise NotImplementedError(code)


def threshold(src, thresh, maxval, typ):
    assert typ == THRESH_BINARY and np.asarray(src).dtype == np.uint8
    t = int(np.floor(thresh))
    return float(t), np.where(np.asarray(src) > t, np.uint8(maxval), np.uint8(0)).astype(np.uint8)


# ----------------------------------------------------------------------------- findContours / moments
_DY = (0, -1, -1, -1, 0, 1, 1, 1)       # direction codes 0..7: right, up-right, up, up-left, left, ...
_DX = (1, 1, 0, -1, -1, -1, 0, 1)


def _fetch_contour(img, y0, x0, nbd, is_hole):
    """icvFetchContour, CHAIN_APPROX_SIMPLE: follows one border from (y0, x0), marks it in img, returns
    the vertex list in padded coordinates."""
    s_end = s = 0 if is_hole else 4
    while True:
        s = (s - 1) & 7
        y1, x1 = y0 + _DY[s], x0 + _DX[s]
        if img[y1, x1] != 0 or s == s_end:
            break
    pts = []
    if s == s_end:
        img[y0, x0] = -nbd
        pts.append((x0, y0))
        return pts
    y3, x3 = y0, x0
    prev_s = s ^ 4
    px, py = x0, y0
    while True:
        s_end = s
        while True:
            s += 1
            y4, x4 = y3 + _DY[s & 7], x3 + _DX[s & 7]
            if img[y4, x4] != 0:
                break
        s &= 7
        if 1 <= s <= s_end:                       # (unsigned)(s - 1) < (unsigned)s_end: the right pixel was examined
            img[y3, x3] = -nbd
        elif img[y3, x3] == 1:
            img[y3, x3] = nbd
        if s != prev_s:
            pts.append((px, py))
            prev_s = s
        px += _DX[s]
        py += _DY[s]
        if (y4, x4) == (y0, x0) and (y3, x3) == (y1, x1):
            break
        y3, x3 = y4, x4
        s = (s + 4) & 7
    return pts


def find_contours(image, mode, method):
    """cv.findContours(image, RETR_TREE, CHAIN_APPROX_SIMPLE) -> (contours, hierarchy); contours are
    int32 arrays of shape (n, 1, 2) in OpenCV's output order."""
    assert mode == RETR_TREE and method == CHAIN_APPROX_SIMPLE
    src = np.asarray(image)
    H, W = src.shape
    img = np.zeros((H + 2, W + 2), dtype=np.int64)
    img[1:-1, 1:-1] = (src != 0)
    info = {}                                     # nbd -> dict(is_hole, parent, pts)
    order = []
    nbd = 1
    ys = np.flatnonzero(img.any(axis=1))
    for y in ys:
        lnbd = 0
        row = img[y]
        x = 1
        prev = 0
        # columns where something changes: scan only around non-zero pixels
        while x < W + 2:
            p = row[x]
            if p != prev:
                is_hole = False
                start = False
                if prev == 0 and p == 1:
                    start = True
                elif p == 0 and prev >= 1:
                    is_hole = True
                    start = True
                if start:
                    if is_hole and prev > 1:      # `if (prev & new_mask) lnbd.x = x - 1`
                        lnbd = prev
                    nbd += 1
                    # Suzuki's parent rule from the border met last on this row
                    if lnbd == 0:
                        parent = 0
                    else:
                        b = info[abs(lnbd)]
                        parent = b["parent"] if b["is_hole"] == is_hole else abs(lnbd)
                    pts = _fetch_contour(img, y, x - 1 if is_hole else x, nbd, is_hole)
                    info[nbd] = {"is_hole": is_hole, "parent": parent, "pts": pts}
                    order.append(nbd)
                    lnbd = nbd
                    p = row[x]                     # the pixel may have been marked just now
                prev = p
                if prev != 0 and prev != 1:
                    lnbd = abs(prev)
            x += 1
    # tree -> output order: children at the head (reverse discovery), pre-order
    children = {0: []}
    for b in order:
        children[b] = []
    for b in order:
        children[info[b]["parent"]].insert(0, b)
    seq = []
    stack = list(reversed(children[0]))
    while stack:
        b = stack.pop()
        seq.append(b)
        stack.extend(reversed(children[b]))
    contours = [np.array([[[px - 1, py - 1]] for (px, py) in info[b]["pts"]], dtype=np.int32) for b in seq]
    index = {b: i for i, b in enumerate(seq)}
    hier = np.full((1, len(seq), 4), -1, dtype=np.int32)
    for i, b in enumerate(seq):
        sib = children[info[b]["parent"]]
        k = sib.index(b)
        hier[0, i, 0] = index[sib[k + 1]] if k + 1 < len(sib) else -1
        hier[0, i, 1] = index[sib[k - 1]] if k > 0 else -1
        hier[0, i, 2] = index[children[b][0]] if children[b] else -1
        hier[0, i, 3] = index[info[b]["parent"]] if info[b]["parent"] else -1
    return contours, hier


def moments(contour):
    """imgproc/moments.cpp contourMoments (integer points): spatial moments up to order 1 are all the
    reference reads (helpers.py:152-155)."""
    pts = np.asarray(contour).reshape(-1, 2).astype(np.float64)
    out = {"m00": 0.0, "m10": 0.0, "m01": 0.0}
    if len(pts) == 0:
        return out
    a00 = a10 = a01 = 0.0
    xi_1, yi_1 = pts[-1]
    for xi, yi in pts:
        dxy = xi_1 * yi - xi * yi_1
        a00 += dxy
        a10 += dxy * (xi_1 + xi)
        a01 += dxy * (yi_1 + yi)
        xi_1, yi_1 = xi, yi
    if abs(a00) > np.finfo(np.float32).eps:
        db1_2, db1_6 = (0.5, 0.16666666666666666666666666666667) if a00 > 0 else (-0.5, -0.16666666666666666666666666666667)
        out["m00"] = a00 * db1_2
        out["m10"] = a10 * db1_6
        out["m01"] = a01 * db1_6
    return out


# ----------------------------------------------------------------------------- drawing: no-ops
def draw_contours(img, *a, **k):
    return img


def put_text(img, *a, **k):
    return img


def circle(img, *a, **k):
    return img


def install(cv2_module):
    """Attach the restated functions to a stub cv2 namespace (oracle/ref_harness.py)."""
    cv2_module.undistort = undistort
    cv2_module.GaussianBlur = gaussian_blur
    cv2_module.filter2D = filter2d
    cv2_module.cvtColor = cvt_color
    cv2_module.threshold = threshold
    cv2_module.findContours = find_contours
    cv2_module.drawContours = draw_contours
    cv2_module.moments = moments
    cv2_module.putText = put_text
    cv2_module.circle = circle
    cv2_module.COLOR_RGB2BGR = COLOR_RGB2BGR
    cv2_module.COLOR_RGB2GRAY = COLOR_RGB2GRAY
    cv2_module.THRESH_BINARY = THRESH_BINARY
    cv2_module.RETR_TREE = RETR_TREE
    cv2_module.CHAIN_APPROX_SIMPLE = CHAIN_APPROX_SIMPLE
    cv2_module.FONT_HERSHEY_SIMPLEX = FONT_HERSHEY_SIMPLEX
