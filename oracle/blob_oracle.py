"""TEST INFRASTRUCTURE ONLY (oracle/) -- never imported by the product path.

CPU restatement of the reference's blob-extraction stage, the step that produces
the `image_points` the hot path consumes:

    Cameras._camera_read   computer_code/api/helpers.py:68-82    per camera: np.rot90, make_square
                                                                (helpers.py:507-523), cv.undistort,
                                                                cv.GaussianBlur 9x9, cv.filter2D 5x5,
                                                                cv.cvtColor RGB2BGR
    Cameras._find_dot      computer_code/api/helpers.py:143-163  grey, threshold 255*0.2, findContours
                                                                RETR_TREE, contour moments, int() centroids

The OpenCV calls are the restatements of oracle/cv_image_restate.py (PARITY UNPINNED there, see its
header).  The reference-owned parts (rot90, make_square, the centroid rule) are pinned by
tests/golden/blobs_*.npz, which oracle/make_golden.py produces by running the reference's own
_camera_read / _find_dot through the stub harness.
"""
import numpy as np

from . import cv_image_restate as ci

SHARPEN = np.array([[-2, -1, -1, -1, -2],
                    [-1, 1, 3, 1, -1],
                    [-1, 3, 4, 3, -1],
                    [-1, 1, 3, 1, -1],
                    [-2, -1, -1, -1, -2]])          # helpers.py:76-80
FEATHER = 8                                         # helpers.py:516


def make_square(img):
    """helpers.py:507-523, restated with integer arithmetic: (1 - alpha) = (7 - i) / 8 is exact, the
    float product is exact and the uint8 store truncates."""
    rows, cols = img.shape[:2]
    size = max(rows, cols)
    ax, ay = (size - cols) // 2, (size - rows) // 2
    if cols != size or ay < FEATHER:
        raise ValueError("the reference's make_square only works for landscape frames with >= 8 rows of padding")
    out = np.zeros((size, size, 3), dtype=np.uint8)
    out[ay:ay + rows, ax:ax + cols] = img
    for i in range(FEATHER):
        out[ay - i - 1, :] = (img[0, :].astype(np.int64) * (FEATHER - 1 - i)) >> 3
        out[ay + rows + i, :] = (img[-1, :].astype(np.int64) * (FEATHER - 1 - i)) >> 3
    return out


_MAP_CACHE = {}


def preprocess(raw, K, dist, rotation=0):
    """helpers.py:71-82 for one camera: raw RGB frame -> the BGR frame the reference streams and feeds
    to _find_dot."""
    f = np.rot90(np.asarray(raw, dtype=np.uint8), k=rotation)
    f = make_square(f)
    key = (tuple(np.asarray(K, dtype=np.float64).ravel()), tuple(np.asarray(dist, dtype=np.float64).ravel()), f.shape[:2])
    if key not in _MAP_CACHE:
        _MAP_CACHE[key] = ci.undistort_map(K, dist, f.shape[0], f.shape[1])
    f = ci.remap_fixed(f, *_MAP_CACHE[key])
    f = ci.gaussian_blur(f, (9, 9), 0)
    f = ci.filter2d(f, -1, SHARPEN)
    return ci.cvt_color(f, ci.COLOR_RGB2BGR)


def binary_mask(frame_bgr):
    """helpers.py:145-146."""
    grey = ci.cvt_color(frame_bgr, ci.COLOR_RGB2GRAY)
    return ci.threshold(grey, 255 * 0.2, 255, ci.THRESH_BINARY)[1]


def centroids_from_mask(mask):
    """helpers.py:147-156: one [x, y] per contour with non-zero area, in findContours' order."""
    contours, _ = ci.find_contours(mask, ci.RETR_TREE, ci.CHAIN_APPROX_SIMPLE)
    pts = []
    for c in contours:
        m = ci.moments(c)
        if m["m00"] != 0:
            pts.append([int(m["m10"] / m["m00"]), int(m["m01"] / m["m00"])])
    return pts


def find_dots(raw_frames, Ks, dists, rotations=None):
    """All cameras of one frame set -> (processed BGR frames, list of per-camera [[x, y], ...]; an empty
    camera yields [] here where the reference substitutes [[None, None]], helpers.py:158-159)."""
    frames, points = [], []
    for i, raw in enumerate(raw_frames):
        f = preprocess(raw, Ks[i], dists[i], 0 if rotations is None else rotations[i])
        frames.append(f)
        points.append(centroids_from_mask(binary_mask(f)))
    return frames, points


def pack_points(points, M_max):
    """Per-camera point lists -> the frame path's input layout: blobs f32 [C][M_max][2], counts i32 [C]."""
    C = len(points)
    blobs = np.zeros((C, M_max, 2), dtype=np.float32)
    counts = np.zeros(C, dtype=np.int32)
    for c, p in enumerate(points):
        counts[c] = len(p)
        n = min(len(p), M_max)
        if n:
            blobs[c, :n] = np.asarray(p[:n], dtype=np.float32)
    return blobs, counts
