"""TEST INFRASTRUCTURE ONLY (oracle/) -- ctypes binding of oracle/c/libmocap_oracle.so,
the plain-C restatement of the reference hot path (see oracle/c/mocap_oracle.c).
Used by tests/ (parity at sizes the Python restatement cannot reach), by
__graft_entry__.smoke() and as bench.py's `cpu_baseline` ("port") leg."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "c", "libmocap_oracle.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, "c", f) for f in ("mocap_oracle.c", "blob_oracle.c")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "c"), "-B", "libmocap_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        vp, i32, i64, dbl = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double
        L.mo_cams_create.restype = vp
        L.mo_cams_create.argtypes = [i32, vp, vp, vp, i32]
        L.mo_cams_destroy.argtypes = [vp]
        L.mo_get_fundamental.argtypes = [vp, vp]
        L.mo_triangulate.argtypes = [vp, i64, vp, vp, vp]
        L.mo_match_triangulate.argtypes = [vp, i64, i32, vp, vp, dbl, i32, i64, vp, vp, vp, vp, vp, vp]
        L.mo_ba_residuals.argtypes = [vp, i32, vp, i64, vp, vp]
        L.bo_cam_create.restype = vp
        L.bo_cam_create.argtypes = [i32, i32, vp, vp, i32]
        L.bo_cam_destroy.argtypes = [vp]
        L.bo_find_dots.restype = i32
        L.bo_find_dots.argtypes = [vp, vp, i32, vp, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class COracle:
    def __init__(self, K, R, t, f32_rounding=True):
        L = _load()
        self.K = np.ascontiguousarray(K, dtype=np.float64).reshape(-1, 9)
        self.C = self.K.shape[0]
        self.R = np.ascontiguousarray(R, dtype=np.float64).reshape(self.C, 9)
        self.t = np.ascontiguousarray(t, dtype=np.float64).reshape(self.C, 3)
        self._h = L.mo_cams_create(self.C, _p(self.K), _p(self.R), _p(self.t), int(f32_rounding))

    def __del__(self):
        if getattr(self, "_h", None):
            _load().mo_cams_destroy(self._h)
            self._h = None

    def fundamental(self):
        F = np.zeros((self.C, self.C, 3, 3))
        _load().mo_get_fundamental(self._h, _p(F))
        return F

    def triangulate(self, obs):
        obs = np.ascontiguousarray(obs, dtype=np.float64)
        N = obs.shape[0]
        xyz = np.empty((N, 3))
        err = np.empty(N)
        _load().mo_triangulate(self._h, N, _p(obs), _p(xyz), _p(err))
        return xyz, err

    def match_triangulate(self, blobs, counts, gate_px=5.0, K_max=None, G_cap=1 << 40):
        blobs = np.ascontiguousarray(blobs, dtype=np.float32)
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        F, C, M, _ = blobs.shape
        K_max = C * M if K_max is None else int(K_max)
        xyz = np.full((F, K_max, 3), np.nan)
        err = np.full((F, K_max), np.nan)
        corr = np.full((F, K_max, C), -1, dtype=np.int16)
        n_out = np.zeros(F, dtype=np.int32)
        status = np.zeros(F, dtype=np.int32)
        n_cand = np.zeros(F, dtype=np.int32)
        _load().mo_match_triangulate(self._h, F, M, _p(blobs), _p(counts), float(gate_px), K_max,
                                     int(G_cap), _p(xyz), _p(err), _p(corr), _p(n_out), _p(status),
                                     _p(n_cand))
        return {"xyz": xyz, "err": err, "corr": corr, "n_out": n_out, "status": status,
                "n_cand": n_cand}

    def ba_residuals(self, params, obs):
        params = np.ascontiguousarray(np.atleast_2d(params), dtype=np.float64)
        obs = np.ascontiguousarray(obs, dtype=np.float64)
        P, N = params.shape[0], obs.shape[0]
        r = np.empty((P, N))
        _load().mo_ba_residuals(self._h, P, _p(params), N, _p(obs), _p(r))
        return r


class BlobOracle:
    """Blob-extraction stage (helpers.py:68-82, 143-163) in plain C, sequential Suzuki-Abe contours
    (oracle/c/blob_oracle.c).  One instance per camera set."""

    def __init__(self, rows, cols, K, dist, rotation=None):
        L = _load()
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(-1, 9)
        self.C = K.shape[0]
        dist = np.ascontiguousarray(dist, dtype=np.float64).reshape(self.C, 5)
        rot = [0] * self.C if rotation is None else list(rotation)
        self.rows, self.cols = int(rows), int(cols)
        self._cams = [L.bo_cam_create(self.rows, self.cols, _p(K[c]), _p(dist[c]), int(rot[c])) for c in range(self.C)]

    def __del__(self):
        for h in getattr(self, "_cams", []):
            _load().bo_cam_destroy(h)
        self._cams = []

    def find_blobs(self, images, M_max=64, want_processed=False):
        """images [F][C][rows][cols][3] -> same dict layout as MocapCore.find_blobs (counts are NOT clipped)."""
        L = _load()
        images = np.ascontiguousarray(images, dtype=np.uint8)
        F, C = images.shape[:2]
        assert C == self.C
        blobs = np.zeros((F, C, M_max, 2), dtype=np.float32)
        counts = np.zeros((F, C), dtype=np.int32)
        ncont = np.zeros((F, C), dtype=np.int32)
        proc = np.zeros((F, C, self.cols, self.cols, 3), dtype=np.uint8) if want_processed else None
        nc = ctypes.c_int()
        for f in range(F):
            for c in range(C):
                counts[f, c] = L.bo_find_dots(self._cams[c], _p(images[f, c]), M_max, _p(blobs[f, c]),
                                              _p(proc[f, c]) if want_processed else None, ctypes.byref(nc))
                ncont[f, c] = nc.value
        out = {"blobs": blobs, "counts": counts, "n_contours": ncont}
        if want_processed:
            out["processed"] = proc
        return out
