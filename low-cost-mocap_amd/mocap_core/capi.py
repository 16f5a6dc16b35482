"""ctypes binding of the C ABI in include/mocap_core.h (lib/libmocap_core.so).

Load order: PyTorch ships its own libamdhip64 under the same SONAME as the system one this library
links to; the first one loaded serves the whole process.  When PyTorch is used alongside (device
buffers, torch.distributed), import torch BEFORE the first MocapCore() so both share PyTorch's copy.

This is the host side of the drop-in boundary.  There is NO CPU fallback: if the shared
library is missing or no MI355X is visible, construction raises.  Marshalling only --
every number comes from the HIP kernels.
"""
import ctypes
import os

import numpy as np

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# MOCAP_CORE_LIB points at another build of the same C ABI (A/B measurements of kernel variants)
LIB_PATH = os.environ.get("MOCAP_CORE_LIB") or os.path.join(_PKG_ROOT, "lib", "libmocap_core.so")

MOCAP_OK = 0
MOCAP_E_NOCONV = -5
ST_ROOT_OVERFLOW = 1
ST_CAND_OVERFLOW = 2
ST_HIT_OVERFLOW = 4
ST_INTRACTABLE = 16      # with ST_CAND_OVERFLOW: a root of more than 2^24 groups the exact search could not bound
ST_FINAL = 32            # the re-submit pass has seen the frame: its status is final
ST_LOG2_GROUPS_SHIFT, ST_LOG2_GROUPS_MASK = 20, 0x1FF
ST_ROUNDED = 8          # informational (mocap_match_triangulate_f64): a coordinate was rounded to float32
BLOB_ST_POINT_OVERFLOW = 1
BLOB_ST_CAP_OVERFLOW = 2
OPT_F32_ROUNDING = 1
OPT_EXHAUSTIVE_WALK = 2
OPT_BOUNDED_RESUBMIT = 4

_vp, _i32, _i64, _dbl, _u32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_uint32

# name -> (restype, argtypes); must list every symbol include/mocap_core.h declares
SIGNATURES = {
    "mocap_create": (_i32, [_i32, ctypes.POINTER(_vp)]),
    "mocap_destroy": (None, [_vp]),
    "mocap_last_error": (ctypes.c_char_p, [_vp]),
    "mocap_version": (ctypes.c_char_p, []),
    "mocap_set_stream": (_i32, [_vp, _vp]),
    "mocap_synchronize": (_i32, [_vp]),
    "mocap_last_frame_kernel": (ctypes.c_char_p, [_vp]),
    "mocap_set_options": (_i32, [_vp, _u32]),
    "mocap_set_tuning": (_i32, [_vp, _i32, _i32, _i32]),
    "mocap_set_frame_limits": (_i32, [_vp, _i32, _i32]),
    "mocap_limits": (None, [ctypes.POINTER(_i32), ctypes.POINTER(_i32)]),
    "mocap_set_cameras": (_i32, [_vp, _i32, _vp, _vp, _vp]),
    "mocap_get_fundamental": (_i32, [_vp, _vp]),
    "mocap_triangulate": (_i32, [_vp, _i64, _vp, _vp, _vp]),
    "mocap_triangulate_dev": (_i32, [_vp, _i64, _vp, _vp, _vp]),
    "mocap_match_triangulate": (_i32, [_vp, _i64, _i32, _vp, _vp, _dbl, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mocap_match_triangulate_dev": (_i32, [_vp, _i64, _i32, _vp, _vp, _dbl, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mocap_match_triangulate_dev_auto": (_i32, [_vp, _i64, _i32, _vp, _vp, _dbl, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mocap_resubmit_dev": (_i32, [_vp, _i64, _i32, _vp, _vp, _dbl, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mocap_match_triangulate_auto": (_i32, [_vp, _i64, _i32, _vp, _vp, _dbl, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mocap_match_triangulate_f64": (_i32, [_vp, _i64, _i32, _vp, _vp, _dbl, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mocap_track_frame": (_i32, [_vp, _i64, _i32, _vp, _vp, _dbl, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "mocap_track_frame_images": (_i32, [_vp, _i64, _vp, _i32, _dbl, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp,
                                        _vp, _vp, _vp]),
    "mocap_track_frame_dev": (_i32, [_vp, _i64, _i32, _vp, _vp, _dbl, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "mocap_set_image_params": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "mocap_set_blob_options": (_i32, [_vp, _i32]),
    "mocap_get_undistort_map": (_i32, [_vp, _i32, _vp]),
    "mocap_find_blobs": (_i32, [_vp, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "mocap_find_blobs_dev": (_i32, [_vp, _i64, _vp, _i32, _vp, _vp, _vp, _vp]),
    "mocap_set_world_transform": (_i32, [_vp, _vp]),
    "mocap_locate_objects": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mocap_locate_objects_dev": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mocap_initial_poses": (_i32, [_vp, _i32, _i64, _vp, _vp, _dbl, _dbl, _i32, _vp, _vp, _vp]),
    "mocap_find_fundamental": (_i32, [_vp, _i64, _vp, _vp, _dbl, _dbl, _i32, _vp, _vp, _vp]),
    "mocap_ba_residuals": (_i32, [_vp, _i32, _vp, _i64, _vp, _vp]),
    "mocap_ba_normal_eq": (_i32, [_vp, _vp, _i64, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "mocap_ba_trust_region_step": (_i32, [_vp, _i32, _i64, _vp, _vp, _dbl, _vp, _i32, _vp, _vp]),
    "mocap_track_record_bytes": (_i32, [_i32]),
    "mocap_compact_tracks_dev": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "mocap_reproject": (_i32, [_vp, _i64, _vp, _vp, _vp]),
    "mocap_set_ba_progress": (_i32, [_vp, _vp, _vp]),
    "mocap_ba_profile": (_i32, [_vp, _vp, _i64, _vp, _i32, _i32, _i32, _vp]),
    "mocap_ba_solve": (_i32, [_vp, _vp, _i64, _vp, _dbl, _dbl, _dbl, _i32, _i32, _i32, _vp]),
    "mocap_ba_solve_ex": (_i32, [_vp, _vp, _i64, _vp, _dbl, _dbl, _dbl, _i32, _i32, _i32, _vp, _i32]),
}

_lib = None


class MocapError(RuntimeError):
    pass


def load_library(path=None):
    """dlopen the core.  Raises (never falls back) when the library is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise MocapError(f"{p} not found: build it with `make -C {_PKG_ROOT}` (hipcc, gfx950); "
                         "there is no CPU fallback")
    lib = ctypes.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError = the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class MocapCore:
    """One context = one GPU (include/mocap_core.h).  Methods mirror the C entry points."""

    def __init__(self, device_id=0):
        self.lib = load_library()
        h = ctypes.c_void_p()
        rc = self.lib.mocap_create(int(device_id), ctypes.byref(h))
        if rc != MOCAP_OK:
            raise MocapError(f"mocap_create(device {device_id}) failed with {rc}: no MI355X visible? "
                             "(there is no CPU fallback)")
        self._h = h
        self.device_id = int(device_id)
        self.C = 0
        self._hit_cap, self._force_wide = 32, False
        self.f32_rounding = True     # MOCAP_OPT_F32_ROUNDING, the library's default

    def close(self):
        if getattr(self, "_h", None):
            self.lib.mocap_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, allow=()):
        if rc != MOCAP_OK and rc not in allow:
            raise MocapError(f"mocap_core error {rc}: {self.lib.mocap_last_error(self._h).decode()}")
        return rc

    def last_frame_kernel(self):
        """Name of the kernel the last frame batch went to (diagnostic; results never depend on it)."""
        return self.lib.mocap_last_frame_kernel(self._h).decode()

    # ------------------------------------------------------------------ configuration
    def set_cameras(self, K, R, t):
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(-1, 9)
        C = K.shape[0]
        R = np.ascontiguousarray(R, dtype=np.float64).reshape(C, 9)
        t = np.ascontiguousarray(t, dtype=np.float64).reshape(C, 3)
        self._check(self.lib.mocap_set_cameras(self._h, C, _p(K), _p(R), _p(t)))
        self.C = C

    def set_options(self, f32_rounding=True, exhaustive_walk=False, bounded_resubmit=False):
        """exhaustive_walk: MOCAP_OPT_EXHAUSTIVE_WALK, the verification mode (every candidate group evaluated in full).
        bounded_resubmit: MOCAP_OPT_BOUNDED_RESUBMIT (no whole-GPU enumeration of roots the exact search gives up on)."""
        self._check(self.lib.mocap_set_options(self._h, (OPT_F32_ROUNDING if f32_rounding else 0) |
                                               (OPT_EXHAUSTIVE_WALK if exhaustive_walk else 0) |
                                               (OPT_BOUNDED_RESUBMIT if bounded_resubmit else 0)))
        self.f32_rounding = bool(f32_rounding)

    def set_tuning(self, frame_threads=0, heavy_threshold=-1, slice_size=0):
        self._check(self.lib.mocap_set_tuning(self._h, int(frame_threads), int(heavy_threshold), int(slice_size)))

    def set_frame_limits(self, hit_cap=0, force_wide=False):
        """hit_cap: gated hits kept per (root, camera) by the wide-frame variant (0 = keep the current
        value); force_wide: run every batch through that variant (tests)."""
        self._apply_frame_limits(hit_cap, force_wide)
        if hit_cap:
            self._hit_cap = int(hit_cap)
        self._force_wide = bool(force_wide)

    def _apply_frame_limits(self, hit_cap, force_wide):
        self._check(self.lib.mocap_set_frame_limits(self._h, int(hit_cap), int(bool(force_wide))))

    def set_stream(self, hip_stream_handle):
        self._check(self.lib.mocap_set_stream(self._h, ctypes.c_void_p(hip_stream_handle or 0)))

    def synchronize(self):
        self._check(self.lib.mocap_synchronize(self._h))

    def fundamental(self):
        F = np.zeros((self.C, self.C, 3, 3))
        self._check(self.lib.mocap_get_fundamental(self._h, _p(F)))
        return F

    # ------------------------------------------------------------------ host-buffer entry points
    def triangulate(self, obs):
        obs = np.ascontiguousarray(obs, dtype=np.float64).reshape(-1, self.C, 2)
        N = obs.shape[0]
        xyz = np.empty((N, 3))
        err = np.empty(N)
        self._check(self.lib.mocap_triangulate(self._h, N, _p(obs), _p(xyz), _p(err)))
        return xyz, err

    def reproject(self, obs, xyz):
        """Reprojection error of given points (calculate_reprojection_errors, helpers.py:203-241); NaN = < 2 views."""
        obs = np.ascontiguousarray(obs, dtype=np.float64).reshape(-1, self.C, 2)
        xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        assert xyz.shape[0] == obs.shape[0]
        err = np.empty(obs.shape[0])
        self._check(self.lib.mocap_reproject(self._h, obs.shape[0], _p(obs), _p(xyz), _p(err)))
        return err

    def match_triangulate(self, blobs, counts, gate_px=5.0, K_max=None, G_cap=1 << 20):
        blobs = np.ascontiguousarray(blobs, dtype=np.float32)
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        F, C, M, _ = blobs.shape
        assert C == self.C and counts.shape == (F, C)
        K_max = min(C * M, 64) if K_max is None else int(K_max)
        xyz = np.full((F, K_max, 3), np.nan)
        err = np.full((F, K_max), np.nan)
        corr = np.full((F, K_max, C), -1, dtype=np.int16)
        n_out = np.zeros(F, dtype=np.int32)
        status = np.zeros(F, dtype=np.int32)
        n_cand = np.zeros(F, dtype=np.int32)
        self._check(self.lib.mocap_match_triangulate(self._h, F, M, _p(blobs), _p(counts), float(gate_px), K_max,
                                                     int(G_cap), _p(xyz), _p(err), _p(corr), _p(n_out),
                                                     _p(status), _p(n_cand)))
        return {"xyz": xyz, "err": err, "corr": corr, "n_out": n_out, "status": status, "n_cand": n_cand}

    def match_triangulate_auto(self, blobs, counts, gate_px=5.0, K_max=None, G_cap=1 << 20):
        """mocap_match_triangulate_auto: frames whose caps overflowed are re-submitted ON THE GPU by the core itself with the
        worst-case root capacity (C*M), the largest candidate cap and (wide frames) an uncapped hit list -- any caller of
        the C ABI gets this, not only Python.  What is left to do here is grow the output arrays when a re-submitted frame
        needs more than K_max slots (the C entry reports how many in n_out)."""
        blobs = np.ascontiguousarray(blobs, dtype=np.float32)
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        F, C, M, _ = blobs.shape
        assert C == self.C and counts.shape == (F, C)
        K_max = min(C * M, 64) if K_max is None else int(K_max)

        def call(b, c, k):
            f = b.shape[0]
            out = {"xyz": np.full((f, k, 3), np.nan), "err": np.full((f, k), np.nan), "corr": np.full((f, k, C), -1, dtype=np.int16),
                   "n_out": np.zeros(f, dtype=np.int32), "status": np.zeros(f, dtype=np.int32), "n_cand": np.zeros(f, dtype=np.int32)}
            nres = ctypes.c_int32()
            self._check(self.lib.mocap_match_triangulate_auto(self._h, f, M, _p(b), _p(c), float(gate_px), k, int(G_cap),
                                                              _p(out["xyz"]), _p(out["err"]), _p(out["corr"]), _p(out["n_out"]),
                                                              _p(out["status"]), _p(out["n_cand"]), ctypes.addressof(nres)))
            out["resubmitted"] = nres.value
            return out

        res = call(blobs, counts, K_max)
        need = np.nonzero((res["status"] == ST_ROOT_OVERFLOW) & (res["n_out"] > K_max))[0]
        if need.size:
            grow = int(res["n_out"][need].max())
            big = call(blobs[need], counts[need], grow)
            for key, fill in (("xyz", np.nan), ("err", np.nan), ("corr", -1)):
                shape = list(res[key].shape)
                shape[1] = grow
                new = np.full(shape, fill, dtype=res[key].dtype)
                new[:, :K_max] = res[key]
                new[need] = big[key]
                res[key] = new
            for key in ("n_out", "status", "n_cand"):
                res[key][need] = big[key]
        return res

    def match_triangulate_f64(self, blobs, counts, gate_px=5.0, K_max=None, G_cap=1 << 20):
        """mocap_match_triangulate_f64: double centroids.  float32-representable coordinates are used exactly; others are
        rounded to the nearest float32 and the frame's status carries ST_ROUNDED (informational)."""
        blobs = np.ascontiguousarray(blobs, dtype=np.float64)
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        F, C, M, _ = blobs.shape
        assert C == self.C and counts.shape == (F, C)
        K_max = min(C * M, 64) if K_max is None else int(K_max)
        while True:
            out = {"xyz": np.full((F, K_max, 3), np.nan), "err": np.full((F, K_max), np.nan), "corr": np.full((F, K_max, C), -1, dtype=np.int16),
                   "n_out": np.zeros(F, dtype=np.int32), "status": np.zeros(F, dtype=np.int32), "n_cand": np.zeros(F, dtype=np.int32)}
            self._check(self.lib.mocap_match_triangulate_f64(self._h, F, M, _p(blobs), _p(counts), float(gate_px), K_max, int(G_cap),
                                                             _p(out["xyz"]), _p(out["err"]), _p(out["corr"]), _p(out["n_out"]),
                                                             _p(out["status"]), _p(out["n_cand"]), None))
            need = ((out["status"] & ~ST_ROUNDED) == ST_ROOT_OVERFLOW) & (out["n_out"] > K_max)
            if need.any():
                K_max = int(out["n_out"][need].max())
                continue
            return out

    # ------------------------------------------------------------------ the live loop in one call
    def _track_outputs(self, F, K_max, O_max):
        O = max(1, int(O_max))
        return {"xyz": np.full((F, K_max, 3), np.nan), "err": np.full((F, K_max), np.nan),
                "corr": np.full((F, K_max, self.C), -1, dtype=np.int16), "n_pts": np.zeros(F, dtype=np.int32),
                "status": np.zeros(F, dtype=np.int32), "pos": np.full((F, O, 3), np.nan), "heading": np.full((F, O), np.nan),
                "error": np.full((F, O), np.nan), "droneIndex": np.full((F, O), -1, dtype=np.int32),
                "n_obj": np.zeros(F, dtype=np.int32)}

    def track_frame(self, blobs, counts, gate_px=5.0, K_max=None, G_cap=1 << 20, O_max=8):
        """mocap_track_frame: match -> world coordinates -> locate_objects for one or a few frames in one call
        (helpers.py:94-133).  O_max = 0 switches the object search off (Cameras.is_locating_objects)."""
        blobs = np.ascontiguousarray(blobs, dtype=np.float32)
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        F, C, M, _ = blobs.shape
        assert C == self.C and counts.shape == (F, C)
        K_max = min(C * M, 64) if K_max is None else int(K_max)
        while True:
            o = self._track_outputs(F, K_max, O_max)
            self._check(self.lib.mocap_track_frame(self._h, F, M, _p(blobs), _p(counts), float(gate_px), K_max, int(G_cap),
                                                   _p(o["xyz"]), _p(o["err"]), _p(o["corr"]), _p(o["n_pts"]), _p(o["status"]),
                                                   int(O_max), _p(o["pos"]), _p(o["heading"]), _p(o["error"]),
                                                   _p(o["droneIndex"]), _p(o["n_obj"])))
            need = (o["status"] & ST_ROOT_OVERFLOW).astype(bool) & (o["n_pts"] > K_max)
            if need.any() and K_max < min(C * M, 256 if O_max else 1024):
                # the core re-ran those frames itself and says how many slots they need (n_pts)
                K_max = min(int(o["n_pts"][need].max()), 256 if O_max else 1024)
                continue
            return o

    def track_frame_images(self, images, M_max=16, gate_px=5.0, K_max=None, G_cap=1 << 20, O_max=8):
        """mocap_track_frame_images: raw camera frames [F][C][rows][cols][3] -> image points, object points, objects."""
        images = np.ascontiguousarray(images, dtype=np.uint8)
        F, C = images.shape[:2]
        assert images.shape == (F, C, self.img_rows, self.img_cols, 3) and C == self.img_C == self.C
        K_max = min(C * M_max, 64) if K_max is None else int(K_max)
        while True:
            o = self._track_outputs(F, K_max, O_max)
            o.update(blobs=np.zeros((F, C, M_max, 2), dtype=np.float32), counts=np.zeros((F, C), dtype=np.int32),
                     blob_status=np.zeros((F, C), dtype=np.int32))
            self._check(self.lib.mocap_track_frame_images(self._h, F, _p(images), int(M_max), float(gate_px), K_max, int(G_cap),
                                                          _p(o["blobs"]), _p(o["counts"]), _p(o["blob_status"]), _p(o["xyz"]),
                                                          _p(o["err"]), _p(o["corr"]), _p(o["n_pts"]), _p(o["status"]), int(O_max),
                                                          _p(o["pos"]), _p(o["heading"]), _p(o["error"]), _p(o["droneIndex"]),
                                                          _p(o["n_obj"])))
            need = (o["status"] & ST_ROOT_OVERFLOW).astype(bool) & (o["n_pts"] > K_max)
            if need.any() and K_max < min(C * M_max, 256 if O_max else 1024):
                K_max = min(int(o["n_pts"][need].max()), 256 if O_max else 1024)
                continue
            return o

    def track_frame_dev(self, n_frames, M_max, d_blobs, d_counts, gate_px, K_max, G_cap, d_xyz, d_err, d_corr, d_n_pts, d_status,
                        O_max=0, d_pos=0, d_heading=0, d_oerr=0, d_drone=0, d_n_obj=0):
        self._check(self.lib.mocap_track_frame_dev(
            self._h, int(n_frames), int(M_max), _vp(d_blobs), _vp(d_counts), float(gate_px), int(K_max), int(G_cap), _vp(d_xyz),
            _vp(d_err), _vp(d_corr), _vp(d_n_pts), _vp(d_status), int(O_max), _vp(d_pos or 0), _vp(d_heading or 0), _vp(d_oerr or 0),
            _vp(d_drone or 0), _vp(d_n_obj or 0)))

    # ------------------------------------------------------------------ before the path
    def set_image_params(self, rows, cols, K, dist, rotation=None):
        """Frame geometry + lens model of every camera (camera-params.json: intrinsic_matrix,
        distortion_coef, rotation) -> the undistortion maps of the blob-extraction stage."""
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(-1, 9)
        C = K.shape[0]
        dist = np.ascontiguousarray(dist, dtype=np.float64).reshape(C, 5)
        rot = None if rotation is None else np.ascontiguousarray(rotation, dtype=np.int32).reshape(C)
        self._check(self.lib.mocap_set_image_params(self._h, C, int(rows), int(cols), _p(K), _p(dist), _p(rot)))
        self.img_C, self.img_rows, self.img_cols = C, int(rows), int(cols)

    def set_blob_options(self, skip_dark_tiles=True):
        """True / 1: activity pre-pass + early-out; False / 0: every tile filtered; 2: early-out decided inside the mask pass."""
        self._check(self.lib.mocap_set_blob_options(self._h, 2 if skip_dark_tiles == 2 and skip_dark_tiles is not True else int(bool(skip_dark_tiles))))

    def undistort_map(self, camera=0):
        m = np.zeros((self.img_cols, self.img_cols), dtype=np.uint32)
        self._check(self.lib.mocap_get_undistort_map(self._h, int(camera), _p(m)))
        return m

    def find_blobs(self, images, M_max=16, want_processed=False):
        """images [F][C][rows][cols][3] uint8 RGB -> blobs f32 [F][C][M_max][2], counts, status
        (the frame path's input layout), optionally the processed BGR frames."""
        images = np.ascontiguousarray(images, dtype=np.uint8)
        F, C = images.shape[:2]
        assert images.shape == (F, C, self.img_rows, self.img_cols, 3) and C == self.img_C
        blobs = np.zeros((F, C, M_max, 2), dtype=np.float32)
        counts = np.zeros((F, C), dtype=np.int32)
        status = np.zeros((F, C), dtype=np.int32)
        ncont = np.zeros((F, C), dtype=np.int32)
        proc = np.zeros((F, C, self.img_cols, self.img_cols, 3), dtype=np.uint8) if want_processed else None
        self._check(self.lib.mocap_find_blobs(self._h, F, _p(images), int(M_max), _p(blobs), _p(counts), _p(status),
                                              _p(proc), _p(ncont)))
        out = {"blobs": blobs, "counts": counts, "status": status, "n_contours": ncont}
        if want_processed:
            out["processed"] = proc
        return out

    def find_blobs_dev(self, n_frames, d_images, M_max, d_blobs, d_counts, d_status, d_processed=0):
        self._check(self.lib.mocap_find_blobs_dev(self._h, int(n_frames), _vp(d_images), int(M_max), _vp(d_blobs),
                                                  _vp(d_counts), _vp(d_status), _vp(d_processed or 0)))

    # ------------------------------------------------------------------ after the path
    def set_world_transform(self, to_world):
        """4x4 to-world matrix (Cameras.to_world_coords_matrix) -> frame-path points leave the kernel in
        world coordinates (helpers.py:96-103); None switches the epilogue off."""
        if to_world is None:
            self._check(self.lib.mocap_set_world_transform(self._h, None))
        else:
            W = np.ascontiguousarray(to_world, dtype=np.float64).reshape(16)
            self._check(self.lib.mocap_set_world_transform(self._h, _p(W)))

    def locate_objects(self, xyz, err, n_pts, O_max=8):
        xyz = np.ascontiguousarray(xyz, dtype=np.float64)
        F, K_max, _ = xyz.shape
        err = np.ascontiguousarray(err, dtype=np.float64).reshape(F, K_max)
        n_pts = np.ascontiguousarray(n_pts, dtype=np.int32).reshape(F)
        out = {"pos": np.full((F, O_max, 3), np.nan), "heading": np.full((F, O_max), np.nan),
               "error": np.full((F, O_max), np.nan), "droneIndex": np.full((F, O_max), -1, dtype=np.int32),
               "lead": np.full((F, O_max), -1, dtype=np.int32), "n_obj": np.zeros(F, dtype=np.int32)}
        self._check(self.lib.mocap_locate_objects(self._h, F, K_max, _p(xyz), _p(err), _p(n_pts), int(O_max),
                                                  _p(out["pos"]), _p(out["heading"]), _p(out["error"]),
                                                  _p(out["droneIndex"]), _p(out["lead"]), _p(out["n_obj"])))
        return out

    # ------------------------------------------------------------------ device-pointer entry points
    def match_triangulate_dev(self, n_frames, M_max, d_blobs, d_counts, gate_px, K_max, G_cap, d_xyz, d_err,
                              d_corr, d_n_out, d_status, d_n_cand=0):
        """Raw device pointers (ints); enqueues on the context's stream and returns."""
        self._check(self.lib.mocap_match_triangulate_dev(
            self._h, int(n_frames), int(M_max), _vp(d_blobs), _vp(d_counts), float(gate_px), int(K_max),
            int(G_cap), _vp(d_xyz), _vp(d_err), _vp(d_corr), _vp(d_n_out), _vp(d_status), _vp(d_n_cand or 0)))

    def match_triangulate_dev_auto(self, n_frames, M_max, d_blobs, d_counts, gate_px, K_max, G_cap, d_xyz, d_err,
                                   d_corr, d_n_out, d_status, d_n_cand=0, d_resubmitted=0):
        """mocap_match_triangulate_dev_auto: the same, then the frames that hit a cap are re-run ON THE DEVICE with the largest
        caps and scattered back -- three more enqueues, no host wait.  d_resubmitted: 0, or a device-accessible int32[2]
        {frames flagged, frames re-run}."""
        self._check(self.lib.mocap_match_triangulate_dev_auto(
            self._h, int(n_frames), int(M_max), _vp(d_blobs), _vp(d_counts), float(gate_px), int(K_max),
            int(G_cap), _vp(d_xyz), _vp(d_err), _vp(d_corr), _vp(d_n_out), _vp(d_status), _vp(d_n_cand or 0),
            _vp(d_resubmitted or 0)))

    def resubmit_dev(self, n_frames, M_max, d_blobs, d_counts, gate_px, K_max, d_xyz, d_err, d_corr, d_n_out, d_status,
                     d_n_cand=0, d_resubmitted=0):
        """mocap_resubmit_dev: the re-submit stage alone, for a batch whose first pass has run (continues where a scratch batch
        smaller than the flagged set stopped: d_resubmitted[0] > d_resubmitted[1])."""
        self._check(self.lib.mocap_resubmit_dev(
            self._h, int(n_frames), int(M_max), _vp(d_blobs), _vp(d_counts), float(gate_px), int(K_max), _vp(d_xyz), _vp(d_err),
            _vp(d_corr), _vp(d_n_out), _vp(d_status), _vp(d_n_cand or 0), _vp(d_resubmitted or 0)))

    def compact_tracks_dev(self, n_frames, K_max, d_n_out, d_xyz, d_err, d_corr, d_offsets, d_records, capacity, d_total=0):
        """Valid points of a frame batch -> fixed-stride records + exclusive prefix of n_out (device pointers)."""
        self._check(self.lib.mocap_compact_tracks_dev(self._h, int(n_frames), int(K_max), _vp(d_n_out), _vp(d_xyz), _vp(d_err),
                                                      _vp(d_corr), _vp(d_offsets), _vp(d_records or 0), int(capacity),
                                                      _vp(d_total or 0)))

    def triangulate_dev(self, N, d_obs, d_xyz, d_err):
        self._check(self.lib.mocap_triangulate_dev(self._h, int(N), _vp(d_obs), _vp(d_xyz), _vp(d_err or 0)))

    # ------------------------------------------------------------------ initial poses
    def find_fundamental(self, p1, p2, threshold=1.0, confidence=0.99999, max_iters=1000):
        p1 = np.ascontiguousarray(p1, dtype=np.float32).reshape(-1, 2)
        p2 = np.ascontiguousarray(p2, dtype=np.float32).reshape(-1, 2)
        n = p1.shape[0]
        F = np.zeros(9)
        mask = np.zeros(n, dtype=np.uint8)
        info = np.zeros(3, dtype=np.int32)
        self._check(self.lib.mocap_find_fundamental(self._h, n, _p(p1), _p(p2), float(threshold), float(confidence),
                                                    int(max_iters), _p(F), _p(mask), _p(info)))
        return F.reshape(3, 3), mask, {"inliers": int(info[0]), "iterations": int(info[1]), "best_iteration": int(info[2])}

    def initial_poses(self, obs, K, threshold=1.0, confidence=0.99999, max_iters=1000):
        """obs (N, C, 2) NaN = unseen, K [C][3][3] -> (R [C][3][3], t [C][3], info [C-1][4])."""
        obs = np.ascontiguousarray(obs, dtype=np.float64)
        N, C, _ = obs.shape
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(C, 9)
        R = np.zeros((C, 3, 3))
        t = np.zeros((C, 3))
        info = np.zeros((max(C - 1, 1), 4), dtype=np.int32)
        self._check(self.lib.mocap_initial_poses(self._h, C, N, _p(obs), _p(K), float(threshold), float(confidence),
                                                 int(max_iters), _p(R), _p(t), _p(info)))
        return R, t, info

    # ------------------------------------------------------------------ bundle adjustment
    def n_params(self):
        return 1 + 7 * (self.C - 1)

    def ba_residuals(self, params, obs):
        params = np.ascontiguousarray(np.atleast_2d(params), dtype=np.float64)
        obs = np.ascontiguousarray(obs, dtype=np.float64).reshape(-1, self.C, 2)
        P, N = params.shape[0], obs.shape[0]
        assert params.shape[1] == self.n_params()
        r = np.empty((P, N))
        self._check(self.lib.mocap_ba_residuals(self._h, P, _p(params), N, _p(obs), _p(r)))
        return r

    def ba_normal_eq(self, x, obs, f32_residuals=False, use_cauchy=True, want_J=False):
        x = np.ascontiguousarray(x, dtype=np.float64)
        obs = np.ascontiguousarray(obs, dtype=np.float64).reshape(-1, self.C, 2)
        n, N = self.n_params(), obs.shape[0]
        JtJ = np.empty((n, n))
        Jtr = np.empty(n)
        cost = ctypes.c_double()
        m = ctypes.c_int64()
        J = np.zeros((N, n)) if want_J else None
        self._check(self.lib.mocap_ba_normal_eq(self._h, _p(x), N, _p(obs), int(f32_residuals), int(use_cauchy),
                                                _p(JtJ), _p(Jtr), ctypes.addressof(cost), _p(J),
                                                ctypes.addressof(m)))
        out = {"JtJ": JtJ, "Jtr": Jtr, "cost": cost.value, "m": m.value}
        if want_J:
            out["J"] = J[:m.value]
        return out

    def ba_trust_region_step(self, JtJ, Jtr, m, Delta, alpha=0.0, method=0):
        """The trust-region subproblem as mocap_ba_solve solves it -> (step, alpha, {"method", "live"})."""
        JtJ = np.ascontiguousarray(JtJ, dtype=np.float64)
        n = JtJ.shape[0]
        Jtr = np.ascontiguousarray(Jtr, dtype=np.float64).reshape(n)
        a = ctypes.c_double(float(alpha))
        step = np.zeros(n)
        info = np.zeros(2, dtype=np.int32)
        self._check(self.lib.mocap_ba_trust_region_step(self._h, n, int(m), _p(JtJ), _p(Jtr), float(Delta),
                                                        ctypes.addressof(a), int(method), _p(step), _p(info)))
        return step, a.value, {"method": int(info[0]), "live": int(info[1])}

    _BA_CB = ctypes.CFUNCTYPE(None, ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_void_p)

    def set_ba_progress(self, fn):
        """fn(x: ndarray) is called once per accepted step of ba_solve (None = off)."""
        if fn is None:
            self._ba_cb = None
            self._check(self.lib.mocap_set_ba_progress(self._h, None, None))
            return
        self._ba_cb = self._BA_CB(lambda px, n, _u: fn(np.ctypeslib.as_array(px, shape=(n,)).copy()))   # keep a reference
        self._check(self.lib.mocap_set_ba_progress(self._h, ctypes.cast(self._ba_cb, ctypes.c_void_p), None))

    def ba_profile(self, x, obs, f32_residuals=True, use_cauchy=True, reps=100):
        x = np.ascontiguousarray(x, dtype=np.float64)
        obs = np.ascontiguousarray(obs, dtype=np.float64).reshape(-1, self.C, 2)
        out = np.zeros(8)
        self._check(self.lib.mocap_ba_profile(self._h, _p(x), obs.shape[0], _p(obs), int(f32_residuals), int(use_cauchy),
                                              int(reps), _p(out)))
        keys = ("gpu_us_per_linearisation", "wall_us_per_linearisation", "host_tr_us", "launches", "m", "NP", "fused", "cost")
        return dict(zip(keys, out.tolist()))

    def ba_solve(self, x0, obs, ftol=1e-2, xtol=1e-8, gtol=1e-8, max_iter=0, f32_residuals=True,
                 use_cauchy=True):
        x = np.array(x0, dtype=np.float64)
        obs = np.ascontiguousarray(obs, dtype=np.float64).reshape(-1, self.C, 2)
        info = np.zeros(10)
        rc = self._check(self.lib.mocap_ba_solve_ex(self._h, _p(x), obs.shape[0], _p(obs), float(ftol), float(xtol),
                                                    float(gtol), int(max_iter), int(f32_residuals), int(use_cauchy),
                                                    _p(info), info.size), allow=(MOCAP_E_NOCONV,))
        keys = ("iterations", "nfev", "status", "cost0", "cost", "optimality", "m", "elapsed_ms", "njev", "relaunches")
        return x, dict(zip(keys, info.tolist()), converged=(rc == MOCAP_OK))
