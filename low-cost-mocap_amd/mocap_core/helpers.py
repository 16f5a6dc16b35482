"""Host-side mirror of the reference's hot-path seam (computer_code/api/helpers.py:203-421).

Same function names, argument conventions, return shapes and error behaviour as the four
module-level functions `api/index.py:1` imports from `helpers`, so the Flask/socket.io
layer, the UI and the drone path are untouched (INTEGRATION.md shows the 6-line patch):

    find_point_correspondance_and_object_points(image_points, camera_poses, frames)   helpers.py:339
    triangulate_points(image_points, camera_poses)                                    helpers.py:330
    calculate_reprojection_errors(image_points, object_points, camera_poses)          helpers.py:203
    bundle_adjustment(image_points, camera_poses, socketio)                           helpers.py:244
  (+ the singular forms triangulate_point / calculate_reprojection_error)

Everything numeric runs in the HIP core through the C ABI (mocap_core.capi); this file only
converts between the reference's nested lists / None sentinels and the packed arrays of
include/mocap_core.h.  There is no CPU fallback: without the .so or a GPU these raise.

Intrinsics: the reference reads them from the `Cameras` singleton (helpers.py:215,295,340);
here `set_camera_params()` takes the same list of {"intrinsic_matrix": 3x3, ...} dicts
(the content of api/camera-params.json).
"""
import os
import threading
import time

import numpy as np

from . import capi

# Bundle adjustment through the reference's seam (helpers.bundle_adjustment, index.py:272) defaults to the mode that
# reproduces the reference's poses BIT FOR BIT: its own scipy.optimize.least_squares call, residuals evaluated on the GPU
# (seconds per calibration; the reference takes minutes).  "resident" -- the whole trust-region loop inside the core,
# milliseconds, the mode the BA iterations/s metric measures -- lands inside the reference's own run-to-run spread but
# not inside north_star's 1e-5 on rigs where the reference itself is not reproducible to 1e-5 (8 cameras: the reference
# moves 1.7e-3 under a 1e-15 nudge of its start vector, resident mode 2.3e-3 from it).  Opt in with
# set_bundle_adjustment_mode("resident").
DEFAULT_BA_MODE = "scipy"

_state = {
    "core": None,
    "ba_core": None,         # bundle adjustment's own context (own stream, own lock inside the library): a calibration never blocks the frame loop
    "ba_lock": threading.Lock(),   # one calibration at a time (the reference's handler is not re-entrant either)
    "camera_params": None,   # list of dicts like api/camera-params.json
    "cam_key": None,         # bytes of (K, R, t) currently uploaded
    "lock": threading.Lock(),
    "ba_mode": DEFAULT_BA_MODE,   # "scipy" (reference optimizer, GPU residuals) | "resident" (LM loop in the core)
    "ba_blas_threads": None,      # None: the process's BLAS settings are left alone (as the reference does); n: pinned for the solve
    "img_key": None,         # (rows, cols, K, dist, rot) of the lens model currently uploaded
    "to_world": None,        # last Cameras.to_world_coords_matrix handed to set_to_world_coords_matrix
}


def get_core(device_id=0):
    if _state["core"] is None:
        _state["core"] = capi.MocapCore(device_id)
    return _state["core"]


def _ba_core():
    """Bundle adjustment runs on its own context of the same GPU: the reference's frame loop (helpers.py:68-135, MJPEG
    thread) keeps running while calculate_camera_pose -> bundle_adjustment (index.py:229-277) runs in a socket handler
    thread; one context means one internal lock and one stream, i.e. a calibration of seconds in front of every frame."""
    if _state["ba_core"] is None:
        _state["ba_core"] = capi.MocapCore(get_core().device_id)
    return _state["ba_core"]


def set_core(core):
    """Use an existing MocapCore (e.g. one per GPU in a frame-sharded run)."""
    with _state["lock"]:
        if _state["ba_core"] is not None and (core is None or core.device_id != _state["ba_core"].device_id):
            _state["ba_core"] = None
        _state["core"] = core
        _state["cam_key"] = None      # cameras, lens model and world transform live in the context:
        _state["img_key"] = None      # re-upload them to the new one on first use
        if core is not None and _state["to_world"] is not None:
            core.set_world_transform(_state["to_world"])


def set_camera_params(camera_params):
    """camera_params: list of {"intrinsic_matrix": 3x3 list, ...} (api/camera-params.json)."""
    _state["camera_params"] = [dict(p) for p in camera_params]
    _state["cam_key"] = None
    _state["img_key"] = None


def set_bundle_adjustment_blas_threads(n):
    """Mode "scipy" spends its host time in SciPy's SVD of the m x n Jacobian.  On a many-core host OpenBLAS's default thread
    count makes that SVD slower, not faster (1 000 points x 50 parameters: 1.1 s per calibration with 64 spinning threads on a
    shared 256-thread box, 0.23 s with one -- bench.py ba.default_mode.one_blas_thread), and its last bits depend on the thread
    count either way.  n = 1 pins the BLAS for the duration of a solve (threadpoolctl); None (default) leaves the process's
    settings alone, like the reference."""
    _state["ba_blas_threads"] = None if n is None else int(n)


def set_bundle_adjustment_mode(mode):
    assert mode in ("resident", "scipy")
    _state["ba_mode"] = mode


class bundle_adjustment_mode:
    """with helpers.bundle_adjustment_mode("resident"): ...  -- the previous mode comes back whatever happens inside."""

    def __init__(self, mode):
        assert mode in ("resident", "scipy")
        self.mode = mode

    def __enter__(self):
        self.prev = _state["ba_mode"]
        _state["ba_mode"] = self.mode
        return self

    def __exit__(self, *exc):
        _state["ba_mode"] = self.prev
        return False


def _intrinsics(C):
    params = _state["camera_params"]
    if params is None:
        raise RuntimeError("set_camera_params() has not been called (the reference loads camera-params.json)")
    if len(params) < C:
        raise IndexError("fewer camera_params entries than camera poses")  # the reference raises IndexError
    return np.array([np.array(params[i]["intrinsic_matrix"], dtype=np.float64) for i in range(C)])


def _pose_arrays(camera_poses):
    R = np.array([np.array(p["R"], dtype=np.float64).reshape(3, 3) for p in camera_poses])
    t = np.array([np.array(p["t"], dtype=np.float64).reshape(3) for p in camera_poses])
    return R, t


def _upload_cameras(camera_poses):
    """mocap_set_cameras only when (K, R, t) changed -- replaces the per-call rebuild of P = K[R|t]
    (helpers.py:305-308, :351-355) and the per-root fundamentalFromProjections (helpers.py:362)."""
    core = get_core()
    R, t = _pose_arrays(camera_poses)
    K = _intrinsics(len(camera_poses))
    key = K.tobytes() + R.tobytes() + t.tobytes()
    if key != _state["cam_key"]:
        core.set_cameras(K, R, t)
        _state["cam_key"] = key
    return core


def _obs_array(image_points, C):
    """(N, C, 2) list / object ndarray with None -> float64 with NaN."""
    arr = np.asarray(image_points, dtype=object).reshape(-1, C, 2)
    out = np.full(arr.shape, np.nan)
    mask = np.vectorize(lambda v: v is not None)(arr) if arr.size else np.zeros(arr.shape, bool)
    out[mask] = arr[mask].astype(np.float64)
    bad = ~(mask[..., 0] & mask[..., 1])
    out[bad] = np.nan
    return out


# ----------------------------------------------------------------------------- triangulation
def triangulate_points(image_points, camera_poses):
    """helpers.py:330-336.  Rows with fewer than two views are [None, None, None]."""
    C = len(camera_poses)
    with _state["lock"]:
        core = _upload_cameras(camera_poses)
        obs = _obs_array(image_points, C)
        if obs.shape[0] == 0:
            return np.array([])
        xyz, _ = core.triangulate(obs)
    missing = np.isnan(xyz[:, 0])
    if not missing.any():
        return xyz
    out = xyz.astype(object)
    out[missing] = None
    return out


def triangulate_point(image_points, camera_poses):
    """helpers.py:293-327."""
    res = triangulate_points([image_points], camera_poses)
    row = res[0]
    return [None, None, None] if row[0] is None else np.asarray(row, dtype=np.float64)


def calculate_reprojection_errors(image_points, object_points, camera_poses):
    """helpers.py:203-211: entries with fewer than two views are skipped (the result may be shorter).
    The errors are those of the object points PASSED IN (mocap_reproject), as the function's contract says; the
    reference's three call sites happen to pass the triangulation of the same observations."""
    C = len(camera_poses)
    with _state["lock"]:
        core = _upload_cameras(camera_poses)
        obs = _obs_array(image_points, C)
        if obs.shape[0] == 0:
            return np.array([])
        op = np.asarray(object_points, dtype=object).reshape(obs.shape[0], 3)
        xyz = np.full(op.shape, np.nan)
        ok = np.vectorize(lambda v: v is not None)(op).all(axis=1)
        xyz[ok] = op[ok].astype(np.float64)
        err = core.reproject(obs, xyz)
    return err[~np.isnan(err)]


def calculate_reprojection_error(image_points, object_point, camera_poses):
    """helpers.py:214-241: None when fewer than two cameras see the point."""
    e = calculate_reprojection_errors([image_points], [object_point], camera_poses)
    return None if e.size == 0 else e[0]


# ----------------------------------------------------------------------------- frame path
def pack_frame(image_points, M_max=None, strict=False):
    """Nested lists of one frame -> (blobs f32 [1][C][M][2], counts i32 [1][C], rounded: bool).

    The C ABI's frame path carries blob coordinates as float32.  The reference measures point-line distances on whatever
    image_points holds (helpers.py:367-373: int64 for _find_dot's int() centroids, float64 for floats).  Integer centroids
    (the reference's own, |x| < 2^24) and float32-valued sub-pixel centroids are carried exactly.  Any other float64
    coordinate is rounded to the nearest float32 (2e-5 px at 320 px; `rounded` says so) -- or refused with strict=True.
    NaN / infinite coordinates are always refused."""
    C = len(image_points)
    n = [len(p) for p in image_points]
    M = max(1, max(n) if n else 1) if M_max is None else M_max
    blobs = np.full((1, C, M, 2), np.nan, dtype=np.float32)
    counts = np.zeros((1, C), dtype=np.int32)
    rounded = False
    for c, pts in enumerate(image_points):
        if pts:
            exact = np.asarray(pts, dtype=np.float64)
            if not np.isfinite(exact).all():
                raise ValueError(f"image point of camera {c} is NaN or infinite")
            as_f32 = exact.astype(np.float32)
            if not np.array_equal(as_f32.astype(np.float64), exact):
                if strict:
                    bad = exact[as_f32.astype(np.float64) != exact][0]
                    raise ValueError(f"image point coordinate {bad!r} (camera {c}) is not representable in float32: the core's "
                                     "blob arrays are float32 (include/mocap_core.h)")
                rounded = True
            blobs[0, c, :len(pts)] = as_f32
        counts[0, c] = len(pts)
    return blobs, counts, rounded


def find_point_correspondance_and_object_points(image_points, camera_poses, frames):
    """helpers.py:339-421.  `image_points` is mutated like the reference does (the [None, None]
    sentinel of an empty camera is removed, helpers.py:342-346).  `frames` is returned untouched:
    the reference only draws debug epipolar lines into it (helpers.py:365)."""
    for image_points_i in image_points:
        try:
            image_points_i.remove([None, None])
        except Exception:
            pass
    with _state["lock"]:
        core = _upload_cameras(camera_poses)
        blobs, counts, _ = pack_frame(image_points)
        res = core.match_triangulate_auto(blobs, counts, gate_px=5.0)
    if int(res["status"][0]) != 0:
        # still over a cap after the worst-case re-submit (> 2^24 candidate groups for one root, > 2^32 per
        # frame, C*M > 1024 roots): the reference would enumerate the full product; an empty answer would be
        # silently wrong
        raise capi.MocapError(_status_message(int(res["status"][0])))
    k = int(res["n_out"][0])
    if k == 0:
        return np.array([]), np.array([]), frames
    return res["err"][0, :k].copy(), res["xyz"][0, :k].copy(), frames


def find_point_correspondance_and_object_points_batch(blobs, counts, camera_poses, gate_px=5.0, K_max=None):
    """Batch form on the packed layout (many frames per call); also returns the correspondence indices."""
    with _state["lock"]:
        core = _upload_cameras(camera_poses)
        return core.match_triangulate_auto(blobs, counts, gate_px=gate_px, K_max=K_max)


# ----------------------------------------------------------------------------- after the path
# ----------------------------------------------------------------------------- before the path
def camera_read_find_dots(raw_frames, M_max=64, want_frames=True):
    """The per-camera body of Cameras._camera_read (helpers.py:71-82) + Cameras._find_dot
    (helpers.py:143-163) for one set of raw frames (what pseyepy's Camera.read() returns):
    returns (frames, image_points) with frames = the processed BGR frames the reference streams
    (without its debug drawings) and image_points = per camera [[x, y], ...] or [[None, None]]
    (helpers.py:158-159) -- ready for find_point_correspondance_and_object_points.
    Lens model and rotation come from set_camera_params() (camera-params.json entries)."""
    raw = np.ascontiguousarray(np.asarray(raw_frames, dtype=np.uint8))
    C, rows, cols = raw.shape[0], raw.shape[1], raw.shape[2]
    params = _state["camera_params"]
    if params is None or len(params) < C:
        raise RuntimeError("set_camera_params() has not been called with one entry per camera")
    K = np.array([np.array(params[i]["intrinsic_matrix"], dtype=np.float64) for i in range(C)])
    dist = np.array([np.array(params[i]["distortion_coef"], dtype=np.float64).ravel()[:5] for i in range(C)])
    rot = np.array([int(params[i].get("rotation", 0)) for i in range(C)], dtype=np.int32)
    with _state["lock"]:
        core = get_core()
        key = (rows, cols, K.tobytes(), dist.tobytes(), rot.tobytes())
        if _state["img_key"] != key:
            core.set_image_params(rows, cols, K, dist, rot)
            _state["img_key"] = key
        res = core.find_blobs(raw[None], M_max=M_max, want_processed=want_frames)
        if (res["status"] & capi.BLOB_ST_POINT_OVERFLOW).any():     # more dots than slots: ask again
            res = core.find_blobs(raw[None], M_max=int(res["n_contours"].max()) + 1, want_processed=want_frames)
    if (res["status"] & capi.BLOB_ST_CAP_OVERFLOW).any():
        cams = np.nonzero(res["status"][0] & capi.BLOB_ST_CAP_OVERFLOW)[0].tolist()
        raise capi.MocapError(f"camera(s) {cams}: more contours than the blob stage's largest tables hold "
                              "(BLOB_ST_CAP_OVERFLOW); no centroids were produced for them")
    image_points = []
    for c in range(C):
        n = int(res["counts"][0, c])
        pts = res["blobs"][0, c, :n].astype(np.int64).tolist()
        image_points.append(pts if n else [[None, None]])
    frames = [f for f in res["processed"][0]] if want_frames else [None] * C
    return frames, image_points


def set_to_world_coords_matrix(to_world_coords_matrix):
    """Cameras.to_world_coords_matrix (helpers.py:40,100): with a matrix set, the frame path returns
    world coordinates -- the loop at helpers.py:96-103 runs fused in the kernel's store.  None = off
    (camera-0 coordinates, exactly what find_point_correspondance_and_object_points returns upstream)."""
    with _state["lock"]:
        _state["to_world"] = None if to_world_coords_matrix is None else np.array(to_world_coords_matrix, dtype=np.float64)
        get_core().set_world_transform(to_world_coords_matrix)


def locate_objects(object_points, errors):
    """helpers.py:424-480: list of {"pos", "heading", "error", "droneIndex"} for one frame."""
    P = np.asarray(object_points, dtype=np.float64).reshape(-1, 3)
    if P.shape[0] == 0:
        return []
    E = np.asarray(errors, dtype=np.float64).reshape(-1)
    with _state["lock"]:
        res = get_core().locate_objects(P[None], E[None], [P.shape[0]], O_max=max(1, P.shape[0]))
    return [{"pos": res["pos"][0, j].copy(), "heading": float(res["heading"][0, j]),
             "error": float(res["error"][0, j]), "droneIndex": int(res["droneIndex"][0, j])}
            for j in range(int(res["n_obj"][0]))]


def _objects_list(res, f=0):
    return [{"pos": res["pos"][f, j].copy(), "heading": float(res["heading"][f, j]),
             "error": float(res["error"][f, j]), "droneIndex": int(res["droneIndex"][f, j])}
            for j in range(min(int(res["n_obj"][f]), res["pos"].shape[1]))]


def track_frame(image_points, camera_poses, is_locating_objects=True, O_max=8):
    """The body of the reference's live loop after _find_dot in ONE core call (helpers.py:94-108):
    find_point_correspondance_and_object_points -> world coordinates (needs set_to_world_coords_matrix; without it the
    points stay in camera-0 coordinates) -> locate_objects.  Returns (errors, object_points, objects) -- exactly the
    three values the loop holds at helpers.py:109 before the Kalman filter -- ready for object_points_payload().
    `image_points` is mutated like the reference does (helpers.py:342-346)."""
    for image_points_i in image_points:
        try:
            image_points_i.remove([None, None])
        except Exception:
            pass
    with _state["lock"]:
        core = _upload_cameras(camera_poses)
        blobs, counts, _ = pack_frame(image_points)
        res = core.track_frame(blobs, counts, gate_px=5.0, O_max=O_max if is_locating_objects else 0)
    if int(res["status"][0]) != 0:
        raise capi.MocapError(_status_message(int(res["status"][0])))
    k = int(res["n_pts"][0])
    if k == 0:
        return np.array([]), np.array([]), []
    return res["err"][0, :k].copy(), res["xyz"][0, :k].copy(), (_objects_list(res) if is_locating_objects else [])


def camera_read_track(raw_frames, camera_poses, M_max=16, is_locating_objects=True, O_max=8):
    """Raw camera frames -> (image_points, errors, object_points, objects): Cameras._camera_read's preprocessing,
    _find_dot, the frame path, the world transform and locate_objects (helpers.py:68-108) in one core call; nothing but the
    payload crosses PCIe on the way back.  image_points is what _find_dot returns per camera ([[None, None]] when a
    camera saw nothing)."""
    raw = np.ascontiguousarray(np.asarray(raw_frames, dtype=np.uint8))
    C, rows, cols = raw.shape[0], raw.shape[1], raw.shape[2]
    params = _state["camera_params"]
    if params is None or len(params) < C:
        raise RuntimeError("set_camera_params() has not been called with one entry per camera")
    K = np.array([np.array(params[i]["intrinsic_matrix"], dtype=np.float64) for i in range(C)])
    dist = np.array([np.array(params[i]["distortion_coef"], dtype=np.float64).ravel()[:5] for i in range(C)])
    rot = np.array([int(params[i].get("rotation", 0)) for i in range(C)], dtype=np.int32)
    with _state["lock"]:
        core = _upload_cameras(camera_poses)
        key = (rows, cols, K.tobytes(), dist.tobytes(), rot.tobytes())
        if _state["img_key"] != key:
            core.set_image_params(rows, cols, K, dist, rot)
            _state["img_key"] = key
        while True:
            res = core.track_frame_images(raw[None], M_max=M_max, O_max=O_max if is_locating_objects else 0)
            if (res["blob_status"] & capi.BLOB_ST_POINT_OVERFLOW).any() and M_max < 256:   # more dots than slots: ask again
                M_max = min(256, 4 * M_max)
                continue
            break
    if (res["blob_status"] & capi.BLOB_ST_CAP_OVERFLOW).any():
        raise capi.MocapError("more contours than the blob stage's largest tables hold (BLOB_ST_CAP_OVERFLOW)")
    if int(res["status"][0]) != 0:
        raise capi.MocapError(f"frame exceeds the core's limits (status {int(res['status'][0])})")
    image_points = []
    for c in range(C):
        n = int(res["counts"][0, c])
        image_points.append(res["blobs"][0, c, :n].astype(np.int64).tolist() if n else [[None, None]])
    k = int(res["n_pts"][0])
    if k == 0:
        return image_points, np.array([]), np.array([]), []
    return (image_points, res["err"][0, :k].copy(), res["xyz"][0, :k].copy(),
            _objects_list(res) if is_locating_objects else [])


def object_points_payload(errors, object_points, objects, filtered_objects=()):
    """The dict the reference emits as the `object-points` socket event (helpers.py:128-133), built from what
    track_frame() / camera_read_track() return.  `filtered_objects` is the caller's Kalman output
    (helpers.py:109-126: already .tolist()-ed there), passed through."""
    return {
        "object_points": np.asarray(object_points).tolist(),
        "errors": np.asarray(errors).tolist(),
        "objects": [{k: (v.tolist() if isinstance(v, np.ndarray) else v) for (k, v) in obj.items()} for obj in objects],
        "filtered_objects": list(filtered_objects),
    }


def _status_message(st):
    """What a frame's status word says once the core's worst-case re-submit has run (include/mocap_core.h MOCAP_ST_*)."""
    if st & capi.ST_INTRACTABLE:
        lg = (st >> capi.ST_LOG2_GROUPS_SHIFT) & capi.ST_LOG2_GROUPS_MASK
        return (f"frame has a root with about 2^{lg} candidate groups (status {st}): the reference would enumerate all of them "
                "(helpers.py:394-400) and never return; no enumeration reaches it and the exact search could not bound it")
    return (f"frame exceeds the core's limits (status {st}): candidate groups / roots over the caps of include/mocap_core.h")


# ----------------------------------------------------------------------------- initial poses (caller of BA)
def initial_camera_poses(image_points):
    """The pose-chaining loop of the `calculate-camera-pose` handler (index.py:234-270): `image_points` is
    the handler's np.array(data["cameraPoints"]) -- (N, C, 2) with None for unseen -- and the result the
    list of {"R": 3x3, "t": (3, 1)} it hands to bundle_adjustment (camera 0 = identity)."""
    arr = np.asarray(image_points, dtype=object)
    C = arr.shape[1]
    obs = _obs_array(arr, C)
    K = _intrinsics(C)
    with _state["lock"]:
        R, t, _ = get_core().initial_poses(obs, K)
    return [{"R": R[i], "t": t[i].reshape(3, 1)} for i in range(C)]


def calculate_camera_pose(data, socketio):
    """index.py:229-281 end to end: initial poses, bundle_adjustment, final error, the `camera-pose` event."""
    image_points = np.array(data["cameraPoints"], dtype=object)
    camera_poses = initial_camera_poses(image_points)
    camera_poses = bundle_adjustment(image_points, camera_poses, socketio)
    object_points = triangulate_points(image_points, camera_poses)
    error = float(np.mean(calculate_reprojection_errors(image_points, object_points, camera_poses)))
    socketio.emit("camera-pose", {"camera_poses": camera_pose_to_serializable(camera_poses)})
    return camera_poses, error


# ----------------------------------------------------------------------------- bundle adjustment
def _ba_x0(camera_poses):
    """helpers.py:278-285 (including its focal-length indexing: entry i+1 takes camera i's focal)."""
    from scipy.spatial.transform import Rotation
    K = _intrinsics(len(camera_poses))
    x0 = [K[0][0, 0]]
    for i, pose in enumerate(camera_poses[1:]):
        rot_vec = Rotation.from_matrix(np.asarray(pose["R"], dtype=np.float64)).as_rotvec().flatten()
        x0 += [K[i][0, 0]] + rot_vec.tolist() + np.asarray(pose["t"], dtype=np.float64).flatten().tolist()
    return np.array(x0, dtype=np.float64)


def _params_to_camera_poses(params):
    """helpers.py:247-262."""
    from scipy.spatial.transform import Rotation
    C = int((params.size - 1) / 7) + 1
    poses = [{"R": np.eye(3), "t": np.array([0, 0, 0], dtype=np.float32)}]
    for i in range(C - 1):
        poses.append({"R": Rotation.from_rotvec(params[i * 7 + 2:i * 7 + 5]).as_matrix(),
                      "t": params[i * 7 + 5:i * 7 + 8]})
    return poses


def camera_pose_to_serializable(camera_poses):
    """helpers.py:526-530."""
    return [{k: np.asarray(v).tolist() for k, v in p.items()} for p in camera_poses]


def bundle_adjustment(image_points, camera_poses, socketio, return_info=False):
    """helpers.py:244-290: least_squares(residual_function, x0, loss="cauchy", ftol=1e-2).

    mode "scipy" (default): the reference's optimizer call verbatim, only the residual evaluations are GPU -- poses
    bit-identical to the reference's on the solver goldens.
    mode "resident": the whole trust-region loop runs in the core (mocap_ba_solve), milliseconds instead of seconds;
    inside the reference's own reproducibility, see DEFAULT_BA_MODE above.
    `socketio.emit("camera-pose", ...)`: the reference streams one per residual evaluation (helpers.py:274; the UI
    animates the cameras while calibrating, App.tsx:255).  Mode "scipy" does exactly that; mode "resident" has no
    per-evaluation host round trip and emits once per accepted step (1 / (n + 2) as often), then the final poses."""
    C = len(camera_poses)
    x0 = _ba_x0(camera_poses)
    obs = _obs_array(image_points, C)
    R, t = _pose_arrays(camera_poses)
    K = _intrinsics(C)

    def emit(params):
        if socketio is not None:
            socketio.emit("camera-pose", {"camera_poses": camera_pose_to_serializable(_params_to_camera_poses(params))})

    # The solve runs on bundle adjustment's OWN context: the module lock -- the one every frame call takes -- is held only to
    # fetch that context, never across a residual evaluation, let alone the whole solve.
    with _state["lock"]:
        core = _ba_core()
        mode = _state["ba_mode"]
        f32_rounding = get_core().f32_rounding      # options live in a context: the calibration's follows the frame path's
    with _state["ba_lock"]:
        if core.f32_rounding != f32_rounding:
            core.set_options(f32_rounding=f32_rounding)
        core.set_cameras(K, R, t)
        if mode == "resident":
            core.set_ba_progress(emit if socketio is not None else None)
            try:
                x, info = core.ba_solve(x0, obs, ftol=1e-2, f32_residuals=True, use_cauchy=True)
            finally:
                core.set_ba_progress(None)
        else:
            from scipy import optimize
            # SciPy's own step rule for jac='2-point' (a PRIVATE helper: its 4-argument form exists in SciPy 1.5 .. 1.15; checked
            # here, not assumed).  Anything else -- import error, another signature, a step that is not what approx_derivative
            # would take -- falls back to jac='2-point' (the reference's call verbatim: n + 1 separate residual evaluations)
            _compute_absolute_step = None
            try:
                import inspect
                from scipy.optimize._numdiff import _compute_absolute_step as _cas
                if list(inspect.signature(_cas).parameters)[:4] == ["rel_step", "x0", "f0", "method"]:
                    probe = _cas(None, np.array([1.0, -2.0, 0.0]), np.zeros(2, dtype=np.float32), "2-point")
                    eps32 = float(np.finfo(np.float32).eps) ** 0.5
                    if np.allclose(probe, [eps32, -2.0 * eps32, eps32], rtol=1e-12, atol=0.0):
                        _compute_absolute_step = _cas
            except Exception:
                _compute_absolute_step = None

            last = {"x": None, "r": None}
            spent = {"core_s": 0.0}                             # wall time inside the core's calls (the rest is SciPy's own)

            def residual_function(params):
                t0 = time.perf_counter()
                r = core.ba_residuals(params, obs)[0]
                spent["core_s"] += time.perf_counter() - t0
                emit(params)                                    # helpers.py:274
                r = r[~np.isnan(r)].astype(np.float32)          # helpers.py:273
                last["x"], last["r"] = np.array(params, dtype=np.float64), r
                return r

            def jacobian(params):
                """The Jacobian least_squares would difference itself (jac='2-point': scipy _numdiff.approx_derivative ->
                _dense_difference), with its n residual evaluations made by ONE call of the core: the n perturbed parameter
                vectors x + h_i e_i go through mocap_ba_residuals as a batch.  Same step vector (SciPy's own
                _compute_absolute_step: float32 residuals -> sqrt(eps_float32) relative step), same residual bits (the batch
                is the single call's kernel per parameter vector), same NumPy expressions for `df / dx` (float32
                difference, float64 quotient) => the same J bit for bit, hence the same iterates, nfev, njev and poses as
                the reference's call on the solver goldens (tests/test_gpu_ba.py).  The reference's residual_function
                emits the poses at every one of those evaluations (helpers.py:274): so does this."""
                x0_ = np.array(params, dtype=np.float64)
                if last["x"] is not None and np.array_equal(last["x"], x0_):
                    f0 = last["r"]                              # trf hands approx_derivative the f it just evaluated at x
                    X = np.repeat(x0_[None, :], x0_.size, axis=0)
                    first = 0
                else:                                           # (not reached from trf: f0 rides along in the batch)
                    X = np.repeat(x0_[None, :], x0_.size + 1, axis=0)
                    first = 1
                    f0 = None
                h = _compute_absolute_step(None, x0_, last["r"] if f0 is None else f0, "2-point")
                idx = np.arange(x0_.size)
                X[first + idx, idx] = x0_ + h                   # x1[i] += h[i]
                dx = X[first + idx, idx] - x0_                  # "recompute dx as exactly representable number"
                t0 = time.perf_counter()
                R = core.ba_residuals(X, obs)
                spent["core_s"] += time.perf_counter() - t0
                if f0 is None:
                    f0 = R[0][~np.isnan(R[0])].astype(np.float32)
                J_T = np.empty((x0_.size, f0.size))
                for i in range(x0_.size):
                    emit(X[first + i])
                    ri = R[first + i]
                    df = ri[~np.isnan(ri)].astype(np.float32) - f0   # (a mask that moved raises here, as it would in the reference)
                    J_T[i] = df / dx[i]
                return J_T.T

            use_batched = os.environ.get("MOCAP_BA_BATCHED_JAC", "1") != "0" and _compute_absolute_step is not None
            import contextlib
            blas = contextlib.nullcontext()
            if _state["ba_blas_threads"] is not None:
                try:
                    from threadpoolctl import threadpool_limits
                    blas = threadpool_limits(limits=_state["ba_blas_threads"])
                except Exception:
                    pass
            with blas:
                res = optimize.least_squares(residual_function, x0, jac=jacobian if use_batched else "2-point", verbose=0,
                                             loss="cauchy", ftol=1e-2)
            x, info = res.x, {"iterations": res.njev, "njev": res.njev, "nfev": res.nfev, "status": res.status,
                       "cost": res.cost, "optimality": res.optimality, "core_s": spent["core_s"]}
    poses = _params_to_camera_poses(x)
    if socketio is not None:
        socketio.emit("camera-pose", {"camera_poses": camera_pose_to_serializable(poses)})
    return (poses, info) if return_info else poses
