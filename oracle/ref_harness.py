"""TEST INFRASTRUCTURE ONLY (oracle/) -- never imported by the product path.

Stub-import harness that runs the reference's OWN hot-path functions
(/root/reference/computer_code/api/helpers.py:203-421) unmodified inside this
container.  It exists only to (a) validate the restatement in
`oracle/mocap_oracle.py` / `oracle/c/` and (b) generate the golden vectors under
`tests/golden/` (see `oracle/make_golden.py`).  /root/reference does not exist on
the GPU box, so nothing that runs there imports this module.

`helpers.py` imports `cv2` and `pseyepy` at module top (helpers.py:3,12); both are
absent here.  We inject:
  * a fake `pseyepy.Camera` whose `.exposure` has one entry per camera, and
  * a fake `cv2` namespace whose `sfm.fundamentalFromProjections`,
    `computeCorrespondEpilines` and `projectPoints` are the NumPy restatements in
    `oracle/cv_restate.py`; `cv2.line` (used by the debug `drawlines`,
    helpers.py:497-504, called from inside the hot loop at helpers.py:365) is a no-op
    and `drawlines` itself is replaced by identity because it raises
    OverflowError on vertical epipolar lines.
"""
import importlib
import os
import sys
import types

import numpy as np

from . import cv_image_restate, cv_pose_restate, cv_restate

REFERENCE_API = "/root/reference/computer_code/api"


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_API, "helpers.py"))


class _FakeKalman:
    def __init__(self, *a, **k):
        pass


def _install_stubs(num_cameras):
    cv2 = types.ModuleType("cv2")
    sfm = types.SimpleNamespace(fundamentalFromProjections=cv_restate.fundamental_from_projections)
    cv2.sfm = sfm
    cv2.computeCorrespondEpilines = cv_restate.compute_correspond_epilines
    cv2.projectPoints = cv_restate.project_points
    cv2.line = lambda img, *a, **k: img
    cv2.KalmanFilter = _FakeKalman
    cv_image_restate.install(cv2)      # blob-extraction stage (helpers.py:68-88, 143-163), SURVEY 8f row 3
    cv_pose_restate.install(cv2)       # initial pose estimation (index.py:229-270), SURVEY 8f row 4
    sys.modules["cv2"] = cv2

    pseyepy = types.ModuleType("pseyepy")

    class Camera:
        RES_SMALL = 0

        def __init__(self, *a, **k):
            self.exposure = [100] * num_cameras
            self.gain = [10] * num_cameras

        def read(self):
            raise RuntimeError("no hardware in the oracle harness")

    pseyepy.Camera = Camera
    sys.modules["pseyepy"] = pseyepy


_helpers = None


def load_reference(num_cameras, intrinsics=None):
    """Import (once) the reference helpers module and size its Cameras singleton.

    intrinsics: optional list of C 3x3 matrices (lists); default = the reference's
    camera-params.json entry 0 (f=320, c=160) repeated."""
    global _helpers
    if not reference_available():
        raise RuntimeError("reference not present (expected only in the build container)")
    _install_stubs(num_cameras)
    if _helpers is None:
        if REFERENCE_API not in sys.path:
            sys.path.insert(0, REFERENCE_API)
        sys.dont_write_bytecode = True
        _helpers = importlib.import_module("helpers")
        _helpers.drawlines = lambda img, lines: img
    cams = _helpers.Cameras.instance()
    cams.num_cameras = num_cameras
    base = cams.camera_params[0]
    params = []
    for i in range(num_cameras):
        K = intrinsics[i] if intrinsics is not None else [[320.0, 0, 160], [0, 320, 160], [0, 0, 1]]
        params.append({
            "intrinsic_matrix": [list(r) for r in K],
            "distortion_coef": list(base["distortion_coef"]),
            "rotation": 0,
        })
    cams.camera_params = params
    return _helpers


def reference_find_dots(H, raw_frames, distortion=None, rotation=None):
    """The reference's own Cameras._camera_read preprocessing (helpers.py:68-82) followed by its
    Cameras._find_dot (helpers.py:143-163) on one set of raw camera frames (list of HxWx3 uint8, what
    pseyepy's Camera.read() hands over).  Returns (processed frames, image_points per camera) exactly as
    the reference produces them ([[None, None]] for an empty camera)."""
    cams = H.Cameras.instance()
    assert len(raw_frames) == cams.num_cameras
    for i in range(cams.num_cameras):
        if distortion is not None:
            cams.camera_params[i]["distortion_coef"] = list(distortion[i])
        if rotation is not None:
            cams.camera_params[i]["rotation"] = int(rotation[i])
    cams.cameras.read = lambda: ([np.array(f, dtype=np.uint8) for f in raw_frames], None)
    cams.is_capturing_points = False
    frames = cams._camera_read()
    out_frames, out_points = [], []
    for f in frames:
        img, pts = cams._find_dot(f)
        out_frames.append(img)
        out_points.append(pts)
    return out_frames, out_points


def reference_initial_poses(H, camera_points):
    """Runs the reference's own `calculate_camera_pose` handler (computer_code/api/index.py:229-281),
    extracted from index.py by its AST (index.py itself imports flask, serial, ruckig: absent here), with
    `bundle_adjustment` replaced by a probe that records the poses the handler hands to it.
    camera_points: the `cameraPoints` payload, (N, C, 2) nested lists with None for unseen.
    Returns the list of {"R", "t"} initial poses (index.py:234-270)."""
    import ast
    src = open(os.path.join(REFERENCE_API, "index.py")).read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "calculate_camera_pose")
    fn.decorator_list = []
    captured = {}

    def probe(image_points, camera_poses, socketio):
        captured["poses"] = [{"R": np.array(p["R"], dtype=np.float64), "t": np.array(p["t"], dtype=np.float64)}
                             for p in camera_poses]
        return camera_poses

    ns = {"np": np, "cv": sys.modules["cv2"], "Cameras": H.Cameras, "triangulate_points": H.triangulate_points,
          "calculate_reprojection_errors": H.calculate_reprojection_errors, "bundle_adjustment": probe,
          "camera_pose_to_serializable": H.camera_pose_to_serializable, "socketio": NullSocket()}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "index.py:calculate_camera_pose", "exec"), ns)
    ns["calculate_camera_pose"]({"cameraPoints": camera_points})
    return captured["poses"]


class NullSocket:
    """socketio stand-in for bundle_adjustment (helpers.py:274)."""

    def __init__(self):
        self.count = 0

    def emit(self, *a, **k):
        self.count += 1
