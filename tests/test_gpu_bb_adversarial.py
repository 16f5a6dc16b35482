"""GPU: the exact branch-and-bound selection (csrc/frame_bb.hip) off the ring rig it was developed on.

Every case runs with the search FORCED ON (MOCAP_BB_MIN_G=0: bound tests on every frame, however few candidates) and is
compared (a) bit for bit with the exhaustive walk (MOCAP_EVAL_BB=0, csrc/frame_kernel.hip: every candidate group of the
Cartesian product triangulated and reprojected, helpers.py:408-421) -- indices, points, errors, counts, status -- and
(b) with the C oracle (indices exact, points to 1e-9).  The bound's allowances scale with the units of the rig and the
size of the pixel coordinates, its centre comes from the optical axes, its ties are broken across blocks and waves:
each of those is attacked below.  A self-check build (-DMOCAP_DEBUG_EIGCHECK) re-evaluates on the device everything
the search cut or dropped and must report nothing.
"""
import os

import numpy as np
import pytest

from mocap_core import capi, synth

pytestmark = pytest.mark.gpu

XYZ_RTOL = 1e-9


def _ctx(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return capi.MocapCore(0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.fixture(scope="module")
def searchers():
    import torch  # noqa: F401  (load order, see conftest.core)
    s = {"forced": _ctx({"MOCAP_BB_MIN_G": "0"}), "forced_small_blocks": _ctx({"MOCAP_BB_MIN_G": "0", "MOCAP_BB_PL": "3"}),
         "default": _ctx({}), "exhaustive": _ctx({"MOCAP_EVAL_BB": "0"})}
    yield s
    for c in s.values():
        c.close()


def _rig(R, t, K, image_size, centre):
    R, t = np.asarray(R, dtype=np.float64), np.asarray(t, dtype=np.float64)
    return {"K": np.repeat(np.asarray(K, dtype=np.float64)[None], len(R), axis=0), "R": R, "t": t,
            "image_size": image_size, "centre": np.asarray(centre, dtype=np.float64), "R0": np.eye(3)}


def _check(searchers, rig, blobs, counts, K_max=None, gate=5.0, oracle_frames=40, min_cand=None):
    from oracle import c_oracle
    out = {}
    for name, c in searchers.items():
        c.set_cameras(rig["K"], rig["R"], rig["t"])
        out[name] = c.match_triangulate(blobs, counts, gate_px=gate, K_max=K_max)
        kern = c.last_frame_kernel()
        assert kern.startswith("frame_kernel" if name == "exhaustive" else "frame_bb_kernel"), (name, kern)
    base = out["exhaustive"]
    assert not base["status"].any()
    K = base["err"].shape[1]
    valid = np.arange(K)[None, :] < base["n_out"][:, None]
    for name in ("forced", "forced_small_blocks", "default"):
        res = out[name]
        for key in ("n_out", "status", "n_cand"):
            assert np.array_equal(res[key], base[key]), (name, key)
        assert np.array_equal(res["corr"][valid], base["corr"][valid]), (name, "corr")
        for key in ("xyz", "err"):
            assert np.array_equal(res[key][valid], base[key][valid], equal_nan=True), (name, key)
    if min_cand is not None:
        assert base["n_cand"].mean() >= min_cand, base["n_cand"].mean()   # the case really exercises the search
    n = min(oracle_frames, blobs.shape[0])
    ref = c_oracle.COracle(rig["K"], rig["R"], rig["t"]).match_triangulate(blobs[:n], counts[:n], gate_px=gate, K_max=K)
    assert np.array_equal(ref["n_out"], base["n_out"][:n])
    vv = valid[:n]
    assert np.array_equal(ref["corr"][vv], base["corr"][:n][vv])
    np.testing.assert_allclose(base["xyz"][:n][vv], ref["xyz"][vv], rtol=XYZ_RTOL, atol=1e-12 * max(1.0, np.abs(ref["xyz"][vv]).max()))
    return base


def _app_tsx_rig(scale=1.0):
    from oracle.make_golden import APP_TSX_POSES, rig_from_poses
    rig = rig_from_poses(APP_TSX_POSES, [synth.DEFAULT_K] * 4)
    rig["t"] = rig["t"] * scale
    rig["centre"] = rig["centre"] * scale
    return rig


def test_app_tsx_rig_16_blobs(searchers):
    """The really calibrated 4-camera rig of the reference UI (App.tsx:44), 16 blobs per camera."""
    rig = _app_tsx_rig()
    blobs, counts, _ = synth.make_blob_stream(rig, 500, 16, seed=301)
    _check(searchers, rig, blobs, counts, min_cand=60)


def test_app_tsx_rig_in_millimetres(searchers):
    """The same rig with translations in millimetres (markers within +-700 mm): the bound's rounding allowances
    (2e-12 trace(B), the float32 slack) and its centre scale with the unit; pixels do not."""
    rig = _app_tsx_rig(scale=1000.0)
    blobs, counts, _ = synth.make_blob_stream(rig, 500, 14, seed=302, half_extent=700.0, min_sep=50.0)
    _check(searchers, rig, blobs, counts, min_cand=60)


def test_parallel_optical_axes(searchers):
    """A row of 5 cameras looking the same way (no point is closest to all axes: mocap_set_cameras falls back to the
    world origin as the centre of the bounds), markers 4 m in front."""
    C = 5
    R = np.repeat(np.eye(3)[None], C, axis=0)
    t = np.array([[-0.6 * i, 0.02 * i, 0.0] for i in range(C)])
    rig = _rig(R, t, synth.DEFAULT_K, (320, 320), centre=[1.2, 0.0, 4.0])
    blobs, counts, _ = synth.make_blob_stream(rig, 400, 16, seed=303, half_extent=0.6)
    _check(searchers, rig, blobs, counts, min_cand=40)


@pytest.mark.parametrize("K,size,C,M,seed,gate", [
    ([[640.0, 0, 320.0], [0, 640.0, 240.0], [0, 0, 1]], (640, 480), 6, 14, 304, 5.0),             # BASELINE "640 x 480" configs
    ([[16000.0, 0, 8000.0], [0, 16000.0, 8000.0], [0, 0, 1]], (16000, 16000), 8, 16, 305, 150.0),  # 16 k-pixel coordinates
    ([[311.0, 0, 150.5], [0, 327.0, 170.25], [0, 0, 1]], (320, 320), 7, 16, 306, 5.0),            # fx != fy, off-centre
])
def test_other_intrinsics(searchers, K, size, C, M, seed, gate):
    """(the 16 k-pixel sensor gets a gate of 150 px = the same angular width as 5 px at f = 320 ... and the same ambiguity)"""
    rig = synth.ring_rig(C, K=K, image_size=size)
    blobs, counts, _ = synth.make_blob_stream(rig, 300, M, seed=seed)
    _check(searchers, rig, blobs, counts, K_max=min(C * M, 64), gate=gate, min_cand=40)


def test_duplicate_pixels_tie_across_blocks(searchers):
    """Two blobs with the SAME pixel in a camera have the same distance to every line and give candidate groups with
    bit-identical errors: exact ties between candidates that sit in different blocks (the duplicated camera is a slow
    digit for some roots) and are evaluated by different waves.  np.argmin keeps the first (helpers.py:418)."""
    rig = synth.ring_rig(8)
    blobs, counts, _ = synth.make_blob_stream(rig, 400, 12, seed=307)
    for cam in (2, 6, 7):
        ok = counts[:, cam] >= 3
        blobs[ok, cam, 2] = blobs[ok, cam, 0]          # blob 2 := blob 0, exactly
    base = _check(searchers, rig, blobs, counts, K_max=64, min_cand=150)
    assert (base["n_out"] > 0).all()


def test_half_the_blobs_missing(searchers):
    """50 % dropout: most roots are created on the way (the chain over the cameras of phase B does the work), groups
    have 2-5 views, many roots have no second view at all."""
    rig = synth.ring_rig(8)
    blobs, counts, _ = synth.make_blob_stream(rig, 500, 16, seed=308, dropout=0.5)
    _check(searchers, rig, blobs, counts, K_max=64)


@pytest.mark.parametrize("C,M,K_max", [(2, 20, 40), (3, 16, 48), (2, 64, 128), (5, 33, 200)])
def test_few_cameras_many_blobs(searchers, C, M, K_max):
    """Two cameras (no chain at all: a root of camera 1 has nothing after it), blob counts that are not powers of two,
    a full wave of blobs per camera, root capacities up to 200."""
    rig = synth.ring_rig(C)
    blobs, counts, _ = synth.make_blob_stream(rig, 150, M, seed=330 + C + M)
    _check(searchers, rig, blobs, counts, K_max=K_max, oracle_frames=15)


def test_largest_layout_16_cameras_48_blob_slots(searchers):
    """16 cameras x 48 blob slots per camera (20 markers in them), 44 roots: the per-blob DLT table alone is 61 KB, the
    workgroup's LDS layout ~ 115 KB of the 128 KB the kernel admits (one workgroup per CU) -- the general narrow layout
    would not fit LDS at all, the search kernel is chosen before that question is asked; the prefetch buffers must stay
    addressable (low 64 KB)."""
    rig = synth.ring_rig(16, K=[[640.0, 0, 320.0], [0, 640.0, 240.0], [0, 0, 1]], image_size=(640, 480))
    blobs, counts, _ = synth.make_blob_stream(rig, 40, 20, seed=340, dropout=0.2, m_max=48)
    _check(searchers, rig, blobs, counts, K_max=44, gate=2.0, oracle_frames=6)
    assert searchers["forced"].last_frame_kernel() == "frame_bb_kernel<CW=2>"


@pytest.mark.parametrize("C,M", [(12, 8), (16, 6), (9, 10)])
def test_more_than_eight_cameras(searchers, C, M):
    """Groups of more than 8 cameras carry their blob indices in two 64-bit words (frame_bb_kernel<CW=2>)."""
    rig = synth.ring_rig(C)
    blobs, counts, _ = synth.make_blob_stream(rig, 200, M, seed=310 + C)
    _check(searchers, rig, blobs, counts, K_max=64, oracle_frames=20)
    assert searchers["forced"].last_frame_kernel() == "frame_bb_kernel<CW=2>"


def test_random_rigs_hypothesis(searchers):
    """Hypothesis-drawn rigs: cameras anywhere on a shell of 1.5-6 m around the markers, rolled, not quite looking at the
    centre; focal length, principal point, blob count, noise, dropout and gate drawn too."""
    from hypothesis import given, settings, strategies as st, HealthCheck

    @settings(max_examples=12, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(st.integers(4, 10), st.integers(9, 16), st.integers(0, 2**31 - 1), st.floats(250.0, 900.0),
           st.floats(0.0, 0.35), st.floats(0.05, 1.2), st.sampled_from([3.0, 5.0, 8.0]))
    def run(C, M, seed, f, dropout, noise, gate):
        rng = np.random.default_rng(seed)
        w = int(rng.integers(300, 1300))
        K = [[f, 0, w / 2 + rng.normal(0, 5)], [0, f * rng.uniform(0.97, 1.03), w / 2 + rng.normal(0, 5)], [0, 0, 1]]
        Rw, tw = [], []
        for _ in range(C):
            d = rng.normal(size=3)
            pos = d / np.linalg.norm(d) * rng.uniform(1.5, 6.0)
            R = synth._look_at(pos, target=rng.normal(0, 0.15, 3))
            a = rng.uniform(-0.5, 0.5)                      # roll about the optical axis
            Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
            R = Rz @ R
            Rw.append(R)
            tw.append(-R @ pos)
        Rw, tw = np.array(Rw), np.array(tw)
        R = np.array([Rw[i] @ Rw[0].T for i in range(C)])
        t = np.array([tw[i] - R[i] @ tw[0] for i in range(C)])
        R[0], t[0] = np.eye(3), 0.0
        rig = {"K": np.repeat(np.array(K)[None], C, axis=0), "R": R, "t": t, "image_size": (w, w), "centre": tw[0], "R0": Rw[0]}
        blobs, counts, _ = synth.make_blob_stream(rig, 60, M, seed=seed % 1000, noise_px=noise, dropout=dropout, half_extent=0.5)
        _check(searchers, rig, blobs, counts, K_max=min(C * M, 64), gate=gate, oracle_frames=10)
    run()


def test_two_candidate_blocks_on_heavy_frames_repeatedly(searchers):
    """Regression for a race of the search's queue (round 5): every lane reads the queue counter at the top of a test
    round; a wave that was through its block tests early (cached bounds, nothing to decode) could bump the counter before the
    slowest wave had read it -- that wave then took the other branch of the flush decision.  It showed with 2-candidate
    blocks (a thousand blocks per heavy frame: many test rounds per frame) as one wrong frame in a few runs of 1 500; the
    pushes now wait behind a barrier.  30 repetitions, bit for bit against the exhaustive walk each time."""
    rig = synth.ring_rig(8)
    blobs, counts, _ = synth.make_blob_stream(rig, 1500, 16, seed=7)
    ex = searchers["exhaustive"]
    ex.set_cameras(rig["K"], rig["R"], rig["t"])
    base = ex.match_triangulate(blobs, counts, K_max=48)
    valid = np.arange(48)[None, :] < base["n_out"][:, None]
    tiny = _ctx({"MOCAP_BB_PL": "2"})
    try:
        tiny.set_cameras(rig["K"], rig["R"], rig["t"])
        for rep in range(30):
            res = tiny.match_triangulate(blobs, counts, K_max=48)
            assert tiny.last_frame_kernel().startswith("frame_bb_kernel")
            for key in ("xyz", "err", "corr"):
                assert np.array_equal(res[key][valid], base[key][valid]), (rep, key)
    finally:
        tiny.close()


def test_the_reference_seam_reaches_the_search_kernel(core):
    """helpers.find_point_correspondance_and_object_points (the mirror of helpers.py:339, what the live loop calls at
    helpers.py:94) passes no K_max: the default min(C M, 64) and the re-submit capacities must land in the
    branch-and-bound kernel -- and match_triangulate_auto's worst-case re-submit too."""
    from mocap_core import helpers
    rig = synth.ring_rig(8)
    blobs, counts, _ = synth.make_blob_stream(rig, 4, 16, seed=320)
    helpers.set_core(core)
    helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in rig["K"]])
    pts = synth.frame_to_reference_lists(blobs[0], counts[0])
    err, xyz, _ = helpers.find_point_correspondance_and_object_points(pts, synth.rig_to_pose_dicts(rig), None)
    assert core.last_frame_kernel() == "frame_bb_kernel<CW=1>" and len(err) == len(xyz) > 0
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    res = core.match_triangulate(blobs, counts, K_max=8 * 16)          # the re-submit's capacity
    assert core.last_frame_kernel() == "frame_bb_kernel<CW=1>" and not res["status"].any()


def test_self_check_build_reports_no_violation():
    """lib/libmocap_core_eigcheck.so (-DMOCAP_DEBUG_EIGCHECK): every candidate whose evaluation the search cut short
    and every candidate of every dropped block is evaluated in full on the device and compared with the bound it was
    cut on; a violation prints an EIGCHECK line.  Counters prove the checks ran."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "low-cost-mocap_amd", "lib", "libmocap_core_eigcheck.so")
    assert os.path.exists(lib), "build it with `make -C low-cost-mocap_amd all` (__graft_entry__.build does)"
    code = r"""
import sys, numpy as np
sys.path[:0] = [%r, %r]
import torch
from mocap_core import capi, synth
core = capi.MocapCore(0)
dev = torch.device("cuda:0")
tot = [0, 0]
for C, M, F, seed, dup in ((8, 16, 1500, 1, False), (4, 16, 600, 2, False), (8, 12, 600, 3, True)):
    rig = synth.ring_rig(C)
    blobs, counts, _ = synth.make_blob_stream(rig, F, M, seed=seed)
    if dup:
        ok = counts[:, 5] >= 3
        blobs[ok, 5, 2] = blobs[ok, 5, 0]
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    K = 48
    d_b, d_c = torch.from_numpy(blobs).to(dev), torch.from_numpy(counts).to(dev)
    xyz = torch.empty((F, K, 3), dtype=torch.float64, device=dev); err = torch.empty((F, K), dtype=torch.float64, device=dev)
    corr = torch.empty((F, K, C), dtype=torch.int16, device=dev); n_out = torch.zeros(F, dtype=torch.int32, device=dev)
    status = torch.zeros(F + 2, dtype=torch.int32, device=dev)          # + the self-check build's two counters
    core.match_triangulate_dev(F, M, d_b.data_ptr(), d_c.data_ptr(), 5.0, K, 1 << 20, xyz.data_ptr(), err.data_ptr(),
                               corr.data_ptr(), n_out.data_ptr(), status.data_ptr())
    core.synchronize()
    assert core.last_frame_kernel().startswith("frame_bb_kernel")
    s = status.cpu().numpy()
    assert not s[:F].any()
    tot[0] += int(s[F]); tot[1] += int(s[F + 1])
print("CHECKED", tot[0], tot[1])
""" % (root, os.path.join(root, "low-cost-mocap_amd"))
    env = dict(os.environ, MOCAP_CORE_LIB=lib, MOCAP_BB_MIN_G="0")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "EIGCHECK" not in p.stdout, p.stdout[:2000]
    checked = [ln for ln in p.stdout.splitlines() if ln.startswith("CHECKED")][-1].split()
    assert int(checked[1]) > 10000 and int(checked[2]) > 100000, checked   # cut candidates, candidates of dropped blocks


@pytest.mark.parametrize("K_max", [40, 48, 57, 64, 65])
def test_fixed_layout_instantiations_equal_the_runtime_layout(searchers, K_max):
    """8 cameras x 16 blob slots with K_max <= 48 / <= 64 go to instantiations whose LDS layout is a compile-time
    constant (laid out for 48 / 64 roots; the root limit and the output stride stay K_max); MOCAP_BB_FIXED_LAYOUT=0 sends
    the same batch to the runtime-layout instantiation.  Same bits either way, and the same as the exhaustive walk --
    also with fewer output slots than roots (root overflow status) and at K_max = 65, which no fixed layout takes."""
    rig = synth.ring_rig(8)
    blobs, counts, _ = synth.make_blob_stream(rig, 300, 16, seed=11)
    assert blobs.shape[2] == 16
    base = _check(searchers, rig, blobs, counts, K_max=K_max, min_cand=1000)
    old = os.environ.get("MOCAP_BB_FIXED_LAYOUT")
    os.environ["MOCAP_BB_FIXED_LAYOUT"] = "0"
    try:
        c = searchers["default"]
        res = c.match_triangulate(blobs, counts, gate_px=5.0, K_max=K_max)
        assert c.last_frame_kernel().startswith("frame_bb_kernel")
    finally:
        if old is None:
            del os.environ["MOCAP_BB_FIXED_LAYOUT"]
        else:
            os.environ["MOCAP_BB_FIXED_LAYOUT"] = old
    valid = np.arange(base["err"].shape[1])[None, :] < base["n_out"][:, None]
    for key in ("n_out", "status", "n_cand"):
        assert np.array_equal(res[key], base[key]), key
    assert np.array_equal(res["corr"][valid], base["corr"][valid])
    for key in ("xyz", "err"):
        assert np.array_equal(res[key][valid], base[key][valid], equal_nan=True), key


def test_fixed_layout_with_more_roots_than_output_slots(searchers):
    """K_max = 20 < roots of a 16-marker frame: the frame is flagged (root overflow) by both layouts and by the
    exhaustive walk alike; frames that fit are unaffected."""
    rig = synth.ring_rig(8)
    blobs, counts, _ = synth.make_blob_stream(rig, 200, 16, seed=12)
    outs = {}
    for name in ("default", "exhaustive"):
        c = searchers[name]
        c.set_cameras(rig["K"], rig["R"], rig["t"])
        outs[name] = c.match_triangulate(blobs, counts, gate_px=5.0, K_max=20)
    os.environ["MOCAP_BB_FIXED_LAYOUT"] = "0"
    try:
        outs["runtime"] = searchers["default"].match_triangulate(blobs, counts, gate_px=5.0, K_max=20)
    finally:
        del os.environ["MOCAP_BB_FIXED_LAYOUT"]
    base = outs["exhaustive"]
    assert base["status"].any() and not base["status"].all()
    for name in ("default", "runtime"):
        assert np.array_equal(outs[name]["status"], base["status"]), name
        assert np.array_equal(outs[name]["n_out"], base["n_out"]), name
        ok = base["status"] == 0
        valid = (np.arange(20)[None, :] < base["n_out"][:, None]) & ok[:, None]
        assert np.array_equal(outs[name]["corr"][valid], base["corr"][valid]), name
        assert np.array_equal(outs[name]["err"][valid], base["err"][valid], equal_nan=True), name
        assert np.array_equal(outs[name]["xyz"][valid], base["xyz"][valid], equal_nan=True), name
