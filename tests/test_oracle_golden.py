"""CPU: the oracle (Python + C restatements) against the golden vectors produced by the
reference's own functions (oracle/make_golden.py).  Pins the checker before it is trusted."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from oracle import c_oracle, mocap_oracle as mo

FRAME_SETS = golden_names("frames_")
DLT_SETS = golden_names("dlt_")


def _corr_xy(blobs_f, corr):
    K, C = corr.shape
    out = np.full((K, C, 2), np.nan)
    for r in range(K):
        for c in range(C):
            if corr[r, c] >= 0:
                out[r, c] = blobs_f[c, corr[r, c]]
    return out


@pytest.mark.parametrize("name", FRAME_SETS)
def test_python_oracle_frames_bit_exact(name):
    g = load_golden(name)
    Ks = [k for k in g["K"]]
    F = g["blobs"].shape[0]
    Ftab = mo.fundamental_table(Ks, g["R"], g["t"])
    for f in range(min(F, 12)):
        o = mo.find_point_correspondance_and_object_points(g["blobs"][f], g["counts"][f], Ks, g["R"], g["t"], Ftab=Ftab)
        k = int(g["ref_n"][f])
        assert len(o["errors"]) == k
        # same arithmetic as the reference -> bit-exact
        assert np.array_equal(o["object_points"], g["ref_xyz"][f, :k])
        assert np.array_equal(o["errors"], g["ref_err"][f, :k])
        assert np.array_equal(_corr_xy(g["blobs"][f], o["corr"]), g["ref_corr_xy"][f, :k], equal_nan=True)


@pytest.mark.parametrize("name", FRAME_SETS)
def test_c_oracle_frames(name):
    g = load_golden(name)
    co = c_oracle.COracle(g["K"], g["R"], g["t"])
    res = co.match_triangulate(g["blobs"], g["counts"])
    assert np.array_equal(res["n_out"], g["ref_n"])
    assert not res["status"].any()
    for f in range(g["blobs"].shape[0]):
        k = int(g["ref_n"][f])
        # correspondence: bit-exact
        assert np.array_equal(_corr_xy(g["blobs"][f], res["corr"][f, :k]), g["ref_corr_xy"][f, :k], equal_nan=True)
        if k:
            # 3-D points: 1e-5 relative is the contract; the restatement is ~1e-12
            np.testing.assert_allclose(res["xyz"][f, :k], g["ref_xyz"][f, :k], rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(res["err"][f, :k], g["ref_err"][f, :k], rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("name", DLT_SETS)
def test_oracles_dlt(name):
    g = load_golden(name)
    Ks = [k for k in g["K"]]
    xyz_py = mo.triangulate_points(g["obs"], Ks, g["R"], g["t"])
    err_py = mo.reprojection_errors(g["obs"], xyz_py, Ks, g["R"], g["t"])
    assert np.array_equal(xyz_py, g["ref_xyz"], equal_nan=True)
    assert np.array_equal(err_py, g["ref_err"], equal_nan=True)
    assert np.array_equal(err_py[np.isfinite(err_py)], g["ref_err_packed"])
    co = c_oracle.COracle(g["K"], g["R"], g["t"])
    xyz_c, err_c = co.triangulate(g["obs"])
    assert np.array_equal(np.isnan(xyz_c), np.isnan(g["ref_xyz"]))
    np.testing.assert_allclose(xyz_c, g["ref_xyz"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(err_c, g["ref_err"], rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("name", golden_names("ba_"))
def test_oracles_ba_residuals(name):
    g = load_golden(name)
    Ks = [k for k in g["K"]]
    co = c_oracle.COracle(g["K"], g["R_init"], g["t_init"])
    r_c = co.ba_residuals(g["xs"], g["obs"])
    for p, x in enumerate(g["xs"]):
        r_py = mo.ba_residuals(x, g["obs"], Ks)
        valid = np.isfinite(r_py)
        np.testing.assert_allclose(r_py[valid], g["res64"][p], rtol=1e-12)
        # the float32 cast of helpers.py:273
        assert np.array_equal(r_py[valid].astype(np.float32), g["res32"][p])
        np.testing.assert_allclose(r_c[p][valid], g["res64"][p], rtol=1e-7)
        assert np.array_equal(np.isfinite(r_c[p]), valid)


def test_f32_rounding_switch_changes_little():
    g = load_golden("frames_c4_m4")
    a = c_oracle.COracle(g["K"], g["R"], g["t"], f32_rounding=True).match_triangulate(g["blobs"], g["counts"])
    b = c_oracle.COracle(g["K"], g["R"], g["t"], f32_rounding=False).match_triangulate(g["blobs"], g["counts"])
    assert np.array_equal(a["n_out"], b["n_out"])
    m = np.isfinite(a["err"]) & np.isfinite(b["err"])
    np.testing.assert_allclose(a["err"][m], b["err"][m], rtol=1e-2, atol=1e-5)


def test_post_rows_oracle_vs_reference_golden():
    """World-coordinate epilogue (helpers.py:96-103, run verbatim from the reference file by
    oracle/make_golden.py) and locate_objects (helpers.py:424-480) against the restatements."""
    from oracle import mocap_oracle as mo
    g = load_golden("post_world_locate")
    for f in range(g["cam_xyz"].shape[0]):
        n = int(g["n_pts"][f])
        if n:
            assert np.array_equal(mo.world_epilogue(g["cam_xyz"][f, :n], g["to_world"]), g["ref_world"][f, :n])
        objs = mo.locate_objects(g["ref_world"][f, :n], g["err"][f, :n])
        assert len(objs) == g["ref_nobj"][f]
        for j, o in enumerate(objs):
            assert np.array_equal(o["pos"], g["ref_pos"][f, j]) and o["droneIndex"] == g["ref_drone"][f, j]
            assert abs(o["heading"] - g["ref_heading"][f, j]) < 1e-12
            assert abs(o["error"] - g["ref_error"][f, j]) <= 1e-15


# ----------------------------------------------------------------------------- blob extraction
BLOB_SETS = golden_names("blobs_")


@pytest.mark.parametrize("name", BLOB_SETS)
def test_blob_oracle_matches_reference_golden(name):
    """oracle/blob_oracle.py (restated rot90 / make_square / _find_dot) against the reference's own
    _camera_read + _find_dot run through the harness: bit-exact frames, points and order."""
    from oracle import blob_oracle as bo
    g = load_golden(name)
    for f in range(g["images"].shape[0]):
        frames, pts = bo.find_dots(g["images"][f], g["K"], g["dist"], g["rotation"])
        assert np.array_equal(np.array(frames), g["ref_frames"][f])
        for c, p in enumerate(pts):
            n = int(g["ref_counts"][f, c])
            assert len(p) == n
            assert np.array_equal(np.array(p, dtype=np.int32).reshape(n, 2), g["ref_points"][f, c, :n])


def test_find_contours_structure_against_scipy_labelling():
    """The restated findContours against an independent definition: one outer border per 8-connected
    foreground component, one hole border per enclosed 4-connected background component, parents =
    geometric containment, RETR_TREE order = pre-order with siblings in reverse raster order."""
    from scipy import ndimage
    from oracle import cv_image_restate as ci
    rng = np.random.default_rng(0)
    for trial in range(6):
        m = (rng.random((40, 48)) < (0.35 + 0.08 * trial)).astype(np.uint8) * 255
        contours, hier = ci.find_contours(m, ci.RETR_TREE, ci.CHAIN_APPROX_SIMPLE)
        fg, n_fg = ndimage.label(m > 0, structure=np.ones((3, 3)))
        pad = np.pad(m == 0, 1, constant_values=True)
        bg, n_bg = ndimage.label(pad)                       # 4-connected; label of the outside = bg[0, 0]
        holes = set(range(1, n_bg + 1)) - {bg[0, 0]}
        is_hole = []
        for i, c in enumerate(contours):
            depth, p = 0, hier[0, i, 3]
            while p >= 0:
                depth, p = depth + 1, hier[0, p, 3]
            is_hole.append(depth % 2 == 1)
        assert sum(1 for h in is_hole if not h) == n_fg
        assert sum(1 for h in is_hole if h) == len(holes)
        for i, c in enumerate(contours):
            x, y = c[0, 0]
            comp = fg[y, x]
            assert comp > 0                                 # contour vertices are foreground pixels
            assert all(fg[py, px] == comp for px, py in c[:, 0])
            par = hier[0, i, 3]
            if is_hole[i]:
                assert par >= 0 and fg[contours[par][0, 0][1], contours[par][0, 0][0]] == comp
            elif par >= 0:
                # an outer border inside a hole: the hole's border belongs to the enclosing component
                assert is_hole[par]
        # siblings at the top level come in reverse raster order of their first (start) vertex
        top = [i for i in range(len(contours)) if hier[0, i, 3] < 0]
        starts = [(contours[i][:, 0, 1].min(), ) for i in top]
        assert starts == sorted(starts, reverse=True) or len(top) < 2 or all(
            starts[k] >= starts[k + 1] for k in range(len(starts) - 1))


@pytest.mark.parametrize("name", BLOB_SETS)
def test_c_blob_oracle_matches_reference_golden(name):
    """oracle/c/blob_oracle.c (sequential Suzuki-Abe with border marks) against the same goldens."""
    g = load_golden(name)
    bo = c_oracle.BlobOracle(g["images"].shape[2], g["images"].shape[3], g["K"], g["dist"], g["rotation"])
    r = bo.find_blobs(g["images"], M_max=g["ref_points"].shape[2], want_processed=True)
    assert np.array_equal(r["counts"], g["ref_counts"])
    assert np.array_equal(r["processed"], g["ref_frames"])
    for f in range(g["images"].shape[0]):
        for c in range(g["images"].shape[1]):
            n = int(g["ref_counts"][f, c])
            assert np.array_equal(r["blobs"][f, c, :n], g["ref_points"][f, c, :n].astype(np.float32))


# ----------------------------------------------------------------------------- initial poses
@pytest.mark.parametrize("name", golden_names("pose_"))
def test_pose_oracle_matches_reference_golden(name):
    """oracle/pose_oracle.py against the reference's own calculate_camera_pose handler (index.py:229-270)
    run through the harness: same arithmetic -> bit-exact."""
    from oracle import pose_oracle
    g = load_golden(name)
    R, t = pose_oracle.initial_poses(g["obs"], [k for k in g["K"]])
    assert np.array_equal(R, g["ref_R"])
    assert np.array_equal(t, g["ref_t"])


def test_cv_rng_and_subset_known_answers():
    """cv::RNG is a documented multiply-with-carry generator: its first outputs from the RANSAC seed are
    fixed numbers (computed by hand from state = 2^64 - 1, coefficient 4164903690)."""
    from oracle import cv_pose_restate as cp
    r = cp.RNG()
    s = 0xFFFFFFFFFFFFFFFF
    for _ in range(5):
        s = ((s & 0xFFFFFFFF) * 4164903690 + (s >> 32)) & 0xFFFFFFFFFFFFFFFF
        assert r.next() == (s & 0xFFFFFFFF)
    r = cp.RNG()
    assert [r.uniform(0, 100) for _ in range(4)] == [(v % 100) for v in _mwc(4)]


def _mwc(n):
    s, out = 0xFFFFFFFFFFFFFFFF, []
    for _ in range(n):
        s = ((s & 0xFFFFFFFF) * 4164903690 + (s >> 32)) & 0xFFFFFFFFFFFFFFFF
        out.append(s & 0xFFFFFFFF)
    return out


def test_seven_point_models_satisfy_their_sample():
    """Every matrix run_7point returns is singular and satisfies x2^T F x1 = 0 on its 7 correspondences."""
    from mocap_core import synth
    from oracle import cv_pose_restate as cp
    rng = np.random.default_rng(0)
    rig = synth.ring_rig(2)
    obs, _ = synth.make_ba_observations(rig, 7, seed=1, dropout=0.0)
    p1, p2 = obs[:, 0], obs[:, 1]
    Fs = cp.run_7point(p1, p2)
    assert 1 <= len(Fs) <= 3
    for F in Fs:
        assert abs(np.linalg.det(F / np.linalg.norm(F))) < 1e-10
        h1 = np.c_[p1, np.ones(7)]
        h2 = np.c_[p2, np.ones(7)]
        assert np.abs(np.einsum("ni,ij,nj->n", h2, F, h1)).max() < 1e-6 * np.abs(F).max() * 320 * 320
    del rng


def test_blob_oracle_recovers_rendered_markers():
    """Physical sanity of the restated pipeline (independent of any OpenCV detail): dots rendered at the
    lens-DISTORTED projection come back at the IDEAL pinhole pixel, i.e. the restated undistortion map
    really inverts the Brown-Conrady model of camera-params.json, and make_square's 40-row offset is right."""
    from mocap_core import synth
    rig = synth.ring_rig(3)
    strong = (-0.35, 0.10, 0.002, -0.001, 0.0)            # enough barrel distortion to move dots by several px
    images, truth = synth.render_camera_frames(rig, 4, 8, seed=77, dropout=0.0, dist=strong, half_extent=1.0)
    bo = c_oracle.BlobOracle(240, 320, rig["K"], [strong] * 3)
    res = bo.find_blobs(images, M_max=32)
    near = total = 0
    worst_shift = 0.0
    for f in range(4):
        for c in range(3):
            uv = truth["uv"][f, c]
            uv = uv[~np.isnan(uv[:, 0])]
            b = res["blobs"][f, c, :res["counts"][f, c]]
            for p in uv:
                total += 1
                d = np.abs(b - p).max(axis=1).min()
                near += d < 1.5
                # how far the lens moved this dot: the test is only meaningful if that exceeds the tolerance
                ud, vd = synth.distort_pixels(p[0], p[1], rig["K"][c], strong)
                worst_shift = max(worst_shift, abs(ud - p[0]), abs(vd - p[1]))
    assert near >= 0.9 * total, (near, total)
    assert worst_shift > 2.0


# ----------------------------------------------------------------------------- known answers of the restated OpenCV calls
def test_find_contours_known_answers():
    """Facts about cv.findContours / cv.moments that are common knowledge among OpenCV users and follow from
    its border-following definition: a filled 3x3 square gives the 4 corner pixels, starting at the top-left
    one and going down first (counter-clockwise on screen), with polygon area 4 (through pixel centres), not 9;
    a single pixel is a 1-point contour of area 0; a 2-pixel bar has area 0; contours come bottom-most first;
    a ring has an outer contour and a hole contour (an octagon: hole borders cut corners) whose parent is the outer one."""
    from oracle import cv_image_restate as ci
    m = np.zeros((7, 8), dtype=np.uint8)
    m[1:4, 1:4] = 255
    cs, h = ci.find_contours(m, ci.RETR_TREE, ci.CHAIN_APPROX_SIMPLE)
    assert len(cs) == 1 and cs[0][:, 0].tolist() == [[1, 1], [1, 3], [3, 3], [3, 1]]
    mo = ci.moments(cs[0])
    assert mo["m00"] == 4.0 and mo["m10"] / mo["m00"] == 2.0 and mo["m01"] / mo["m00"] == 2.0
    m = np.zeros((6, 6), dtype=np.uint8)
    m[2, 3] = 255
    cs, _ = ci.find_contours(m, ci.RETR_TREE, ci.CHAIN_APPROX_SIMPLE)
    assert len(cs) == 1 and cs[0][:, 0].tolist() == [[3, 2]] and ci.moments(cs[0])["m00"] == 0.0
    m[2, 4] = 255
    cs, _ = ci.find_contours(m, ci.RETR_TREE, ci.CHAIN_APPROX_SIMPLE)
    assert len(cs) == 1 and ci.moments(cs[0])["m00"] == 0.0          # a 2-pixel bar encloses nothing
    m = np.zeros((10, 10), dtype=np.uint8)
    m[1:3, 1:3] = 255
    m[6:9, 5:8] = 255
    cs, _ = ci.find_contours(m, ci.RETR_TREE, ci.CHAIN_APPROX_SIMPLE)
    assert [c[0, 0].tolist() for c in cs] == [[5, 6], [1, 1]]          # the lower blob first
    m = np.zeros((9, 9), dtype=np.uint8)
    m[1:8, 1:8] = 255
    m[3:6, 3:6] = 0
    cs, h = ci.find_contours(m, ci.RETR_TREE, ci.CHAIN_APPROX_SIMPLE)
    assert len(cs) == 2 and h[0, 0, 3] == -1 and h[0, 1, 3] == 0 and h[0, 0, 2] == 1
    # the hole border runs over the foreground pixels 8-connected around the hole: it cuts the four corners
    assert cs[1][:, 0].tolist() == [[2, 3], [3, 2], [5, 2], [6, 3], [6, 5], [5, 6], [3, 6], [2, 5]]
    assert ci.moments(cs[0])["m00"] == 36.0 and ci.moments(cs[1])["m00"] == 14.0


def test_image_filters_known_answers():
    """The fixed-point pieces against values that follow from their definitions: the 9-tap kernel sums to 256
    and is symmetric; blurring / sharpening a constant image leaves it constant (the sharpening kernel sums to 0:
    constant -> 0); a lens without distortion has the identity map; BT.601 grey of (255, 255, 255) is 255;
    the threshold is strict at floor(255 * 0.2) = 51."""
    from oracle import blob_oracle as bo
    from oracle import cv_image_restate as ci
    k = ci.gaussian_kernel_fixed(9, 0)
    assert sum(k) == 256 and k == k[::-1] and k[4] == max(k)
    img = np.full((24, 24, 3), 77, dtype=np.uint8)
    assert (ci.gaussian_blur(img, (9, 9), 0) == 77).all()
    assert (ci.filter2d(img, -1, bo.SHARPEN) == 0).all() and int(bo.SHARPEN.sum()) == 0
    sx, sy, fx, fy = ci.undistort_map([[300.0, 0, 100], [0, 310.0, 90], [0, 0, 1]], [0, 0, 0, 0, 0], 64, 64)
    yy, xx = np.mgrid[0:64, 0:64]
    assert np.array_equal(sx, xx) and np.array_equal(sy, yy) and not fx.any() and not fy.any()
    ramp = np.arange(64 * 64 * 3, dtype=np.uint32).reshape(64, 64, 3).astype(np.uint8)
    assert np.array_equal(ci.undistort(ramp, [[300.0, 0, 100], [0, 310.0, 90], [0, 0, 1]], [0, 0, 0, 0, 0]), ramp)
    white = np.full((2, 2, 3), 255, dtype=np.uint8)
    assert (ci.cvt_color(white, ci.COLOR_RGB2GRAY) == 255).all()
    g = np.array([[50, 51, 52]], dtype=np.uint8)
    assert ci.threshold(g, 255 * 0.2, 255, ci.THRESH_BINARY)[1].tolist() == [[0, 0, 255]]


def test_pose_restatement_known_answers():
    """Geometry the restated calibration calls must satisfy whatever OpenCV's internals are: on exact
    correspondences RANSAC keeps every point and its F is the rig's fundamental matrix up to scale; E = K2^T F K1
    has singular values (s, s, 0); the four candidates of motionFromEssential contain the true relative pose."""
    from mocap_core import synth
    from oracle import cv_pose_restate as cp
    rig = synth.ring_rig(2)
    obs, _ = synth.make_ba_observations(rig, 80, seed=4, noise_px=0.0, dropout=0.0)
    p1, p2 = obs[:, 0].astype(np.float32), obs[:, 1].astype(np.float32)
    F, mask, info = cp.find_fundamental_mat(p1, p2, cp.FM_RANSAC, 1.0, 0.99999, return_info=True)
    assert info["inliers"] == 80 and mask.all()
    K, R, t = rig["K"][0], rig["R"][1], rig["t"][1]
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    F_true = np.linalg.inv(K).T @ tx @ R @ np.linalg.inv(K)
    Fn, Ft = F / np.linalg.norm(F), F_true / np.linalg.norm(F_true)
    if (Fn * Ft).sum() < 0:
        Fn = -Fn
    np.testing.assert_allclose(Fn, Ft, atol=5e-3)              # up to scale and sign; points are float32 pixels
    E = cp.essential_from_fundamental(F, K, K)
    sv = np.linalg.svd(E, compute_uv=False)
    assert abs(sv[0] - sv[1]) < 2e-3 * sv[0] and sv[2] < 2e-3 * sv[0]
    Rs, ts = cp.motion_from_essential(E)
    tn = t / np.linalg.norm(t)
    best = min(max(np.abs(Rc - R).max(), np.abs(tc.ravel() - tn).max()) for Rc, tc in zip(Rs, ts))
    assert best < 5e-3
    for Rc in Rs:
        assert abs(np.linalg.det(Rc) - 1) < 1e-9


def test_hot_path_cv_restatements_known_answers():
    """The three OpenCV calls of the hot path, against what their definitions imply for exact data:
    fundamentalFromProjections gives x2^T F x1 = 0 for every projected 3-D point; computeCorrespondEpilines
    returns the unit-normal line F x through the matching point; projectPoints equals K (R X + t)."""
    from mocap_core import synth
    from oracle import cv_restate as cr
    rig = synth.ring_rig(3)
    rng = np.random.default_rng(2)
    X = rng.uniform(-0.5, 0.5, (20, 3)) @ rig["R0"].T + rig["centre"]
    P = [mo.projection_matrix(rig["K"][c], rig["R"][c], rig["t"][c]) for c in range(3)]
    F = cr.fundamental_from_projections(P[0], P[2])
    uv = []
    for c in (0, 2):
        h = (P[c] @ np.c_[X, np.ones(20)].T).T
        uv.append(h[:, :2] / h[:, 2:3])
    h1, h2 = np.c_[uv[0], np.ones(20)], np.c_[uv[1], np.ones(20)]
    assert np.abs(np.einsum("ni,ij,nj->n", h2, F, h1)).max() < 1e-9 * np.abs(F).max() * 320 * 320
    saved = cr.F32_ROUNDING
    try:
        cr.F32_ROUNDING = False
        lines = cr.compute_correspond_epilines(uv[0].astype(np.float32), 1, F)[:, 0]
        assert np.allclose(np.hypot(lines[:, 0], lines[:, 1]), 1.0)
        assert np.abs(np.einsum("ni,ni->n", lines, h2)).max() < 1e-3          # float32 input pixels
        proj, _ = cr.project_points(X, rig["R"][2], rig["t"][2], rig["K"][2], [])
        assert np.allclose(proj[:, 0], uv[1], atol=1e-9)
    finally:
        cr.F32_ROUNDING = saved
