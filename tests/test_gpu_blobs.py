"""GPU parity of the blob-extraction stage (Cameras._camera_read preprocessing +
Cameras._find_dot, reference helpers.py:68-82, 143-163) against the oracle and the reference-pinned
golden set.  Everything in this stage is integer arithmetic: the bar is BIT-EXACT (processed frames,
centroids, their order, counts)."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from mocap_core import capi, synth
from oracle import blob_oracle as bo

pytestmark = pytest.mark.gpu

REF_K = np.array([[320.0, 0, 160], [0, 320, 160], [0, 0, 1]])


def _oracle_points(frames_set, Ks, dists, rots=None):
    frames, pts = bo.find_dots(frames_set, Ks, dists, rots)
    return np.array(frames), pts


def _check_against_oracle(core, images, Ks, dists, rots=None, M_max=64, check_frames=True):
    F, C = images.shape[:2]
    core.set_image_params(images.shape[2], images.shape[3], Ks, dists, rots)
    res = core.find_blobs(images, M_max=M_max, want_processed=check_frames)
    for f in range(F):
        frames, pts = _oracle_points(images[f], Ks, dists, rots)
        if check_frames:
            assert np.array_equal(res["processed"][f], frames), f"processed frame differs (frame set {f})"
        for c in range(C):
            n = len(pts[c])
            assert res["counts"][f, c] == min(n, M_max), (f, c, res["counts"][f, c], n)
            k = min(n, M_max)
            got = res["blobs"][f, c, :k].astype(np.int64).tolist()
            assert got == pts[c][:k], (f, c)
            assert bool(res["status"][f, c] & capi.BLOB_ST_POINT_OVERFLOW) == (n > M_max)
    return res


def test_undistort_map_matches_oracle(core):
    from oracle import cv_image_restate as ci
    core.set_image_params(240, 320, [REF_K], [synth.REFERENCE_DISTORTION])
    m = core.undistort_map(0)
    sx, sy, fx, fy = ci.undistort_map(REF_K, synth.REFERENCE_DISTORTION, 320, 320)
    outside = (sx >= 320) | (sx + 1 < 0) | (sy >= 320) | (sy + 1 < 0)
    want = np.where(outside, 2047 << 10, fx | (fy << 5) | ((sx + 1) << 10) | ((sy + 1) << 21)).astype(np.uint32)
    assert np.array_equal(m, want)


@pytest.mark.parametrize("name", golden_names("blobs_"))
def test_blobs_match_reference_golden(core, name):
    """Golden = the reference's own _camera_read + _find_dot run through the stub harness."""
    g = load_golden(name)
    images = g["images"]
    F, C = images.shape[:2]
    core.set_image_params(images.shape[2], images.shape[3], g["K"], g["dist"], g["rotation"])
    res = core.find_blobs(images, M_max=g["ref_points"].shape[2], want_processed=True)
    assert np.array_equal(res["counts"], g["ref_counts"])
    assert np.array_equal(res["processed"], g["ref_frames"])
    for f in range(F):
        for c in range(C):
            n = g["ref_counts"][f, c]
            assert np.array_equal(res["blobs"][f, c, :n], g["ref_points"][f, c, :n].astype(np.float32))


def test_helpers_mirror_returns_the_reference_lists(core):
    """helpers.camera_read_find_dots = the per-camera body of Cameras._camera_read + _find_dot: same frames,
    same nested lists ([[None, None]] for an empty camera) as the reference produced for the golden frames."""
    from mocap_core import helpers
    g = load_golden("blobs_c3_calib_rot")
    helpers.set_core(core)
    helpers.set_camera_params([{"intrinsic_matrix": g["K"][c].tolist(), "distortion_coef": g["dist"][c].tolist(),
                                "rotation": int(g["rotation"][c])} for c in range(3)])
    frames, image_points = helpers.camera_read_find_dots(list(g["images"][0]))
    assert np.array_equal(np.array(frames), g["ref_frames"][0])
    for c in range(3):
        n = int(g["ref_counts"][0, c])
        want = g["ref_points"][0, c, :n].tolist() if n else [[None, None]]
        assert image_points[c] == want
    # an empty camera yields the reference's sentinel
    blank = np.zeros_like(g["images"][0])
    _, pts = helpers.camera_read_find_dots(list(blank), want_frames=False)
    assert pts == [[[None, None]]] * 3


def test_synthetic_rig_frames(core):
    rig = synth.ring_rig(4)
    images, truth = synth.render_camera_frames(rig, 3, 8, seed=11)
    dists = [synth.REFERENCE_DISTORTION] * 4
    res = _check_against_oracle(core, images, rig["K"], dists)
    # and the centroids are the markers: visible markers have a blob within 1.5 px of their ideal pixel
    # (spots that overlap in a view merge into one blob, as they do for the reference)
    near = total = 0
    for f in range(3):
        for c in range(4):
            uv = truth["uv"][f, c]
            uv = uv[~np.isnan(uv[:, 0])]
            b = res["blobs"][f, c, :res["counts"][f, c]]
            for p in uv:
                total += 1
                near += np.min(np.abs(b - p).max(axis=1)) < 1.5
    assert near >= 0.85 * total, (near, total)


def test_rotation_and_distinct_intrinsics(core):
    rng = np.random.default_rng(5)
    rig = synth.ring_rig(3)
    images, _ = synth.render_camera_frames(rig, 2, 6, seed=12)
    Ks = np.array([[[268.66976067, 0, 123.58679484], [0, 268.57495496, 167.56126939], [0, 0, 1]],
                   [[269.95158059, 0, 139.37072352], [0, 270.09608831, 160.36482761], [0, 0, 1]],
                   REF_K])
    dists = np.array([synth.REFERENCE_DISTORTION, [-0.2, 0.1, 0.002, -0.001, 0.05], [0, 0, 0, 0, 0]])
    _check_against_oracle(core, images, Ks, dists, rots=[2, 0, 2])
    del rng


def test_noisy_frames_holes_nesting_and_caps(core):
    """Bright noise makes hundreds of contours with holes and nesting: exercises the parent/ordering rules,
    the zero-area filter, the large-table re-run and the point-overflow flag."""
    rng = np.random.default_rng(6)
    rig = synth.ring_rig(2)
    images, _ = synth.render_camera_frames(rig, 2, 10, seed=13, spot_sigma=(2.0, 9.0), peak=600.0)
    images[0, 1] = np.maximum(images[0, 1], rng.integers(0, 110, images[0, 1].shape, dtype=np.uint8))
    images[1, 0] = np.maximum(images[1, 0], rng.integers(0, 70, images[1, 0].shape, dtype=np.uint8))
    dists = [synth.REFERENCE_DISTORTION] * 2
    res = _check_against_oracle(core, images, rig["K"], dists, M_max=1024)
    assert res["n_contours"].max() > 256          # the small tables overflowed and the re-run produced the result
    _check_against_oracle(core, images, rig["K"], dists, M_max=8, check_frames=False)


def test_batch_against_c_oracle_sequential_contours(core):
    """A batch that spans several internal chunks (camera phase, workspace reuse) with frames from clean to
    heavily noisy, against the C oracle: the kernel follows border cycles in parallel, the oracle runs the
    sequential Suzuki-Abe raster scan with marks -- two algorithms, identical output required."""
    from oracle import c_oracle
    rng = np.random.default_rng(31)
    C, F = 3, 24
    rig = synth.ring_rig(C)
    images, _ = synth.render_camera_frames(rig, F, 12, seed=32, spot_sigma=(0.8, 5.0), peak=500.0)
    for f in range(0, F, 3):
        amp = int(rng.integers(20, 140))
        images[f, f % C] = np.maximum(images[f, f % C], rng.integers(0, amp, images[f, 0].shape, dtype=np.uint8))
    dists = [synth.REFERENCE_DISTORTION, [-0.2, 0.1, 0.002, -0.001, 0.05], synth.REFERENCE_DISTORTION]
    rots = [0, 2, 0]
    core.set_image_params(240, 320, rig["K"], dists, rots)
    M_max = 128
    res = core.find_blobs(images, M_max=M_max, want_processed=True)
    ref = c_oracle.BlobOracle(240, 320, rig["K"], dists, rots).find_blobs(images, M_max=M_max, want_processed=True)
    assert np.array_equal(res["processed"], ref["processed"])
    assert np.array_equal(res["n_contours"], ref["n_contours"])
    assert np.array_equal(res["counts"], np.minimum(ref["counts"], M_max))
    assert np.array_equal(res["blobs"], ref["blobs"])
    assert ref["n_contours"].max() > 300 and (ref["counts"] > M_max).any()   # the stress frames are in


def test_dark_tile_early_out_does_not_change_results(core):
    """mocap_set_blob_options(skip_dark_tiles): an exact early-out, so masks / centroids must be bit-identical
    with it on and off -- clean frames (most tiles skipped), noise of range 2 / 3 / 4 around the bound of the
    proof, a bright offset (range small but values high), dots on tile borders."""
    rng = np.random.default_rng(41)
    rig = synth.ring_rig(2)
    images, _ = synth.render_camera_frames(rig, 6, 10, seed=42, noise_levels=1)
    images[1] = np.maximum(images[1], rng.integers(0, 3, images[1].shape, dtype=np.uint8))     # range 2
    images[2] = np.maximum(images[2], rng.integers(0, 4, images[2].shape, dtype=np.uint8))     # range 3
    images[3] = np.maximum(images[3], rng.integers(0, 5, images[3].shape, dtype=np.uint8))     # range 4
    images[4] = np.clip(images[4].astype(np.int32) + 200, 0, 255).astype(np.uint8)            # bright, flat
    images[5, 0, 100:104, 62:66] = 255                                                         # dot across a tile border
    images[5, 1, 76:80, 126:130] = 255
    dists = [synth.REFERENCE_DISTORTION] * 2
    core.set_image_params(240, 320, rig["K"], dists, [0, 2])
    try:
        core.set_blob_options(skip_dark_tiles=True)
        on = core.find_blobs(images, M_max=64)
        core.set_blob_options(skip_dark_tiles=False)
        off = core.find_blobs(images, M_max=64)
        core.set_blob_options(skip_dark_tiles=2)        # (round 6) the early-out decided inside the mask pass: no activity pass
        fold = core.find_blobs(images, M_max=64)
    finally:
        core.set_blob_options(skip_dark_tiles=True)
    for k in ("blobs", "counts", "n_contours"):
        assert np.array_equal(fold[k], off[k]), ("fold", k)
    for k in ("blobs", "counts", "status", "n_contours"):
        assert np.array_equal(on[k], off[k]), k
    assert on["counts"].sum() > 0
    _check_against_oracle(core, images[[0, 4, 5]], rig["K"], dists, [0, 2])


@pytest.mark.parametrize("rows,cols", [(240, 352), (200, 352), (480, 640)])   # same edge, fewer rows: stale workspace rows would show
def test_other_frame_geometries(core, rows, cols):
    """Frame edges that are not a multiple of the 64-px tile (352) and VGA: partial tiles, other mask
    strides, LDS table sizes chosen per geometry."""
    from oracle import c_oracle
    rng = np.random.default_rng(rows)
    C, F = 2, 2
    K = np.array([[cols * 1.0, 0, cols / 2], [0, cols * 1.0, cols / 2], [0, 0, 1]])
    images = rng.integers(0, 3, (F, C, rows, cols, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:rows, 0:cols]
    for f in range(F):
        for c in range(C):
            img = images[f, c].astype(np.float32)
            for _ in range(12):
                cx, cy, sg = rng.uniform(0, cols), rng.uniform(0, rows), rng.uniform(1.0, 4.0)
                img += (400 * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * sg * sg)))[..., None]
            images[f, c] = np.clip(img, 0, 255).astype(np.uint8)
    images[1, 1] = np.maximum(images[1, 1], rng.integers(0, 100, images[1, 1].shape, dtype=np.uint8))
    dists = [synth.REFERENCE_DISTORTION, [-0.2, 0.1, 0.002, -0.001, 0.05]]
    core.set_image_params(rows, cols, [K, K], dists, [0, 2])
    res = core.find_blobs(images, M_max=256, want_processed=True)
    ref = c_oracle.BlobOracle(rows, cols, [K, K], dists, [0, 2]).find_blobs(images, M_max=256, want_processed=True)
    assert np.array_equal(res["processed"], ref["processed"])
    over = (res["status"] & capi.BLOB_ST_CAP_OVERFLOW) != 0
    # more borders than the LDS tables of this geometry hold (noise, not blobs): flagged, never silently wrong
    assert (ref["n_contours"][over] > 512).all() and (res["counts"][over] == 0).all() and (res["n_contours"][over] == -1).all()
    ok = ~over
    assert ok.sum() >= 3
    assert np.array_equal(res["n_contours"][ok], ref["n_contours"][ok])
    assert np.array_equal(res["counts"][ok], np.minimum(ref["counts"][ok], 256))
    assert np.array_equal(res["blobs"][ok], ref["blobs"][ok])


@pytest.mark.parametrize("rows,cols", [(240, 352), (480, 640), (600, 832)])
def test_dark_tile_early_out_on_wide_frames(core, rows, cols):
    """Rows wider than 1024 bytes (cols > 341): the pre-pass's activity map has more than 64 segments per
    row.  Sparse frames (the default configuration: early-out on, no processed output) must give the same
    centroids with the early-out on and off and equal the C oracle -- dots are placed in the right-hand part
    of the frame, whose segments a 64-entry activity row would never have written (round-1 advisor finding)."""
    from oracle import c_oracle
    rng = np.random.default_rng(cols)
    C, F = 2, 2
    K = np.array([[cols * 1.0, 0, cols / 2], [0, cols * 1.0, cols / 2], [0, 0, 1]])
    images = np.zeros((F, C, rows, cols, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:rows, 0:cols]
    for f in range(F):
        for c in range(C):
            img = np.zeros((rows, cols), dtype=np.float32)
            for k in range(10):
                # most dots beyond byte 1024 of a row, a few on the left
                cx = rng.uniform(345, cols - 8) if (k < 8 and cols > 360) else rng.uniform(8, cols - 8)
                cy, sg = rng.uniform(8, rows - 8), rng.uniform(1.0, 3.0)
                img += 400 * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * sg * sg))
            images[f, c] = np.clip(img, 0, 255).astype(np.uint8)[..., None]
    images[1, 0] = np.maximum(images[1, 0], rng.integers(0, 3, images[1, 0].shape, dtype=np.uint8))   # range 2: still skippable
    dists = [synth.REFERENCE_DISTORTION, [-0.2, 0.1, 0.002, -0.001, 0.05]]
    core.set_image_params(rows, cols, [K, K], dists, [0, 0])
    try:
        core.set_blob_options(skip_dark_tiles=True)
        on = core.find_blobs(images, M_max=64)
        core.set_blob_options(skip_dark_tiles=False)
        off = core.find_blobs(images, M_max=64)
        core.set_blob_options(skip_dark_tiles=2)        # (round 6) the early-out decided inside the mask pass: no activity pass
        fold = core.find_blobs(images, M_max=64)
    finally:
        core.set_blob_options(skip_dark_tiles=True)
    for k in ("blobs", "counts", "n_contours"):
        assert np.array_equal(fold[k], off[k]), ("fold", k)
    ref = c_oracle.BlobOracle(rows, cols, [K, K], dists, [0, 0]).find_blobs(images, M_max=64)
    assert ref["counts"].min() >= 6
    for k in ("blobs", "counts", "n_contours"):
        assert np.array_equal(on[k], off[k]), k
        assert np.array_equal(on[k], ref[k]), k


def test_argument_errors_mirror_the_reference_domain():
    """Geometries for which the reference's make_square raises (square frames, portrait after rot90, too little
    padding) are rejected, not guessed at; calls before the lens model is set fail loudly."""
    from mocap_core.capi import MocapCore, MocapError
    fresh = MocapCore(0)
    with pytest.raises(MocapError):                             # MOCAP_E_NOCAMS: no lens model yet
        fresh._check(fresh.lib.mocap_find_blobs(fresh._h, 1, None, 4, None, None, None, None, None))
    d = [synth.REFERENCE_DISTORTION]
    for rows, cols, rot in ((320, 320, [0]), (310, 320, [0]), (240, 320, [1]), (240, 320, [3]), (320, 240, [0]), (240, 324, [0])):
        with pytest.raises(MocapError):
            fresh.set_image_params(rows, cols, [REF_K], d, rot)
    fresh.set_image_params(240, 320, [REF_K], d, [2])          # the valid domain still works afterwards
    out = fresh.find_blobs(np.zeros((0, 1, 240, 320, 3), dtype=np.uint8))
    assert out["counts"].shape == (0, 1)
    fresh.close()


def test_blank_and_saturated_frames(core):
    images = np.zeros((1, 2, 240, 320, 3), dtype=np.uint8)
    images[0, 1] = 255                             # one huge blob touching every border of the frame area
    res = _check_against_oracle(core, images, [REF_K, REF_K], [synth.REFERENCE_DISTORTION] * 2)
    assert res["counts"][0, 0] == 0


def test_images_chain_into_frame_path_on_device(core):
    """find_blobs_dev writes the frame path's input layout: images -> blobs -> 3-D points without
    leaving HBM, and the points are the markers."""
    import torch
    rig = synth.ring_rig(4)
    F, M = 6, 6
    images, truth = synth.render_camera_frames(rig, F, M, seed=21, dropout=0.0)
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    core.set_image_params(240, 320, rig["K"], [synth.REFERENCE_DISTORTION] * 4)
    dev = torch.device("cuda", 0)
    K_max, M_max = 32, 16
    d_img = torch.from_numpy(images).to(dev)
    d_blobs = torch.zeros((F, 4, M_max, 2), dtype=torch.float32, device=dev)
    d_counts = torch.zeros((F, 4), dtype=torch.int32, device=dev)
    d_bst = torch.zeros((F, 4), dtype=torch.int32, device=dev)
    d_xyz = torch.zeros((F, K_max, 3), dtype=torch.float64, device=dev)
    d_err = torch.zeros((F, K_max), dtype=torch.float64, device=dev)
    d_corr = torch.zeros((F, K_max, 4), dtype=torch.int16, device=dev)
    d_n = torch.zeros(F, dtype=torch.int32, device=dev)
    d_st = torch.zeros(F, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    core.find_blobs_dev(F, d_img.data_ptr(), M_max, d_blobs.data_ptr(), d_counts.data_ptr(), d_bst.data_ptr())
    core.match_triangulate_dev(F, M_max, d_blobs.data_ptr(), d_counts.data_ptr(), 5.0, K_max, 1 << 20,
                               d_xyz.data_ptr(), d_err.data_ptr(), d_corr.data_ptr(), d_n.data_ptr(), d_st.data_ptr())
    core.synchronize()
    host = core.find_blobs(images, M_max=M_max)
    assert np.array_equal(d_counts.cpu().numpy(), host["counts"])
    assert np.array_equal(d_blobs.cpu().numpy(), host["blobs"])
    n = d_n.cpu().numpy()
    xyz = d_xyz.cpu().numpy()
    assert not d_st.cpu().numpy().any()
    hit = 0
    for f in range(F):
        pts = xyz[f, :n[f]]
        for X in truth["points_cam0"][f]:
            hit += np.linalg.norm(pts - X, axis=1).min() < 0.03      # int() centroids: ~1 px at 3 m
    assert hit >= 0.8 * F * M, hit                                    # merged spots lose a marker
