"""GPU: the live loop's body in one call (reference helpers.py:94-133): mocap_track_frame /
mocap_track_frame_images / helpers.track_frame / helpers.object_points_payload against what the reference's own code
returned for the same frames (tests/golden/track_apptsx_chain.npz: find_point_correspondance_and_object_points -> the
world-coordinate loop -> locate_objects on the reference UI's rig), against the separate entry points bit for bit,
and the C-level re-submit (mocap_match_triangulate_auto)."""
import json
import os

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture()
def chain(core):
    g = load_golden("track_apptsx_chain")
    core.set_cameras(g["K"], g["R"], g["t"])
    core.set_world_transform(g["to_world"])
    yield g
    core.set_world_transform(None)


def test_track_frame_vs_reference_chain_golden(core, chain):
    g = chain
    F = g["blobs"].shape[0]
    objects = 0
    for f in range(F):
        res = core.track_frame(g["blobs"][f:f + 1], g["counts"][f:f + 1], gate_px=5.0, O_max=8)
        assert int(res["status"][0]) == 0
        k = int(g["ref_n"][f])
        assert int(res["n_pts"][0]) == k, f
        if not k:
            assert int(res["n_obj"][0]) == 0
            continue
        # points: north_star's 1e-5 relative; errors 1e-3 (a float32 rounding of cv.projectPoints can flip)
        scale = np.abs(g["ref_world"][f, :k]).max()
        assert np.abs(res["xyz"][0, :k] - g["ref_world"][f, :k]).max() <= 1e-5 * scale, f
        np.testing.assert_allclose(res["err"][0, :k], g["ref_err"][f, :k], rtol=1e-3, atol=1e-9)
        no = int(g["ref_nobj"][f])
        assert int(res["n_obj"][0]) == no, f
        objects += no
        assert np.array_equal(res["droneIndex"][0, :no], g["ref_drone"][f, :no])
        np.testing.assert_allclose(res["pos"][0, :no], g["ref_pos"][f, :no], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(res["heading"][0, :no], g["ref_heading"][f, :no], rtol=0, atol=1e-5)
        np.testing.assert_allclose(res["error"][0, :no], g["ref_error"][f, :no], rtol=1e-3, atol=1e-9)
    assert objects > 40        # the golden really holds drone patterns


def test_track_frame_equals_the_separate_entry_points_bitwise(core, chain):
    g = chain
    F = g["blobs"].shape[0]
    K = 48
    one = core.track_frame(g["blobs"], g["counts"], K_max=K, O_max=8)          # all frames in one call
    sep = core.match_triangulate(g["blobs"], g["counts"], K_max=K)
    assert np.array_equal(one["n_pts"], sep["n_out"]) and np.array_equal(one["status"], sep["status"])
    valid = np.arange(K)[None, :] < sep["n_out"][:, None]
    for key in ("xyz", "err", "corr"):
        assert np.array_equal(one[key][valid], sep[key][valid]), key
    assert np.isnan(one["xyz"][~valid]).all()                                   # slots beyond n_pts keep the caller's fill
    loc = core.locate_objects(sep["xyz"], sep["err"], sep["n_out"], O_max=8)
    assert np.array_equal(one["n_obj"], loc["n_obj"])
    have = np.arange(8)[None, :] < loc["n_obj"][:, None]
    for key in ("pos", "heading", "error", "droneIndex"):
        assert np.array_equal(one[key][have], loc[key][have]), key
    off = core.track_frame(g["blobs"][:5], g["counts"][:5], K_max=K, O_max=0)   # is_locating_objects off
    assert np.array_equal(off["xyz"][valid[:5]], sep["xyz"][:5][valid[:5]]) and not off["n_obj"].any()


def test_wave_and_lane_object_search_agree(core):
    """locate_objects: one wave per frame (small batches, the live path) and one lane per frame (big batches) are the same
    function -- bit for bit, up to the 256 points per frame both accept."""
    from mocap_core import synth
    for k_max, n in ((24, 700), (200, 40), (256, 12)):
        xyz, err, n_pts = synth.make_object_frames(n, k_max, seed=5 + k_max)
        out = {}
        for which in ("lane", "wave"):
            os.environ["MOCAP_LOCATE_KERNEL"] = which
            try:
                out[which] = core.locate_objects(xyz, err, n_pts, O_max=12)
            finally:
                del os.environ["MOCAP_LOCATE_KERNEL"]
        assert out["lane"]["n_obj"].sum() > 0
        for key in ("n_obj", "pos", "heading", "error", "droneIndex", "lead"):
            assert np.array_equal(out["lane"][key], out["wave"][key], equal_nan=True), (k_max, key)


def test_object_points_payload_is_the_reference_event(core, chain):
    """helpers.track_frame + object_points_payload: the dict of helpers.py:128-133, JSON-serialisable like the socket
    event, from the reference's own nested-list frame."""
    from mocap_core import helpers, synth
    g = chain
    helpers.set_core(core)
    helpers.set_camera_params([{"intrinsic_matrix": g["K"][i].tolist()} for i in range(4)])
    helpers.set_to_world_coords_matrix(g["to_world"])
    try:
        poses = [{"R": g["R"][i].tolist(), "t": g["t"][i].tolist()} for i in range(4)]
        f = int(np.argmax(g["ref_nobj"]))
        ip = synth.frame_to_reference_lists(g["blobs"][f], g["counts"][f], as_int=True)
        errors, object_points, objects = helpers.track_frame(ip, poses)
        payload = helpers.object_points_payload(errors, object_points, objects, filtered_objects=[])
        assert list(payload) == ["object_points", "errors", "objects", "filtered_objects"]
        json.dumps(payload)                                                      # what socketio.emit serialises
        k, no = int(g["ref_n"][f]), int(g["ref_nobj"][f])
        assert len(payload["object_points"]) == k and len(payload["errors"]) == k and len(payload["objects"]) == no
        np.testing.assert_allclose(np.array(payload["object_points"]), g["ref_world"][f, :k], rtol=1e-5, atol=1e-7)
        for j, o in enumerate(payload["objects"]):
            assert set(o) == {"pos", "heading", "error", "droneIndex"} and isinstance(o["pos"], list)
            assert o["droneIndex"] == int(g["ref_drone"][f, j])
            np.testing.assert_allclose(o["pos"], g["ref_pos"][f, j], rtol=1e-5, atol=1e-7)
        # a frame in which no camera saw anything: empty lists, like np.array([]).tolist() upstream
        errors, object_points, objects = helpers.track_frame([[[None, None]] for _ in range(4)], poses)
        assert helpers.object_points_payload(errors, object_points, objects) == \
            {"object_points": [], "errors": [], "objects": [], "filtered_objects": []}
    finally:
        helpers.set_to_world_coords_matrix(None)


def test_images_to_payload_chain_equals_the_staged_calls(core):
    """mocap_track_frame_images (raw frames -> blobs -> points -> world -> objects in one enqueue) against the same
    stages called one by one."""
    from mocap_core import synth
    g = load_golden("post_world_locate")
    C = 4
    rig = synth.ring_rig(C)
    images, _ = synth.render_camera_frames(rig, 3, 6, seed=21)
    dists = [synth.REFERENCE_DISTORTION] * C
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    core.set_image_params(240, 320, rig["K"], dists)
    core.set_world_transform(g["to_world"])
    try:
        one = core.track_frame_images(images, M_max=16, K_max=32, O_max=4)
        bl = core.find_blobs(images, M_max=16)
        assert np.array_equal(one["counts"], bl["counts"]) and np.array_equal(one["blob_status"], bl["status"])
        slot = np.arange(16)[None, None, :] < bl["counts"][:, :, None]
        assert np.array_equal(one["blobs"][slot], bl["blobs"][slot]) and bl["counts"].sum() > 20
        mt = core.match_triangulate(bl["blobs"], bl["counts"], K_max=32)
        assert np.array_equal(one["n_pts"], mt["n_out"]) and mt["n_out"].sum() > 0
        valid = np.arange(32)[None, :] < mt["n_out"][:, None]
        for key in ("xyz", "err", "corr"):
            assert np.array_equal(one[key][valid], mt[key][valid]), key
        loc = core.locate_objects(mt["xyz"], mt["err"], mt["n_out"], O_max=4)
        assert np.array_equal(one["n_obj"], loc["n_obj"])
    finally:
        core.set_world_transform(None)


def test_c_level_resubmit_of_overflowed_frames(core):
    """mocap_match_triangulate_auto: frames over a cap come back as if the caps had been large from the start -- for any
    caller of the C ABI (the re-submit used to live in Python); a frame that needs more slots than K_max says how many."""
    from mocap_core import capi, synth
    rig = synth.ring_rig(8)
    blobs, counts, _ = synth.make_blob_stream(rig, 300, 16, seed=3)
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    want = core.match_triangulate(blobs, counts, K_max=128, G_cap=1 << 24)
    assert not want["status"].any()
    tight = core.match_triangulate(blobs, counts, K_max=128, G_cap=64)
    assert (tight["status"] & capi.ST_CAND_OVERFLOW).any()
    got = core.match_triangulate_auto(blobs, counts, K_max=128, G_cap=64)
    assert got["resubmitted"] == int(np.count_nonzero(tight["status"])) and not got["status"].any()
    valid = np.arange(128)[None, :] < want["n_out"][:, None]
    assert np.array_equal(got["n_out"], want["n_out"])
    for key in ("xyz", "err", "corr"):
        assert np.array_equal(got[key][valid], want[key][valid]), key
    # root capacity: K_max = 8 is too small for 16-marker frames -> the C entry reports the need, the binding grows
    small = core.match_triangulate_auto(blobs[:20], counts[:20], K_max=8)
    assert small["xyz"].shape[1] == int(want["n_out"][:20].max()) and not small["status"].any()
    v20 = np.arange(small["xyz"].shape[1])[None, :] < want["n_out"][:20, None]
    assert np.array_equal(small["xyz"][v20], want["xyz"][:20, :small["xyz"].shape[1]][v20])
    # the live call re-submits by itself too
    f = int(np.nonzero(tight["status"])[0][0])
    live = core.track_frame(blobs[f:f + 1], counts[f:f + 1], K_max=128, G_cap=64, O_max=0)
    assert int(live["status"][0]) == 0 and int(live["n_pts"][0]) == int(want["n_out"][f])
    k = int(want["n_out"][f])
    assert np.array_equal(live["xyz"][0, :k], want["xyz"][f, :k])


def test_double_precision_centroids_at_the_boundary(core):
    """mocap_match_triangulate_f64: float64 blob arrays (the reference measures on whatever its lists hold,
    helpers.py:367-373).  float32-representable coordinates -- the reference's own int() centroids -- give bit for bit
    the float32 entry's results; sub-pixel float64 centroids are rounded to the nearest float32 (frames flagged
    ST_ROUNDED, outputs valid) and agree with the C oracle run on the rounded coordinates exactly and with the
    unrounded ones to north_star's 1e-5."""
    from mocap_core import capi, synth
    from oracle import c_oracle
    rig = synth.ring_rig(8)
    blobs, counts, _ = synth.make_blob_stream(rig, 200, 16, seed=9)
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    want = core.match_triangulate_auto(blobs, counts, K_max=64)
    got = core.match_triangulate_f64(blobs.astype(np.float64), counts, K_max=64)
    assert not got["status"].any()
    valid = np.arange(64)[None, :] < want["n_out"][:, None]
    assert np.array_equal(got["n_out"], want["n_out"])
    for key in ("xyz", "err", "corr"):
        assert np.array_equal(got[key][valid], want[key][valid]), key
    # sub-pixel float64 centroids
    sub, counts2, _ = synth.make_blob_stream(rig, 200, 16, seed=10, truncate=False)
    sub64 = sub.astype(np.float64) + np.random.default_rng(1).uniform(-1e-5, 1e-5, sub.shape)
    sub64[np.isnan(sub64)] = 0.0
    got = core.match_triangulate_f64(sub64, counts2, K_max=64)
    assert (got["status"] == capi.ST_ROUNDED).all()
    ref = c_oracle.COracle(rig["K"], rig["R"], rig["t"]).match_triangulate(sub64.astype(np.float32), counts2, K_max=64)
    valid = np.arange(64)[None, :] < ref["n_out"][:, None]
    assert np.array_equal(got["n_out"], ref["n_out"]) and np.array_equal(got["corr"][valid], ref["corr"][valid])
    np.testing.assert_allclose(got["xyz"][valid], ref["xyz"][valid], rtol=1e-9, atol=1e-12)
    with pytest.raises(capi.MocapError, match="NaN"):
        bad = sub64.copy()
        bad[3, 2, 0, 1] = np.nan
        core.match_triangulate_f64(bad, counts2, K_max=64)


def test_track_frame_without_object_search_takes_more_than_256_points(core):
    """The export wave stages points through LDS only for the object search (256 points, refused beyond by the host);
    without it (Cameras.is_locating_objects off) a frame may carry up to 1 024 roots straight through."""
    from mocap_core import synth
    rig = synth.ring_rig(8)
    blobs, counts, _ = synth.make_blob_stream(rig, 6, 48, seed=14, dropout=0.5)      # heavy dropout: many roots beyond camera 0
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    K = 384
    sep = core.match_triangulate(blobs, counts, K_max=K, G_cap=1 << 22)
    one = core.track_frame(blobs, counts, K_max=K, G_cap=1 << 22, O_max=0)
    ok = sep["status"] == 0
    assert ok.any() and np.array_equal(one["status"], sep["status"]) and np.array_equal(one["n_pts"], sep["n_out"])
    valid = (np.arange(K)[None, :] < sep["n_out"][:, None]) & ok[:, None]
    for key in ("xyz", "err", "corr"):
        assert np.array_equal(one[key][valid], sep[key][valid]), key
    with pytest.raises(Exception, match="256"):
        core.track_frame(blobs, counts, K_max=K, O_max=4)                                # with the search on: refused, not overrun


# ----------------------------------------------------------------------------- re-submit on the device (round 5)
def _dev_call(core, fn, blobs, counts, gate, K, G_cap, with_info=True):
    """Run a _dev frame entry point on torch buffers; returns numpy copies after a synchronise."""
    import torch
    dev = torch.device("cuda", 0)
    F, C, M, _ = blobs.shape
    d_b, d_c = torch.from_numpy(blobs).to(dev), torch.from_numpy(counts).to(dev)
    o = dict(xyz=torch.full((F, K, 3), float("nan"), dtype=torch.float64, device=dev),
             err=torch.full((F, K), float("nan"), dtype=torch.float64, device=dev),
             corr=torch.full((F, K, C), -1, dtype=torch.int16, device=dev), n_out=torch.zeros(F, dtype=torch.int32, device=dev),
             status=torch.zeros(F, dtype=torch.int32, device=dev), n_cand=torch.zeros(F, dtype=torch.int32, device=dev),
             info=torch.full((2,), -1, dtype=torch.int32, device=dev))
    torch.cuda.synchronize(dev)
    extra = (o["info"].data_ptr(),) if with_info else ()
    fn(F, M, d_b.data_ptr(), d_c.data_ptr(), gate, K, G_cap, o["xyz"].data_ptr(), o["err"].data_ptr(), o["corr"].data_ptr(),
       o["n_out"].data_ptr(), o["status"].data_ptr(), o["n_cand"].data_ptr(), *extra)
    core.synchronize()
    return {k: v.cpu().numpy() for k, v in o.items()}


def test_device_side_resubmit_equals_an_uncapped_run(core):
    """mocap_match_triangulate_dev_auto: frames over the candidate cap are gathered, re-run with the largest caps and
    scattered back ON THE DEVICE (no host wait): afterwards the batch equals, bit for bit, a run whose caps were large
    from the start -- the reference has no caps at all (helpers.py:394-400).  The plain _dev entry leaves them flagged."""
    from mocap_core import capi, synth
    rig = synth.ring_rig(8)
    blobs, counts, _ = synth.make_blob_stream(rig, 3000, 16, seed=3)
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    want = core.match_triangulate(blobs, counts, K_max=64, G_cap=1 << 24)
    assert not want["status"].any()
    plain = _dev_call(core, core.match_triangulate_dev, blobs, counts, 5.0, 64, 64, with_info=False)
    n_flag = int(np.count_nonzero(plain["status"]))
    assert n_flag > 20 and ((plain["status"] & capi.ST_CAND_OVERFLOW) != 0).sum() == n_flag and not plain["n_out"][plain["status"] != 0].any()
    got = _dev_call(core, core.match_triangulate_dev_auto, blobs, counts, 5.0, 64, 64)
    assert got["info"].tolist() == [n_flag, n_flag]
    assert not got["status"].any() and np.array_equal(got["n_out"], want["n_out"]) and np.array_equal(got["n_cand"], want["n_cand"])
    valid = np.arange(64)[None, :] < want["n_out"][:, None]
    for key in ("xyz", "err", "corr"):
        assert np.array_equal(got[key][valid], want[key][valid]), key
    assert np.isnan(got["xyz"][~valid]).all()                    # nothing written beyond the valid slots
    # nothing flagged: the three extra enqueues find an empty list
    clean = _dev_call(core, core.match_triangulate_dev_auto, blobs, counts, 5.0, 64, 1 << 24)
    assert clean["info"].tolist() == [0, 0] and np.array_equal(clean["xyz"][valid], want["xyz"][valid])
    # twice in a row (the two alternating device counters), the second time flagged again
    again = _dev_call(core, core.match_triangulate_dev_auto, blobs, counts, 5.0, 64, 64)
    assert again["info"].tolist() == [n_flag, n_flag] and np.array_equal(again["xyz"][valid], want["xyz"][valid])


def test_device_side_resubmit_root_capacity(core):
    """Root capacity: K_max = 14 on 16-marker frames with 60 % dropout -- many frames have more ROOTS than slots (single-view
    blobs are roots that yield no point).  The second pass runs with C * M root slots in scratch: frames whose POINTS fit
    K_max come back complete, the others report ROOT_OVERFLOW and how many slots they need (n_out > K_max; nothing of
    them is written, consumers count them as empty)."""
    from mocap_core import capi, synth
    from mocap_core import dist as mdist
    rig = synth.ring_rig(8)
    blobs, counts, _ = synth.make_blob_stream(rig, 400, 16, seed=13, dropout=0.6)
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    want = core.match_triangulate(blobs, counts, K_max=128, G_cap=1 << 24)
    K = 14
    plain = _dev_call(core, core.match_triangulate_dev, blobs, counts, 5.0, K, 1 << 20, with_info=False)
    flagged = (plain["status"] & capi.ST_ROOT_OVERFLOW) != 0
    fits = want["n_out"] <= K
    assert (flagged & fits).sum() > 50 and (flagged & ~fits).sum() > 50 and (~flagged).sum() > 20    # (C oracle: 106 / 233 / 61)
    assert not (~flagged & ~fits).any()
    got = _dev_call(core, core.match_triangulate_dev_auto, blobs, counts, 5.0, K, 1 << 20)
    assert got["info"].tolist() == [int(flagged.sum())] * 2
    assert np.array_equal(got["n_out"], want["n_out"])
    assert not got["status"][fits].any() and (got["status"][~fits] == capi.ST_ROOT_OVERFLOW).all()
    valid = (np.arange(K)[None, :] < want["n_out"][:, None]) & fits[:, None]
    for key in ("xyz", "err", "corr"):
        assert np.array_equal(got[key][valid], want[key][:, :K][valid]), key
    assert np.isnan(got["xyz"][~fits]).all()                     # frames that need more slots: untouched
    # downstream consumers treat n_out > K_max as "no valid slot"
    rec, offsets = mdist.compact_tracks_reference(got["n_out"], got["xyz"], got["err"], got["corr"])
    assert offsets[-1] == int(want["n_out"][fits].sum()) == rec.shape[0]
    # the host-buffer form: the same, and the binding grows K_max to what the core says the frames need
    auto = core.match_triangulate_auto(blobs, counts, K_max=K)
    kk = int(want["n_out"].max())
    assert auto["xyz"].shape[1] == kk and not auto["status"].any() and np.array_equal(auto["n_out"], want["n_out"])
    vk = np.arange(kk)[None, :] < want["n_out"][:, None]
    assert np.array_equal(auto["xyz"][vk], want["xyz"][:, :kk][vk]) and np.array_equal(auto["corr"][vk], want["corr"][:, :kk][vk])
    # the live call: roots > K_max, points <= K_max: repaired in place ...
    g = int(np.nonzero(flagged & fits & (want["n_out"] > 0))[0][0])
    live = core.track_frame(blobs[g:g + 1], counts[g:g + 1], K_max=K, O_max=0)
    k = int(want["n_out"][g])
    assert int(live["status"][0]) == 0 and live["xyz"].shape[1] == K and np.array_equal(live["xyz"][0, :k], want["xyz"][g, :k])
    # ... points > K_max: the core says how many, the binding calls again with that
    f = int(np.nonzero(~fits)[0][0])
    live = core.track_frame(blobs[f:f + 1], counts[f:f + 1], K_max=K, O_max=0)
    k = int(want["n_out"][f])
    assert int(live["status"][0]) == 0 and int(live["n_pts"][0]) == k and live["xyz"].shape[1] == k
    assert np.array_equal(live["xyz"][0, :k], want["xyz"][f, :k])


def test_device_side_resubmit_at_the_stress_shape_and_scratch_budget(core, monkeypatch):
    """64 cameras x 256 blobs through the wide variant: candidate-cap overflow repaired on the device = the host-buffer
    re-submit.  Then with a scratch budget that holds only a few frames: the frames that did not fit keep their status,
    the counters say so (flagged > re-run), and a second call with the default budget repairs them."""
    from mocap_core import capi, synth
    rig = synth.stress_rig(64)
    blobs, counts, _ = synth.make_stress_stream(rig, 10, 256, seed=31)
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    gate = synth.STRESS_GATE_PX
    auto = core.match_triangulate_auto(blobs, counts, gate_px=gate, K_max=384, G_cap=2)
    assert auto["resubmitted"] == 10
    got = _dev_call(core, core.match_triangulate_dev_auto, blobs, counts, gate, 384, 2)
    assert core.last_frame_kernel() in ("frame_kernel<512, wide>", "frame_kernel<1024, wide>")
    assert got["info"].tolist() == [10, 10] and np.array_equal(got["status"], auto["status"]) and np.array_equal(got["n_out"], auto["n_out"])
    ok = auto["status"] == 0
    assert ok.sum() >= 8
    valid = (np.arange(384)[None, :] < auto["n_out"][:, None]) & ok[:, None]
    for key in ("xyz", "err", "corr"):
        assert np.array_equal(got[key][valid], auto[key][valid]), key
    monkeypatch.setenv("MOCAP_RESUBMIT_SCRATCH_MB", "1")          # ~ 3 frames of this shape
    part = _dev_call(core, core.match_triangulate_dev_auto, blobs, counts, gate, 384, 2)
    flagged, rerun = part["info"].tolist()
    assert flagged == 10 and 1 <= rerun < 10
    left = (part["status"] & capi.ST_CAND_OVERFLOW) != 0
    assert 10 - rerun <= left.sum() <= 10 - rerun + (~ok).sum() and not part["n_out"][left].any()
    done = ~left & ok
    v2 = (np.arange(384)[None, :] < auto["n_out"][:, None]) & done[:, None]
    assert np.array_equal(part["xyz"][v2], auto["xyz"][v2])


def test_track_frame_dev_resubmits_before_the_object_search(core):
    """mocap_track_frame_dev: frame kernel -> device-side re-submit -> locate_objects, all queued; equal to the same call
    with caps that never bind."""
    import torch
    from mocap_core import synth
    rig = synth.ring_rig(8)
    blobs, counts, _ = synth.make_blob_stream(rig, 500, 16, seed=3)
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    dev = torch.device("cuda", 0)
    F, C, M, K, O = 500, 8, 16, 64, 4
    d_b, d_c = torch.from_numpy(blobs).to(dev), torch.from_numpy(counts).to(dev)
    outs = {}
    for G in (1 << 24, 64):
        o = dict(xyz=torch.zeros((F, K, 3), dtype=torch.float64, device=dev), err=torch.zeros((F, K), dtype=torch.float64, device=dev),
                 corr=torch.zeros((F, K, C), dtype=torch.int16, device=dev), n=torch.zeros(F, dtype=torch.int32, device=dev),
                 st=torch.zeros(F, dtype=torch.int32, device=dev), pos=torch.zeros((F, O, 3), dtype=torch.float64, device=dev),
                 head=torch.zeros((F, O), dtype=torch.float64, device=dev), oerr=torch.zeros((F, O), dtype=torch.float64, device=dev),
                 drone=torch.zeros((F, O), dtype=torch.int32, device=dev), nobj=torch.zeros(F, dtype=torch.int32, device=dev))
        core.track_frame_dev(F, M, d_b.data_ptr(), d_c.data_ptr(), 5.0, K, G, o["xyz"].data_ptr(), o["err"].data_ptr(),
                             o["corr"].data_ptr(), o["n"].data_ptr(), o["st"].data_ptr(), O, o["pos"].data_ptr(), o["head"].data_ptr(),
                             o["oerr"].data_ptr(), o["drone"].data_ptr(), o["nobj"].data_ptr())
        core.synchronize()
        outs[G] = {k: v.cpu().numpy() for k, v in o.items()}
    a, b = outs[1 << 24], outs[64]
    assert not a["st"].any() and not b["st"].any() and np.array_equal(a["n"], b["n"]) and np.array_equal(a["nobj"], b["nobj"])
    valid = np.arange(K)[None, :] < a["n"][:, None]
    assert np.array_equal(a["xyz"][valid], b["xyz"][valid]) and np.array_equal(a["corr"][valid], b["corr"][valid])


def test_resubmit_continues_where_a_small_scratch_stopped(core, monkeypatch):
    """Round-5 advice: the re-submit's scratch batch is sized for the flagged share one expects (one frame in eight), and a
    batch with more flagged frames must still be repaired.  With a scratch of 7 frames (MOCAP_RESUBMIT_SCRATCH_FRAMES): the
    device entry repairs 7 and says so; mocap_resubmit_dev continues with the rest -- every call makes progress, repaired
    frames are not picked again -- until the batch equals, bit for bit, a run whose caps never bound.  The host-buffer entry
    loops by itself."""
    import torch
    from mocap_core import devcheck, synth
    rig = synth.ring_rig(8)
    blobs, counts, _ = synth.make_blob_stream(rig, 3000, 16, seed=3)
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    dev = torch.device("cuda", 0)
    d_b, d_c = torch.from_numpy(blobs).to(dev), torch.from_numpy(counts).to(dev)
    want = devcheck.FrameOutputs(3000, 64, 8, dev)
    want.run(core, 16, d_b, d_c, 5.0, 1 << 24)
    core.synchronize()
    assert not want.status.any().item()
    monkeypatch.setenv("MOCAP_RESUBMIT_SCRATCH_FRAMES", "7")
    got = devcheck.FrameOutputs(3000, 64, 8, dev)
    got.run(core, 16, d_b, d_c, 5.0, 64)
    core.synchronize()
    flagged, rerun = got.info.cpu().tolist()
    assert flagged > 20 and rerun == 7
    assert int((got.status != 0).sum().item()) == flagged - 7 and not got.n_out[got.status != 0].any().item()
    rounds, left = 0, flagged - 7
    while left > 0:
        core.resubmit_dev(3000, 16, d_b.data_ptr(), d_c.data_ptr(), 5.0, 64, got.xyz.data_ptr(), got.err.data_ptr(), got.corr.data_ptr(),
                          got.n_out.data_ptr(), got.status.data_ptr(), got.n_cand.data_ptr(), got.info.data_ptr())
        core.synchronize()
        f2, r2 = got.info.cpu().tolist()
        assert f2 == left and r2 == min(left, 7)           # only the frames still flagged are picked; each round takes 7
        left -= r2
        rounds += 1
    assert rounds == -(-(flagged - 7) // 7)
    cmp = devcheck.compare_bitwise(got, want)
    assert cmp["frames_differing"] == 0, cmp
    host = core.match_triangulate_auto(blobs, counts, K_max=64, G_cap=64)
    assert host["resubmitted"] == flagged and not host["status"].any()
    valid = np.arange(64)[None, :] < host["n_out"][:, None]
    assert np.array_equal(host["n_out"], want.n_out.cpu().numpy()) and np.array_equal(host["xyz"][valid], want.xyz.cpu().numpy()[valid])
