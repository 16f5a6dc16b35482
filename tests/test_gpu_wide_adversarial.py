"""GPU: the wide variant of the frame path (csrc/frame_kernel.hip match_pairs_wide; BASELINE.json configs[4], 64 cameras x
256 markers) drops most (root, camera, blob) triples on a float32 pre-test with a rigorous error bound before the exact
double decision of helpers.py:373,375.  "Dropped without evaluating" is where a bit-exact path can go wrong silently, so
it is attacked here the way tests/test_gpu_bb_adversarial.py attacks the branch and bound:
  * 256 stress frames against the C oracle (indices exact, points to 1e-9);
  * blobs placed on float32 neighbours either side of distance == gate, duplicate pixels, heavy dropout, rigs in
    millimetres, 16 k-pixel coordinates, random rigs (Hypothesis), all against the C oracle;
  * a self-check build (-DMOCAP_DEBUG_PRETEST) that takes the exact decision on the device for every blob the pre-test
    rejected and counts false negatives (must be 0; counters prove it ran);
  * candidate-cap overflow at the stress shape, repaired by the re-submit path."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compare(core, rig, blobs, counts, gate, K_max, G_cap=1 << 20, force_wide=False, min_points=1):
    from oracle import c_oracle
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    try:
        if force_wide:
            core.set_frame_limits(hit_cap=32, force_wide=True)
        res = core.match_triangulate_auto(blobs, counts, gate_px=gate, K_max=K_max, G_cap=G_cap)
        assert core.last_frame_kernel() in ("frame_kernel<512, wide>", "frame_kernel<1024, wide>"), core.last_frame_kernel()
    finally:
        if force_wide:
            core.set_frame_limits(hit_cap=32, force_wide=False)
    # the oracle walks every candidate group (as the reference does): frames above 2^20 groups for one root would take it
    # minutes each and are left out of the comparison (status 2 on its side)
    ref = c_oracle.COracle(rig["K"], rig["R"], rig["t"]).match_triangulate(blobs, counts, gate_px=gate, K_max=K_max, G_cap=1 << 20)
    ok = ref["status"] == 0
    assert ok.any()
    assert np.array_equal(res["status"][ok], ref["status"][ok])
    assert np.array_equal(res["n_out"][ok], ref["n_out"][ok]) and ref["n_out"][ok].sum() >= min_points
    assert np.array_equal(res["n_cand"][ok], ref["n_cand"][ok])
    kk = min(res["corr"].shape[1], ref["corr"].shape[1])
    valid = (np.arange(kk)[None, :] < ref["n_out"][:, None]) & ok[:, None]
    assert np.array_equal(res["corr"][:, :kk][valid], ref["corr"][:, :kk][valid])          # bit-exact indices
    np.testing.assert_allclose(res["xyz"][:, :kk][valid], ref["xyz"][:, :kk][valid], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(res["err"][:, :kk][valid], ref["err"][:, :kk][valid], rtol=1e-3, atol=1e-12)
    return res, ref


def test_256_stress_frames_vs_c_oracle(core):
    """BASELINE.json configs[4] at its own shape, 256 frames (round 3 checked 3)."""
    from mocap_core import synth
    rig = synth.stress_rig(64)
    blobs, counts, _ = synth.make_stress_stream(rig, 256, 256, seed=4242)
    res, ref = _compare(core, rig, blobs, counts, synth.STRESS_GATE_PX, K_max=384, min_points=256 * 200)
    assert ref["n_cand"].max() > 3000          # frames with real multi-hit pairs are in the set


def _gate_straddlers(rig, blobs, counts, gate, rng, pairs_per_frame=6):
    """Overwrite blobs of cameras i >= 1 with float32 points whose exact distance (helpers.py:373) from the epipolar line
    of a camera-0 blob is just below / just at-or-above the gate: for each chosen (root, camera) two float32 neighbours
    x_lo, x_hi (same y) with d(x_lo) < gate <= d(x_hi), found by bisection over the float32 ordinals."""
    from oracle import mocap_oracle as mo
    F, C, M, _ = blobs.shape
    Ftab = mo.fundamental_table(rig["K"], rig["R"], rig["t"])
    made = 0
    for f in range(F):
        for _ in range(pairs_per_frame):
            r = int(rng.integers(0, max(1, counts[f, 0])))
            i = int(rng.integers(1, C))
            if counts[f, 0] == 0 or counts[f, i] < 4:
                continue
            a, b, c = mo.epiline(Ftab[0, i], blobs[f, 0, r])
            if abs(a) < 0.05:
                continue
            den = np.sqrt(a ** 2 + b ** 2)

            def dist(x32, y32):
                return abs(a * float(x32) + b * float(y32) + c) / den

            y = np.float32(rng.uniform(2000, 14000)) if blobs[f].max() > 1000 else np.float32(rng.uniform(40, 280))
            side = 1.0 if rng.random() < 0.5 else -1.0
            x_on = (-(b * float(y) + c)) / a                       # on the line
            x_in = np.float32(x_on + side * 0.5 * gate * den / abs(a))
            x_out = np.float32(x_on + side * 1.5 * gate * den / abs(a))
            if not (dist(x_in, y) < gate <= dist(x_out, y)):
                continue
            lo, hi = x_in.view(np.int32).item(), x_out.view(np.int32).item()
            if (lo < 0) != (hi < 0):
                continue
            step = 1 if hi > lo else -1
            while abs(hi - lo) > 1:
                mid = (lo + hi) // 2
                if dist(np.int32(mid).view(np.float32), y) < gate:
                    lo = mid
                else:
                    hi = mid
            x_lo, x_hi = np.int32(lo).view(np.float32), np.int32(hi).view(np.float32)
            assert dist(x_lo, y) < gate <= dist(x_hi, y) and step
            k = int(counts[f, i]) - 1 - 2 * int(rng.integers(0, 2))     # replace two existing blobs near the end
            blobs[f, i, k] = (x_lo, y)
            blobs[f, i, k - 1] = (x_hi, y)
            made += 1
    return made


@pytest.mark.parametrize("shape", ["stress64x256", "px320_forced_wide"])
def test_blobs_on_either_side_of_the_gate(core, shape):
    from mocap_core import synth
    rng = np.random.default_rng(7)
    if shape == "stress64x256":
        rig = synth.stress_rig(64)
        blobs, counts, _ = synth.make_stress_stream(rig, 24, 256, seed=77)
        gate, K_max, fw = synth.STRESS_GATE_PX, 512, False
    else:
        rig = synth.ring_rig(8)
        blobs, counts, _ = synth.make_blob_stream(rig, 200, 16, seed=78)
        gate, K_max, fw = 5.0, 128, True
    made = _gate_straddlers(rig, blobs, counts, gate, rng)
    assert made > 60
    _compare(core, rig, blobs, counts, gate, K_max=K_max, G_cap=1 << 24, force_wide=fw)


def test_duplicate_pixels_and_heavy_dropout(core):
    """Duplicate coordinates inside a camera (claimed by value, helpers.py:391) and 50 % dropout (many roots from cameras
    after the first: the chain that creates roots is where the wide variant differs most from the narrow one)."""
    from mocap_core import synth
    rig = synth.stress_rig(64)
    blobs, counts, _ = synth.make_blob_stream(rig, 16, 256, seed=5, noise_px=0.02, dropout=0.5, half_extent=1.5, min_sep=0.05,
                                              truncate=False)
    rng = np.random.default_rng(9)
    for f in range(blobs.shape[0]):
        for i in rng.choice(64, size=20, replace=False):
            n = int(counts[f, i])
            if n >= 6:
                a, b = rng.choice(n, size=2, replace=False)
                blobs[f, i, a] = blobs[f, i, b]
    res, ref = _compare(core, rig, blobs, counts, synth.STRESS_GATE_PX, K_max=600, G_cap=1 << 24)
    assert ref["n_out"].max() > 256            # kept roots created by cameras 1..: more points than camera 0 has blobs


def _run_wide(core_cls, rig, blobs, counts, gate, K_max, hit_cap, spec):
    """One forced-wide pass in a fresh context, with the speculative chain on or off (MOCAP_WIDE_SPEC is read per launch)."""
    old = os.environ.get("MOCAP_WIDE_SPEC")
    os.environ["MOCAP_WIDE_SPEC"] = "1" if spec else "0"
    try:
        core = core_cls(0)
        core.set_cameras(rig["K"], rig["R"], rig["t"])
        core.set_frame_limits(hit_cap=hit_cap, force_wide=True)
        res = core.match_triangulate(blobs, counts, gate_px=gate, K_max=K_max, G_cap=1 << 20)
        assert core.last_frame_kernel() in ("frame_kernel<512, wide>", "frame_kernel<1024, wide>")
        return res
    finally:
        if old is None:
            os.environ.pop("MOCAP_WIDE_SPEC", None)
        else:
            os.environ["MOCAP_WIDE_SPEC"] = old


def test_speculative_chain_on_many_small_frames(core):
    """Round 6: once few blobs are left unclaimed, the wide variant matches all of them at once as PROVISIONAL roots and
    replays the reference's camera-by-camera claims afterwards (frame_kernel.hip spec_begin / spec_finish).  What can go
    wrong there is order and existence -- a provisional root claimed by an earlier provisional root, rows moving down, a
    hit list that holds a duplicate of the closest hit (the frame must fall back to the sequential chain), a hit list over
    the cap on a provisional root that turns out not to exist (must not flag the frame).  Thousands of small forced-wide
    frames with integer pixels (duplicates), dropout (roots from many cameras), wide gates (multi-hit pairs), one run with a
    hit cap of 2: every output bit equal to the strictly sequential chain (MOCAP_WIDE_SPEC=0), and the uncapped runs equal
    to the C oracle."""
    from mocap_core import capi, synth
    from oracle import c_oracle
    for C, M, gate, dropout, seed, hit_cap in ((8, 16, 3.0, 0.3, 11, 32), (12, 12, 5.0, 0.15, 12, 32), (16, 24, 2.0, 0.1, 13, 32),
                                               (8, 16, 1.5, 0.3, 14, 2)):
        K = [[320.0, 0.0, 160.0], [0.0, 320.0, 160.0], [0.0, 0.0, 1.0]]
        rig = synth.ring_rig(C, K=K, image_size=(320, 320))
        blobs, counts, _ = synth.make_blob_stream(rig, 600, M, seed=seed, noise_px=0.3, dropout=dropout, truncate=True)
        res = _run_wide(capi.MocapCore, rig, blobs, counts, gate, C * M, hit_cap, spec=True)
        seq = _run_wide(capi.MocapCore, rig, blobs, counts, gate, C * M, hit_cap, spec=False)
        for k in ("status", "n_out", "n_cand"):
            assert np.array_equal(res[k], seq[k]), (C, M, k, np.flatnonzero(res[k] != seq[k])[:8])
        valid = np.arange(res["corr"].shape[1])[None, :] < res["n_out"][:, None]
        assert np.array_equal(res["corr"][valid], seq["corr"][valid])
        assert np.array_equal(res["xyz"][valid].view(np.uint64), seq["xyz"][valid].view(np.uint64))
        assert np.array_equal(res["err"][valid].view(np.uint64), seq["err"][valid].view(np.uint64))
        if hit_cap == 2:
            assert (res["status"] != 0).sum() > 10 and (res["status"] == 0).sum() > 10   # (the capped run flags frames, not all)
            continue
        ref = c_oracle.COracle(rig["K"], rig["R"], rig["t"]).match_triangulate(blobs, counts, gate_px=gate, K_max=C * M, G_cap=1 << 20)
        assert np.array_equal(res["status"], ref["status"]) and np.array_equal(res["n_out"], ref["n_out"])
        kk = min(res["corr"].shape[1], ref["corr"].shape[1])
        valid = (np.arange(kk)[None, :] < ref["n_out"][:, None]) & (ref["status"] == 0)[:, None]
        assert valid.sum() > 1000
        assert np.array_equal(res["corr"][:, :kk][valid], ref["corr"][:, :kk][valid])
        np.testing.assert_allclose(res["xyz"][:, :kk][valid], ref["xyz"][:, :kk][valid], rtol=1e-9, atol=1e-12)


def test_speculative_chain_equals_the_sequential_one_at_the_stress_shape(core):
    """... and 256 stress frames (64 x 256), where the speculative pass is what runs for nearly every frame."""
    from mocap_core import capi, synth
    rig = synth.stress_rig(64)
    blobs, counts, _ = synth.make_stress_stream_chunked(rig, 256, 256, seed=77)
    res = _run_wide(capi.MocapCore, rig, blobs, counts, synth.STRESS_GATE_PX, 384, 32, spec=True)
    seq = _run_wide(capi.MocapCore, rig, blobs, counts, synth.STRESS_GATE_PX, 384, 32, spec=False)
    for k in ("status", "n_out", "n_cand"):
        assert np.array_equal(res[k], seq[k]), k
    valid = np.arange(res["corr"].shape[1])[None, :] < res["n_out"][:, None]
    assert valid.sum() > 256 * 200
    assert np.array_equal(res["corr"][valid], seq["corr"][valid])
    assert np.array_equal(res["xyz"][valid].view(np.uint64), seq["xyz"][valid].view(np.uint64))
    assert np.array_equal(res["err"][valid].view(np.uint64), seq["err"][valid].view(np.uint64))


def test_rig_in_millimetres_and_tiny_gate(core):
    """World units do not enter the pre-test's bound (the line is normalised), coordinates do: a rig in millimetres with
    16 k-pixel coordinates, and a gate far below the float32 resolution of the coordinates' products."""
    from mocap_core import synth
    rig = synth.ring_rig(64, radius=3000.0, height=1500.0, K=synth.STRESS_K, image_size=(16000, 16000))
    blobs, counts, _ = synth.make_blob_stream(rig, 12, 256, seed=6, noise_px=0.02, dropout=0.05, half_extent=1500.0,
                                              min_sep=50.0, truncate=False)
    _compare(core, rig, blobs, counts, 0.5, K_max=384)
    # float32 coordinates near 8 000 px are quantised to 2^-10 px: a 0.02 px gate is ~20 ulps of a coordinate
    blobs, counts, _ = synth.make_blob_stream(rig, 12, 256, seed=7, noise_px=0.002, dropout=0.05, half_extent=1500.0,
                                              min_sep=50.0, truncate=False)
    _compare(core, rig, blobs, counts, 0.02, K_max=600, G_cap=1 << 24)


def test_random_rigs_forced_wide_vs_c_oracle(core):
    """Hypothesis: random camera counts, blob counts, focal lengths, gates, integer or float centroids -- every batch forced
    through the wide variant and compared with the C oracle."""
    from hypothesis import given, settings, strategies as st, HealthCheck
    from mocap_core import synth

    @settings(max_examples=20, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(C=st.integers(2, 12), M=st.integers(1, 40), f=st.sampled_from([200.0, 320.0, 900.0, 12000.0]),
           gate=st.sampled_from([0.5, 2.0, 5.0]), trunc=st.booleans(), seed=st.integers(0, 10 ** 6),
           dropout=st.sampled_from([0.0, 0.05, 0.4]))
    def run(C, M, f, gate, trunc, seed, dropout):
        K = [[f, 0.0, f / 2], [0.0, f, f / 2], [0.0, 0.0, 1.0]]
        rig = synth.ring_rig(C, K=K, image_size=(int(f), int(f)))
        blobs, counts, _ = synth.make_blob_stream(rig, 12, M, seed=seed, noise_px=0.05 * gate, dropout=dropout,
                                                  truncate=trunc)
        _compare(core, rig, blobs, counts, gate, K_max=min(C * M, 1024), G_cap=1 << 24, force_wide=True, min_points=0)

    run()


def test_candidate_cap_overflow_at_the_stress_shape_is_resubmitted(core):
    """Frames over the candidate cap come back empty with status 2 from the plain call and complete from the re-submit
    (G_cap = 2^24 per root, uncapped hit lists): equal to the oracle, which has no caps."""
    from mocap_core import capi, synth
    from oracle import c_oracle
    rig = synth.stress_rig(64)
    blobs, counts, _ = synth.make_stress_stream(rig, 12, 256, seed=31)
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    tight = core.match_triangulate(blobs, counts, gate_px=synth.STRESS_GATE_PX, K_max=384, G_cap=2)
    assert core.last_frame_kernel() in ("frame_kernel<512, wide>", "frame_kernel<1024, wide>")
    assert (tight["status"] & capi.ST_CAND_OVERFLOW).all() and not tight["n_out"].any()
    auto = core.match_triangulate_auto(blobs, counts, gate_px=synth.STRESS_GATE_PX, K_max=384, G_cap=2)
    ref = c_oracle.COracle(rig["K"], rig["R"], rig["t"]).match_triangulate(blobs, counts, gate_px=synth.STRESS_GATE_PX, K_max=384,
                                                                           G_cap=1 << 20)
    ok = ref["status"] == 0         # (a frame with a root of more than 2^20 groups is left to the core alone: minutes on the CPU)
    assert auto["resubmitted"] == 12 and ok.sum() >= 10 and not auto["status"][ok].any()
    assert np.array_equal(auto["n_out"][ok], ref["n_out"][ok])
    # (a root the second pass hands to the heavy-root search -- more than 4096 groups -- counts one candidate)
    small = ok & (ref["n_cand"] <= 4096)
    assert small.sum() >= 6 and np.array_equal(auto["n_cand"][small], ref["n_cand"][small]) and (auto["n_cand"][ok] <= ref["n_cand"][ok]).all()
    valid = (np.arange(384)[None, :] < ref["n_out"][:, None]) & ok[:, None]
    assert np.array_equal(auto["corr"][valid], ref["corr"][valid])
    np.testing.assert_allclose(auto["xyz"][valid], ref["xyz"][valid], rtol=1e-9, atol=1e-12)
    # hit-list cap: one hit per pair kept -> frames with a multi-hit pair are flagged and repaired
    try:
        core.set_frame_limits(hit_cap=1)
        capped = core.match_triangulate(blobs, counts, gate_px=synth.STRESS_GATE_PX, K_max=384)
        assert (capped["status"] & capi.ST_HIT_OVERFLOW).any()
        rep = core.match_triangulate_auto(blobs, counts, gate_px=synth.STRESS_GATE_PX, K_max=384)
        assert not rep["status"][ok].any() and np.array_equal(rep["corr"][valid], ref["corr"][valid])
    finally:
        core.set_frame_limits(hit_cap=32)


def test_heavy_root_search_equals_the_enumeration(core):
    """csrc/heavy_bb.hip against the enumeration, root by root, where the enumeration is feasible: with
    MOCAP_RESUBMIT_G_CAP=8 the second pass hands every root with more than eight groups to the search (first pass: G_cap = 1,
    every frame with a choice is flagged).  Points, errors and correspondences must equal the plain call's bit for bit: the
    search drops groups on a rigorous bound only and scores the leaves with the path's own device function."""
    import os
    from mocap_core import capi, synth
    rig = synth.stress_rig(64)
    blobs, counts, _ = synth.make_stress_stream(rig, 24, 256, seed=77)
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    plain = core.match_triangulate(blobs, counts, gate_px=synth.STRESS_GATE_PX, K_max=384, G_cap=1 << 20)
    ok = plain["status"] == 0
    assert ok.sum() >= 20
    valid = (np.arange(384)[None, :] < plain["n_out"][:, None]) & ok[:, None]
    # second leg: a frontier of 64 nodes -- the search gives up on some of these roots and its fall-back enumerates them in
    # place (products up to 2^16): the same bits again.  Third leg (round 6): a frontier of ONE node and no in-place fall-back
    # (MOCAP_HEAVY_ENUM_CAP=0) -- every root with a level that keeps two nodes goes to heavy_enum_kernel, the enumeration over the whole
    # GPU that keeps the re-submit exact up to 2^24 groups per root: the same bits once more.
    for ncap, ecap in ((None, None), ("64", None), ("1", "0")):
        os.environ["MOCAP_RESUBMIT_G_CAP"] = "8"
        if ncap:
            os.environ["MOCAP_HEAVY_NCAP"] = ncap
        if ecap:
            os.environ["MOCAP_HEAVY_ENUM_CAP"] = ecap
        try:
            auto = core.match_triangulate_auto(blobs, counts, gate_px=synth.STRESS_GATE_PX, K_max=384, G_cap=1)
        finally:
            del os.environ["MOCAP_RESUBMIT_G_CAP"]
            os.environ.pop("MOCAP_HEAVY_NCAP", None)
            os.environ.pop("MOCAP_HEAVY_ENUM_CAP", None)
        assert auto["resubmitted"] >= 20 and not auto["status"][ok].any(), ncap
        assert np.array_equal(auto["n_out"][ok], plain["n_out"][ok])
        assert np.array_equal(auto["corr"][valid], plain["corr"][valid]), ncap
        assert np.array_equal(auto["xyz"][valid], plain["xyz"][valid]) and np.array_equal(auto["err"][valid], plain["err"][valid]), ncap
        # the search did run: roots with a choice count one candidate each now
        assert (auto["n_cand"][ok] < plain["n_cand"][ok]).any()
    # MOCAP_OPT_BOUNDED_RESUBMIT: the same forced give-ups are NOT enumerated -- their frames stay flagged (candidate overflow +
    # FINAL, not INTRACTABLE: the roots have fewer than 2^24 groups) and report no point; every frame that is returned is exact
    # (with the third leg's settings: that these frames stay flagged HERE is also the proof that the third leg's roots did go
    # through heavy_enum_kernel)
    os.environ.update({"MOCAP_RESUBMIT_G_CAP": "8", "MOCAP_HEAVY_NCAP": "1", "MOCAP_HEAVY_ENUM_CAP": "0"})
    try:
        core.set_options(bounded_resubmit=True)
        bounded = core.match_triangulate_auto(blobs, counts, gate_px=synth.STRESS_GATE_PX, K_max=384, G_cap=1)
    finally:
        core.set_options(bounded_resubmit=False)
        for k in ("MOCAP_RESUBMIT_G_CAP", "MOCAP_HEAVY_NCAP", "MOCAP_HEAVY_ENUM_CAP"):
            del os.environ[k]
    left = bounded["status"] != 0
    assert left.any() and (bounded["status"][left] & (capi.ST_CAND_OVERFLOW | capi.ST_FINAL | capi.ST_INTRACTABLE) == (capi.ST_CAND_OVERFLOW | capi.ST_FINAL)).all()
    assert not bounded["n_out"][left].any()
    done = ok & ~left
    v3 = (np.arange(384)[None, :] < plain["n_out"][:, None]) & done[:, None]
    assert np.array_equal(bounded["n_out"][done], plain["n_out"][done]) and np.array_equal(bounded["xyz"][v3], plain["xyz"][v3])


def test_two_markers_behind_each_other_as_seen_from_camera_0(core):
    """The shape no enumeration finishes (the reference included, helpers.py:394-400): marker B on camera 0's ray through
    marker A -- both lie on the epipolar line of either root in EVERY other camera: two hits per camera, 2^60 groups per
    root.  The plain call flags the frame; the re-submit solves it (heavy-root search) and both roots come back with the
    blobs of their own marker in every camera and the marker's position."""
    from mocap_core import capi, synth
    rig = synth.stress_rig(64)

    def behind(w):
        w = w.copy()
        for f in range(w.shape[0]):
            a0 = w[f, 0] @ rig["R0"].T + rig["centre"]                 # marker 0 in camera-0 coordinates (camera 0 at the origin)
            w[f, 1] = (a0 * (1.15 + 0.05 * f) - rig["centre"]) @ rig["R0"]   # marker 1 further out on the same ray
        return w

    blobs, counts, truth = synth.make_blob_stream(rig, 4, 256, seed=5, noise_px=0.02, dropout=0.0, half_extent=1.5, min_sep=0.05,
                                                  truncate=False, world=behind)
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    plain = core.match_triangulate(blobs, counts, gate_px=synth.STRESS_GATE_PX, K_max=384, G_cap=1 << 24)
    assert (plain["status"] & capi.ST_CAND_OVERFLOW).all()
    auto = core.match_triangulate_auto(blobs, counts, gate_px=synth.STRESS_GATE_PX, K_max=384, G_cap=1 << 20)
    assert auto["resubmitted"] == 4 and not auto["status"].any()
    # The same frames with marker A's blob missing in three cameras where B's is there (what 5 % dropout does to such a pair in
    # the stress stream): B's blob is then the ONLY hit of A's root in those cameras -- forced views hundreds of pixels off, every
    # group of the root carries them, its best error is ~1e5 px^2 and no bound separates the 2^57 mixtures.  No enumeration
    # reaches that, the reference's own included (helpers.py:394-400 would not return): the frame says so -- candidate overflow
    # + INTRACTABLE + FINAL, log2(groups) of the root in bits 20..28 -- and reports no point.
    b2, c2 = blobs.copy(), counts.copy()
    for f in range(4):
        for cam in (5, 17, 40):
            k = int(np.nonzero(truth["ident"][f, cam] == 0)[0][0])
            n = int(c2[f, cam])
            b2[f, cam, k:n - 1] = b2[f, cam, k + 1:n]
            b2[f, cam, n - 1] = np.nan
            c2[f, cam] = n - 1
    lost = core.match_triangulate_auto(b2, c2, gate_px=synth.STRESS_GATE_PX, K_max=384, G_cap=1 << 20)
    want_bits = capi.ST_CAND_OVERFLOW | capi.ST_INTRACTABLE | capi.ST_FINAL
    # (where B is only a little behind A the forced views are a few tens of pixels off and the search still bounds the root:
    # those frames come back solved; the frame with B furthest out does not)
    hard = lost["status"] != 0
    assert hard.any() and ((lost["status"][hard] & want_bits) == want_bits).all() and not lost["n_out"][hard].any(), lost["status"]
    lg = (lost["status"][hard] >> capi.ST_LOG2_GROUPS_SHIFT) & capi.ST_LOG2_GROUPS_MASK
    assert (lg >= 45).all() and (lg <= 63).all(), lg
    assert (lost["n_out"][~hard] >= 250).all()
    X0 = truth["points_cam0"]
    for f in range(4):
        n = int(auto["n_out"][f])
        assert n >= 250
        found = 0
        for k in range(n):
            k0 = auto["corr"][f, k, 0]
            if k0 < 0 or int(truth["ident"][f, 0, k0]) not in (0, 1):
                continue                                              # (the two camera-0 roots of the pair are what is tested)
            mk = int(truth["ident"][f, 0, k0])
            ids = {int(truth["ident"][f, c, auto["corr"][f, k, c]]) for c in range(1, 64) if auto["corr"][f, k, c] >= 0}
            # one marker's blobs in all 63 other cameras, never a mixture.  WHICH marker's is decided by 0.02 px of noise in
            # camera 0 (the two share the root's pixel): the winner is the consistent group with the smaller error
            assert len(ids) == 1 and ids <= {0, 1}, (f, k, ids)
            assert (auto["corr"][f, k] >= 0).sum() == 64
            np.testing.assert_allclose(auto["xyz"][f, k], X0[f, ids.pop()], atol=5e-5)
            found += 1
        assert found == 2


def test_a_camera_whose_every_blob_is_inside_the_gate(core):
    """256 hits for one (root, camera) pair -- the longest list a pair can have (only the uncapped re-submit keeps it) --
    and a product of exactly 256 candidates for the root."""
    from mocap_core import synth
    from oracle import mocap_oracle as mo
    rig = synth.ring_rig(3, K=synth.STRESS_K, image_size=(16000, 16000))
    rng = np.random.default_rng(3)
    M = 256
    blobs = np.zeros((2, 3, M, 2), dtype=np.float32)
    counts = np.zeros((2, 3), dtype=np.int32)
    Ftab = mo.fundamental_table(rig["K"], rig["R"], rig["t"])
    for f in range(2):
        blobs[f, 0, 0] = (7000.0 + 100 * f, 8100.0)
        counts[f, 0] = 1
        a, b, c = mo.epiline(Ftab[0, 1], blobs[f, 0, 0])
        den = np.hypot(a, b)
        for k in range(M):                              # camera 1: every blob within 0.4 px of the root's line
            if abs(b) > abs(a):
                x = rng.uniform(3000, 13000)
                y = (-(a * x + c)) / b
            else:
                y = rng.uniform(3000, 13000)
                x = (-(b * y + c)) / a
            off = rng.uniform(-0.4, 0.4)
            blobs[f, 1, k] = (x + off * a / den, y + off * b / den)
        counts[f, 1] = M
        a2, b2, c2 = mo.epiline(Ftab[0, 2], blobs[f, 0, 0])
        blobs[f, 2, 0] = (8000.0, (-(a2 * 8000.0 + c2)) / b2) if abs(b2) > 1e-3 else ((-(c2 + b2 * 8000.0)) / a2, 8000.0)
        counts[f, 2] = 1
    res, ref = _compare(core, rig, blobs, counts, 0.5, K_max=600, G_cap=1 << 24)
    assert ref["n_cand"].min() >= 200


def test_pretest_self_check_build_reports_no_false_negative():
    """lib/libmocap_core_pretest.so (-DMOCAP_DEBUG_PRETEST): for every (root, camera) pair the exact double decision is
    taken on the device for every blob the float32 pre-test rejected; a blob inside the gate among them is counted.  Runs the stress shape, the gate-straddling sets, a millimetre rig and integer-pixel frames."""
    lib = os.path.join(ROOT, "low-cost-mocap_amd", "lib", "libmocap_core_pretest.so")
    assert os.path.exists(lib), "build it with `make -C low-cost-mocap_amd all` (__graft_entry__.build does)"
    code = r"""
import sys, numpy as np
sys.path[:0] = [%r, %r, %r]
import torch
from mocap_core import capi, synth
import test_gpu_wide_adversarial as adv
core = capi.MocapCore(0)
dev = torch.device("cuda:0")
tot = [0, 0]
rng = np.random.default_rng(11)
sets = []
rig = synth.stress_rig(64)
b, c, _ = synth.make_stress_stream(rig, 48, 256, seed=55)
adv._gate_straddlers(rig, b, c, synth.STRESS_GATE_PX, rng)
sets.append((rig, b, c, synth.STRESS_GATE_PX, 512, False))
rig = synth.ring_rig(64, radius=3000.0, height=1500.0, K=synth.STRESS_K, image_size=(16000, 16000))
b, c, _ = synth.make_blob_stream(rig, 8, 256, seed=7, noise_px=0.002, dropout=0.05, half_extent=1500.0, min_sep=50.0, truncate=False)
sets.append((rig, b, c, 0.02, 600, False))
rig = synth.ring_rig(8)
b, c, _ = synth.make_blob_stream(rig, 400, 16, seed=78)
adv._gate_straddlers(rig, b, c, 5.0, rng)
sets.append((rig, b, c, 5.0, 128, True))
for rig, blobs, counts, gate, K, fw in sets:
    F, C, M, _ = blobs.shape
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    core.set_frame_limits(hit_cap=32, force_wide=fw)
    d_b, d_c = torch.from_numpy(blobs).to(dev), torch.from_numpy(counts).to(dev)
    xyz = torch.empty((F, K, 3), dtype=torch.float64, device=dev); err = torch.empty((F, K), dtype=torch.float64, device=dev)
    corr = torch.empty((F, K, C), dtype=torch.int16, device=dev); n_out = torch.zeros(F, dtype=torch.int32, device=dev)
    status = torch.zeros(F + 2, dtype=torch.int32, device=dev)          # + the self-check build's two counters
    core.match_triangulate_dev(F, M, d_b.data_ptr(), d_c.data_ptr(), gate, K, 1 << 24, xyz.data_ptr(), err.data_ptr(),
                               corr.data_ptr(), n_out.data_ptr(), status.data_ptr())
    core.synchronize()
    assert core.last_frame_kernel() in ("frame_kernel<512, wide>", "frame_kernel<1024, wide>"), core.last_frame_kernel()
    s = status.cpu().numpy()
    tot[0] += int(s[F]); tot[1] += int(s[F + 1])
print("CHECKED", tot[0], tot[1])
""" % (ROOT, os.path.join(ROOT, "low-cost-mocap_amd"), os.path.join(ROOT, "tests"))
    env = dict(os.environ, MOCAP_CORE_LIB=lib)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    checked = [ln for ln in p.stdout.splitlines() if ln.startswith("CHECKED")][-1].split()
    assert int(checked[1]) > 500000 and int(checked[2]) == 0, checked        # (root, camera) pairs checked, false negatives
