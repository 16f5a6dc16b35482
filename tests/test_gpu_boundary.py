"""GPU: the drop-in boundary's contracts beyond the numbers -- arguments honoured as the reference's functions
define them, the context shared by several threads (Flask-SocketIO handlers and the MJPEG generator run
unlocked against the same Cameras state), per-thread error text, progress events."""
import threading

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def test_reprojection_errors_of_given_object_points(core):
    """calculate_reprojection_errors(image_points, object_points, camera_poses) (helpers.py:203-241) scores the
    points it is HANDED, not a re-triangulation: moved points get the oracle's errors for the moved points; the
    triangulated points reproduce mocap_triangulate's own error output bit for bit; the helpers mirror drops
    entries with fewer than two views."""
    from mocap_core import helpers, synth
    from oracle import mocap_oracle as mo
    g = load_golden("dlt_c8")
    K, R, t = g["K"], g["R"], g["t"]
    obs = g["obs"]
    core.set_cameras(K, R, t)
    xyz, err = core.triangulate(obs)
    again = core.reproject(obs, xyz)
    assert np.array_equal(np.isnan(err), np.isnan(again))
    assert np.array_equal(err[~np.isnan(err)], again[~np.isnan(err)])
    rng = np.random.default_rng(3)
    moved = np.where(np.isnan(xyz), 0.0, xyz) + rng.normal(0, 0.02, xyz.shape)
    got = core.reproject(obs, moved)
    want = mo.reprojection_errors(obs, moved, [k for k in K], R, t)
    seen = (~np.isnan(obs[..., 0])).sum(axis=1) >= 2
    assert np.array_equal(~np.isnan(got), seen)
    w = np.array([np.nan if e is None else e for e in want], dtype=np.float64)
    np.testing.assert_allclose(got[seen], w[seen], rtol=1e-9)
    assert (np.abs(got[seen] - err[seen]) > 1e-3 * err[seen]).mean() > 0.9       # really the moved points' errors
    # the reference-shaped mirror
    helpers.set_core(core)
    helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in K])
    poses = [{"R": R[i], "t": t[i]} for i in range(len(K))]
    e2 = helpers.calculate_reprojection_errors(synth.obs_to_reference_array(obs), moved.tolist(), poses)
    assert e2.shape == (int(seen.sum()),)
    assert np.array_equal(e2, got[seen])


def test_context_shared_by_two_threads(core):
    """One thread keeps asking for a frame's points (the MJPEG generator's find_point_correspondance... call,
    helpers.py:94) while another keeps replacing the camera set and the to-world matrix (the socket handlers,
    helpers.py:171-175, :100).  Every answer must be exactly one of the single-threaded answers: calls are
    serialised per context and tables are swapped whole."""
    from mocap_core import capi, synth
    own = capi.MocapCore(0)
    rig_a = synth.ring_rig(4)
    rig_b = synth.ring_rig(4, radius=2.5)
    blobs, counts, _ = synth.make_blob_stream(rig_a, 4, 4, seed=90)
    W = np.array(synth.APP_TSX_TO_WORLD, dtype=np.float64)
    answers = []
    for rig in (rig_a, rig_b):
        for world in (None, W):
            own.set_cameras(rig["K"], rig["R"], rig["t"])
            own.set_world_transform(world)
            r = own.match_triangulate(blobs, counts, K_max=16)
            answers.append((r["n_out"].copy(), r["corr"].copy(), r["xyz"].copy()))
    stop = threading.Event()
    errors = []

    def flipper():
        i = 0
        try:
            while not stop.is_set():
                rig = rig_a if i & 1 else rig_b
                own.set_cameras(rig["K"], rig["R"], rig["t"])
                own.set_world_transform(W if i & 2 else None)
                i += 1
        except Exception as e:  # pragma: no cover
            errors.append(e)

    th = threading.Thread(target=flipper)
    th.start()
    seen = set()
    try:
        for _ in range(400):
            r = own.match_triangulate(blobs, counts, K_max=16)
            hit = [k for k, (n, c, x) in enumerate(answers)
                   if np.array_equal(n, r["n_out"]) and np.array_equal(c, r["corr"]) and np.array_equal(x, r["xyz"], equal_nan=True)]
            assert hit, "an answer that belongs to no single camera set / transform"
            seen.add(hit[0])
    finally:
        stop.set()
        th.join()
        own.close()
    assert not errors
    assert len(seen) >= 2          # the flips really interleaved with the frame calls


def test_error_text_belongs_to_the_calling_thread(core):
    """mocap_last_error copies the message into a buffer of the calling thread: a second thread failing
    differently cannot rewrite the text the first one is about to read."""
    from mocap_core import capi
    lib = capi.load_library()
    fresh = capi.MocapCore(0)
    rc = lib.mocap_triangulate(fresh._h, 1, None, None, None)              # no cameras yet
    assert rc != 0
    first = lib.mocap_last_error(fresh._h)
    seen = {}

    def other():
        rc2 = lib.mocap_find_blobs(fresh._h, 1, None, 4, None, None, None, None, None)   # no lens model yet
        seen["rc"] = rc2
        seen["text"] = lib.mocap_last_error(fresh._h)

    th = threading.Thread(target=other)
    th.start()
    th.join()
    assert seen["rc"] != 0 and seen["text"] != first                       # the context now holds the other message
    assert b"mocap_set_cameras" in first                                    # this thread's copy is what it was
    fresh.close()


def test_bundle_adjustment_streams_camera_poses(core):
    """helpers.py:274: the reference emits "camera-pose" on every residual evaluation.  Mode "scipy" does the same
    (one emit per evaluation of the reference's optimizer), mode "resident" one per accepted step + the final one."""
    from mocap_core import helpers, synth
    helpers.set_core(core)
    rig = synth.ring_rig(4)
    rng = np.random.default_rng(70)
    obs, _ = synth.make_ba_observations(rig, 120, seed=70)
    init = synth.perturb_rig(rig, rng)
    helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in rig["K"]])
    poses0 = [{"R": init["R"][i], "t": init["t"][i].reshape(3, 1)} for i in range(4)]

    class Sock:
        def __init__(self):
            self.n = 0
            self.last = None

        def emit(self, name, payload):
            assert name == "camera-pose" and len(payload["camera_poses"]) == 4
            self.n += 1
            self.last = payload
    counts = {}
    for mode in ("resident", "scipy"):
        s = Sock()
        with helpers.bundle_adjustment_mode(mode):
            out, info = helpers.bundle_adjustment(synth.obs_to_reference_array(obs), poses0, s, return_info=True)
        counts[mode] = (s.n, info)
        np.testing.assert_allclose(np.array(s.last["camera_poses"][1]["R"]), np.asarray(out[1]["R"]), atol=0)
    n_res, info_res = counts["resident"]
    assert n_res == int(info_res["njev"]) - 1 + 1                           # accepted steps + the final emit
    n_sp, info_sp = counts["scipy"]
    assert n_sp == info_sp["nfev"] + info_sp["njev"] * 22 + 1               # every evaluation (n = 22 FD probes per Jacobian)


def test_stream_handover_is_ordered_on_the_device(core):
    """mocap_set_stream: work a _dev entry point left running on the previous stream is ordered in front of the first launch
    on the new one by an event (no host block, the previous stream -- the caller's, possibly gone -- is never touched
    again).  Alternating streams over one context's shared work queues must give the one-stream results."""
    import gc
    import torch
    from mocap_core import synth
    rig = synth.ring_rig(8)
    F, M, K = 3000, 16, 48
    blobs, counts, _ = synth.make_blob_stream(rig, F, M, seed=31)
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    want = core.match_triangulate(blobs, counts, K_max=K)
    dev = torch.device("cuda", 0)
    d_b, d_c = torch.from_numpy(blobs).to(dev), torch.from_numpy(counts).to(dev)

    def outputs():
        return dict(xyz=torch.empty((F, K, 3), dtype=torch.float64, device=dev), err=torch.empty((F, K), dtype=torch.float64, device=dev),
                    corr=torch.empty((F, K, 8), dtype=torch.int16, device=dev), n=torch.zeros(F, dtype=torch.int32, device=dev),
                    st=torch.zeros(F, dtype=torch.int32, device=dev))

    outs = []
    try:
        for rep in range(6):
            s = torch.cuda.Stream(dev)                    # a fresh stream per launch; the previous one is dropped below
            core.set_stream(s.cuda_stream)
            o = outputs()
            core.match_triangulate_dev(F, M, d_b.data_ptr(), d_c.data_ptr(), 5.0, K, 1 << 20, o["xyz"].data_ptr(), o["err"].data_ptr(),
                                       o["corr"].data_ptr(), o["n"].data_ptr(), o["st"].data_ptr())
            outs.append(o)
            del s
            gc.collect()
        core.synchronize()
    finally:
        core.set_stream(0)
    torch.cuda.synchronize(dev)
    valid = np.arange(K)[None, :] < want["n_out"][:, None]
    for o in outs:
        assert np.array_equal(o["n"].cpu().numpy(), want["n_out"]) and not o["st"].cpu().numpy().any()
        assert np.array_equal(o["xyz"].cpu().numpy()[valid], want["xyz"][valid])
        assert np.array_equal(o["corr"].cpu().numpy()[valid], want["corr"][valid])


def test_frame_calls_stay_fast_while_a_calibration_runs(core):
    """The reference's frame loop (helpers.py:68-135, MJPEG thread) keeps running while calculate_camera_pose ->
    bundle_adjustment (index.py:229-277) runs in a handler thread.  The mirror runs the calibration on its own context /
    stream and never holds the module lock across a residual evaluation: frame calls issued while default-mode (scipy)
    8-camera bundle adjustments are in flight keep their sub-millisecond latency and none of them waits for a solve.
    (Round 5: a default-mode solve is ~0.1 s since its Jacobian is one call -- the calibration thread repeats it for the
    length of the measurement, and is in the library, GIL released, most of that time: a p99 bound holds again.)"""
    import sys
    import threading
    import time
    from conftest import load_golden
    from mocap_core import helpers, synth
    g = load_golden("ba_c8_n100_solved")
    C = g["K"].shape[0]
    helpers.set_core(core)
    helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in g["K"]])
    rig = synth.ring_rig(8)
    blobs, counts, _ = synth.make_blob_stream(rig, 64, 16, seed=4)
    frame_poses = synth.rig_to_pose_dicts(rig)
    frames = [synth.frame_to_reference_lists(blobs[f], counts[f], as_int=True) for f in range(64)]
    want = [helpers.find_point_correspondance_and_object_points([list(p) for p in fr], frame_poses, None)[1] for fr in frames]
    ref_obs = synth.obs_to_reference_array(g["obs"])
    result = {"solves": 0}

    def one_solve():
        poses0 = [{"R": g["R_init"][i].copy(), "t": g["t_init"][i].copy()} for i in range(C)]
        return helpers.bundle_adjustment(ref_obs, poses0, None, return_info=True)

    one_solve()                            # the calibration context exists, its kernels have run once

    def calibrate():
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 1.0:
            result["poses"], result["info"] = one_solve()
            result["solves"] += 1
        result["seconds"] = time.perf_counter() - t0

    old = sys.getswitchinterval()
    sys.setswitchinterval(1e-4)            # hand the GIL over quickly
    try:
        th = threading.Thread(target=calibrate)
        th.start()
        lat, k = [], 0
        while th.is_alive():
            fr = [list(p) for p in frames[k % 64]]
            t0 = time.perf_counter()
            _, xyz, _ = helpers.find_point_correspondance_and_object_points(fr, frame_poses, None)
            lat.append(time.perf_counter() - t0)
            assert np.array_equal(xyz, want[k % 64])          # the two contexts do not disturb each other's cameras
            k += 1
            time.sleep(0.002)
        th.join()
    finally:
        sys.setswitchinterval(old)
    # the calibrations really overlapped (one solve is 0.1-0.2 s on a quiet host; on a bench box whose host cores are shared --
    # round 6 -- a single solve has been seen to take the whole second: the frame calls still ran THROUGH it, which is the point)
    assert result["solves"] >= 1 and len(lat) > 100, (result, len(lat))
    lat = np.sort(np.array(lat)) * 1e3
    p50, p90, p99 = lat[len(lat) // 2], lat[int(len(lat) * 0.90)], lat[int(len(lat) * 0.99)]
    print(f"frame calls during calibrations: n {len(lat)} p50 {p50:.3f} ms p90 {p90:.3f} p99 {p99:.3f} ms max {lat[-1]:.3f} ms; "
          f"{result['solves']} solves in {result['seconds']:.2f} s")
    # No frame call waits behind a calibration (one shared lock would put a whole solve in front of it).  The calls go
    # through Python (list marshalling under the GIL, which the other thread's SciPy code also wants): bounded well below
    # one solve and far below the 8 ms of a 120 fps frame; the library's own share is measured without Python in between in
    # test_two_contexts_raw_ctypes_frame_latency below.
    # (round 5: with one GPU call per Jacobian the calibration thread spends most of its time in SciPy's own Python code, i.e.
    # holding the GIL -- one or two of ~140 calls wait 5-70 ms for it (seen: p99 5.6 and 69 ms in two of five runs of the suite).
    # That tail is the interpreter's; what this test can hold against the LIBRARY is: the typical call is undisturbed (p50, p90,
    # p95) and nothing ever waits for a whole solve, which one shared lock would make the rule.)
    p95 = lat[int(len(lat) * 0.95)]
    assert p50 < 0.5 and p90 < 1.0 and p95 < 4.0, (p50, p90, p95, p99, lat[-1])
    assert lat[-1] < 0.6 * 1e3 * result["seconds"] / result["solves"], (lat[-1], result)
    # and the calibration's own answer is the reference's, as when it runs alone
    R = np.array([np.asarray(p["R"], dtype=np.float64) for p in result["poses"]])
    assert np.abs(R - g["R_ba"]).max() == 0.0 and int(result["info"]["nfev"]) == int(g["ba_stats"][0])


def _raw_two_thread_latency(core, ba_call, seconds=1.0):
    """Thread A: `ba_call()` in a tight loop (a ctypes call: the GIL is released for its whole duration).  Thread B (this
    one): mocap_track_frame on the frame context through raw ctypes with every argument prepared beforehand -- no
    marshalling, no NumPy between calls.  Returns B's latencies (ms, sorted) and A's call count."""
    import ctypes
    import threading
    import time
    from mocap_core import capi, synth
    rig = synth.ring_rig(8)
    blobs, counts, _ = synth.make_blob_stream(rig, 64, 16, seed=4)
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    K = 64
    o = core._track_outputs(1, K, 0)
    fn = core.lib.mocap_track_frame
    args = [[core._h, 1, 16, capi._p(blobs[f:f + 1]), capi._p(counts[f:f + 1]), ctypes.c_double(5.0), K, 1 << 20, capi._p(o["xyz"]),
             capi._p(o["err"]), capi._p(o["corr"]), capi._p(o["n_pts"]), capi._p(o["status"]), 0, None, None, None, None, None]
            for f in range(64)]
    want = [core.track_frame(blobs[f:f + 1], counts[f:f + 1], K_max=K, O_max=0) for f in range(64)]
    for a in args[:8]:
        assert fn(*a) == 0
    stop = threading.Event()
    calls = {"n": 0, "err": None}

    def hammer():
        try:
            while not stop.is_set():
                ba_call()
                calls["n"] += 1
        except Exception as e:               # noqa: BLE001 -- reported by the caller
            calls["err"] = e

    th = threading.Thread(target=hammer)
    th.start()
    lat = []
    t_end = time.perf_counter() + seconds
    k = 0
    try:
        while time.perf_counter() < t_end:
            a = args[k % 64]
            t0 = time.perf_counter()
            rc = fn(*a)
            lat.append(time.perf_counter() - t0)
            assert rc == 0
            n = int(o["n_pts"][0])
            assert n == int(want[k % 64]["n_pts"][0]) and np.array_equal(o["xyz"][0, :n], want[k % 64]["xyz"][0, :n])
            k += 1
    finally:
        stop.set()
        th.join()
    assert calls["err"] is None, calls["err"]
    return np.sort(np.array(lat)) * 1e3, calls["n"]


def test_two_contexts_raw_ctypes_frame_latency(core):
    """What the LIBRARY adds to a frame call while a calibration's residual evaluations run on another context (own
    stream, own lock): nothing of the other context's work is ever waited for.  Raw ctypes on both threads, arguments
    prepared beforehand: p99 < 0.5 ms, and -- one outlier tolerated for the runtime's once-per-process dispatch hole
    (profiles/r03_ba_dispatch_hole.txt) -- no call above 2 ms.  Reference: the frame loop and the calculate-camera-pose
    handler run concurrently (helpers.py:68-135 vs index.py:229-277)."""
    import ctypes
    from mocap_core import capi, helpers, synth
    rig = synth.ring_rig(8)
    obs, _ = synth.make_ba_observations(rig, 1000, seed=7)
    init = synth.perturb_rig(rig, np.random.default_rng(7))
    ba = capi.MocapCore(core.device_id)
    try:
        ba.set_cameras(rig["K"], init["R"], init["t"])
        helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in rig["K"]])
        x0 = helpers._ba_x0([{"R": init["R"][i], "t": init["t"][i]} for i in range(8)])
        X = np.ascontiguousarray(np.repeat(x0[None, :], 51, axis=0))
        obs_c = np.ascontiguousarray(obs, dtype=np.float64)
        r = np.empty((51, obs_c.shape[0]))
        fa = ba.lib.mocap_ba_residuals
        a_args = (ba._h, 51, capi._p(X), obs_c.shape[0], capi._p(obs_c), capi._p(r))
        assert fa(*a_args) == 0

        def ba_call():
            assert fa(*a_args) == 0

        lat, n_ba = _raw_two_thread_latency(core, ba_call)
    finally:
        ba.close()
    p50, p99 = lat[len(lat) // 2], lat[int(len(lat) * 0.99)]
    print(f"raw ctypes: {len(lat)} frame calls next to {n_ba} batched residual evaluations: p50 {p50:.3f} ms p99 {p99:.3f} ms "
          f"max {lat[-1]:.3f} ms")
    assert n_ba > 100 and len(lat) > 1000
    assert p99 < 0.5 and lat[-2] < 2.0, (p50, p99, lat[-3:])


def test_frame_calls_next_to_the_resident_solver(core):
    """The same with mode "resident" on the other context: mocap_ba_solve keeps a launched-ahead linearisation kernel
    spinning on its mailbox (device watchdog: 2 s) while frame kernels from this context share the GPU.  The solves must
    keep converging to the same answer, none of them through a relaunch, and the frame calls keep their latency."""
    from mocap_core import capi, helpers, synth
    rig = synth.ring_rig(8)
    obs, _ = synth.make_ba_observations(rig, 1000, seed=7)
    init = synth.perturb_rig(rig, np.random.default_rng(7))
    ba = capi.MocapCore(core.device_id)
    try:
        ba.set_cameras(rig["K"], init["R"], init["t"])
        helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in rig["K"]])
        x0 = helpers._ba_x0([{"R": init["R"][i], "t": init["t"][i]} for i in range(8)])
        x_alone, info_alone = ba.ba_solve(x0, obs, ftol=1e-2)
        seen = []

        def ba_call():
            x, info = ba.ba_solve(x0, obs, ftol=1e-2)
            seen.append((x, info))

        lat, n_ba = _raw_two_thread_latency(core, ba_call)
    finally:
        ba.close()
    p50, p99 = lat[len(lat) // 2], lat[int(len(lat) * 0.99)]
    relaunches = sum(int(i.get("relaunches", 0)) for _, i in seen)
    print(f"resident solver next to {len(lat)} frame calls: {n_ba} solves, {relaunches} relaunched linearisations; frame p50 {p50:.3f} ms "
          f"p99 {p99:.3f} ms max {lat[-1]:.3f} ms")
    assert n_ba >= 5
    for x, info in seen:
        assert int(info["status"]) == int(info_alone["status"]) and np.array_equal(x, x_alone)
    assert relaunches == 0
    assert p99 < 1.0 and lat[-2] < 5.0, (p50, p99, lat[-3:])
