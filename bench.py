#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native marker-tracking core.

Metric (BASELINE.json): triangulated 3-D markers/sec at 8 cams x 16 markers (synthetic blob
streams), plus BA iterations/sec (8 cams, 1k points) as a secondary figure in the same line.

A "step" = one pass of the hot path (mocap_match_triangulate_dev: epipolar matching + candidate
triangulation + per-root selection) over one batch of FRAMES_PER_GPU synthetic frames that is
ALREADY RESIDENT in HBM when the timed region starts.  With N > 1 (one process per GPU, launched
by torch.distributed.run) every rank owns its own contiguous block of frames (weak scaling) and a
step ends with the single exchange of the path: the gather of the packed track records on rank 0
(RCCL over xGMI).  value = markers produced by all ranks / max-over-ranks time.

    python bench.py [--gpus N] [--steps K] [--warmup W]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "low-cost-mocap_amd"))

from mocap_core import capi, dist as mdist, synth  # noqa: E402

CAMS, MARKERS = 8, 16
FRAMES_PER_GPU = 100_000      # BASELINE.json configs[2] / SURVEY.md 8d cfg3
K_MAX = 48
G_CAP = 1 << 20           # groups per root a pass enumerates unless the workload says otherwise (WORKLOADS[...]["g_cap"], MOCAP_BENCH_G_CAP)
# --workload selects one of BASELINE.json's configs; the default (the driver's) is the metric's own
# configuration, 8 cams x 16 markers.  The others are secondary measurements of the same path.
WORKLOADS = {
    "8x16": dict(C=8, M=16, frames=100_000, K_max=48, gate=5.0, stress=False,
                 desc="8 cams x 16 markers, synthetic ring rig f=320 c=160 (camera-params.json), "
                      "int-truncated blobs, sigma=0.3px, 5% dropout (BASELINE.json configs[2])"),
    "4x4": dict(C=4, M=4, frames=1_000_000, K_max=16, gate=5.0, stress=False, oracle_frames=300,
                desc="4 cams x 4 markers, synthetic ring rig f=320 c=160, int-truncated blobs, sigma=0.3px, "
                     "5% dropout (BASELINE.json configs[1])"),
    "64x256": dict(C=64, M=256, frames=12_500, K_max=384, gate=None, stress=True, oracle_frames=16,
                   desc="stress: 64 virtual cams x 256 markers, 16k x 16k px virtual sensor, float centroids, "
                        "sigma=0.02px, gate 0.5px (bounded ambiguity), 5% dropout (BASELINE.json configs[4]: 100 k frames over 8 GPUs "
                        "= 12 500 frames per GPU, generated in chunks of 256 on the host's cores)"),
}
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_TF = 78.6      # MI355X FP64 vector peak
FP64_MFMA_PEAK_TF = 78.6      # FP64 matrix peak (v_mfma_f64_16x16x4_f64, 32 flop/clk/SIMD)


def algorithmic_bytes(counts, n_out, C, M):
    """per frame 8*C*M + 4*C in, K*(24 + 8 + 2*C) out (K = points produced)."""
    F = counts.shape[0]
    return F * (8 * C * M + 4 * C) + int(n_out.sum()) * (24 + 8 + 2 * C)


PROFILE_TAGS = ("r06", "r05")  # profiles/<tag>_hbm_traffic.json, <tag>_fp64_mix.json (scripts/profile_frame_pmc.sh): newest first


def kernel_source_hash():
    """sha256 (first 16 hex digits) over the sources the library is built from.  The PMC summaries under profiles/
    carry the hash of the sources they were measured on; a bench run on different sources marks every figure it derives
    from them `stale` instead of passing the old counters off as this kernel's."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "low-cost-mocap_amd", "csrc", "*")) + [os.path.join(ROOT, "include", "mocap_core.h"),
                   os.path.join(ROOT, "low-cost-mocap_amd", "Makefile")])
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load_profile(name):
    """(summary dict or None, stale flag, path) of the newest profiles/<tag>_<name>.json."""
    path = None
    for tag in PROFILE_TAGS:
        path = os.path.join("profiles", f"{tag}_{name}.json")
        try:
            with open(os.path.join(ROOT, path)) as f:
                t = json.load(f)
        except Exception:
            continue
        return t, t.get("kernel_source_sha16") != kernel_source_hash(), path
    return None, None, path


def measured_traffic(frames):
    """HBM bytes per launch from the committed PMC pass (rocprofv3 FETCH_SIZE x2 [gfx950 correction] + WRITE_SIZE,
    separate passes, taken at this bench's own frame count), scaled to this launch's frame count.  PMC counters
    cannot be read inside the timed run, so this is the profile's figure, not a live one: the summary names the
    sources and kernel it was taken on, and `stale` says whether those are the sources of this run."""
    t, stale, path = load_profile("hbm_traffic")
    if t is None:
        return None, {"traffic_from_profiles_source": None}
    return float(t["hbm_bytes_per_frame"]) * frames, {
        "traffic_is_from_profiles": True,   # a builder-run counter pass scaled to this launch, NOT measured by this run
        "traffic_from_profiles_source": path, "traffic_stale": bool(stale), "traffic_measured_on": {
            "kernel": t.get("kernel"), "git_head": t.get("git_head"), "kernel_source_sha16": t.get("kernel_source_sha16"),
            "frames_per_launch": t.get("frames_per_launch")},
        "traffic_over_algorithmic_from_profiles": t.get("traffic_over_algorithmic"),
        "write_bytes_over_output_bytes_from_profiles": t.get("write_bytes_over_output_bytes")}


def executed_fp64(candidates, kernel_ms):
    """Executed FP64 work of the frame kernel from the committed instruction-mix counter pass
    (SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64 per candidate on this kernel), scaled to this launch's candidate count and
    divided by this run's kernel time.  None if the summary is absent."""
    mix, stale, path = load_profile("fp64_mix")
    if mix is None:
        return None
    flop = float(mix["fp64_flop_per_candidate"]) * candidates
    tf = flop / (kernel_ms * 1e-3) / 1e12
    lane_util = mix.get("vector_lane_utilisation")
    return {"achieved": tf, "frac": tf / FP64_VALU_PEAK_TF,
            # frac counts every lane of an issued wave instruction; frac_active_lanes = frac x the measured share of lanes that
            # were switched on (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)): the useful fraction
            "frac_active_lanes": (tf / FP64_VALU_PEAK_TF * lane_util) if lane_util else None,
            "is_from_profiles": True,   # flop per candidate, lane utilisation: builder-run counter passes; the time is this run's
            "stale": bool(stale),
            "flop_per_candidate_from_profiles": mix["fp64_flop_per_candidate"],
            "vector_lane_utilisation_from_profiles": lane_util,
            "valu_lane_instructions_per_candidate_from_profiles": mix["valu_lane_instructions_per_candidate"],
            "fp64_share_of_valu_instructions_from_profiles": mix["fp64_share_of_valu_instructions"],
            "valu_issue_utilisation_from_profiles": mix.get("valu_issue_utilisation"),
            "measured_on": {"kernel": mix.get("kernel"), "git_head": mix.get("git_head"),
                            "kernel_source_sha16": mix.get("kernel_source_sha16"), "frames_per_launch": mix.get("frames_per_launch")},
            "source": path + " (rocprofv3 PMC instruction mix of this kernel, FMA = 2 flop, MUL/ADD/TRANS = 1; per "
                             "candidate group of the Cartesian product, evaluated or dropped)"}


def cpu_baseline(rig, blobs, counts, budget_s=15.0, gpu=None):
    """The oracle's C restatement of the reference path ("port"), single thread, on a bounded
    prefix of the SAME frames.  Reported baseline only -- never the thing measured or shipped.
    gpu: the timed run's outputs (numpy: n_out, corr, xyz); the all-cores run's oracle results are then also
    compared with them frame by frame -> `wide_parity` (tens of thousands of frames instead of the 300-frame gate)."""
    from oracle import c_oracle
    co = c_oracle.COracle(rig["K"], rig["R"], rig["t"])
    co.match_triangulate(blobs[:20], counts[:20])          # warm
    n, done, pts, t0 = 500, 0, 0, time.perf_counter()
    while done < blobs.shape[0]:
        hi = min(done + n, blobs.shape[0])
        r = co.match_triangulate(blobs[done:hi], counts[done:hi])
        pts += int(r["n_out"].sum())
        done = hi
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    out = {"value": pts / dt, "unit": "markers/s", "cores": 1, "kind": "port",
           "sample": f"first {done} frames of the bench batch, C restatement (oracle/c), 1 thread, {dt:.1f}s"}
    # what "port" is worth against the reference itself: measured once where /root/reference exists (the build container),
    # same frames, one thread each (scripts/calibrate_cpu_baseline.py -> profiles/r05_cpu_baseline_calibration.json)
    try:
        with open(os.path.join(ROOT, "profiles", "r05_cpu_baseline_calibration.json")) as f:
            cal = json.load(f)
    except Exception:
        cal = None
    if cal:
        k = cal["c_port"]["speed_vs_reference"]
        out["sample"] += (f"; calibration (profiles/r05_cpu_baseline_calibration.json, build container, 1 thread): this C port runs "
                          f"{k:.0f} x the reference's own find_point_correspondance_and_object_points on the same frames "
                          f"(reference {cal['reference']['markers_per_s']:.1f}, Python port {cal['python_port']['markers_per_s']:.1f}, "
                          f"C port {cal['c_port']['markers_per_s']:.0f} markers/s there)")
        out["calibration"] = {"c_port_speed_vs_reference": k, "python_port_speed_vs_reference": cal["python_port"]["speed_vs_reference"],
                              "reference_markers_per_s_build_container": cal["reference"]["markers_per_s"],
                              "reference_equivalent_markers_per_s_this_core": (pts / dt) / k,
                              "source": "profiles/r05_cpu_baseline_calibration.json"}
    # the same port frame-sharded over every host core (the reference itself is single-threaded; this is
    # the "whole box" figure SURVEY 8d asks for): one thread per core, ctypes releases the GIL in the C call
    try:
        from concurrent.futures import ThreadPoolExecutor
        import threading
        ncore = os.cpu_count() or 1
        chunk = 8
        n_chunks = min(ncore * 16, blobs.shape[0] // chunk)       # small chunks, pulled dynamically:
        nthreads = max(1, min(ncore, n_chunks))                   # candidate counts per frame are heavy-tailed
        local = threading.local()

        def work(i):
            if not hasattr(local, "co"):
                local.co = c_oracle.COracle(rig["K"], rig["R"], rig["t"])
            lo, hi = i * chunk, (i + 1) * chunk
            r = local.co.match_triangulate(blobs[lo:hi], counts[lo:hi])
            bad, dev = 0, 0.0
            if gpu is not None:          # checker role: the GPU's answer for the very same frames
                kk = gpu["corr"].shape[1]
                vv = np.arange(kk)[None, :] < r["n_out"][:, None]
                same = np.array_equal(r["n_out"], gpu["n_out"][lo:hi]) and np.array_equal(r["corr"][:, :kk][vv], gpu["corr"][lo:hi][vv])
                bad = 0 if same else 1
                if same and vv.any():
                    ref = r["xyz"][:, :kk][vv]
                    dev = float(np.abs(gpu["xyz"][lo:hi][vv] - ref).max() / np.abs(ref).max())
            return int(r["n_out"].sum()), bad, dev
        t2 = time.perf_counter()
        with ThreadPoolExecutor(nthreads) as ex:
            res = list(ex.map(work, range(n_chunks)))
        dt2 = time.perf_counter() - t2
        tot = sum(r[0] for r in res)
        if gpu is not None:
            out["wide_parity"] = {"frames_checked": n_chunks * chunk, "chunks_with_any_index_difference": sum(r[1] for r in res),
                                  "corr_bit_exact": not any(r[1] for r in res), "xyz_max_rel": max(r[2] for r in res),
                                  "against": "oracle/c (the all-cores baseline run's own results)"}
        out["all_cores"] = {"value": tot / dt2, "unit": "markers/s", "cores": nthreads,
                            "sample": f"first {n_chunks * chunk} frames of the bench batch in chunks of {chunk}, "
                                      f"{nthreads} threads, {dt2:.1f}s"}
    except Exception as e:  # pragma: no cover
        out["all_cores_error"] = repr(e)
    # the NumPy/Python restatement keeps the reference's own structure (Python loops + LAPACK SVD per
    # candidate) and is bit-exact against it: its rate is the closest stand-in for the reference itself.  BASELINE.md section 3:
    # >= 200 frames, spread over the host's cores (separate interpreters, oracle/port_pool.py)
    try:
        from oracle import port_pool
        pp = port_pool.python_port_rate(rig, blobs, counts, frames=int(os.environ.get("MOCAP_BENCH_PYPORT_FRAMES", "208")))
        out["python_port_markers_per_s"] = pp["markers_per_s_per_core"]
        out["python_port_all_workers_markers_per_s"] = pp["markers_per_s_all_workers"]
        out["python_port_sample"] = (f"{pp['frames']} frames (frame i on worker i mod {pp['workers']}), oracle/mocap_oracle.py (bit-exact vs the "
                                     f"reference), {pp['workers']} single-thread workers, {pp['wall_s_incl_startup']:.1f}s incl. interpreter start-up; "
                                     "per-core rate = markers / summed worker CPU seconds")
    except Exception as e:  # pragma: no cover
        out["python_port_error"] = repr(e)
    return out


def frame_latency(core, blobs, counts, n=300):
    """Live-tracking use (BASELINE.json configs[2], "@120 fps"): ONE frame per call through the
    host-buffer C entry point the reference's per-frame seam maps to (helpers.py:94 ->
    mocap_match_triangulate): H2D of the frame's blobs, the kernels, D2H of the points, sync.
    Wall-clock per call, successive frames of the bench stream."""
    core.match_triangulate(blobs[:1], counts[:1], K_max=K_MAX)          # warm (allocations)
    ts = []
    for f in range(1, n + 1):
        t0 = time.perf_counter()
        core.match_triangulate(blobs[f:f + 1], counts[f:f + 1], K_max=K_MAX)
        ts.append(time.perf_counter() - t0)
    ts = np.sort(np.array(ts)) * 1e3
    return {"calls": n, "p50_ms": float(ts[n // 2]), "p99_ms": float(ts[int(n * 0.99)]), "max_ms": float(ts[-1]),
            "frame_budget_ms_at_120fps": 1e3 / 120.0,
            "path": "mocap_match_triangulate (host buffers, 1 frame per call, incl. PCIe copies and sync; "
                    "includes the Python/ctypes call overhead)"}


def chain_latency(core, blobs, counts, n=200, distinct=16):
    """latency.chain: the live loop's body per frame set, wall clock per call incl. Python/ctypes and the payload dict.
    `track`: image points (host) -> `object-points` payload (helpers.py:94-133: match, world transform, locate_objects,
    the dict) through ONE core call (mocap_track_frame).  `chain`: raw 8-camera frame set as pseyepy hands it over (host
    memory, 1.8 MB) -> the same payload through mocap_track_frame_images (helpers.py:68-133), PCIe upload included."""
    from mocap_core import helpers

    def payload(res):
        k = int(res["n_pts"][0])
        objs = helpers._objects_list(res)
        return helpers.object_points_payload(res["err"][0, :k], res["xyz"][0, :k], objs)

    def stats(ts):
        ts = np.sort(np.array(ts)) * 1e3
        return {"calls": len(ts), "p50_ms": float(ts[len(ts) // 2]), "p99_ms": float(ts[int(len(ts) * 0.99)]), "max_ms": float(ts[-1])}

    out = {}
    core.set_world_transform(synth.APP_TSX_TO_WORLD)
    try:
        ts = []
        for f in range(n + 10):
            t0 = time.perf_counter()
            pl = payload(core.track_frame(blobs[f:f + 1], counts[f:f + 1], K_max=K_MAX, O_max=8))
            ts.append(time.perf_counter() - t0)
        out["track"] = dict(stats(ts[10:]), points_last=len(pl["object_points"]),
                            path="mocap_track_frame: image points (host) -> match -> world -> locate_objects -> payload dict; "
                                 "one enqueue, one event wait, zero-copy through pinned memory")
        C, M_max = CAMS, 32
        rig = synth.ring_rig(C)
        images, _ = synth.render_camera_frames(rig, distinct, MARKERS, seed=1)
        core.set_cameras(rig["K"], rig["R"], rig["t"])
        core.set_image_params(240, 320, rig["K"], [synth.REFERENCE_DISTORTION] * C)
        ts = []
        for i in range(n + 10):
            img = images[i % distinct][None]
            t0 = time.perf_counter()
            pl = payload(core.track_frame_images(img, M_max=M_max, K_max=K_MAX, O_max=8))
            ts.append(time.perf_counter() - t0)
        out["chain"] = dict(stats(ts[10:]), points_last=len(pl["object_points"]), image_bytes=int(images[0].nbytes),
                            path="mocap_track_frame_images: 8 raw 240x320 RGB frames (pageable host memory) -> H2D -> blob stage -> "
                                 "match -> world -> locate_objects -> payload dict; one enqueue, one event wait")
    finally:
        core.set_world_transform(None)
    out["frame_budget_ms_at_120fps"] = 1e3 / 120.0
    return out


def blob_stage_bench(core, dev, stream, steps=5, frames=1024, distinct=16):
    """The row before the path: raw 8-camera PS3-Eye frame sets (240 x 320 RGB, resident in
    HBM) -> blob centroids (mocap_find_blobs_dev), and the chain images -> blobs -> 3-D markers without
    leaving HBM.  Secondary figures; the headline metric starts from blobs."""
    import torch
    from oracle import c_oracle
    C, M, M_max, K_MAX = CAMS, MARKERS, 32, 48
    rig = synth.ring_rig(C)
    images, _ = synth.render_camera_frames(rig, distinct, M, seed=1)
    dists = [synth.REFERENCE_DISTORTION] * C
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    core.set_image_params(240, 320, rig["K"], dists)
    core.set_stream(stream.cuda_stream)
    F = frames
    d_img = torch.from_numpy(images).to(dev).repeat((F + distinct - 1) // distinct, 1, 1, 1, 1)[:F].contiguous()
    d_blobs = torch.zeros((F, C, M_max, 2), dtype=torch.float32, device=dev)
    d_counts = torch.zeros((F, C), dtype=torch.int32, device=dev)
    d_bst = torch.zeros((F, C), dtype=torch.int32, device=dev)
    d_xyz = torch.empty((F, K_MAX, 3), dtype=torch.float64, device=dev)
    d_err = torch.empty((F, K_MAX), dtype=torch.float64, device=dev)
    d_corr = torch.empty((F, K_MAX, C), dtype=torch.int16, device=dev)
    d_nout = torch.zeros(F, dtype=torch.int32, device=dev)
    d_st = torch.zeros(F, dtype=torch.int32, device=dev)

    def blobs_only():
        core.find_blobs_dev(F, d_img.data_ptr(), M_max, d_blobs.data_ptr(), d_counts.data_ptr(), d_bst.data_ptr())

    def chain():
        blobs_only()
        core.match_triangulate_dev(F, M_max, d_blobs.data_ptr(), d_counts.data_ptr(), 5.0, K_MAX, G_CAP,
                                   d_xyz.data_ptr(), d_err.data_ptr(), d_corr.data_ptr(), d_nout.data_ptr(),
                                   d_st.data_ptr())

    def timed(fn):
        fn()
        torch.cuda.synchronize(dev)
        ts = []
        for _ in range(steps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            fn()
            b.record(stream)
            torch.cuda.synchronize(dev)
            ts.append(a.elapsed_time(b))
        return float(np.median(ts))

    ms_b = timed(blobs_only)
    ms_c = timed(chain)
    # live use: ONE 8-camera frame set per call, raw frames already on the device, images -> 3-D points
    lat = []
    F_saved = F
    for i in range(60):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        core.find_blobs_dev(1, d_img.data_ptr(), M_max, d_blobs.data_ptr(), d_counts.data_ptr(), d_bst.data_ptr())
        core.match_triangulate_dev(1, M_max, d_blobs.data_ptr(), d_counts.data_ptr(), 5.0, K_MAX, G_CAP,
                                   d_xyz.data_ptr(), d_err.data_ptr(), d_corr.data_ptr(), d_nout.data_ptr(),
                                   d_st.data_ptr())
        torch.cuda.synchronize(dev)
        lat.append((time.perf_counter() - t0) * 1e3)
    lat = np.sort(np.array(lat[10:]))
    assert F_saved == F
    core.set_blob_options(skip_dark_tiles=False)       # every tile filtered: the dense-work figure
    ms_dense = timed(blobs_only)
    core.set_blob_options(skip_dark_tiles=True)
    n_img = F * C
    in_bytes = n_img * 240 * 320 * 3
    out_bytes = int(d_counts.sum().item()) * 8 + n_img * 8
    # parity gate: the first frame sets against the C oracle (sequential contours)
    nchk = 2
    ref = c_oracle.BlobOracle(240, 320, rig["K"], dists).find_blobs(images[:nchk], M_max=M_max)
    got_c, got_b = d_counts[:nchk].cpu().numpy(), d_blobs[:nchk].cpu().numpy()
    t0 = time.perf_counter()
    ncpu = 0
    bo = c_oracle.BlobOracle(240, 320, rig["K"], dists)
    while time.perf_counter() - t0 < 4.0 and ncpu < distinct:
        bo.find_blobs(images[ncpu:ncpu + 1], M_max=M_max)
        ncpu += 1
    cpu_rate = ncpu * C / (time.perf_counter() - t0)
    return {"metric": "camera images/s, raw RGB frames -> blob centroids (helpers.py:68-82,143-163)",
            "value": n_img / ms_b * 1e3, "frame_sets_per_s": F / ms_b * 1e3, "ms_per_batch": ms_b,
            "images_per_batch": n_img, "markers_per_frame_set": MARKERS,
            "all_tiles_filtered": {"value": n_img / ms_dense * 1e3, "ms_per_batch": ms_dense,
                                   "note": "mocap_set_blob_options(0): without the exact dark-tile early-out "
                                           "(background noise here is uniform in [0, 3): range 2, the bound of the proof)"},
            "roofline": {"bound": "hbm", "achieved": (in_bytes + out_bytes) / ms_b / 1e6, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": (in_bytes + out_bytes) / ms_b / 1e6 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_image": 240 * 320 * 3 + 8,
                         "note": "integer-VALU bound (9x9 + 5x5 filters on 3 channels: ~180 ops/pixel); "
                                 "profiles/r01_blob_* (instruction mix), profiles/r02_blob_pmc_traffic.csv (HBM traffic)"},
            "frame_set_latency_ms": {"p50": float(lat[len(lat) // 2]), "max": float(lat[-1]),
                                     "path": "1 frame set (8 images on the device) -> blobs -> 3-D points, "
                                             "5 kernel launches + sync, wall clock incl. Python/ctypes"},
            "chain_images_to_markers": {"markers_per_s": float(d_nout.sum().item()) / ms_c * 1e3,
                                        "frame_sets_per_s": F / ms_c * 1e3, "ms_per_batch": ms_c,
                                        "overflow_frames": int((d_st != 0).sum().item())},
            "parity": {"frame_sets_checked": nchk, "counts_equal": bool(np.array_equal(got_c, ref["counts"])),
                       "centroids_bit_exact": bool(np.array_equal(got_b, ref["blobs"]))},
            "cpu_baseline": {"value": cpu_rate, "unit": "images/s", "cores": 1, "kind": "port",
                             "sample": f"{ncpu} frame sets of {C} images, scalar C restatement (oracle/c/blob_oracle.c); "
                                       "OpenCV's own SIMD paths would be faster than this port"}}


def ba_timed_solves(core, x0, obs, iters, n_runs=5, warm_s=0.25):
    """n_runs timed solves behind a warm-up of at least warm_s seconds of back-to-back solves, sorted by time.
    Why a timed warm-up and not one solve: a process shows ONE hole of 30-80 ms in kernel dispatch 5-50 ms after its first
    burst of bundle-adjustment launches (rocprofv3 kernel trace: no kernel running while the host has long submitted
    the next one; it comes with the launch-ahead on or off, the one-launch kernel or the five-launch chain, the copy
    engine on or off, and never again in that process -- not after idle gaps, reallocations or a second context:
    scripts/diag_ba_stall.py, diag_gpu_wake.py, diag_alloc_stall.py, profiles/r03_ba_dispatch_hole.txt).  One warm-up
    solve of a few ms ends before the hole opens, which then lands in the timed solve."""
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < warm_s or k < 3:
        core.ba_solve(x0, obs, ftol=0.0, xtol=0.0, gtol=0.0, max_iter=iters)
        k += 1
    runs = []
    for _ in range(n_runs):
        t1 = time.perf_counter()
        _, info = core.ba_solve(x0, obs, ftol=0.0, xtol=0.0, gtol=0.0, max_iter=iters)
        runs.append((time.perf_counter() - t1, info))
    runs.sort(key=lambda r: r[0])
    return runs, k


def ba_bench_16k(core, iters=60):
    """BASELINE.json configs[3]: 8 cams, 2000 frames x 8 markers = 16 000 calibration points."""
    from mocap_core import helpers
    rig = synth.ring_rig(CAMS)
    rng = np.random.default_rng(9)
    obs, _ = synth.make_ba_observations(rig, 16000, seed=9)
    init = synth.perturb_rig(rig, rng)
    core.set_cameras(rig["K"], init["R"], init["t"])
    helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in rig["K"]])
    x0 = helpers._ba_x0([{"R": init["R"][i], "t": init["t"][i]} for i in range(CAMS)])
    runs, n_warm = ba_timed_solves(core, x0, obs, iters)
    dt, info = runs[len(runs) // 2]
    _, ref = core.ba_solve(x0, obs, ftol=1e-2)
    return {"points": int(info["m"]), "value": info["iterations"] / dt, "ms_per_iter": 1e3 * dt / max(info["iterations"], 1),
            "runs_ms": [round(1e3 * r[0], 2) for r in runs], "statistic": "median of 5 solves", "warmup_solves": n_warm,
            "iterations": info["iterations"],
            "reference_rule_run": {"iterations": ref["iterations"], "status": ref["status"], "cost0": ref["cost0"],
                                   "cost": ref["cost"], "elapsed_ms": ref["elapsed_ms"]}}


def ba_cpu_baseline(rig, init, obs, x0, budget_nfev=150):
    """CPU baseline of the BA half of the metric, timed on this box in this run: the reference's own optimizer
    call (helpers.py:287: scipy least_squares, loss="cauchy", 2-point Jacobian, float32 residuals) driven by
    the oracle's C restatement of residual_function ("port"; single thread, BLAS pinned to 1 thread).  One
    iteration = one accepted-or-rejected trust-region step, the unit the GPU figure counts."""
    from scipy import optimize
    from oracle import c_oracle
    co = c_oracle.COracle(rig["K"], init["R"], init["t"])

    def fun(x):
        r = co.ba_residuals(x, obs)[0]
        return r[~np.isnan(r)].astype(np.float32)
    import contextlib
    try:
        from threadpoolctl import threadpool_limits
        single = threadpool_limits(limits=1)
    except Exception:  # pragma: no cover
        single = contextlib.nullcontext()
    with single:
        fun(x0)
        t0 = time.perf_counter()
        # tolerances at scipy's floor: the run is ended by its evaluation budget, like the GPU figure's
        res = optimize.least_squares(fun, x0, loss="cauchy", ftol=1e-15, xtol=None, gtol=None, max_nfev=budget_nfev)
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        for _ in range(5):
            fun(x0)
        t_eval = (time.perf_counter() - t1) / 5
    out = {"value": (res.nfev - 1) / dt, "unit": "LM iterations/s", "cores": 1, "kind": "port",
           "sample": f"{res.nfev - 1} trust-region steps ({res.njev} Jacobians of {x0.size + 1} residual evaluations) of "
                     f"scipy.optimize.least_squares on the C restatement of residual_function (oracle/c), "
                     f"8 cams x {obs.shape[0]} points, {dt:.1f}s", "ms_per_residual_evaluation": 1e3 * t_eval}
    # the NumPy/Python restatement keeps the reference's structure (per-point Python loops, LAPACK SVD per point) and
    # is bit-exact against it: one residual evaluation timed, an iteration costs n + 2 of them
    try:
        from oracle import mocap_oracle as mo
        sub = obs[:100]
        t2 = time.perf_counter()
        mo.ba_residuals(x0, sub, [k for k in rig["K"]])
        t_py = (time.perf_counter() - t2) * obs.shape[0] / sub.shape[0]
        out["python_port_iterations_per_s_est"] = 1.0 / ((x0.size + 2) * t_py)
        out["python_port_sample"] = (f"one residual evaluation of 100 points by oracle/mocap_oracle.py scaled to {obs.shape[0]}, "
                                     f"x (n + 2) evaluations per accepted step")
    except Exception as e:  # pragma: no cover
        out["python_port_error"] = repr(e)
    return out


def ba_parity(core):
    """Reference-mode bundle adjustment on the solver goldens (tests/golden/ba_*: the reference's own poses and
    OptimizeResult statistics, plus its reproducibility under a 1e-15 nudge of the start vector): pose deltas of
    mode "scipy" (reference optimizer, GPU residuals) and mode "resident" (mocap_ba_solve)."""
    from mocap_core import helpers
    out = {}
    gdir = os.path.join(ROOT, "tests", "golden")
    for name in ("ba_c3_n24", "ba_c4_n60_solved", "ba_c6_n80_solved", "ba_c8_n100_solved"):
        path = os.path.join(gdir, name + ".npz")
        if not os.path.exists(path):
            continue
        g = np.load(path)
        C = g["K"].shape[0]
        helpers.set_core(core)
        helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in g["K"]])
        poses0 = [{"R": g["R_init"][i].copy(), "t": g["t_init"][i].copy()} for i in range(C)]
        row = {"reference": {"nfev": int(g["ba_stats"][0]), "njev": int(g["ba_stats"][1]),
                             "self_dR_max": float(g["self_dR"].max()), "self_dt_rel_max": float(g["self_dt"].max())}}
        for mode in ("scipy", "resident"):
            with helpers.bundle_adjustment_mode(mode):
                poses, info = helpers.bundle_adjustment(synth.obs_to_reference_array(g["obs"]), poses0, None, return_info=True)
            R = np.array([np.asarray(p["R"], dtype=np.float64) for p in poses])
            t = np.array([np.asarray(p["t"], dtype=np.float64).reshape(3) for p in poses])
            row[mode] = {"nfev": int(info["nfev"]), "njev": int(info["njev"]), "dR_max": float(np.abs(R - g["R_ba"]).max()),
                         "dt_rel_max": float(np.abs(t - g["t_ba"]).max() / np.abs(g["t_ba"]).max())}
        out[name] = row
    return out


def ba_bench(core, iters=200, cpu=True):
    """Secondary metric: LM iterations/sec, 8 cams x 1000 points, reference settings
    (cauchy loss, float32 residual cast, 2-point Jacobian incl. the dead focal columns)."""
    from mocap_core import helpers
    rig = synth.ring_rig(CAMS)
    rng = np.random.default_rng(7)
    obs, _ = synth.make_ba_observations(rig, 1000, seed=7)
    init = synth.perturb_rig(rig, rng)
    core.set_cameras(rig["K"], init["R"], init["t"])
    helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in rig["K"]])
    x0 = helpers._ba_x0([{"R": init["R"][i], "t": init["t"][i]} for i in range(CAMS)])
    # tolerances 0: the loop runs until its evaluation budget is spent.  One iteration = one
    # accepted-or-rejected trust-region step including its Jacobian (n+1 residual evaluations of all
    # points, robust scaling, the MFMA J^T J / J^T f, the n x n subproblem and the trial evaluation).
    runs, n_warm = ba_timed_solves(core, x0, obs, iters)
    dt, info = runs[len(runs) // 2]
    _, info_ref = core.ba_solve(x0, obs, ftol=1e-2)                             # reference stopping rule
    prof = core.ba_profile(x0, obs, reps=200)
    n, m = int(x0.size), int(info["m"])
    gram_flop = 2.0 * m * n * n                                                  # J^T J, 2 m n^2
    mfma_tf = gram_flop / (prof["gpu_us_per_linearisation"] * 1e-6) / 1e12
    roofline = {"bound": "mfma", "achieved": mfma_tf, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": mfma_tf / FP64_MFMA_PEAK_TF, "traffic": None,
                "kernel": ("mocap::ba_fused_kernel: one launch per linearisation (camera tables, (n_live + 1) x N residuals, "
                           "float32 differencing, Cauchy scaling, v_mfma_f64_16x16x4_f64 Gram, reduction, zero-copy hand-over)")
                          if prof["fused"] else "chain of 5 launches (build cameras, residuals, Jacobian, Gram, reduce)",
                "kernel_us": prof["gpu_us_per_linearisation"], "launches_per_linearisation": int(prof["launches"]),
                "algorithmic_flop_per_launch": gram_flop,
                "algorithmic_bytes_per_launch": 16 * CAMS * int(obs.shape[0]) + 8 * (n + 1) * (n + 2) // 2,
                "iteration_breakdown_us": {"gpu_linearisation": prof["gpu_us_per_linearisation"],
                                           "launch_to_result_on_host": prof["wall_us_per_linearisation"],
                                           "host_trust_region_subproblem": prof["host_tr_us"],
                                           "whole_iteration": 1e6 * dt / max(info["iterations"], 1)},
                "note": "latency bound by construction: 5 Mflop of J^T J per iteration is microseconds on the matrix cores; "
                        "the iteration is a chain of dependent steps (launch, residual phase, two cross-XCD hand-offs, "
                        "PCIe hand-over, host subproblem).  MFMA issue counters: profiles/r02_ba_pmc_mfma.csv"}
    out_cpu = ba_cpu_baseline(rig, init, obs, x0) if cpu else None
    # what index.py:272 reaches by DEFAULT: helpers.bundle_adjustment in mode "scipy" -- the reference's own optimizer call,
    # residuals on the GPU (poses bit-identical to the reference's on the solver goldens) -- on the same 8 cams x 1 000 points
    ref_obs = synth.obs_to_reference_array(obs)
    poses0 = [{"R": init["R"][i].copy(), "t": init["t"][i].copy()} for i in range(CAMS)]
    helpers.set_core(core)
    d_runs = []
    with helpers.bundle_adjustment_mode("scipy"):
        # one untimed solve first: bundle adjustment runs on its own lazily created context (mocap_create, allocations,
        # first launches of every kernel) -- a one-off per process that is not part of a calibration's cost
        helpers.bundle_adjustment(ref_obs, [dict(p) for p in poses0], None, return_info=True)
        for _ in range(3):
            t0 = time.perf_counter()
            _, dinfo = helpers.bundle_adjustment(ref_obs, [dict(p) for p in poses0], None, return_info=True)
            d_runs.append((time.perf_counter() - t0, float(dinfo.get("core_s", 0.0))))
    d_dt, d_core = sorted(d_runs)[1]
    # The same solve with the host's BLAS pinned to ONE thread for its duration.  SciPy's share of this mode is an SVD of the
    # m x n Jacobian per iteration; on a many-core host whose cores are shared (the bench boxes: 256 threads, load average 30-50)
    # OpenBLAS's 64 spinning threads make that SVD slower, starve the HIP runtime's own threads (inside_core_calls_s grows from
    # ~0.01 s of GPU work to 0.4-0.8 s) and change the result's last bits with the thread count.  Measured, not imposed: the mirror
    # leaves the process's BLAS settings alone, as the reference does.
    one_thread = None
    try:
        from threadpoolctl import threadpool_limits
        o_runs = []
        with threadpool_limits(limits=1), helpers.bundle_adjustment_mode("scipy"):
            for _ in range(3):
                t0 = time.perf_counter()
                _, oinfo = helpers.bundle_adjustment(ref_obs, [dict(p) for p in poses0], None, return_info=True)
                o_runs.append((time.perf_counter() - t0, float(oinfo.get("core_s", 0.0))))
        o_dt, o_core = sorted(o_runs)[1]
        one_thread = {"wall_s": o_dt, "iterations_per_s": oinfo["njev"] / o_dt, "inside_core_calls_s": o_core, "scipy_own_s": o_dt - o_core,
                      "njev": int(oinfo["njev"]), "nfev": int(oinfo["nfev"]), "runs_s": [round(r[0], 4) for r in o_runs],
                      "how": "threadpoolctl.threadpool_limits(limits=1) around helpers.bundle_adjustment"}
    except Exception as e:  # pragma: no cover
        one_thread = {"error": repr(e)}
    default_mode = {"mode": helpers.DEFAULT_BA_MODE, "measured_mode": "scipy", "wall_s": d_dt, "njev": int(dinfo["njev"]),
                    "nfev": int(dinfo["nfev"]), "iterations_per_s": dinfo["njev"] / d_dt,
                    "residual_evaluations_per_s": (dinfo["nfev"] + dinfo["njev"] * x0.size) / d_dt,
                    "runs_s": [round(r[0], 4) for r in d_runs], "statistic": "median of 3 solves after one untimed",
                    "inside_core_calls_s": d_core, "scipy_own_s": d_dt - d_core, "one_blas_thread": one_thread,
                    "host_load_average": (os.getloadavg()[0] if hasattr(os, "getloadavg") else None),
                    "note": "the seam's default: scipy.optimize.least_squares drives; a trial point is one mocap_ba_residuals "
                            "call, a Jacobian is ONE call too (jac= callable: the n perturbed parameter vectors of scipy's "
                            "2-point rule as a batch, J formed with scipy's own float32-difference / float64-quotient "
                            "expressions: same bits as the reference's n + 1 separate evaluations); `value` above is mode "
                            "\"resident\" (mocap_ba_solve), opt-in via helpers.set_bundle_adjustment_mode.  inside_core_calls_s = wall time spent "
                            "inside mocap_ba_residuals (marshalling, GPU, copy back); scipy_own_s = the rest: SciPy's own "
                            "per-iteration work on the host (SVD of the m x n Jacobian, Cauchy scaling, the step) which "
                            "bit-identical poses oblige this mode to keep"}
    return {"metric": "BA iters/sec (8 cams, 1k pts), mode \"resident\" (mocap_ba_solve: whole LM loop on the GPU, opt-in; inside the "
                      "reference's own reproducibility, NOT 1e-5 at 8 cameras -- DESIGN 4.1)",
            "value": info["iterations"] / dt, "measured_mode": "resident",
            # the seam's DEFAULT (mode "scipy": the reference's optimizer call, GPU residuals + batched Jacobian; poses
            # bit-identical to the reference on the solver goldens) beside it, not below it:
            "default_mode_iterations_per_s": default_mode["iterations_per_s"],
            "default_mode_metric": "BA iters/sec (8 cams, 1k pts), mode \"scipy\" (helpers.bundle_adjustment default; bit-identical poses)",
            "default_mode": default_mode, "roofline": roofline,
            "cpu_baseline": out_cpu, "parity": ba_parity(core),
            "iterations": info["iterations"], "nfev": info["nfev"], "ms_per_iter": 1e3 * dt / max(info["iterations"], 1),
            "runs_ms": [round(1e3 * r[0], 2) for r in runs], "statistic": "median of 5 solves", "warmup_solves": n_warm,
            "params": int(x0.size), "points": int(info["m"]),
            "reference_rule_run": {"iterations": info_ref["iterations"], "status": info_ref["status"],
                                   "cost0": info_ref["cost0"], "cost": info_ref["cost"],
                                   "elapsed_ms": info_ref["elapsed_ms"]}}


def device_identity(dev):
    """A string that tells two GPUs apart (UUID where torch exposes it, else PCI location + name)."""
    import torch
    pr = torch.cuda.get_device_properties(dev)
    for attr in ("uuid", "pci_bus_id"):
        v = getattr(pr, attr, None)
        if v is not None and str(v) not in ("", "0"):
            return f"{attr}:{v}"
    return f"name:{pr.name}/index:{dev.index}/pci:{getattr(pr, 'pci_domain_id', '?')}:{getattr(pr, 'pci_device_id', '?')}"


def full_batch_parity(local_rank, dev, stream, rig, M, gate, g_cap, d_blobs, d_counts, shipped):
    """AFTER the timed region: the whole timed batch once more through the exhaustive walk (MOCAP_OPT_EXHAUSTIVE_WALK: every
    candidate group triangulated and reprojected, no bound drops or cuts anything -- helpers.py:408-421 as written) on a second
    context, compared ON THE DEVICE with what the timed run left in its output buffers: every bit of n_out, status, corr, xyz,
    err of every frame.  `shipped` = a devcheck.FrameOutputs view of the timed run's buffers."""
    import torch
    from mocap_core import devcheck
    walk = capi.MocapCore(local_rank)
    try:
        walk.set_stream(stream.cuda_stream)
        walk.set_options(exhaustive_walk=True)
        walk.set_cameras(rig["K"], rig["R"], rig["t"])
        ref = devcheck.FrameOutputs(shipped.F, shipped.K, shipped.C, dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        ref.run(walk, M, d_blobs, d_counts, gate, g_cap)
        b.record(stream)
        cmp = devcheck.compare_bitwise(shipped, ref)
        kern = walk.last_frame_kernel()
        return {"frames_checked": cmp["frames"], "frames_differing": cmp["frames_differing"], "fields_differing": cmp["fields"],
                "first_differing_frames": cmp["first_differing_frames"], "exhaustive_kernel": kern, "exhaustive_ms": a.elapsed_time(b)}
    finally:
        torch.cuda.synchronize(dev)
        walk.close()


def config_record(name, local_rank, dev, stream, steps, warmup, frames=0):
    """A sub-record of the default line for one of BASELINE.json's other frame configs (configs[1] = "4x4", configs[4] =
    "64x256"): the same hot-path call as the headline (mocap_match_triangulate_dev_auto, inputs resident in HBM, HIP events on
    the launch stream), its own core context and buffers.  Carries ms_per_step, frames/s, markers/s, the overflow counts, an HBM
    roofline on SURVEY 8d's algorithmic bytes, the executed-FP64 figure where a committed counter pass exists, parity against
    the C oracle on a prefix and run-to-run bitwise equality of the whole batch."""
    import torch
    from mocap_core import devcheck
    from oracle import c_oracle
    wl = WORKLOADS[name]
    C, M, K_MAX = wl["C"], wl["M"], wl["K_max"]
    F = int(frames or wl.get("sub_frames", wl["frames"]))
    t_gen = time.perf_counter()
    if wl["stress"]:
        rig = synth.stress_rig(C)
        blobs, counts, _ = synth.make_stress_stream_chunked(rig, F, M, seed=1)
        gate = synth.STRESS_GATE_PX
    else:
        rig = synth.ring_rig(C)
        blobs, counts, _ = synth.make_blob_stream(rig, F, M, seed=1)
        gate = wl["gate"]
    t_gen = time.perf_counter() - t_gen
    g_cap = wl.get("g_cap", G_CAP)
    core = capi.MocapCore(local_rank)
    try:
        core.set_stream(stream.cuda_stream)
        core.set_cameras(rig["K"], rig["R"], rig["t"])
        d_blobs, d_counts = torch.from_numpy(blobs).to(dev), torch.from_numpy(counts).to(dev)
        out, again = devcheck.FrameOutputs(F, K_MAX, C, dev), devcheck.FrameOutputs(F, K_MAX, C, dev)
        out.run(core, M, d_blobs, d_counts, gate, g_cap, auto=False)      # (untimed: names the first pass's kernel)
        torch.cuda.synchronize(dev)
        first_kernel = core.last_frame_kernel()
        for _ in range(warmup):
            out.run(core, M, d_blobs, d_counts, gate, g_cap)
        torch.cuda.synchronize(dev)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for a, b in ev:
            a.record(stream)
            out.run(core, M, d_blobs, d_counts, gate, g_cap)
            b.record(stream)
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
        kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        again.run(core, M, d_blobs, d_counts, gate, g_cap)                  # run-to-run: every bit of the batch once more
        rr = devcheck.compare_bitwise(again, out)
        n_out, status = out.n_out.cpu().numpy(), out.status.cpu().numpy()
        n_cand = out.n_cand.cpu().numpy()
        info = out.info.cpu().numpy()
        markers = float(n_out[status == 0].sum())
        abytes = algorithmic_bytes(counts, np.where(status == 0, n_out, 0), C, M)
        ach = abytes / (kernel_ms * 1e-3) / 1e9
        nchk = min(F, wl.get("oracle_frames", 16))
        ref = c_oracle.COracle(rig["K"], rig["R"], rig["t"]).match_triangulate(blobs[:nchk], counts[:nchk], gate_px=gate, K_max=K_MAX)
        vv = (np.arange(K_MAX)[None, :] < ref["n_out"][:, None]) & (status[:nchk, None] == 0)
        xyz, corr = out.xyz[:nchk].cpu().numpy(), out.corr[:nchk].cpu().numpy()
        rec = {"workload": wl["desc"], "frames_per_gpu": F, "cams": C, "markers": M, "K_max": K_MAX, "gate_px": gate,
               "first_pass_G_cap": g_cap, "steps": steps, "warmup": warmup,
               "ms_per_step": 1e3 * wall / steps, "kernel_ms": kernel_ms, "frames_per_s": F * steps / wall,
               "value": markers * steps / wall, "unit": "markers/s", "markers_per_frame": markers / F,
               "candidates_per_frame": float(n_cand.mean()),
               "overflow_frames": int((status != 0).sum()), "flagged_by_first_pass": int(info[0]), "resubmitted_frames": int(info[1]),
               "overflow_by_cap": {"roots_K_max": int(((status & 1) != 0).sum()), "candidates_G_cap": int(((status & 2) != 0).sum()),
                                   "hits_per_root_and_camera": int(((status & 4) != 0).sum()),
                                   "intractable_roots_over_2^24_groups": int(((status & 16) != 0).sum())},
               "kernel": first_kernel + " (first pass) + device-side re-submit", "host_generation_s": t_gen,
               "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                            "traffic": None, "algorithmic_bytes_per_launch": abytes, "kernel_ms": kernel_ms,
                            "note": "all launches of one hot-path call (first pass + device-side re-submit), HIP events"},
               "parity": {"frames_checked": nchk, "against": "oracle/c",
                          "n_out_equal": bool(np.array_equal(ref["n_out"][status[:nchk] == 0], n_out[:nchk][status[:nchk] == 0])),
                          "corr_bit_exact": bool(np.array_equal(ref["corr"][vv], corr[vv])),
                          "xyz_max_rel": float(np.abs(xyz[vv] - ref["xyz"][vv]).max() / np.abs(ref["xyz"][vv]).max()) if vv.any() else None,
                          "run_to_run": {"frames_checked": rr["frames"], "frames_differing": rr["frames_differing"],
                                         "bit_identical": rr["frames_differing"] == 0}}}
        if wl["stress"]:
            # the same batch with MOCAP_OPT_BOUNDED_RESUBMIT: roots of 2^16 .. 2^24 groups the exact search gives up on are flagged
            # instead of enumerated by the whole GPU (~3 ms each): the price of the default's last few frames, measured
            core.set_options(bounded_resubmit=True)
            out.run(core, M, d_blobs, d_counts, gate, g_cap)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(steps):
                out.run(core, M, d_blobs, d_counts, gate, g_cap)
            torch.cuda.synchronize(dev)
            w2 = time.perf_counter() - t1
            st2, no2 = out.status.cpu().numpy(), out.n_out.cpu().numpy()
            rec["bounded_resubmit"] = {"option": "MOCAP_OPT_BOUNDED_RESUBMIT", "ms_per_step": 1e3 * w2 / steps, "frames_per_s": F * steps / w2,
                                       "value": float(no2[st2 == 0].sum()) * steps / w2, "overflow_frames": int((st2 != 0).sum())}
            core.set_options(bounded_resubmit=False)
        mix, stale, path = load_profile(f"fp64_mix_{name}")
        if mix is not None:
            tf = float(mix["fp64_flop_per_frame"]) * F / (kernel_ms * 1e-3) / 1e12
            lu = mix.get("vector_lane_utilisation")
            rec["roofline_fp64"] = {"bound": "fp64_valu", "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s", "achieved": tf,
                                    "frac": tf / FP64_VALU_PEAK_TF, "frac_active_lanes": tf / FP64_VALU_PEAK_TF * lu if lu else None,
                                    "is_from_profiles": True, "stale": bool(stale), "from_profiles_source": path,
                                    "flop_per_frame_from_profiles": mix["fp64_flop_per_frame"]}
            if mix.get("hbm_bytes_per_frame"):
                rec["roofline"]["traffic"] = float(mix["hbm_bytes_per_frame"]) * F
                rec["roofline"]["traffic_is_from_profiles"] = True
        return rec
    finally:
        torch.cuda.synchronize(dev)
        core.close()


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher: re-run this script under torch.distributed.run, one rank per
    GPU on this node (rendezvous on 127.0.0.1, a free port), same arguments; returns the launcher's exit code.
    The ranks print nothing but rank 0's JSON line, which passes through on stdout."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n_gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_run(args):
    """MOCAP_BENCH_DRY=1: the N-rank control flow of this script WITHOUT GPUs -- launcher, rank bookkeeping, the
    compact count-first exchange with two transfers in flight and buffer reuse, max-over-ranks timing, one rank-0
    line -- over gloo with made-up track records.  A functional check of the multi-process path (tests/), never a
    measurement: the line says "dry_run": true and carries no value."""
    import torch
    import torch.distributed as dist
    rank, _, world = mdist.init_process_group(backend="gloo")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    wl = WORKLOADS[args.workload]
    C, K = wl["C"], (16 if args.workload == "8x16" else min(wl["K_max"], 40))   # the record stride follows the camera count (64 x 256: 160 B)
    F = (args.frames or 64) + rank                       # uneven shards
    frames_per_rank = [(args.frames or 64) + r for r in range(world)]
    stride = mdist.track_record_bytes(C)
    t0 = time.perf_counter()
    pending, sums, sent = [], [], 0
    for step in range(args.warmup + args.steps):
        rng = np.random.default_rng(1000 * step + rank)
        n_out = rng.integers(0, K + 1, size=F).astype(np.int32)
        xyz, err = rng.standard_normal((F, K, 3)), rng.random((F, K))
        corr = rng.integers(-1, 16, size=(F, K, C)).astype(np.int16)
        rec, off = mdist.compact_tracks_reference(n_out, xyz, err, corr)
        cap = torch.zeros((F * K, stride), dtype=torch.uint8)
        cap[:rec.shape[0]] = torch.from_numpy(rec)
        sums.append(int(rec.astype(np.int64).sum()) + int(n_out.sum()))
        sent += rec.shape[0]
        pending.append(mdist.gather_compact_async(torch.from_numpy(n_out), cap, int(off[-1]), frames_per_rank, dst=0))
        while len(pending) > 2:
            pending.pop(0).result()
    results = [h.result() for h in pending]               # the last two exchanges, still in flight
    elapsed = time.perf_counter() - t0
    local = torch.tensor([float(sums[-1]), float(sums[-2]) if len(sums) > 1 else 0.0, elapsed, float(sent)], dtype=torch.float64)
    allv = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(allv, local)
    allv = torch.stack(allv).numpy()
    if rank == 0:
        ok = True
        for h, col in zip(results[::-1], (0, 1)):
            n_all, r_all = h
            ok = ok and n_all.shape[0] == sum(frames_per_rank)
            ok = ok and int(r_all.numpy().astype(np.int64).sum()) + int(n_all.numpy().sum()) == int(allv[:, col].sum())
        print(json.dumps({"metric": "triangulated 3D markers/sec at 8 cams x 16 markers", "value": None, "unit": "markers/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True, "backend": "gloo",
                          "data": "made-up track records (no GPU): control flow and exchange only",
                          "config": {"frames_per_rank": frames_per_rank, "workload": args.workload, "record_bytes": int(stride),
                                     "exchange": {"format": "compact records, count in the first point-to-point message",
                                                  "payload_checksums_match": bool(ok),
                                                  "records_sent_all_ranks": int(allv[:, 3].sum())}},
                          "max_rank_seconds": float(allv[:, 2].max())}), flush=True)
        if not ok:
            raise SystemExit("dry run: gathered payload differs from what the ranks sent")
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="8x16")
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU per step (0 = the workload's default)")
    ap.add_argument("--chunks", type=int, default=0,
                    help="N > 1: sub-batches a rank's shard is cut into per step, so that the gather of chunk k travels while chunk "
                         "k + 1 is computed inside ONE step (0 = automatic: 1 = the whole shard at once when the run has >= 4 steps, whose "
                         "exchanges travel under the NEXT step's kernels; a shorter run: 4, or 2 at 64 x 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--no-blobs", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the 4x4 and 64x256 sub-records of the default line")
    ap.add_argument("--no-full-parity", action="store_true", help="skip the full-batch comparison with the exhaustive walk")
    args = ap.parse_args()

    # `python bench.py --gpus N` on its own (no launcher): start the N ranks here, one process per GPU, and pass
    # rank 0's line through
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    if os.environ.get("MOCAP_BENCH_DRY"):
        return dry_run(args)
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > torch.cuda.device_count() and not os.environ.get("MOCAP_DIST_BACKEND"):
        raise SystemExit(f"WORLD_SIZE={os.environ['WORLD_SIZE']} ranks but only {torch.cuda.device_count()} GPU(s) visible: "
                         "one process per GPU (RCCL refuses two ranks on one device)")
    rank, local_rank, world = mdist.init_process_group(backend=os.environ.get("MOCAP_DIST_BACKEND") or None)
    # MOCAP_DIST_BACKEND=gloo: a functional dry run of the N > 1 control flow on a box with fewer GPUs than ranks
    # (ranks share devices; RCCL itself refuses two ranks on one device) -- never a measurement
    local_rank = local_rank % torch.cuda.device_count()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    # ---- synthetic workload: each rank owns its own block of frames (seed differs per rank)
    wl = WORKLOADS[args.workload]
    C, M, F = wl["C"], wl["M"], (args.frames or wl["frames"])
    K_MAX = wl["K_max"]
    # groups per root the FIRST pass enumerates; the device-side re-submit repairs what it flags (swept at the stress shape,
    # a smaller cap makes the first pass faster and leaves more frames to a search that cannot solve all of them)
    g_cap = int(os.environ["MOCAP_BENCH_G_CAP"]) if "MOCAP_BENCH_G_CAP" in os.environ else wl.get("g_cap", G_CAP)
    if wl["stress"]:
        rig = synth.stress_rig(C)
        blobs, counts, _ = synth.make_stress_stream_chunked(rig, F, M, seed=1 + rank)
        gate = synth.STRESS_GATE_PX
    else:
        rig = synth.ring_rig(C)
        blobs, counts, _ = synth.make_blob_stream(rig, F, M, seed=1 + rank)
        gate = wl["gate"]
    default_wl = args.workload == "8x16"
    core = capi.MocapCore(local_rank)
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    stream = torch.cuda.current_stream(dev)
    core.set_stream(stream.cuda_stream)

    d_blobs = torch.from_numpy(blobs).to(dev)
    d_counts = torch.from_numpy(counts).to(dev)
    d_xyz = torch.empty((F, K_MAX, 3), dtype=torch.float64, device=dev)
    d_err = torch.empty((F, K_MAX), dtype=torch.float64, device=dev)
    d_corr = torch.empty((F, K_MAX, C), dtype=torch.int16, device=dev)
    d_nout = torch.zeros(F, dtype=torch.int32, device=dev)
    d_status = torch.zeros(F, dtype=torch.int32, device=dev)
    d_ncand = torch.zeros(F, dtype=torch.int32, device=dev)

    # {frames flagged by the first pass, frames re-run} per call of the hot path (device-side; read after the timed region)
    d_resub = torch.zeros((64, 2), dtype=torch.int32, device=dev)

    def hot_path(lo=0, hi=None, chunk=0):
        # the product path for device buffers: frame kernel, then -- queued behind it, no host wait -- the frames that hit a
        # cap re-run on the device with the largest caps (the reference has none, helpers.py:394-400) and scattered back.
        # Inside the timed region, so a figure never excludes its heaviest frames.
        hi = F if hi is None else hi
        core.match_triangulate_dev_auto(hi - lo, M, d_blobs[lo:].data_ptr(), d_counts[lo:].data_ptr(), gate, K_MAX, g_cap,
                                        d_xyz[lo:].data_ptr(), d_err[lo:].data_ptr(), d_corr[lo:].data_ptr(),
                                        d_nout[lo:].data_ptr(), d_status[lo:].data_ptr(), d_ncand[lo:].data_ptr(),
                                        d_resub[chunk % 64].data_ptr())

    # N > 1: the one exchange of the path: final tracks -> rank 0.  Only the valid points travel:
    # mocap_compact_tracks_dev packs them into 32 + 2C-byte records behind the frame kernel (1.1 KB instead of the
    # 2.3 KB of K_max-padded arrays per 8 x 16 frame), the record count lands in pinned host memory, and the
    # transfer of step i is posted -- on its own stream -- once step i + 1's kernels are queued, so it hides behind
    # them; only the last step's exchange is exposed.  MOCAP_BENCH_EXCHANGE=1 runs the same code path on one GPU.
    multi = world > 1 or bool(int(os.environ.get("MOCAP_BENCH_EXCHANGE", "0")))
    # A rank's shard is cut into sub-batches: the gather of chunk k is posted once chunk k + 1's kernels are queued, so the
    # exchange overlaps compute INSIDE a step too (a one-step run used to expose all of it: at 64 x 256 that is 532 MB per
    # rank, 3.7 GB into the root).  Each chunk has its own compactor buffers.
    # (runs of one to three steps: 4 sub-batches, 2 at the stress shape, where every sub-batch pays a kernel tail of ~1 ms and its
    # own re-submit -- gather, second pass, heavy-root search: 1.7 ms)
    # Round 6, measured on one GPU with the exchange code path on (scripts/gpu_r06_chunks.sh, 20 steps of 8 x 16): 1 / 2 / 4
    # sub-batches = 4.50 / 4.89 / 5.38 ms per step against 4.18 without the exchange; 64 x 256: 35.5 (1) / 40.8 (2) against 34.7.
    # A sub-batch costs ~0.3 ms (8 x 16) / ~5 ms (64 x 256) on EVERY step; what it buys -- a smaller exposed tail after the LAST
    # kernel -- is paid once per run, because step i's exchange is posted behind step i + 1's kernels anyway.  So: the whole shard
    # at once when there are steps to hide behind (>= 4), sub-batches only for a run of a few steps.
    auto_chunks = 1 if args.steps >= 4 else (2 if C * M >= 4096 else 4)
    n_chunks = 1 if not multi else max(1, min(args.chunks or auto_chunks, F))
    cb = [mdist.shard_bounds(F, c, n_chunks) for c in range(n_chunks)]
    comps = [mdist.TrackCompactor(core, hi - lo, K_MAX, C, dev) for lo, hi in cb] if multi else []
    comm = torch.cuda.Stream(dev) if multi else None
    exchanged = {"records": 0, "bytes": 0}

    def post_exchange(c, i):
        comp = comps[c]
        n = comp.count(i)                  # waits for this chunk's compaction only
        exchanged["records"] += n
        exchanged["bytes"] += n * comp.stride + 4 * comp.F
        with torch.cuda.stream(comm):
            comm.wait_event(comp.events[i])
            # the handle stays attached to buffer i: compact() will not overwrite it while the exchange reads it
            return comp.attach(i, mdist.gather_compact_async(comp.n_out[i], comp.records[i], n, [comp.F] * world, dst=0))

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    exposed = {"ms": None}

    def run(n_steps, events=None):
        prev, pending = None, []
        for i in range(n_steps):
            if events:
                events[i][0].record(stream)
            if not multi:
                hot_path()
            for c, (lo, hi) in enumerate(cb if multi else []):
                hot_path(lo, hi, c)
                cur = comps[c].compact(d_nout[lo:hi], d_xyz[lo:hi], d_err[lo:hi], d_corr[lo:hi], stream)
                if prev is not None:               # chunk k's transfer is posted behind chunk k + 1's kernels
                    pending.append(post_exchange(*prev))
                    while len(pending) > 2:        # at most two exchanges in flight (bounds the staging memory)
                        pending.pop(0).result()
                prev = (c, cur)
            if events:
                events[i][1].record(stream)
        if multi and prev is not None:
            # what is left when the last kernel has finished = the exposed part of the exchange
            e_comp, e_all = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e_comp.record(stream)
            pending.append(post_exchange(*prev))
            for h in pending:
                h.result()
            pending = []
            stream.wait_stream(comm)
            e_all.record(stream)
            e_all.synchronize()
            exposed["ms"] = e_comp.elapsed_time(e_all)
        for h in pending:
            h.result()

    run(args.warmup)
    fence()
    exchanged["records"] = exchanged["bytes"] = 0
    # kernel time with HIP events on the stream the kernel is launched on (torch's current stream)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    run(args.steps, ev)
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    # parity at the scale of the metric (round-5 verdict, item 1b): the batch the timed run just produced against the exhaustive
    # walk of the same batch, every bit of every frame, on the device.  Outside the timed region; on every rank.
    full_par = None
    if not wl["stress"] and not args.no_full_parity:
        from mocap_core import devcheck
        shipped = devcheck.FrameOutputs.__new__(devcheck.FrameOutputs)
        shipped.F, shipped.K, shipped.C = F, K_MAX, C
        shipped.xyz, shipped.err, shipped.corr, shipped.n_out, shipped.status, shipped.n_cand = d_xyz, d_err, d_corr, d_nout, d_status, d_ncand
        full_par = full_batch_parity(local_rank, dev, stream, rig, M, gate, g_cap, d_blobs, d_counts, shipped)
    # which devices took part: one identity string per rank, gathered over the process group
    ident = device_identity(dev)
    ident_t = torch.zeros(96, dtype=torch.uint8, device=dev)
    raw = ident.encode()[:96]
    ident_t[:len(raw)] = torch.tensor(list(raw), dtype=torch.uint8, device=dev)

    n_out = d_nout.cpu().numpy()
    status = d_status.cpu().numpy()
    n_cand = d_ncand.cpu().numpy()
    resub = d_resub[:max(n_chunks, 1)].cpu().numpy()        # the last step's calls
    local = torch.tensor([float(n_out.sum()), elapsed, float(status.astype(bool).sum()), float((status & 1).astype(bool).sum()),
                          float((status & 2).astype(bool).sum()), float((status & 4).astype(bool).sum()),
                          float(exposed["ms"] or 0.0), float(resub[:, 0].sum()), float(resub[:, 1].sum()),
                          float(full_par["frames_checked"] if full_par else 0), float(full_par["frames_differing"] if full_par else 0),
                          float((status & 16).astype(bool).sum())],
                         dtype=torch.float64, device=dev)
    if world > 1:
        allv = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(allv, local)
        allv = torch.stack(allv).cpu().numpy()
        ids = [torch.zeros_like(ident_t) for _ in range(world)]
        dist.all_gather(ids, ident_t)
        idents = [bytes(t.cpu().numpy().tobytes()).rstrip(b"\0").decode(errors="replace") for t in ids]
    else:
        allv = local.cpu().numpy()[None]
        idents = [ident]
    total_markers = float(allv[:, 0].sum())
    t_max = float(allv[:, 1].max())

    if rank == 0:
        value = total_markers * args.steps / t_max
        abytes = algorithmic_bytes(counts, n_out, C, M)
        ach = abytes / (kernel_ms * 1e-3) / 1e9
        # FP64 work model: per candidate 92*v + 1500 flop; v = views of the kept groups
        corr = d_corr.cpu().numpy()
        valid = np.arange(K_MAX)[None, :] < n_out[:, None]
        v_mean = float((corr[valid] >= 0).sum(axis=1).mean()) if valid.any() else float(C)
        flops = float(n_cand.sum()) * (92.0 * v_mean + 1500.0)
        traffic, traffic_meta = measured_traffic(F) if default_wl else (None, {})
        executed = executed_fp64(float(n_cand.sum()), kernel_ms) if default_wl else None
        kernel_name = core.last_frame_kernel() if hasattr(core, "last_frame_kernel") else "mocap::frame_kernel"
        line = {
            "metric": "triangulated 3D markers/sec at 8 cams x 16 markers" if default_wl
                      else f"triangulated 3D markers/sec at {C} cams x {M} markers",
            "value": value, "unit": "markers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            # who took part: a SCALE line proves N distinct GPUs by N distinct identities gathered over the process group
            "world_size": world, "backend": (dist.get_backend() if world > 1 else None),
            "devices": idents, "distinct_devices": len(set(idents)),
            "ms_per_step": 1e3 * t_max / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl["desc"],
                       "frames_per_gpu": F, "cams": C, "markers": M, "K_max": K_MAX, "gate_px": gate, "first_pass_G_cap": g_cap,
                       "parallelism": f"frame-shard x{world}", "frames_per_s": F * world * args.steps / t_max,
                       "markers_per_frame": total_markers / (F * world),
                       "candidates_per_frame": float(n_cand.mean()), "overflow_frames": int(allv[:, 2].sum()),
                       "resubmitted_frames": int(allv[:, 8].sum()), "flagged_by_first_pass": int(allv[:, 7].sum()),
                       "overflow_by_cap": {"roots_K_max": int(allv[:, 3].sum()), "candidates_G_cap": int(allv[:, 4].sum()),
                                           "hits_per_root_and_camera": int(allv[:, 5].sum()),
                                           "intractable_roots_over_2^24_groups": int(allv[:, 11].sum()),
                                           "note": "AFTER the device-side re-submit (mocap_match_triangulate_dev_auto, inside the timed "
                                                   f"region): frames the first pass flagged (G_cap = {g_cap} groups per root, K_max "
                                                   "roots, hit cap) are re-run per step with C x M roots and every hit; a root of more "
                                                   "than 4096 groups goes to the heavy-root search (csrc/heavy_bb.hip: exact branch "
                                                   "and bound over its multi-hit cameras, whatever the size of the product -- 2^60 for "
                                                   "two markers behind each other); a root the search gives up on is enumerated -- in "
                                                   "place up to 2^16 groups, by the whole GPU (heavy_enum_kernel) up to 2^24.  What is still "
                                                   "counted here: frames with a root of MORE than 2^24 groups the search could not bound "
                                                   "(MOCAP_ST_INTRACTABLE: a marker dropped out of cameras where ANOTHER marker's blob is the "
                                                   "root's only hit, every group carries views hundreds of pixels off, no bound separates "
                                                   "the mixtures; the reference would enumerate 2^25 .. 2^55 groups and not return)"},
                       "exchange": ({"format": "compact records (32 + 2C bytes per valid point) + n_out per frame, count-first "
                                               "point-to-point gather on rank 0",
                                     "bytes_per_rank_per_step": exchanged["bytes"] / max(args.steps, 1),
                                     "chunks_per_step": n_chunks,
                                     "exposed_ms": float(allv[:, 6].max()),
                                     "exposed_note": "per run, max over ranks: from the last kernel's end to the last transfer's "
                                                     "completion on the root (the final chunk's gather; every other chunk travels "
                                                     "under the next chunk's kernels)",
                                     "padded_format_bytes_per_step": F * (4 + K_MAX * (32 + 2 * C))} if multi else None)},
            "roofline": dict({"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                              "kernel": kernel_name + " (one persistent launch per pass, HIP events)",
                              "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": abytes,
                              "note": "path is FP64-VALU bound (~1e3 flop/byte), see roofline_fp64"}, **traffic_meta),
            # frac = EXECUTED FP64 flop (PMC instruction counts of this kernel) / FP64 vector peak; the SURVEY 8d work
            # model (what the reference computes per candidate) is reported as a rate without a fraction: the kernel
            # drops most candidate groups on exact eigenvalue bounds, so that rate is not work the hardware did
            "roofline_fp64": dict({"bound": "fp64_valu", "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s"},
                                  **(executed if executed else {"achieved": None, "frac": None}),
                                  reference_work_rate_tflops=flops / (kernel_ms * 1e-3) / 1e12,
                                  reference_work_model="SURVEY 8d: candidates x (92 v + 1500) flop, v = mean views: the "
                                                       "REFERENCE's work disposed of per second (may exceed the peak), "
                                                       "not work executed", v_mean=v_mean,
                                  candidates_per_launch=float(n_cand.sum())),
        }
        if world == 1:
            # parity gate next to the number: a prefix of the very batch that was timed, vs the oracle
            from oracle import c_oracle
            nchk = 300
            nchk = nchk if default_wl else min(F, 300 if C * M <= 64 else 64)
            ref = c_oracle.COracle(rig["K"], rig["R"], rig["t"]).match_triangulate(blobs[:nchk], counts[:nchk],
                                                                                   gate_px=gate, K_max=K_MAX)
            vv = valid[:nchk]
            xyz = d_xyz[:nchk].cpu().numpy()
            line["parity"] = {
                "oracle_prefix_frames": nchk,
                # the WHOLE timed batch vs the exhaustive walk on the device (bench.py full_batch_parity): the headline check
                "frames_checked": int(allv[:, 9].sum()) if full_par else nchk,
                "full_batch_vs_exhaustive_bit_exact": (bool(allv[:, 10].sum() == 0) if full_par else None),
                "full_batch": full_par,
                "n_out_equal": bool(np.array_equal(ref["n_out"], n_out[:nchk])),
                "corr_bit_exact": bool(np.array_equal(ref["corr"][vv], corr[:nchk][vv])),
                "xyz_max_rel": float(np.abs(xyz[vv] - ref["xyz"][vv]).max() / np.abs(ref["xyz"][vv]).max()),
            }
            if not args.no_cpu_baseline and default_wl:
                line["cpu_baseline"] = cpu_baseline(rig, blobs, counts, gpu={"n_out": n_out, "corr": corr, "xyz": d_xyz.cpu().numpy()})
                if "wide_parity" in line["cpu_baseline"]:
                    line["parity"]["wide"] = line["cpu_baseline"].pop("wide_parity")
                line["config"]["host_cores"] = os.cpu_count()
            if default_wl and not args.no_latency:
                core.set_stream(0)
                line["latency"] = frame_latency(core, blobs, counts)
                line["latency"].update(chain_latency(core, blobs, counts))
                core.set_cameras(rig["K"], rig["R"], rig["t"])
            if not args.no_blobs and default_wl:
                line["blob_stage"] = blob_stage_bench(core, dev, stream)
            if default_wl and not args.no_configs:
                # BASELINE.json configs[1] and configs[4] in the driver's own line (round-5 verdict, item 2)
                core.set_stream(stream.cuda_stream)
                line["configs"] = {}
                for name, st_, wu_ in (("4x4", 5, 2), ("64x256", 3, 1)):
                    try:
                        line["configs"][name] = config_record(name, local_rank, dev, stream, st_, wu_)
                    except Exception as e:  # pragma: no cover
                        line["configs"][name] = {"error": repr(e)}
            if not args.no_ba and default_wl:
                core.set_stream(0)
                line["ba"] = ba_bench(core, cpu=not args.no_cpu_baseline)
                line["ba"]["calibration_16k_points"] = ba_bench_16k(core)
        if world > 1 and full_par:
            line["parity"] = {"frames_checked": int(allv[:, 9].sum()), "full_batch_vs_exhaustive_bit_exact": bool(allv[:, 10].sum() == 0),
                              "frames_differing_per_rank": [int(v) for v in allv[:, 10]],
                              "note": "every rank: its whole timed shard vs the exhaustive walk (MOCAP_OPT_EXHAUSTIVE_WALK) on the device"}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
