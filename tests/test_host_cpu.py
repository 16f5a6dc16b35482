"""CPU-only checks (no GPU in this container): the C-ABI library loads and exports every symbol the
header declares, the product path fails loudly without a GPU (no silent fallback), host-side
marshalling, the synthetic generator, and the frame-sharded N>1 path over gloo (world_size 2)."""
import os
import re
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, gpu_available, load_golden

HEADER = os.path.join(ROOT, "include", "mocap_core.h")


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mocap_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from mocap_core import capi
    syms = _declared_symbols()
    assert len(syms) >= 17
    lib = capi.load_library()                       # dlopen works without a GPU
    for name in syms:
        assert name in capi.SIGNATURES, f"{name} declared in the header but not bound"
        assert getattr(lib, name) is not None
    assert set(capi.SIGNATURES) == set(syms)
    assert lib.mocap_version().decode().startswith("mocap_core")


@pytest.mark.skipif(gpu_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly_not_silently():
    from mocap_core import capi, helpers
    with pytest.raises(capi.MocapError):
        capi.MocapCore(0)
    helpers.set_camera_params([{"intrinsic_matrix": [[320, 0, 160], [0, 320, 160], [0, 0, 1]]}] * 2)
    poses = [{"R": np.eye(3).tolist(), "t": [0, 0, 0]}, {"R": np.eye(3).tolist(), "t": [1, 0, 0]}]
    helpers._state["core"] = None
    with pytest.raises(capi.MocapError):            # the seam has no CPU path to fall back to
        helpers.triangulate_points([[[1, 2], [3, 4]]], poses)


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "low-cost-mocap_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f"{f} mentions the oracle"


def test_synth_is_deterministic_and_well_formed():
    from mocap_core import synth
    rig = synth.ring_rig(8)
    assert np.allclose(rig["R"][0], np.eye(3)) and np.allclose(rig["t"][0], 0)
    for c in range(8):
        assert np.allclose(rig["R"][c] @ rig["R"][c].T, np.eye(3), atol=1e-12)
    a = synth.make_blob_stream(rig, 50, 16, seed=3)
    b = synth.make_blob_stream(rig, 50, 16, seed=3)
    assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1])
    blobs, counts, _ = a
    assert blobs.dtype == np.float32 and counts.dtype == np.int32 and blobs.shape == (50, 8, 16, 2)
    for f in range(50):
        for c in range(8):
            n = counts[f, c]
            assert not np.isnan(blobs[f, c, :n]).any() and np.isnan(blobs[f, c, n:]).all()
            assert (blobs[f, c, :n] == np.trunc(blobs[f, c, :n])).all()      # int() like _find_dot
    lists = synth.frame_to_reference_lists(blobs[0], counts[0])
    assert all(isinstance(v, int) for cam in lists for pt in cam for v in pt if v is not None)
    g = load_golden("frames_c8_m16")           # the generator reproduces the committed golden inputs
    again = synth.make_blob_stream(synth.ring_rig(8), 6, 16, seed=0)
    assert np.array_equal(again[0], g["blobs"], equal_nan=True) and np.array_equal(again[1], g["counts"])


def test_host_marshalling():
    from mocap_core import helpers
    pts = [[[10, 20], [30, 40]], [], [[5, 6]]]
    blobs, counts, rounded = helpers.pack_frame(pts)
    assert not rounded and blobs.shape == (1, 3, 2, 2) and counts.tolist() == [[2, 0, 1]]
    assert blobs[0, 0, 1].tolist() == [30.0, 40.0] and np.isnan(blobs[0, 1]).all()
    obs = helpers._obs_array([[[1, 2], [None, None]], [[3.5, 4.5], [6, 7]]], 2)
    assert obs.shape == (2, 2, 2) and np.isnan(obs[0, 1]).all() and obs[1, 0, 0] == 3.5
    arr = np.empty((1, 2, 2), dtype=object)
    arr[0, 0] = [1, 2]
    arr[0, 1] = [None, None]
    assert np.isnan(helpers._obs_array(arr, 2)[0, 1]).all()


def test_shard_bounds_partition():
    from mocap_core import dist as mdist
    for n in (0, 1, 7, 100, 100_001):
        for w in (1, 2, 3, 8):
            spans = [mdist.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "low-cost-mocap_amd"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from mocap_core import dist as mdist, synth
    from oracle import c_oracle
    mdist.init_process_group(backend="gloo")
    C, M, F, K = 4, 4, 65, 16                 # 65 frames over 2 ranks: uneven shards (33 + 32)
    rig = synth.ring_rig(C)
    blobs, counts, _ = synth.make_blob_stream(rig, F, M, seed=9)
    lo, hi = mdist.shard_bounds(F, rank, world)
    frames_per_rank = [b - a for a, b in (mdist.shard_bounds(F, r, world) for r in range(world))]
    # the per-rank compute is the GPU core in production; here (no GPU) the oracle stands in for it
    res = c_oracle.COracle(rig["K"], rig["R"], rig["t"]).match_triangulate(blobs[lo:hi], counts[lo:hi], K_max=K)
    # ---- compact exchange (what bench.py runs at N > 1): valid records only, count-first point-to-point gather.
    # mocap_compact_tracks_dev produces the records on the GPU; its host restatement stands in here (the GPU test
    # test_compaction_kernel_matches_reference pins the two against each other).
    rec_np, offsets = mdist.compact_tracks_reference(res["n_out"], res["xyz"], res["err"], res["corr"])
    cap = torch.zeros((res["err"].size, mdist.track_record_bytes(C)), dtype=torch.uint8)   # capacity F_r * K
    cap[:rec_np.shape[0]] = torch.from_numpy(rec_np)
    n_t = torch.from_numpy(res["n_out"].astype(np.int32))
    handles = [mdist.gather_compact_async(n_t, cap, int(offsets[-1]), frames_per_rank, dst=0) for _ in range(2)]
    compact = [h.result() for h in handles]
    # ---- fixed-capacity formats (kept for callers that want them): packed records, per-array gathers
    if hi - lo == max(frames_per_rank):
        pad = 0
    else:
        pad = max(frames_per_rank) - (hi - lo)      # gather needs equal shapes: pad the short shard with empty frames
    padded = {k: np.concatenate([v, np.zeros((pad,) + v.shape[1:], dtype=v.dtype)]) for k, v in res.items()
              if k in ("n_out", "xyz", "err", "corr")}
    rec = mdist.pack_records(torch.from_numpy(padded["n_out"]), torch.from_numpy(padded["xyz"]),
                             torch.from_numpy(padded["err"]), torch.from_numpy(padded["corr"]))
    allrec = mdist.gather_records(rec, dst=0)
    later = [h.result() for h in [mdist.gather_records_async(rec.clone(), dst=0) for _ in range(2)]]
    arrays = tuple(torch.from_numpy(padded[k]) for k in ("n_out", "xyz", "err", "corr"))
    parts = [h.result() for h in mdist.gather_tracks_async(arrays, dst=0)]
    if rank == 0:
        n_all, r_all = compact[0]
        assert all(torch.equal(n_all, c[0]) and torch.equal(r_all, c[1]) for c in compact)
        assert n_all.shape[0] == F
        got = mdist.unpack_compact(n_all.numpy(), r_all.numpy(), C, K)
        np.savez(out_path, **got, payload_bytes=r_all.numel() + 4 * F, padded_bytes=allrec.numel())
        # the fixed-capacity paths carry the same numbers (modulo the padding frames of the short shard)
        keep = np.concatenate([np.arange(max(frames_per_rank) * r, max(frames_per_rank) * r + frames_per_rank[r])
                               for r in range(world)])
        fixed = mdist.unpack_records(allrec, C, K)
        assert all(torch.equal(x, allrec) for x in later)
        valid = np.arange(K)[None, :] < got["n_out"][:, None]
        assert np.array_equal(fixed["n_out"][keep], got["n_out"])
        for key in ("xyz", "err", "corr"):
            assert np.array_equal(fixed[key][keep][valid], got[key][valid])
        nF = allrec.shape[0]
        assert np.array_equal(parts[0].numpy().view(np.int32).reshape(nF), fixed["n_out"])
        assert np.array_equal(parts[3].numpy().view(np.int16).reshape(nF, K, C), fixed["corr"])
    else:
        assert all(c is None for c in compact) and allrec is None and all(x is None for x in later + parts)
    dist.barrier()
    dist.destroy_process_group()


def test_frame_sharding_gloo_world2(tmp_path):
    """N-rank output == 1-rank output, bit for bit, through the single exchange -- compact format (valid records
    only), uneven shards (65 frames over 2 ranks)."""
    import torch.multiprocessing as mp
    from mocap_core import synth
    from oracle import c_oracle
    out = str(tmp_path / "gathered.npz")
    mp.start_processes(_worker, args=(2, _free_port(), out), nprocs=2, join=True, start_method="spawn")
    got = dict(np.load(out))
    rig = synth.ring_rig(4)
    blobs, counts, _ = synth.make_blob_stream(rig, 65, 4, seed=9)
    ref = c_oracle.COracle(rig["K"], rig["R"], rig["t"]).match_triangulate(blobs, counts, K_max=16)
    assert np.array_equal(got["n_out"], ref["n_out"])
    valid = np.arange(16)[None, :] < ref["n_out"][:, None]
    for key in ("xyz", "err", "corr"):
        assert np.array_equal(got[key][valid], ref[key][valid])
    assert got["payload_bytes"] < 0.6 * got["padded_bytes"]       # what the compaction is for


def test_compact_records_roundtrip():
    from mocap_core import dist as mdist
    rng = np.random.default_rng(1)
    F, K, C = 9, 7, 3
    n_out = rng.integers(0, K + 1, F).astype(np.int32)
    n_out[2] = 0
    xyz, err = rng.normal(size=(F, K, 3)), rng.normal(size=(F, K))
    corr = rng.integers(-1, 9, (F, K, C)).astype(np.int16)
    rec, off = mdist.compact_tracks_reference(n_out, xyz, err, corr)
    assert rec.shape == (int(n_out.sum()), mdist.track_record_bytes(C)) and off[-1] == n_out.sum()
    assert np.array_equal(np.diff(off), n_out)
    back = mdist.unpack_compact(n_out, rec, C, K)
    valid = np.arange(K)[None, :] < n_out[:, None]
    assert np.array_equal(back["n_out"], n_out)
    for k, v in (("xyz", xyz), ("err", err), ("corr", corr)):
        assert np.array_equal(back[k][valid], v[valid])
    assert np.isnan(back["xyz"][~valid]).all() and (back["corr"][~valid] == -1).all()


def test_record_pack_roundtrip():
    import torch
    from mocap_core import dist as mdist
    rng = np.random.default_rng(0)
    F, K, C = 5, 7, 3
    n_out = rng.integers(0, K, F).astype(np.int32)
    xyz, err = rng.normal(size=(F, K, 3)), rng.normal(size=(F, K))
    corr = rng.integers(-1, 9, (F, K, C)).astype(np.int16)
    rec = mdist.pack_records(*(torch.from_numpy(a) for a in (n_out, xyz, err, corr)))
    assert rec.shape == (F, mdist.record_bytes(C, K))
    back = mdist.unpack_records(rec, C, K)
    assert all(np.array_equal(back[k], v) for k, v in (("n_out", n_out), ("xyz", xyz), ("err", err), ("corr", corr)))


def test_pack_frame_carries_exactly_what_float32_can_and_says_when_it_rounds():
    """helpers.py:367-373 computes distances on image_points as given (int64 / float64); the C ABI's frame path carries
    float32.  Integer centroids (the reference's own, helpers.py:153-154) and float32-valued sub-pixel centroids go
    through unchanged; any other float64 coordinate is rounded to the nearest float32 and REPORTED (strict=True refuses
    it instead); NaN / inf are always refused -- the reference's seam accepts float64 lists, so the default must not raise."""
    import pytest
    from mocap_core import helpers
    b, c, rounded = helpers.pack_frame([[[12, 250], [319, 0]], [], [[1.5, 2.25]]])
    assert not rounded and b.dtype == np.float32 and c.tolist() == [[2, 0, 1]] and b[0, 2, 0].tolist() == [1.5, 2.25]
    b, _, rounded = helpers.pack_frame([[[float(np.float32(100.1)), 7.0]]])        # a float32 value held in a Python float
    assert not rounded and b[0, 0, 0, 0] == np.float32(100.1)
    b, _, rounded = helpers.pack_frame([[[100.1, 7.0]]])                            # 100.1 is not a float32 value: rounded, flagged
    assert rounded and b[0, 0, 0, 0] == np.float32(100.1)
    assert helpers.pack_frame([[[2 ** 24 + 1, 0]]])[2]                              # nor is this integer
    with pytest.raises(ValueError, match="float32"):
        helpers.pack_frame([[[100.1, 7.0]]], strict=True)
    with pytest.raises(ValueError, match="NaN"):
        helpers.pack_frame([[[float("nan"), 7.0]]])


def test_object_points_payload_is_the_reference_dict():
    """helpers.py:128-133: {"object_points": object_points.tolist(), "errors": errors.tolist(), "objects": [{k: (v.tolist() if
    ndarray else v)}], "filtered_objects": filtered_objects} -- host formatting only, no GPU."""
    import json
    from mocap_core import helpers
    pts = np.array([[0.1, 0.2, 0.3], [1.0, 2.0, 3.0]])
    errs = np.array([0.5, 1.5])
    objects = [{"pos": np.array([0.55, 1.1, 1.65]), "heading": -0.25, "error": 1.0, "droneIndex": 1}]
    filtered = [{"pos": [0.5, 1.0, 1.6], "vel": [0.0, 0.0, 0.0], "heading": -0.2, "droneIndex": 1}]
    got = helpers.object_points_payload(errs, pts, objects, filtered)
    want = {"object_points": pts.tolist(), "errors": errs.tolist(),
            "objects": [{k: (v.tolist() if isinstance(v, np.ndarray) else v) for (k, v) in o.items()} for o in objects],
            "filtered_objects": filtered}
    assert got == want and list(got) == ["object_points", "errors", "objects", "filtered_objects"]
    json.dumps(got)
    empty = helpers.object_points_payload(np.array([]), np.array([]), [])
    assert empty == {"object_points": [], "errors": [], "objects": [], "filtered_objects": []}


def test_bundle_adjustment_mode_context_manager_restores_the_mode():
    from mocap_core import helpers
    assert helpers._state["ba_mode"] == helpers.DEFAULT_BA_MODE == "scipy"
    try:
        with helpers.bundle_adjustment_mode("resident"):
            assert helpers._state["ba_mode"] == "resident"
            raise RuntimeError("inside")
    except RuntimeError:
        pass
    assert helpers._state["ba_mode"] == "scipy"


def test_divmod_tiny_is_exact_for_every_operand_the_block_decode_can_see():
    """frame_common.hpp divmod_tiny: q = trunc(fma(float(rem), rcp(float(n)), 2^-8)) for rem < 2^13, n <= 64 (a candidate's
    offset inside its block against a hit count, frame_bb.hip) -- the true quotient for every operand pair, also with the
    hardware reciprocal (v_rcp_f32: 1 ulp) off by an ulp in either direction.  float64 holds rem * inv + 2^-8 exactly
    (13 + 24 bits), so rounding it to float32 IS the fused multiply-add."""
    rem = np.arange(1 << 13, dtype=np.int64)[:, None]
    n = np.arange(1, 65, dtype=np.int64)[None, :]
    inv0 = (np.float32(1.0) / n.astype(np.float32)).astype(np.float32)
    for inv in (inv0, np.nextafter(inv0, np.float32(0)), np.nextafter(inv0, np.float32(2))):
        q = (rem.astype(np.float64) * inv.astype(np.float64) + 2.0 ** -8).astype(np.float32).astype(np.int64)
        assert np.array_equal(q, rem // n)
    # the launch keeps the operands inside that range: block sizes are lowered until bb_pl * M_max < 2^13 (capi.hip), and
    # the search kernel takes M_max <= 64 only (frame_bb_fits)
    src = open(os.path.join(ROOT, "low-cost-mocap_amd", "csrc", "capi.hip")).read()
    assert "(size_t)a.bb_pl * M_max * 2 * 256 >= ((size_t)1 << 22)" in src
    assert "M <= 64" in open(os.path.join(ROOT, "low-cost-mocap_amd", "csrc", "frame_bb.hip")).read()


def test_device_side_comparison_counts_exactly_the_valid_slots():
    """mocap_core/devcheck.compare_bitwise (bench.py's full-batch parity field, tests/test_gpu_bench_scale.py) on CPU tensors:
    equal batches compare equal whatever sits beyond n_out; one flipped bit in a valid slot, a NaN with another payload, a
    differing n_out or status each make their frame differ; NaNs with equal bits do not."""
    import torch
    from mocap_core import devcheck
    dev = torch.device("cpu")
    F, K, C = 50, 6, 3
    g = torch.Generator().manual_seed(5)
    a, b = devcheck.FrameOutputs(F, K, C, dev), devcheck.FrameOutputs(F, K, C, dev)
    a.n_out[:] = torch.randint(0, K + 1, (F,), generator=g, dtype=torch.int32)
    a.xyz[:] = torch.randn((F, K, 3), generator=g, dtype=torch.float64)
    a.err[:] = torch.rand((F, K), generator=g, dtype=torch.float64)
    a.corr[:] = torch.randint(-1, 9, (F, K, C), generator=g, dtype=torch.int16)
    a.err[3, 0] = float("nan")
    a.n_out[3] = max(int(a.n_out[3]), 1)
    for t in ("n_out", "xyz", "err", "corr", "status"):
        getattr(b, t)[:] = getattr(a, t)
    valid = torch.arange(K)[None, :] < a.n_out[:, None]
    b.xyz[~valid] = 123.0                                     # garbage beyond n_out: not compared
    b.corr[~valid] = 77
    assert devcheck.compare_bitwise(a, b, chunk=16)["frames_differing"] == 0
    f = int(torch.nonzero(a.n_out >= 2)[0])
    b.xyz.view(torch.int64)[f, 1, 2] ^= 1                     # one bit of one coordinate of a valid slot
    b.err.view(torch.int64)[3, 0] ^= 1                        # a NaN with another payload
    g2 = int(torch.nonzero(a.n_out >= 1)[5])
    b.corr[g2, 0, 1] += 1
    b.status[40] = 2
    b.n_out[41] += 1
    r = devcheck.compare_bitwise(a, b, chunk=16)
    assert r["frames_differing"] == len({f, 3, g2, 40, 41}) and r["fields"]["xyz"] >= 1 and r["fields"]["err"] >= 1
    assert r["fields"]["status"] == 1 and r["fields"]["n_out"] == 1 and r["fields"]["corr"] >= 1
    assert sorted(r["first_differing_frames"]) == sorted({f, 3, g2, 40, 41})
