"""GPU parity of the rows right after the hot path: the world-coordinate
epilogue fused into the frame kernel (reference helpers.py:96-103) and the object locator
(helpers.py:424-480), against the reference's own outputs (tests/golden/post_world_locate.npz)
and the oracle restatement on larger seeded sets."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def test_locate_objects_vs_reference_golden(core):
    g = load_golden("post_world_locate")
    res = core.locate_objects(g["ref_world"], g["err"], g["n_pts"], O_max=8)
    assert np.array_equal(res["n_obj"], g["ref_nobj"])
    have = np.arange(8)[None, :] < g["ref_nobj"][:, None]
    assert np.array_equal(res["droneIndex"][have], g["ref_drone"][have])
    np.testing.assert_allclose(res["pos"][have], g["ref_pos"][have], rtol=1e-14, atol=0)
    np.testing.assert_allclose(res["error"][have], g["ref_error"][have], rtol=1e-14, atol=0)
    np.testing.assert_allclose(res["heading"][have], g["ref_heading"][have], rtol=0, atol=1e-12)


def test_locate_objects_vs_oracle_large(core):
    from mocap_core import synth
    from oracle import mocap_oracle as mo
    xyz, err, n_pts = synth.make_object_frames(1500, 32, seed=77)
    res = core.locate_objects(xyz, err, n_pts, O_max=10)
    for f in range(xyz.shape[0]):
        n = int(n_pts[f])
        objs = mo.locate_objects(xyz[f, :n], err[f, :n])
        assert res["n_obj"][f] == len(objs), f
        for j, o in enumerate(objs):
            assert res["lead"][f, j] == o["lead"] and res["droneIndex"][f, j] == o["droneIndex"]
            np.testing.assert_allclose(res["pos"][f, j], o["pos"], rtol=1e-14, atol=0)
            assert abs(res["heading"][f, j] - o["heading"]) < 1e-12
    # the mirror of the reference function (one frame, list of dicts)
    from mocap_core import helpers
    helpers.set_core(core)
    n = int(n_pts[3])
    got = helpers.locate_objects(xyz[3, :n], err[3, :n])
    want = mo.locate_objects(xyz[3, :n], err[3, :n])
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert a["droneIndex"] == b["droneIndex"] and abs(a["heading"] - b["heading"]) < 1e-12


def test_world_epilogue_fused_in_frame_kernel(core):
    """With a to-world matrix set, the frame path's points are the reference loop helpers.py:96-103
    applied to the points it returns without one (1e-12: the 4x4 product is a BLAS call upstream)."""
    from mocap_core import synth
    from oracle import mocap_oracle as mo
    g = load_golden("post_world_locate")
    rig = synth.ring_rig(8)
    blobs, counts, _ = synth.make_blob_stream(rig, 400, 16, seed=71)
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    base = core.match_triangulate(blobs, counts, K_max=48)
    try:
        core.set_world_transform(g["to_world"])
        res = core.match_triangulate(blobs, counts, K_max=48)
    finally:
        core.set_world_transform(None)
    assert np.array_equal(res["n_out"], base["n_out"]) and np.array_equal(res["err"], base["err"], equal_nan=True)
    valid = np.arange(48)[None, :] < base["n_out"][:, None]
    assert np.array_equal(res["corr"][valid], base["corr"][valid])
    want = mo.world_epilogue(base["xyz"][valid], g["to_world"])
    np.testing.assert_allclose(res["xyz"][valid], want, rtol=1e-12, atol=1e-12)
    # and against the reference's own epilogue outputs
    F = g["cam_xyz"].shape[0]
    for f in range(0, F, 17):
        n = int(g["n_pts"][f])
        if n:
            np.testing.assert_allclose(mo.world_epilogue(g["cam_xyz"][f, :n], g["to_world"]), g["ref_world"][f, :n],
                                       rtol=0, atol=0)
    again = core.match_triangulate(blobs, counts, K_max=48)      # switched off again: camera-0 coordinates
    assert np.array_equal(again["xyz"][valid], base["xyz"][valid])


def test_rccl_gather_api_world1():
    """The exchange step through the real RCCL backend (world size 1 is all a 1-GPU box offers):
    init, async gather of packed device records, completion, unpack.  The N > 1 logic is covered by
    the gloo world-2 test in test_host_cpu.py; this pins the calls RCCL itself accepts."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from mocap_core import dist as mdist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    try:
        F, K, C = 64, 16, 4
        rng = np.random.default_rng(5)
        n_out = torch.from_numpy(rng.integers(0, K, F).astype(np.int32)).to(dev)
        xyz = torch.from_numpy(rng.normal(size=(F, K, 3))).to(dev)
        err = torch.from_numpy(rng.random((F, K))).to(dev)
        corr = torch.from_numpy(rng.integers(-1, 4, (F, K, C)).astype(np.int16)).to(dev)
        rec = mdist.pack_records(n_out, xyz, err, corr)
        # world == 1 short-circuits inside gather_records*, so call the collective itself too
        out = torch.empty((1,) + tuple(rec.shape), dtype=rec.dtype, device=dev)
        work = dist.gather(rec, gather_list=list(out.unbind(0)), dst=0, async_op=True)
        work.wait()
        torch.cuda.synchronize()
        assert torch.equal(out[0], rec)
        got = mdist.unpack_records(mdist.gather_records_async(rec).result(), C, K)
        assert np.array_equal(got["n_out"], n_out.cpu().numpy()) and np.array_equal(got["xyz"], xyz.cpu().numpy())
        assert np.array_equal(got["corr"], corr.cpu().numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("F,K,C", [(1, 4, 2), (700, 16, 4), (1024, 8, 3), (5000, 48, 8), (300, 90, 64), (37, 70, 5)])
def test_compaction_kernel_matches_reference(core, F, K, C):
    """mocap_compact_tracks_dev (exclusive prefix sum over n_out + scatter into 32 + 2C-byte records) against its
    host restatement mocap_core.dist.compact_tracks_reference: offsets, total, every record byte.  Sizes cross the
    1 024-frame scan block, include empty frames, frames at full capacity and out-of-range counts (negative, or > K: a
    re-submitted frame that needs more slots than it was given, nothing of it was written -- both count as empty)."""
    import torch
    from mocap_core import dist as mdist, synth
    rig = synth.ring_rig(C)
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    rng = np.random.default_rng(F)
    n_out = rng.integers(0, K + 1, F).astype(np.int32)
    n_out[rng.random(F) < 0.2] = 0
    if F > 10:
        n_out[3], n_out[7] = K + 5, -2          # out of range: no valid slot
    xyz, err = rng.normal(size=(F, K, 3)), rng.random((F, K))
    corr = rng.integers(-1, 16, (F, K, C)).astype(np.int16)
    dev = torch.device("cuda", 0)
    d = [torch.from_numpy(a).to(dev) for a in (n_out, xyz, err, corr)]
    stride = mdist.track_record_bytes(C)
    d_off = torch.full((F + 1,), -1, dtype=torch.int64, device=dev)
    d_rec = torch.full((F * K, stride), 0xAB, dtype=torch.uint8, device=dev)
    total = torch.zeros(1, dtype=torch.int64).pin_memory()
    torch.cuda.synchronize()
    core.compact_tracks_dev(F, K, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d_off.data_ptr(),
                            d_rec.data_ptr(), F * K, total.data_ptr())
    core.synchronize()
    rec_ref, off_ref = mdist.compact_tracks_reference(n_out, xyz, err, corr)
    assert int(total[0]) == int(off_ref[-1])
    assert np.array_equal(d_off.cpu().numpy(), off_ref)
    assert np.array_equal(d_rec.cpu().numpy()[:rec_ref.shape[0]], rec_ref)
    assert (d_rec.cpu().numpy()[rec_ref.shape[0]:] == 0xAB).all()         # nothing written past the last record
    back = mdist.unpack_compact(n_out, d_rec.cpu().numpy()[:rec_ref.shape[0]], C, K)
    valid = np.arange(K)[None, :] < np.where((n_out < 0) | (n_out > K), 0, n_out)[:, None]
    assert np.array_equal(back["xyz"][valid], xyz[valid]) and np.array_equal(back["corr"][valid], corr[valid])
    # a record buffer SHORTER than the total: the records that fit are written (a frame's run may be cut in the middle), nothing
    # lands past the capacity, offsets and the total still describe the whole batch
    cap = int(off_ref[-1]) * 2 // 3
    if cap > 0:
        d_rec.fill_(0xAB)
        d_off.fill_(-1)
        core.compact_tracks_dev(F, K, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d_off.data_ptr(),
                                d_rec.data_ptr(), cap, total.data_ptr())
        core.synchronize()
        assert int(total[0]) == int(off_ref[-1]) and np.array_equal(d_off.cpu().numpy(), off_ref)
        assert np.array_equal(d_rec.cpu().numpy()[:cap], rec_ref[:cap])
        assert (d_rec.cpu().numpy()[cap:] == 0xAB).all()


def test_compact_exchange_through_rccl_world1(core):
    """TrackCompactor + gather_compact_async through the real RCCL backend on the frame kernel's own outputs (world
    size 1 is all a 1-GPU box offers; the N > 1 protocol is the gloo world-2 test in test_host_cpu.py)."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from mocap_core import dist as mdist, synth
    C, M, F, K = 4, 4, 300, 16
    rig = synth.ring_rig(C)
    blobs, counts, _ = synth.make_blob_stream(rig, F, M, seed=12)
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    ref = core.match_triangulate(blobs, counts, K_max=K)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    try:
        stream = torch.cuda.current_stream(dev)
        core.set_stream(stream.cuda_stream)
        d_blobs, d_counts = torch.from_numpy(blobs).to(dev), torch.from_numpy(counts).to(dev)
        d_xyz = torch.empty((F, K, 3), dtype=torch.float64, device=dev)
        d_err = torch.empty((F, K), dtype=torch.float64, device=dev)
        d_corr = torch.empty((F, K, C), dtype=torch.int16, device=dev)
        d_n = torch.zeros(F, dtype=torch.int32, device=dev)
        d_st = torch.zeros(F, dtype=torch.int32, device=dev)
        comp = mdist.TrackCompactor(core, F, K, C, dev)
        for _ in range(3):                       # buffers rotate
            core.match_triangulate_dev(F, M, d_blobs.data_ptr(), d_counts.data_ptr(), 5.0, K, 1 << 20, d_xyz.data_ptr(),
                                       d_err.data_ptr(), d_corr.data_ptr(), d_n.data_ptr(), d_st.data_ptr())
            i = comp.compact(d_n, d_xyz, d_err, d_corr, stream)
            n = comp.count(i)
            assert n == int(ref["n_out"].sum())
            n_all, r_all = mdist.gather_compact_async(comp.n_out[i], comp.records[i], n, [F], dst=0).result()
            got = mdist.unpack_compact(n_all.cpu().numpy(), r_all.cpu().numpy(), C, K)
            valid = np.arange(K)[None, :] < ref["n_out"][:, None]
            assert np.array_equal(got["n_out"], ref["n_out"])
            for key in ("xyz", "err", "corr"):
                assert np.array_equal(got[key][valid], ref[key][valid])
        # the collective calls RCCL itself must accept at N > 1: one int64 all-gather + batched isend / irecv
        mine = torch.tensor([n], dtype=torch.int64, device=dev)
        outl = [torch.zeros_like(mine)]
        dist.all_gather(outl, mine)
        assert int(outl[0].item()) == n
        dist.barrier()
    finally:
        core.set_stream(0)
        dist.destroy_process_group()
