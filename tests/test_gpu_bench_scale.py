"""GPU: parity AT THE SCALE THE METRIC IS QUOTED ON (round-5 verdict, item 1).

Two data races of `frame_bb_kernel` gave about one wrong frame in 1e5 through rounds 3-4 under a green suite: the gates
sampled a few hundred frames.  Here the whole bench stream -- the 100 000 frames of BASELINE.json configs[2] that bench.py
times (same generator, same seed, same K_max / G_cap / entry point: mocap_match_triangulate_dev_auto) -- goes through the
SHIPPED configuration 20 times, and every output bit of every frame (n_out, status, correspondence indices, points, errors)
must equal the exhaustive walk's (MOCAP_OPT_EXHAUSTIVE_WALK: every candidate group of the Cartesian product triangulated
and reprojected, helpers.py:408-421 as written; no bound drops or cuts anything).  2 M frame evaluations per run of this test:
a race at the 1e-5 rate shows ~20 times.  Plus run-to-run bitwise equality where no second algorithm exists to compare with:
2 048 stress frames (64 x 256: wide first pass, re-submit, heavy-root search) and 10^6 frames of 4 x 4, each also pinned on
the C oracle for a prefix.  Comparisons run on the device (mocap_core/devcheck.py).
"""
import numpy as np
import pytest

from mocap_core import capi, devcheck, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev)
    cores = []

    def make(**opts):
        c = capi.MocapCore(0)
        c.set_stream(stream.cuda_stream)
        if opts:
            c.set_options(**opts)
        cores.append(c)
        return c
    yield dev, make
    torch.cuda.synchronize(dev)
    for c in cores:
        c.close()


def _oracle_prefix(rig, blobs, counts, out, n, gate, K_max):
    from oracle import c_oracle
    ref = c_oracle.COracle(rig["K"], rig["R"], rig["t"]).match_triangulate(blobs[:n], counts[:n], gate_px=gate, K_max=K_max)
    n_out = out.n_out[:n].cpu().numpy()
    assert np.array_equal(ref["n_out"], n_out)
    vv = np.arange(K_max)[None, :] < n_out[:, None]
    assert np.array_equal(ref["corr"][vv], out.corr[:n].cpu().numpy()[vv])
    xyz = out.xyz[:n].cpu().numpy()
    assert np.abs(xyz[vv] - ref["xyz"][vv]).max() <= 1e-9 * np.abs(ref["xyz"][vv]).max()      # contract: 1e-5 relative


def test_bench_stream_100k_frames_20_repetitions_equal_the_exhaustive_walk_bit_for_bit(gpu):
    import torch
    dev, make = gpu
    C, M, F, K_MAX, G_CAP, GATE = 8, 16, 100_000, 48, 1 << 20, 5.0             # bench.py's default workload, rank 0
    rig = synth.ring_rig(C)
    blobs, counts, _ = synth.make_blob_stream(rig, F, M, seed=1)
    d_blobs, d_counts = torch.from_numpy(blobs).to(dev), torch.from_numpy(counts).to(dev)
    shipped, walk = make(), make(exhaustive_walk=True)
    for c in (shipped, walk):
        c.set_cameras(rig["K"], rig["R"], rig["t"])
    ref = devcheck.FrameOutputs(F, K_MAX, C, dev)
    ref.run(walk, M, d_blobs, d_counts, GATE, G_CAP)
    torch.cuda.synchronize(dev)
    assert walk.last_frame_kernel() == "frame_kernel<256>"
    assert int((ref.status != 0).sum().item()) == 0 and int(ref.n_out.sum().item()) > 2_000_000
    _oracle_prefix(rig, blobs, counts, ref, 200, GATE, K_MAX)                   # the walk itself is pinned on the oracle
    out = devcheck.FrameOutputs(F, K_MAX, C, dev)
    for rep in range(20):
        out.zero_()
        out.run(shipped, M, d_blobs, d_counts, GATE, G_CAP)
        cmp = devcheck.compare_bitwise(out, ref)
        assert cmp["frames_differing"] == 0, (rep, cmp)
        assert torch.equal(out.n_cand, ref.n_cand)
    assert shipped.last_frame_kernel() == "frame_bb_kernel<CW=1>"


def test_other_seeds_and_the_runtime_layout_kernel_equal_the_walk(gpu):
    """20 000 frames each of three other seeds (the multi-GPU ranks' streams: seed 1 + rank) and of K_max = 96 (the
    runtime-layout instantiation the reference seam's re-submit capacity lands in), 3 repetitions."""
    import torch
    dev, make = gpu
    C, M, F = 8, 16, 20_000
    rig = synth.ring_rig(C)
    shipped, walk = make(), make(exhaustive_walk=True)
    for c in (shipped, walk):
        c.set_cameras(rig["K"], rig["R"], rig["t"])
    for seed, K_max in ((2, 48), (3, 48), (8, 48), (2, 96)):
        blobs, counts, _ = synth.make_blob_stream(rig, F, M, seed=seed)
        d_blobs, d_counts = torch.from_numpy(blobs).to(dev), torch.from_numpy(counts).to(dev)
        ref, out = devcheck.FrameOutputs(F, K_max, C, dev), devcheck.FrameOutputs(F, K_max, C, dev)
        ref.run(walk, M, d_blobs, d_counts, 5.0, 1 << 20)
        for rep in range(3):
            out.zero_()
            out.run(shipped, M, d_blobs, d_counts, 5.0, 1 << 20)
            cmp = devcheck.compare_bitwise(out, ref)
            assert cmp["frames_differing"] == 0, (seed, K_max, rep, cmp)
        torch.cuda.synchronize(dev)


def test_stress_frames_run_to_run_bit_identical_through_wide_and_heavy_kernels(gpu):
    """2 048 frames of 64 cameras x 256 markers: wide first pass, device-side re-submit, heavy-root search.  Four runs, every
    bit equal (status and the flagged frames' emptied slots included); 16 frames pinned on the C oracle."""
    import torch
    dev, make = gpu
    C, M, F, K_MAX = 64, 256, 2048, 384
    rig = synth.stress_rig(C)
    blobs, counts, _ = synth.make_stress_stream(rig, F, M, seed=1)
    d_blobs, d_counts = torch.from_numpy(blobs).to(dev), torch.from_numpy(counts).to(dev)
    core = make()
    core.set_cameras(rig["K"], rig["R"], rig["t"])
    first = devcheck.FrameOutputs(F, K_MAX, C, dev)
    first.run(core, M, d_blobs, d_counts, synth.STRESS_GATE_PX, 1 << 20)
    torch.cuda.synchronize(dev)
    assert core.last_frame_kernel() in ("frame_kernel<512, wide>", "frame_kernel<1024, wide>")
    flagged, rerun = first.info.cpu().numpy()
    assert flagged >= 1 and rerun == flagged           # the stream holds frames over the first pass's cap: the repair path runs
    ok = (first.status == 0)
    assert int(ok.sum().item()) >= F - 16
    _oracle_prefix(rig, blobs, counts, first, 16, synth.STRESS_GATE_PX, K_MAX)
    out = devcheck.FrameOutputs(F, K_MAX, C, dev)
    for rep in range(3):
        out.zero_()
        out.run(core, M, d_blobs, d_counts, synth.STRESS_GATE_PX, 1 << 20)
        cmp = devcheck.compare_bitwise(out, first)
        assert cmp["frames_differing"] == 0, (rep, cmp)


def test_million_frames_of_4x4_run_to_run_bit_identical(gpu):
    """BASELINE.json configs[1] at bench.py's size (10^6 frames, one-wave workgroups, three-launch schedule): three runs,
    every bit equal; 300 frames pinned on the C oracle; the walk without cut-offs agrees too."""
    import torch
    dev, make = gpu
    C, M, F, K_MAX = 4, 4, 1_000_000, 16
    rig = synth.ring_rig(C)
    blobs, counts, _ = synth.make_blob_stream(rig, F, M, seed=1)
    d_blobs, d_counts = torch.from_numpy(blobs).to(dev), torch.from_numpy(counts).to(dev)
    core, walk = make(), make(exhaustive_walk=True)
    for c in (core, walk):
        c.set_cameras(rig["K"], rig["R"], rig["t"])
    first = devcheck.FrameOutputs(F, K_MAX, C, dev)
    first.run(core, M, d_blobs, d_counts, 5.0, 1 << 20)
    torch.cuda.synchronize(dev)
    assert core.last_frame_kernel() == "frame_kernel<64>"
    _oracle_prefix(rig, blobs, counts, first, 300, 5.0, K_MAX)
    out = devcheck.FrameOutputs(F, K_MAX, C, dev)
    for rep, c in enumerate((core, core, walk)):
        out.zero_()
        out.run(c, M, d_blobs, d_counts, 5.0, 1 << 20)
        cmp = devcheck.compare_bitwise(out, first)
        assert cmp["frames_differing"] == 0, (rep, cmp)
