"""Device-side verification helpers: a frame batch's outputs as torch tensors, and bit-for-bit comparison of two
result sets without leaving HBM.  Used by bench.py (`parity.full_batch_vs_exhaustive_bit_exact`: the whole timed batch
against the exhaustive walk, MOCAP_OPT_EXHAUSTIVE_WALK) and by tests/test_gpu_bench_scale.py (the bench stream x 20
repetitions; run-to-run equality at 64 x 256 and 4 x 4).  Both sides of a comparison are the core itself -- what is checked is
that the shipped selection (exact branch and bound, csrc/frame_bb.hip) and its synchronisation return, at the scale the
metric is quoted on, every bit the walk over all candidate groups returns (helpers.py:408-421)."""
import torch


class FrameOutputs:
    """xyz f64 [F][K][3], err f64 [F][K], corr i16 [F][K][C], n_out / status / n_cand i32 [F] on `dev`, zero-filled."""

    def __init__(self, F, K_max, C, dev):
        self.F, self.K, self.C = int(F), int(K_max), int(C)
        self.xyz = torch.zeros((F, K_max, 3), dtype=torch.float64, device=dev)
        self.err = torch.zeros((F, K_max), dtype=torch.float64, device=dev)
        self.corr = torch.zeros((F, K_max, C), dtype=torch.int16, device=dev)
        self.n_out = torch.zeros(F, dtype=torch.int32, device=dev)
        self.status = torch.zeros(F, dtype=torch.int32, device=dev)
        self.n_cand = torch.zeros(F, dtype=torch.int32, device=dev)
        self.info = torch.zeros(2, dtype=torch.int32, device=dev)

    def zero_(self):
        for t in (self.xyz, self.err, self.corr, self.n_out, self.status, self.n_cand, self.info):
            t.zero_()
        return self

    def run(self, core, M, d_blobs, d_counts, gate, g_cap, auto=True):
        """One pass of the hot path into these buffers (enqueue only)."""
        fn = core.match_triangulate_dev_auto if auto else core.match_triangulate_dev
        args = [self.F, M, d_blobs.data_ptr(), d_counts.data_ptr(), gate, self.K, g_cap, self.xyz.data_ptr(), self.err.data_ptr(),
                self.corr.data_ptr(), self.n_out.data_ptr(), self.status.data_ptr(), self.n_cand.data_ptr()]
        if auto:
            args.append(self.info.data_ptr())
        fn(*args)


def compare_bitwise(a, b, chunk=1 << 16):
    """Every output bit of the slots a frame reports (k < n_out; the kernels write nothing beyond) + n_out + status, `a` against
    `b`, on the device.  -> dict: frames, frames_differing (any of the fields), per-field counts, first differing frames."""
    assert (a.F, a.K, a.C) == (b.F, b.K, b.C)
    bad_any = torch.zeros(a.F, dtype=torch.bool, device=a.xyz.device)
    fields = {"n_out": 0, "status": 0, "corr": 0, "xyz": 0, "err": 0}
    k = torch.arange(a.K, device=a.xyz.device)[None, :]
    for lo in range(0, a.F, chunk):
        hi = min(a.F, lo + chunk)
        s = slice(lo, hi)
        dn = a.n_out[s] != b.n_out[s]
        ds = a.status[s] != b.status[s]
        n = torch.minimum(a.n_out[s], b.n_out[s]).clamp(0, a.K)
        valid = k < n[:, None]
        dc = ((a.corr[s] != b.corr[s]).any(dim=2) & valid).any(dim=1)
        dx = ((a.xyz[s].view(torch.int64) != b.xyz[s].view(torch.int64)).any(dim=2) & valid).any(dim=1)   # bit patterns: NaN-safe
        de = ((a.err[s].view(torch.int64) != b.err[s].view(torch.int64)) & valid).any(dim=1)
        for name, d in (("n_out", dn), ("status", ds), ("corr", dc), ("xyz", dx), ("err", de)):
            fields[name] += int(d.sum().item())
        bad_any[s] = dn | ds | dc | dx | de
    nbad = int(bad_any.sum().item())
    return {"frames": a.F, "frames_differing": nbad, "fields": fields,
            "first_differing_frames": torch.nonzero(bad_any)[:8, 0].tolist() if nbad else []}
