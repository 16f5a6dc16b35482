#!/usr/bin/env python3
"""Sweep the frame kernel's scheduling knobs on the bench workload (GPU box): one data set, many
(frame_threads, heavy_threshold, slice_size) settings, kernel time by HIP events.  Results are
bit-identical across settings by construction (tests/test_gpu_parity.py); this only times them."""
import os, sys, itertools
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "low-cost-mocap_amd"))
import torch
from mocap_core import capi, synth

C, M, K_MAX = (int(v) for v in os.environ.get("SHAPE", "8,16,48").split(","))
F = int(os.environ.get("FRAMES", 100000))
rig = synth.ring_rig(C)
blobs, counts, _ = synth.make_blob_stream(rig, F, M, seed=1)
dev = torch.device("cuda", 0)
core = capi.MocapCore(0)
core.set_cameras(rig["K"], rig["R"], rig["t"])
stream = torch.cuda.current_stream(dev)
core.set_stream(stream.cuda_stream)
d_blobs = torch.from_numpy(blobs).to(dev); d_counts = torch.from_numpy(counts).to(dev)
d_xyz = torch.empty((F, K_MAX, 3), dtype=torch.float64, device=dev)
d_err = torch.empty((F, K_MAX), dtype=torch.float64, device=dev)
d_corr = torch.empty((F, K_MAX, C), dtype=torch.int16, device=dev)
d_i = [torch.zeros(F, dtype=torch.int32, device=dev) for _ in range(3)]

def run():
    core.match_triangulate_dev(F, M, d_blobs.data_ptr(), d_counts.data_ptr(), 5.0, K_MAX, 1 << 20,
                               d_xyz.data_ptr(), d_err.data_ptr(), d_corr.data_ptr(),
                               d_i[0].data_ptr(), d_i[1].data_ptr(), d_i[2].data_ptr())

def timeit(n=4):
    run(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(n):
        run()
    b.record(stream); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

settings = [tuple(int(x) for x in s.split(",")) for s in sys.argv[1:]] or \
    [(256, h, s) for h, s in itertools.product((4096, 8192, 16384, 32768), (2048, 4096, 8192))]
for T, h, s in settings:
    core.set_tuning(T, h, s)
    print(f"T={T} heavy_threshold={h} slice_size={s}: {timeit():.3f} ms", flush=True)
