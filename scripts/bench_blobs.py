#!/usr/bin/env python3
"""Throughput of the blob-extraction stage (images -> image points) on one GPU: synthetic 8-camera
PS3-Eye frame sets resident in HBM, HIP-event timing of mocap_find_blobs_dev.
    python scripts/bench_blobs.py [--frames 256] [--steps 5] [--processed]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "low-cost-mocap_amd"))
import torch  # noqa: E402
from mocap_core import capi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--distinct", type=int, default=16)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--cams", type=int, default=8)
    ap.add_argument("--markers", type=int, default=16)
    ap.add_argument("--processed", action="store_true")
    ap.add_argument("--no-skip", action="store_true", help="filter every tile (no dark-tile early-out)")
    ap.add_argument("--fold", action="store_true", help="mocap_set_blob_options(2): the early-out decided inside the mask pass (no activity pass)")
    ap.add_argument("--noise", type=int, default=3, help="background noise levels: uniform in [0, noise)")
    args = ap.parse_args()
    C, M_max = args.cams, 32
    rig = synth.ring_rig(C)
    images, _ = synth.render_camera_frames(rig, args.distinct, args.markers, seed=1, noise_levels=args.noise)
    dev = torch.device("cuda", 0)
    core = capi.MocapCore(0)
    core.set_image_params(240, 320, rig["K"], [synth.REFERENCE_DISTORTION] * C)
    core.set_blob_options(skip_dark_tiles=2 if args.fold else (not args.no_skip))
    stream = torch.cuda.current_stream(dev)
    core.set_stream(stream.cuda_stream)
    F = args.frames
    d_img = torch.from_numpy(images).to(dev).repeat((F + args.distinct - 1) // args.distinct, 1, 1, 1, 1)[:F].contiguous()
    d_blobs = torch.zeros((F, C, M_max, 2), dtype=torch.float32, device=dev)
    d_counts = torch.zeros((F, C), dtype=torch.int32, device=dev)
    d_st = torch.zeros((F, C), dtype=torch.int32, device=dev)
    d_proc = torch.zeros((F, C, 320, 320, 3), dtype=torch.uint8, device=dev) if args.processed else None

    def run():
        core.find_blobs_dev(F, d_img.data_ptr(), M_max, d_blobs.data_ptr(), d_counts.data_ptr(), d_st.data_ptr(),
                            d_proc.data_ptr() if d_proc is not None else 0)
    run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        run()
        b.record(stream)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms = float(np.median(ts))
    n_img = F * C
    in_bytes = n_img * 240 * 320 * 3
    out_bytes = n_img * (M_max * 8 + 8) + (n_img * 320 * 320 * 3 if args.processed else 0)
    print(json.dumps({"images": n_img, "ms": ms, "runs_ms": ts, "images_per_s": n_img / ms * 1e3,
                      "frame_sets_per_s": F / ms * 1e3, "GBps_algorithmic": (in_bytes + out_bytes) / ms / 1e6,
                      "frac_of_8TBps": (in_bytes + out_bytes) / ms / 1e6 / 8000, "points": int(d_counts.sum().item()),
                      "status_nonzero": int((d_st != 0).sum().item()), "processed": args.processed, "skip_dark_tiles": not args.no_skip, "fold": args.fold,
                      "centroid_checksum": int(d_blobs.nan_to_num().double().sum().item() * 8) ^ int(d_counts.sum().item()), "noise_levels": args.noise}))


if __name__ == "__main__":
    main()
