"""Calibrates the CPU baselines bench.py reports ("kind": "port") against the REFERENCE ITSELF, once, on the build
container (the only machine that has /root/reference; the GPU box does not, so bench.py can only time ports there).

Same frames for all three (the first N frames of bench.py's 8 cameras x 16 markers stream, seed 1 = rank 0's batch):
  * reference: /root/reference/computer_code/api/helpers.py find_point_correspondance_and_object_points, unmodified,
    imported through oracle/ref_harness.py (cv2 restated by oracle/cv_restate.py: NumPy, so the three OpenCV calls
    inside are slower than OpenCV's C++ would be -- which flatters the ports' ratio; stated in the output);
  * Python port: oracle/mocap_oracle.py (bit-exact against the reference's outputs);
  * C port: oracle/c (the figure bench.py's `cpu_baseline.value` quotes).
One thread each.  Writes profiles/r05_cpu_baseline_calibration.json.  usage: calibrate_cpu_baseline.py [frames=24]"""
import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "low-cost-mocap_amd"))
from mocap_core import synth                      # noqa: E402
from oracle import c_oracle, mocap_oracle as mo   # noqa: E402
from oracle import ref_harness                    # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
C, M = 8, 16
rig = synth.ring_rig(C)
blobs, counts, _ = synth.make_blob_stream(rig, max(N, 2000), M, seed=1)

# ---- the reference itself
H = ref_harness.load_reference(C)
poses = synth.rig_to_pose_dicts(rig)
ref_pts, ref_n = [], []
t0 = time.perf_counter()
for f in range(N):
    ip = synth.frame_to_reference_lists(blobs[f], counts[f], as_int=True)
    err, pts, _ = H.find_point_correspondance_and_object_points(ip, poses, [None] * C)
    ref_n.append(len(err))
    ref_pts.append(np.asarray(pts, dtype=np.float64).reshape(-1, 3))
t_ref = time.perf_counter() - t0
n_ref = int(sum(ref_n))

# ---- the Python port on the same frames
Ks = [k for k in rig["K"]]
Ftab = mo.fundamental_table(Ks, rig["R"], rig["t"])
t0 = time.perf_counter()
n_py, same_py = 0, True
for f in range(N):
    o = mo.find_point_correspondance_and_object_points(blobs[f], counts[f], Ks, rig["R"], rig["t"], Ftab=Ftab)
    n_py += len(o["errors"])
    same_py &= len(o["errors"]) == ref_n[f] and np.array_equal(np.asarray(o["object_points"], dtype=np.float64).reshape(-1, 3), ref_pts[f])
t_py = time.perf_counter() - t0

# ---- the C port: the same frames (for the result check), then enough frames for a stable rate
co = c_oracle.COracle(rig["K"], rig["R"], rig["t"])
r = co.match_triangulate(blobs[:N], counts[:N])
same_c = bool(np.array_equal(r["n_out"], np.array(ref_n)))
dev_c = 0.0
for f in range(N):
    if ref_n[f]:
        dev_c = max(dev_c, float(np.abs(r["xyz"][f, :ref_n[f]] - ref_pts[f]).max() / np.abs(ref_pts[f]).max()))
t0 = time.perf_counter()
r = co.match_triangulate(blobs[:N], counts[:N])
t_c_same = time.perf_counter() - t0
t0 = time.perf_counter()
r2 = co.match_triangulate(blobs[:2000], counts[:2000])
t_c = time.perf_counter() - t0
n_c = int(r2["n_out"].sum())

out = {
    "what": "one-off calibration of bench.py's CPU baselines (kind 'port') against the reference's own function, same frames, "
            "same machine, one thread each (scripts/calibrate_cpu_baseline.py; the GPU box has no /root/reference)",
    "machine": {"where": "build container (no GPU)", "cpu": platform.processor() or platform.machine(),
                "model": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"),
                "logical_cores": os.cpu_count(), "threads_used": 1},
    "workload": f"first {N} frames of bench.py's 8 cams x 16 markers stream (synth.make_blob_stream(ring_rig(8), ., 16, seed=1))",
    "reference": {"function": "computer_code/api/helpers.py:339 find_point_correspondance_and_object_points (unmodified, via "
                              "oracle/ref_harness.py; cv2 calls restated in NumPy by oracle/cv_restate.py)",
                  "frames": N, "markers": n_ref, "seconds": t_ref, "markers_per_s": n_ref / t_ref, "frames_per_s": N / t_ref},
    "python_port": {"function": "oracle/mocap_oracle.py", "frames": N, "markers": n_py, "seconds": t_py,
                    "markers_per_s": n_py / t_py, "points_bitwise_equal_to_reference": bool(same_py),
                    "speed_vs_reference": (n_py / t_py) / (n_ref / t_ref)},
    "c_port": {"function": "oracle/c/mocap_oracle.c (what bench.py's cpu_baseline.value times)", "frames": 2000, "markers": n_c,
               "seconds": t_c, "markers_per_s": n_c / t_c, "same_frames_seconds": t_c_same,
               "n_out_equal_to_reference": same_c, "xyz_max_rel_vs_reference": dev_c,
               "speed_vs_reference": (n_c / t_c) / (n_ref / t_ref)},
    "reading": "bench.py's cpu_baseline.value (C port) divided by c_port.speed_vs_reference is the reference's own rate on the "
               "same core; the GPU/CPU ratio against the REFERENCE is the bench line's ratio against the C port times that factor",
}
path = os.path.join(ROOT, "profiles", "r05_cpu_baseline_calibration.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
