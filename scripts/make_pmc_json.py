#!/usr/bin/env python3
"""Turn the counter summaries of scripts/profile_frame_pmc.sh into the two JSON files bench.py reads
(profiles/<tag>_fp64_mix.json, <tag>_hbm_traffic.json).  Each carries the kernel it was measured on, the git HEAD
passed in and the hash of the library sources (bench.kernel_source_hash): bench.py marks derived figures `stale` when
it runs on other sources.  usage: make_pmc_json.py <dir with *_pmc.csv / *_kernel_stats.csv / bench_line.json> <out prefix> [git head]"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def counters(path, kernel_substr):
    out = {}
    with open(path) as f:
        for row in csv.reader(f):
            if len(row) >= 5 and kernel_substr in row[0]:
                out[row[1]] = float(row[4])     # avg per dispatch
                out["_kernel"] = row[0]
    return out


def kernel_ns(path, kernel_substr):
    with open(path) as f:
        for row in csv.reader(f):
            if len(row) >= 4 and kernel_substr in row[0]:
                return float(row[3]), int(row[1])
    return None, 0


d, prefix = sys.argv[1], sys.argv[2]
head = sys.argv[3] if len(sys.argv) > 3 else None
line = json.loads(open(os.path.join(d, "bench_line.json")).read())
frames = line["config"]["frames_per_gpu"]
cands = line["roofline_fp64"]["candidates_per_launch"]
kern = "frame_bb_kernel"
mix = counters(os.path.join(d, "frame_mix_pmc.csv"), kern)
issue = counters(os.path.join(d, "frame_issue_pmc.csv"), kern)
fetch = counters(os.path.join(d, "frame_fetch_pmc.csv"), kern)
write = counters(os.path.join(d, "frame_write_pmc.csv"), kern)
ns, calls = kernel_ns(os.path.join(d, "frame_mix_kernel_stats.csv"), kern)
sha = bench.kernel_source_hash()
common = {"kernel": mix.get("_kernel"), "git_head": head, "kernel_source_sha16": sha, "frames_per_launch": frames,
          "command": "rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ba "
                     "--no-blobs --no-latency (scripts/profile_frame_pmc.sh; one counter set per run, average per dispatch)"}
fma, mul, add, tr = (mix.get("SQ_INSTS_VALU_" + k + "_F64", 0.0) for k in ("FMA", "MUL", "ADD", "TRANS"))
valu = mix.get("SQ_INSTS_VALU", 0.0)
flop = (2 * fma + mul + add + tr) * 64
cycles_q = 1024 * (ns * 1e-9) * 2.4e9 / 4 if ns else None
out_mix = dict(common, candidates_per_launch=cands, kernel_ns=ns, dispatches=calls,
               wave_instructions={k: v for k, v in mix.items() if not k.startswith("_")},
               issue_counters={k: v for k, v in issue.items() if not k.startswith("_")},
               valu_lane_instructions_per_candidate=valu * 64 / cands, fp64_share_of_valu_instructions=(fma + mul + add + tr) / valu,
               fp64_flop_per_candidate=flop / cands,
               flop_convention="FMA = 2, MUL / ADD / TRANS = 1 per lane; every lane of an issued wave instruction counted; per "
                               "candidate group of the Cartesian product, whether evaluated or dropped with its block",
               valu_issue_utilisation=mix.get("SQ_ACTIVE_INST_VALU", 0.0) / cycles_q if cycles_q else None,
               valu_issue_utilisation_formula="SQ_ACTIVE_INST_VALU / (1024 SIMDs x kernel cycles at 2.4 GHz / 4)",
               active_over_simd_time=(issue.get("SQ_ACTIVE_INST_ANY", 0.0) / cycles_q) if cycles_q and issue else None,
               scalar_share_of_issue=(issue.get("SQ_ACTIVE_INST_SCA", 0.0) / issue["SQ_ACTIVE_INST_ANY"]) if issue.get("SQ_ACTIVE_INST_ANY") else None,
               vector_lane_utilisation=(issue.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64 * mix["SQ_ACTIVE_INST_VALU"])) if issue.get("SQ_THREAD_CYCLES_VALU") and mix.get("SQ_ACTIVE_INST_VALU") else None)
json.dump(out_mix, open(prefix + "_fp64_mix.json", "w"), indent=1)
fk, wk = fetch.get("FETCH_SIZE", 0.0), write.get("WRITE_SIZE", 0.0)
abytes = line["roofline"]["algorithmic_bytes_per_launch"]
out_bytes = line["config"]["markers_per_frame"] * frames * (24 + 8 + 2 * line["config"]["cams"]) + 12 * frames
out_tr = dict(common, FETCH_SIZE_KB_raw=fk, WRITE_SIZE_KB_raw=wk,
              correction="MI355X_MICROARCH.md (HBM): gfx950 rocprofv3 FETCH_SIZE tallies 128-B requests at 64 B -> x2; WRITE_SIZE taken as reported",
              hbm_bytes_per_frame=(2 * fk + wk) * 1024 / frames, algorithmic_bytes_per_frame=abytes / frames,
              traffic_over_algorithmic=(2 * fk + wk) * 1024 / abytes,
              write_bytes_per_frame=wk * 1024 / frames, output_bytes_per_frame=out_bytes / frames,
              write_bytes_over_output_bytes=wk * 1024 / out_bytes)
json.dump(out_tr, open(prefix + "_hbm_traffic.json", "w"), indent=1)
print(json.dumps({"fp64_flop_per_candidate": out_mix["fp64_flop_per_candidate"], "valu_issue": out_mix["valu_issue_utilisation"],
                  "traffic_over_algorithmic": out_tr["traffic_over_algorithmic"], "write_over_output": out_tr["write_bytes_over_output_bytes"]}))
