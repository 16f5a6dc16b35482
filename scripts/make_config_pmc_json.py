#!/usr/bin/env python3
"""Counter summaries of scripts/profile_configs_pmc.sh -> profiles/<tag>_fp64_mix_<config>.json, the file bench.py's sub-records
(configs["4x4"], configs["64x256"]) read: executed FP64 flop and HBM bytes PER FRAME over ALL kernels of one hot-path call (first
pass + device-side re-submit), lane utilisation of the dominant kernel, stamped with the source hash.
usage: make_config_pmc_json.py <dir> <config name> <calls per run> <out prefix> [git head]"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

d, name, calls, prefix = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
head = sys.argv[5] if len(sys.argv) > 5 else None


def table(path):
    """{kernel: {counter: (dispatches, sum)}} of a rocpd_summary pmc csv; only the path's own kernels."""
    out = {}
    with open(path) as f:
        for row in csv.reader(f):
            if len(row) >= 5 and "mocap::" in row[0]:
                out.setdefault(row[0], {})[row[1]] = (int(row[2]), float(row[3]))
    return out


def stats(path):
    out = {}
    with open(path) as f:
        for row in csv.reader(f):
            if len(row) >= 4 and "mocap::" in row[0]:
                out[row[0]] = (int(row[1]), float(row[2]))
    return out


line = json.loads(open(os.path.join(d, f"{name}_bench_line.json")).read())
frames = line["config"]["frames_per_gpu"]
mix, issue = table(os.path.join(d, f"{name}_mix_pmc.csv")), table(os.path.join(d, f"{name}_issue_pmc.csv"))
fetch, write = table(os.path.join(d, f"{name}_fetch_pmc.csv")), table(os.path.join(d, f"{name}_write_pmc.csv"))
st = stats(os.path.join(d, f"{name}_kernel_stats.csv"))


def total(tab, counter):
    return sum(v[counter][1] for v in tab.values() if counter in v) / calls


fma, mul, add, tr = (total(mix, "SQ_INSTS_VALU_" + k + "_F64") for k in ("FMA", "MUL", "ADD", "TRANS"))
flop = (2 * fma + mul + add + tr) * 64
dom = max(st, key=lambda k: st[k][1])
lane_util = None
if dom in issue and dom in mix and "SQ_THREAD_CYCLES_VALU" in issue[dom] and "SQ_ACTIVE_INST_VALU" in mix[dom]:
    a = issue[dom]["SQ_THREAD_CYCLES_VALU"][1] / issue[dom]["SQ_THREAD_CYCLES_VALU"][0]
    b = mix[dom]["SQ_ACTIVE_INST_VALU"][1] / mix[dom]["SQ_ACTIVE_INST_VALU"][0]
    lane_util = a / (64 * b)
fk, wk = total(fetch, "FETCH_SIZE"), total(write, "WRITE_SIZE")
abytes = line["roofline"]["algorithmic_bytes_per_launch"]
out = {"config": name, "git_head": head, "kernel_source_sha16": bench.kernel_source_hash(), "frames_per_launch": frames,
       "command": f"rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --workload {name} --steps 2 --warmup 1 --no-cpu-baseline "
                  "(scripts/profile_configs_pmc.sh; one counter set per run; sums over every mocap:: kernel of a hot-path call / calls)",
       "hot_path_calls_per_run": calls, "dominant_kernel": dom,
       "kernels_ns_per_call": {k: v[1] / calls for k, v in sorted(st.items(), key=lambda kv: -kv[1][1])},
       "fp64_wave_instructions_per_call": {"FMA": fma, "MUL": mul, "ADD": add, "TRANS": tr},
       "fp64_flop_per_frame": flop / frames,
       "flop_convention": "FMA = 2, MUL / ADD / TRANS = 1 per lane; every lane of an issued wave instruction counted",
       "vector_lane_utilisation": lane_util,
       "FETCH_SIZE_KB_per_call_raw": fk, "WRITE_SIZE_KB_per_call_raw": wk,
       "correction": "MI355X_MICROARCH.md (HBM): gfx950 rocprofv3 FETCH_SIZE tallies 128-B requests at 64 B -> x2; WRITE_SIZE as reported",
       "hbm_bytes_per_frame": (2 * fk + wk) * 1024 / frames, "algorithmic_bytes_per_frame": abytes / frames,
       "traffic_over_algorithmic": (2 * fk + wk) * 1024 / abytes}
json.dump(out, open(f"{prefix}_fp64_mix_{name}.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("config", "fp64_flop_per_frame", "vector_lane_utilisation", "traffic_over_algorithmic", "dominant_kernel")}))
