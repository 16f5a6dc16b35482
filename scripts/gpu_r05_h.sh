#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r05h; mkdir -p $O
timeout 900 python bench.py --workload 64x256 --frames 12500 --steps 3 --warmup 1 > $O/bench_64x256.json 2> $O/bench_64x256.err; echo "bench64 rc $?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r05h/bench_64x256.json"))
c=d["config"]
print("ms_per_step", d["ms_per_step"], "frames/s", c["frames_per_s"], "overflow", c["overflow_frames"], "resub", c["resubmitted_frames"], c["overflow_by_cap"], d["parity"])
PY
tail -3 $O/bench_64x256.err
MOCAP_NO_HEAVY_BB=1 timeout 900 python bench.py --workload 64x256 --frames 12500 --steps 3 --warmup 1 > $O/bench_64x256_noheavy.json 2>/dev/null; python - <<'PY'
import json
d=json.load(open("gpurun_out/r05h/bench_64x256_noheavy.json"))
c=d["config"]
print("NO HEAVY: ms_per_step", d["ms_per_step"], "overflow", c["overflow_frames"])
PY
