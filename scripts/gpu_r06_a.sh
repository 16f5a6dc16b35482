#!/bin/bash
# round 6, call A: the GPU suite with the bench-scale parity tests, the default bench line (sub-records, full-batch parity),
# the heavy-root search's per-root log on 12 500 chunk-generated stress frames
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06a; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 900 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?"
grep '^{"metric"' $O/bench.log > $O/bench_line.json
python - <<'PY'
import json
l=json.load(open("gpurun_out/r06a/bench_line.json"))
print({k:l[k] for k in ("value","ms_per_step","world_size","devices")})
print("parity", {k:v for k,v in l["parity"].items() if k!="wide"})
for n,c in l.get("configs",{}).items(): print(n, {k:c.get(k) for k in ("ms_per_step","frames_per_s","value","overflow_frames","flagged_by_first_pass","host_generation_s","error")}, c.get("parity"))
print("ba", l["ba"]["value"], l["ba"]["default_mode_iterations_per_s"], "pyport", l["cpu_baseline"].get("python_port_sample"), l["cpu_baseline"].get("python_port_markers_per_s"), l["cpu_baseline"].get("python_port_error"))
PY
MOCAP_HEAVY_DEBUG=1 timeout 600 python bench.py --workload 64x256 --frames 12500 --steps 1 --warmup 0 --no-cpu-baseline > $O/heavy_debug.log 2>&1
grep -c "^HEAVY" $O/heavy_debug.log; grep "^HEAVY" $O/heavy_debug.log | awk '{print $9, $15, $17}' | sort | uniq -c | sort -k2n | head -80
grep '^{"metric"' $O/heavy_debug.log | python -c "import json,sys; l=json.loads(sys.stdin.read()); print(l['ms_per_step'], l['config']['overflow_frames'], l['config']['flagged_by_first_pass'])"
