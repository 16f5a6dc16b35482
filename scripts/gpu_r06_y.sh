#!/bin/bash
# round 6, call Y: the bench lines alone, AFTER refresh_profiles.sh has put the counter summaries of the same sources into profiles/
# (bench.py reads them: a line taken inside the refresh call itself still carries the previous summaries and marks them stale)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06/final; mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench_final.log 2>&1; echo "bench rc=$?"
grep '^{"metric"' $O/bench_final.log > $O/bench_line_final.json
timeout 400 python bench.py --workload 64x256 --frames 12500 --steps 3 --warmup 1 > $O/bench_64x256.log 2>&1; grep '^{"metric"' $O/bench_64x256.log > $O/bench_line_64x256.json
timeout 400 python bench.py --workload 4x4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_4x4.log 2>&1; grep '^{"metric"' $O/bench_4x4.log > $O/bench_line_4x4.json
python - <<'PY'
import json
O="gpurun_out/r06/final/"
l=json.load(open(O+"bench_line_final.json"))
print("8x16", l["value"], l["ms_per_step"], l["roofline"]["kernel_ms"], l["parity"]["full_batch_vs_exhaustive_bit_exact"], l["roofline"]["frac"], l["roofline"].get("traffic_stale"), l["roofline_fp64"].get("frac"), l["roofline_fp64"].get("frac_active_lanes"), l["roofline_fp64"].get("stale"))
PY
