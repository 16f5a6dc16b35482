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
MOCAP_BENCH_EXCHANGE=1 timeout 400 python bench.py --workload 64x256 --frames 12500 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_64x256_exchange.log 2>&1; grep '^{"metric"' $O/bench_64x256_exchange.log > $O/bench_line_64x256_exchange.json
MOCAP_BENCH_EXCHANGE=1 timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ba --no-blobs --no-latency --no-configs > $O/bench_8x16_exchange.log 2>&1; grep '^{"metric"' $O/bench_8x16_exchange.log > $O/bench_line_8x16_exchange.json
MOCAP_BENCH_EXCHANGE=1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ba --no-blobs --no-latency --no-configs > $O/bench_8x16_exchange_20steps.log 2>&1; grep '^{"metric"' $O/bench_8x16_exchange_20steps.log > $O/bench_line_8x16_exchange_20steps.json
python - <<'PY'
import json
O="gpurun_out/r06/final/"
l=json.load(open(O+"bench_line_final.json"))
print("8x16", l["value"], l["ms_per_step"], l["roofline"]["kernel_ms"], l["parity"]["full_batch_vs_exhaustive_bit_exact"], l["roofline"]["frac"], l["roofline"].get("traffic_stale"), l["roofline_fp64"].get("frac"), l["roofline_fp64"].get("frac_active_lanes"), l["roofline_fp64"].get("stale"))
for f in ("bench_line_64x256_exchange.json","bench_line_8x16_exchange.json","bench_line_8x16_exchange_20steps.json"):
    l=json.load(open(O+f)); c=l["config"]; print(f, l["ms_per_step"], c["frames_per_s"], c["overflow_frames"], (c.get("exchange") or {}).get("exposed_ms"), (c.get("exchange") or {}).get("chunks_per_step"), l["roofline"].get("traffic_stale"))
PY
