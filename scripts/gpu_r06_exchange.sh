#!/bin/bash
# round 6, last session: the exchange code path on one GPU with the whole shard per step (bench.py's automatic choice for runs of
# >= 4 steps) -- kernel trace at 8 x 16, the two exchange bench lines, the multirank / launch tests
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06/exchange; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MOCAP_BENCH_EXCHANGE=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/p -o p -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-full-parity --no-ba --no-blobs --no-latency --no-configs > $O/trace_8x16.log 2>&1
DB=$(find $O/p -name "*.db" | head -1); python $R/scripts/rocpd_summary.py stats $DB > $O/kernel_stats_8x16_exchange.csv; find $O/p -name "*.db" -delete
head -12 $O/kernel_stats_8x16_exchange.csv | cut -c1-160
cd $R
MOCAP_BENCH_EXCHANGE=1 timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ba --no-blobs --no-latency --no-configs > $O/bench_8x16_exchange.log 2>&1; grep '^{"metric"' $O/bench_8x16_exchange.log > $O/bench_line_8x16_exchange.json
MOCAP_BENCH_EXCHANGE=1 timeout 400 python bench.py --workload 64x256 --frames 12500 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_64x256_exchange.log 2>&1; grep '^{"metric"' $O/bench_64x256_exchange.log > $O/bench_line_64x256_exchange.json
python - <<'PY'
import json
O="gpurun_out/r06/exchange/"
for f in ("bench_line_64x256_exchange.json","bench_line_8x16_exchange.json"):
    l=json.load(open(O+f)); c=l["config"]; print(f, l["ms_per_step"], c["frames_per_s"], c["overflow_frames"], (c.get("exchange") or {}).get("exposed_ms"), (c.get("exchange") or {}).get("chunks_per_step"), l.get("parity",{}).get("full_batch_vs_exhaustive_bit_exact"))
PY
timeout 600 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_post.py tests/test_gpu_track.py -m gpu -q > $O/pytest_exchange.txt 2>&1; grep -E "passed|failed|error" $O/pytest_exchange.txt | tail -3
