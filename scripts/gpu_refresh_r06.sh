#!/bin/bash
# GPU box, one call: everything profiles/r06_* is made from, on the CURRENT sources (bench.py marks counter-derived figures
# `stale` when the source hash moves).   usage: gpu_refresh_r06.sh <git head>
set -u
HEAD=${1:-unknown}; TAG=r06
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG/final; rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/gpu_suite.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/gpu_suite.txt
bash scripts/profile_frame_pmc.sh $HEAD $TAG 2>&1 | tail -2
bash scripts/profile_configs_pmc.sh $HEAD $TAG 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for m in skip fold; do
  [ $m = fold ] && X=--fold || X=""
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/bp -o p -- python $R/scripts/bench_blobs.py --frames 1024 --steps 5 $X > $O/blob_$m.log 2>&1
  DB=$(find $O/bp -name "*.db" | head -1); python $R/scripts/rocpd_summary.py stats $DB | grep "name,calls\|mocap::" > $O/blob_kernel_stats_$m.csv; find $O/bp -name "*.db" -delete
  rm -f $O/blob_pmc_traffic_$m.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O/bp -o p -- python $R/scripts/bench_blobs.py --frames 1024 --steps 2 $X > /dev/null 2>&1
    DB=$(find $O/bp -name "*.db" | head -1); python $R/scripts/rocpd_summary.py pmc $DB | grep "kernel,counter\|mocap::" >> $O/blob_pmc_traffic_$m.csv; find $O/bp -name "*.db" -delete
  done
  tail -1 $O/blob_$m.log | cut -c1-200
done
cd $R
timeout 900 python bench.py > $O/bench_final.log 2>&1; echo "bench rc=$?"
grep '^{"metric"' $O/bench_final.log > $O/bench_line_final.json
timeout 400 python bench.py --workload 64x256 --frames 12500 --steps 3 --warmup 1 > $O/bench_64x256.log 2>&1; grep '^{"metric"' $O/bench_64x256.log > $O/bench_line_64x256.json
MOCAP_BENCH_EXCHANGE=1 timeout 400 python bench.py --workload 64x256 --frames 12500 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_64x256_exchange.log 2>&1; grep '^{"metric"' $O/bench_64x256_exchange.log > $O/bench_line_64x256_exchange.json
MOCAP_BENCH_EXCHANGE=1 timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ba --no-blobs --no-latency --no-configs > $O/bench_8x16_exchange.log 2>&1; grep '^{"metric"' $O/bench_8x16_exchange.log > $O/bench_line_8x16_exchange.json
python - <<'PY'
import json
O="gpurun_out/r06/final/"
l=json.load(open(O+"bench_line_final.json"))
print("8x16", l["value"], l["ms_per_step"], l["parity"]["full_batch_vs_exhaustive_bit_exact"], l["roofline"]["frac"], l["roofline_fp64"].get("frac"), l["roofline_fp64"].get("frac_active_lanes"), l["roofline_fp64"].get("stale"))
for n,c in l.get("configs",{}).items(): print(n, {k:c.get(k) for k in ("ms_per_step","frames_per_s","value","overflow_frames","flagged_by_first_pass","error")}, c.get("bounded_resubmit"), c.get("roofline_fp64",{}).get("frac"), c.get("parity",{}).get("run_to_run"))
d=l["ba"]["default_mode"]; print("ba", l["ba"]["value"], d["iterations_per_s"], d["inside_core_calls_s"], d.get("one_blas_thread"))
for f in ("bench_line_64x256.json","bench_line_64x256_exchange.json","bench_line_8x16_exchange.json"):
    l=json.load(open(O+f)); c=l["config"]; print(f, l["ms_per_step"], c["frames_per_s"], c["overflow_frames"], (c.get("exchange") or {}).get("exposed_ms"), (c.get("exchange") or {}).get("chunks_per_step"))
PY
