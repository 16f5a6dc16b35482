#!/bin/bash
# round 6, call D: whole GPU suite; blob stage A/B (activity pre-pass / folded early-out / none) with kernel trace + HBM
# counters of the folded mode; the exchange code path at the stress shape on one GPU; counter summaries of the sub-record configs
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06d; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for m in "" "--fold" "--no-skip"; do
  echo "blobs [$m] $(timeout 200 python scripts/bench_blobs.py --frames 1024 --steps 7 $m 2>&1 | tail -1 | cut -c1-420)"
done
cd /tmp && export TMPDIR=/tmp
for m in skip fold; do
  [ $m = fold ] && X=--fold || X=""
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/bp -o p -- python $R/scripts/bench_blobs.py --frames 1024 --steps 5 $X > $O/blob_$m.log 2>&1
  DB=$(find $O/bp -name "*.db" | head -1); python $R/scripts/rocpd_summary.py stats $DB | grep "mocap::" > $O/blob_kernel_stats_$m.csv; find $O/bp -name "*.db" -delete
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O/bp -o p -- python $R/scripts/bench_blobs.py --frames 1024 --steps 2 $X > /dev/null 2>&1
    DB=$(find $O/bp -name "*.db" | head -1); python $R/scripts/rocpd_summary.py pmc $DB | grep "mocap::" >> $O/blob_pmc_traffic_$m.csv; find $O/bp -name "*.db" -delete
  done
  cat $O/blob_kernel_stats_$m.csv $O/blob_pmc_traffic_$m.csv
done
cd $R
for v in default bounded; do
  [ $v = bounded ] && export MOCAP_NO_HEAVY_ENUM=1 || unset MOCAP_NO_HEAVY_ENUM
  MOCAP_BENCH_EXCHANGE=1 timeout 400 python bench.py --workload 64x256 --frames 12500 --steps 3 --warmup 1 --no-cpu-baseline > $O/exch_$v.log 2>&1
  grep '^{"metric"' $O/exch_$v.log > $O/exch_line_$v.json
  python -c "import json; l=json.load(open('$O/exch_line_$v.json')); c=l['config']; print('exchange $v', l['ms_per_step'], c['overflow_frames'], c['exchange']['chunks_per_step'], c['exchange']['exposed_ms'])"
done
unset MOCAP_NO_HEAVY_ENUM
bash scripts/profile_configs_pmc.sh $1 r06 2>&1 | tail -4
