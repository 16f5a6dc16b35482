#!/bin/bash
# A/B PMC pass on the GPU box: the frame kernel under two settings of one environment variable.
#   bash scripts/pmc_ab.sh <ENVVAR> "<value A> <value B>" "<counters>" [frames]
set -u
VAR=$1; VALS=$2; CTRS=$3; FR=${4:-20000}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in $VALS; do
  OUT=$R/gpurun_out/pmc_ab_${VAR}_$v
  rm -rf $OUT; mkdir -p $OUT
  env $VAR=$v timeout 300 rocprofv3 --kernel-trace --pmc $CTRS -d $OUT -o p -- python $R/bench.py --steps 2 --warmup 1 --frames $FR --no-cpu-baseline --no-ba --no-blobs --no-latency > $OUT/log.txt 2>&1
  DB=$(find $OUT -name "*.db" | head -1)
  echo "== $VAR=$v"
  python $R/scripts/rocpd_summary.py pmc $DB | grep "frame_kernel\|counter" | cut -c1-60,200-
  python $R/scripts/rocpd_summary.py stats $DB | grep "frame_kernel" | cut -c1-40,200-
  find $OUT -name "*.db" -size +8M -delete
done
