#!/bin/bash
# Quick PMC pass on the GPU box: kernel-trace + one SQ counter set (own run, no other trace domains).
#   bash scripts/pmc_quick.sh <tag> "<counters>" [frames]
set -u
TAG=$1; CTRS=$2; FR=${3:-20000}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc $CTRS -d $OUT -o p -- python $R/bench.py --steps 2 --warmup 1 --frames $FR --no-cpu-baseline --no-ba > $OUT/log.txt 2>&1
DB=$(find $OUT -name "*.db" | head -1)
python $R/scripts/rocpd_summary.py pmc $DB | grep -v "rocclr\|at::native" > $OUT/summary.csv
python $R/scripts/rocpd_summary.py stats $DB | grep -v "rocclr\|at::native" >> $OUT/summary.csv
find $OUT -name "*.db" -size +8M -delete
cat $OUT/summary.csv
