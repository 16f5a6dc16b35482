#!/bin/bash
# PMC pass for the blob-extraction kernels on the GPU box:  bash scripts/pmc_blobs.sh <tag> "<counters>"
set -u
TAG=$1; CTRS=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc $CTRS -d $OUT -o p -- python $R/scripts/bench_blobs.py --frames 256 --steps 2 > $OUT/log.txt 2>&1
DB=$(find $OUT -name "*.db" | head -1)
python $R/scripts/rocpd_summary.py pmc $DB | grep -v "rocclr\|at::native" > $OUT/summary.csv
find $OUT -name "*.db" -delete
cat $OUT/summary.csv
