#!/bin/bash
# GPU box: kernel-trace summaries of the other BASELINE configs on the current tree -> gpurun_out/<tag>/cfg/
#   64 x 256 at its per-GPU share (12 500 frames) and 4 x 4 (1 M frames).   usage: profile_configs.sh [tag]
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG/cfg; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
summ() { DB=$(find $1 -name "*.db" | head -1); python $R/scripts/rocpd_summary.py stats $DB | grep -v "rocclr\|at::native\|rocprim" > $O/$2_kernel_stats.csv; find $1 -name "*.db" -delete; }
timeout 300 rocprofv3 --kernel-trace --stats -d $O/w -o p -- python $R/bench.py --workload 64x256 --frames 12500 --steps 3 --warmup 1 > $O/w.log 2>&1; summ $O/w 64x256
timeout 300 rocprofv3 --kernel-trace --stats -d $O/s -o p -- python $R/bench.py --workload 4x4 --steps 5 --warmup 2 > $O/s.log 2>&1; summ $O/s 4x4
grep '^{"metric"' $O/s.log | tail -1 > $O/bench_line_4x4.json
head -3 $O/64x256_kernel_stats.csv $O/4x4_kernel_stats.csv
