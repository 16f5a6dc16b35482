#!/bin/bash
# 4 x 4 (BASELINE.json configs[1]) on the GPU box: JSON line + kernel stats -> gpurun_out/r02cfg/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02cfg
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/p4x4 -o p -- python $R/bench.py --workload 4x4 --steps 5 --warmup 2 > $OUT/bench_4x4.log 2>&1
python $R/scripts/rocpd_summary.py stats $(find $OUT/p4x4 -name "*.db" | head -1) | grep -v "rocclr\|at::native" > $OUT/kernel_stats_4x4.csv
find $OUT -name "*.db" -delete
cat $OUT/kernel_stats_4x4.csv
