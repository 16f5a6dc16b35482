#!/bin/bash
# The other BASELINE.json configurations on the GPU box: 4 x 4 (configs[1]) and the 64 x 256 stress set at its full
# per-GPU frame count (configs[4]: 100 k frames over 8 GPUs = 12.5 k per GPU).  JSON lines + kernel stats -> gpurun_out/r02cfg/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02cfg
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/p4x4 -o p -- python $R/bench.py --workload 4x4 --steps 5 --warmup 2 > $OUT/bench_4x4.log 2>&1
python $R/scripts/rocpd_summary.py stats $(find $OUT/p4x4 -name "*.db" | head -1) | grep -v "rocclr\|at::native" > $OUT/kernel_stats_4x4.csv
timeout 1500 rocprofv3 --kernel-trace -d $OUT/p64 -o p -- python $R/bench.py --workload 64x256 --frames 12500 --steps 2 --warmup 1 > $OUT/bench_64x256.log 2>&1
python $R/scripts/rocpd_summary.py stats $(find $OUT/p64 -name "*.db" | head -1) | grep -v "rocclr\|at::native" > $OUT/kernel_stats_64x256.csv
find $OUT -name "*.db" -delete
tail -c 1500 $OUT/bench_4x4.log; echo; tail -c 1500 $OUT/bench_64x256.log; cat $OUT/kernel_stats_*.csv
