#!/bin/bash
# where the one-GPU exchange line's extra time goes at the stress shape: kernel trace of the exchange code path
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05l; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MOCAP_BENCH_EXCHANGE=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/ex -o p -- python $R/bench.py --workload 64x256 --frames 12500 --steps 2 --warmup 1 > $O/ex.log 2>&1
DB=$(find $O/ex -name "*.db" | head -1)
python $R/scripts/rocpd_summary.py stats $DB > $O/exchange_kernel_stats.csv
find $O/ex -name "*.db" -delete
head -25 $O/exchange_kernel_stats.csv | cut -c1-200
grep -o '"ms_per_step": [0-9.]*' $O/ex.log | head -2
