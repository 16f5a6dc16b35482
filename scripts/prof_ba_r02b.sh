#!/bin/bash
# BA kernel durations on the GPU box: per phase (MOCAP_BA_DEBUG_STOP) and of the whole launch without launch-ahead.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02ba; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for st in 1 2 3 4 5 6 7 0; do
  MOCAP_BA_DEBUG_STOP=$st timeout 100 rocprofv3 --kernel-trace -d $OUT/ph$st -o p -- python $R/scripts/prof_ba_phases.py 1000 > $OUT/ph$st.log 2>&1
  echo -n "stop $st: "; python $R/scripts/rocpd_summary.py stats $(find $OUT/ph$st -name "*.db" | head -1) | grep ba_fused | cut -d, -f2-6
done
for n in 1000 16000; do
  MOCAP_BA_NO_PREARM=1 timeout 200 rocprofv3 --kernel-trace -d $OUT/k$n -o p -- python $R/scripts/prof_ba.py $n > $OUT/k$n.log 2>&1
  python $R/scripts/rocpd_summary.py stats $(find $OUT/k$n -name "*.db" | head -1) | grep -v "rocclr\|at::native" > $OUT/ba_${n}_kernel_stats.csv
  cat $OUT/ba_${n}_kernel_stats.csv; tail -1 $OUT/k$n.log
done
find $OUT -name "*.db" -delete
