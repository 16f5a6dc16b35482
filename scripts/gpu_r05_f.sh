#!/bin/bash
# round 5: instruction counters of the shipped headline kernel against the probe-scheme variant (fewer full evaluations, slower):
# does the variant execute fewer instructions?  gpurun_out/r05f/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in base pr; do
  [ $v = base ] && unset MOCAP_CORE_LIB || export MOCAP_CORE_LIB=$R/low-cost-mocap_amd/lib/libmocap_core_$v.so
  timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d $O/ph_$v -o p -- python $R/scripts/time_frame.py 100000 3 > $O/ph_$v.log 2>&1
  DB=$(find $O/ph_$v -name "*.db" | head -1)
  echo "== $v"; python $R/scripts/rocpd_summary.py pmc $DB | grep "frame_bb" | sed 's/.*FrameArgs)",//'
  python $R/scripts/rocpd_summary.py stats $DB | grep "frame_bb" | sed 's/.*FrameArgs)",//'
  find $O/ph_$v -name "*.db" -delete
done 2>&1 | tee $O/summary.txt
