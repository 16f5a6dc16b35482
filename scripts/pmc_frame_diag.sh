#!/bin/bash
# Where do the frame kernel's wave-cycles go?  Three SQ counter passes (own runs, kernel-trace only) over
# scripts/time_frame.py (100 k resident frames of the 8 x 16 bench stream).  usage: pmc_frame_diag.sh <tag>
set -u
TAG=${1:-diag}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # <name> <counters...>
  n=$1; shift
  timeout 120 rocprofv3 --kernel-trace --pmc "$@" -d $O/pmc_${TAG}_$n -o p -- python $R/scripts/time_frame.py 100000 3 > $O/pmc_${TAG}_$n.log 2>&1
  DB=$(find $O/pmc_${TAG}_$n -name "*.db" | head -1)
  python $R/scripts/rocpd_summary.py pmc $DB | grep -v "rocclr\|at::native" > $O/pmc_${TAG}_$n.csv
  python $R/scripts/rocpd_summary.py stats $DB | grep -v "rocclr\|at::native" | head -3 >> $O/pmc_${TAG}_$n.csv
  find $O/pmc_${TAG}_$n -name "*.db" -delete
  sed 's/.*FrameArgs)",//' $O/pmc_${TAG}_$n.csv
}
run s1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run s2 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU
run s3 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC
