#!/bin/bash
# round 6, call F: where the wide kernel's camera-0 pass spends its time: counters on a variant that runs ONLY that pass
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06f; mkdir -p $O
cd $R
python scripts/time_wide.py 12500 1 > /dev/null 2>&1
export MOCAP_CORE_LIB=$R/low-cost-mocap_amd/lib/libmocap_core_skip6.so
echo "skip6: $(python scripts/time_wide.py 12500 5 2>&1 | tail -1 | cut -c1-120)"
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_VMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC"; do
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/p -o p -- python $R/scripts/time_wide.py 12500 2 > $O/log.txt 2>&1
  DB=$(find $O/p -name "*.db" | head -1); python $R/scripts/rocpd_summary.py pmc $DB | grep "frame_kernel" | sed 's/.*FrameArgs)",//' ; find $O/p -name "*.db" -delete
done
