#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats of the bench command + PMC passes.
# Outputs land under gpurun_out/prof_<tag>/ ; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ba --no-blobs --no-latency"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- $CMD > $OUT/stats.log 2>&1
# PMC passes (separate runs, kernel-trace only), smaller batch to keep them short
CMDS="python $R/bench.py --steps 2 --warmup 1 --frames 20000 --no-cpu-baseline --no-ba --no-blobs --no-latency"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $CMDS > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- $CMDS > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d $OUT/pmc_sq -o p -- $CMDS > $OUT/pmc_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM -d $OUT/pmc_lds -o p -- $CMDS > $OUT/pmc_lds.log 2>&1
timeout 60 rocprofv3 -L > $OUT/counters_list.txt 2>&1
find $OUT -name "*.db" -size +20M -delete
ls -R $OUT | head -50
