#!/bin/bash
# Round-2 measurement pass on the GPU box (via gpurun): bench line, kernel stats, BA kernel stats + MFMA counters,
# frame-kernel FP64 instruction-mix counters.  Outputs under gpurun_out/r02/ ; summaries are copied to profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
summ() {  # <dir> <tag>
  DB=$(find $1 -name "*.db" | head -1)
  python $R/scripts/rocpd_summary.py stats $DB | grep -v "rocclr\|at::native" > $OUT/$2_kernel_stats.csv
  python $R/scripts/rocpd_summary.py pmc $DB | grep -v "rocclr\|at::native" > $OUT/$2_pmc.csv
  find $1 -name "*.db" -size +8M -delete
}
timeout 400 python $R/bench.py > $OUT/bench.log 2>&1
CMD="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ba --no-blobs --no-latency"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- $CMD > $OUT/stats.log 2>&1; summ $OUT/stats bench
timeout 200 rocprofv3 --kernel-trace -d $OUT/ba1k -o p -- python $R/scripts/prof_ba.py 1000 > $OUT/ba1k.log 2>&1; summ $OUT/ba1k ba_1k
timeout 200 rocprofv3 --kernel-trace -d $OUT/ba16k -o p -- python $R/scripts/prof_ba.py 16000 > $OUT/ba16k.log 2>&1; summ $OUT/ba16k ba_16k
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES -d $OUT/ba_mfma -o p -- python $R/scripts/prof_ba.py 1000 > $OUT/ba_mfma.log 2>&1; summ $OUT/ba_mfma ba_pmc_mfma
CMDS="python $R/bench.py --steps 2 --warmup 1 --frames 20000 --no-cpu-baseline --no-ba --no-blobs --no-latency"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES -d $OUT/frame_f64 -o p -- $CMDS > $OUT/frame_f64.log 2>&1; summ $OUT/frame_f64 frame_pmc_f64
# HBM traffic of the frame kernel: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md)
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/frame_fetch -o p -- $CMDS > $OUT/frame_fetch.log 2>&1; summ $OUT/frame_fetch frame_pmc_fetch
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/frame_write -o p -- $CMDS > $OUT/frame_write.log 2>&1; summ $OUT/frame_write frame_pmc_write
ls -la $OUT
