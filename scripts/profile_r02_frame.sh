#!/bin/bash
# Frame-kernel part of scripts/profile_r02.sh on its own (kernel stats, FP64 instruction mix, HBM traffic).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02f
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
summ() {  # <dir> <tag>
  DB=$(find $1 -name "*.db" | head -1)
  python $R/scripts/rocpd_summary.py stats $DB | grep -v "rocclr\|at::native" > $OUT/$2_kernel_stats.csv
  python $R/scripts/rocpd_summary.py pmc $DB | grep -v "rocclr\|at::native" > $OUT/$2_pmc.csv
  find $1 -name "*.db" -delete
}
CMD="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ba --no-blobs --no-latency"
timeout 60 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- $CMD > $OUT/stats.log 2>&1; summ $OUT/stats bench
CMDS="python $R/bench.py --steps 2 --warmup 1 --frames 20000 --no-cpu-baseline --no-ba --no-blobs --no-latency"
timeout 60 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES -d $OUT/frame_f64 -o p -- $CMDS > $OUT/frame_f64.log 2>&1; summ $OUT/frame_f64 frame_pmc_f64
timeout 60 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/frame_fetch -o p -- $CMDS > $OUT/frame_fetch.log 2>&1; summ $OUT/frame_fetch frame_pmc_fetch
timeout 60 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/frame_write -o p -- $CMDS > $OUT/frame_write.log 2>&1; summ $OUT/frame_write frame_pmc_write
cat $OUT/bench_kernel_stats.csv
