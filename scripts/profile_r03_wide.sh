#!/bin/bash
# Wide variant (BASELINE configs[4], 64 cams x 256 markers): kernel stats + PMC passes (instruction mix, VALU issue,
# LDS bank conflicts, HBM traffic), each counter set in its own run.  Summaries -> gpurun_out/r03/wide_*.csv
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03
FR=${1:-1024}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
summ() {  # <dir> <tag>
  DB=$(find $1 -name "*.db" | head -1)
  python $R/scripts/rocpd_summary.py stats $DB | grep -v "rocclr\|at::native" > $OUT/$2_kernel_stats.csv
  python $R/scripts/rocpd_summary.py pmc $DB | grep -v "rocclr\|at::native" > $OUT/$2_pmc.csv
  find $1 -name "*.db" -delete
}
CMD="python $R/bench.py --workload 64x256 --frames $FR --steps 2 --warmup 1 --no-cpu-baseline --no-ba --no-blobs --no-latency"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/w_stats -o p -- $CMD > $OUT/wide_stats.log 2>&1; summ $OUT/w_stats wide_stats
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU -d $OUT/w_sq -o p -- $CMD > $OUT/wide_sq.log 2>&1; summ $OUT/w_sq wide_sq
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $OUT/w_lds -o p -- $CMD > $OUT/wide_lds.log 2>&1; summ $OUT/w_lds wide_lds
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/w_fetch -o p -- $CMD > $OUT/wide_fetch.log 2>&1; summ $OUT/w_fetch wide_fetch
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/w_write -o p -- $CMD > $OUT/wide_write.log 2>&1; summ $OUT/w_write wide_write
tail -c 600 $OUT/wide_stats.log; cat $OUT/wide_*_kernel_stats.csv | head -8; cat $OUT/wide_*_pmc.csv
