#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats + PMC passes (HBM traffic, SQ) of the blob-extraction stage.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_blob_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/bench_blobs.py --frames 1024 --steps 3 ${BLOB_ARGS:---no-skip}"
run() {  # name, extra rocprof args
  local n=$1; shift
  timeout 300 rocprofv3 --kernel-trace "$@" -d $OUT/$n -o p -- $CMD > $OUT/$n.log 2>&1
  local DB=$(find $OUT/$n -name "*.db" | head -1)
  python $R/scripts/rocpd_summary.py ${MODE:-pmc} $DB | grep -v "rocclr\|at::native" > $OUT/$n.csv
  rm -rf $OUT/$n
}
MODE=stats run kernel_stats --stats
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
run pmc_lds --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_LDS
tail -n +1 $OUT/*.csv
