#!/bin/bash
# The frame kernel's tracked profile of a round, at bench.py's own 100 k frames: kernel stats of the default bench
# command, then one counter set per run (FP64 instruction mix; issue mix; FETCH_SIZE; WRITE_SIZE), and the two JSON
# summaries bench.py reads (stamped with git HEAD + source hash).  usage: profile_frame_pmc.sh <git head> [tag]
set -u
HEAD=${1:-unknown}; TAG=${2:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG/prof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
summ() {  # <dir> <tag>
  DB=$(find $1 -name "*.db" | head -1)
  python $R/scripts/rocpd_summary.py stats $DB | grep -v "rocclr\|at::native\|rocprim" > $O/$2_kernel_stats.csv
  python $R/scripts/rocpd_summary.py pmc $DB | grep -v "rocclr\|at::native\|rocprim" > $O/$2_pmc.csv
  find $1 -name "*.db" -delete
}
CMD5="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ba --no-blobs --no-latency --no-configs --no-full-parity"
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ba --no-blobs --no-latency --no-configs --no-full-parity"
timeout 120 rocprofv3 --kernel-trace --stats -d $O/stats -o p -- $CMD5 > $O/stats.log 2>&1; summ $O/stats bench
grep '^{"metric"' $O/stats.log | tail -1 > $O/bench_line.json
timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES -d $O/mix -o p -- $CMD > $O/mix.log 2>&1; summ $O/mix frame_mix
timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU -d $O/issue -o p -- $CMD > $O/issue.log 2>&1; summ $O/issue frame_issue
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o p -- $CMD > $O/fetch.log 2>&1; summ $O/fetch frame_fetch
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o p -- $CMD > $O/write.log 2>&1; summ $O/write frame_write
python $R/scripts/make_pmc_json.py $O $O/$TAG $HEAD
head -3 $O/bench_kernel_stats.csv
