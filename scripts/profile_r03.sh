#!/bin/bash
# Round-3 tracked profiles other than the frame kernel's counter passes (scripts/profile_frame_pmc.sh): BA kernel
# durations (launch-ahead off), the 4 x 4 and 64 x 256 configurations with kernel stats, and the counter passes of the
# wide variant after its rewrite.  Summaries -> gpurun_out/r03/cfg/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/cfg; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stats() { python $R/scripts/rocpd_summary.py stats $(find $1 -name "*.db" | head -1) | grep -v "rocclr\|at::native\|rocprim" > $2; }
pmc() { python $R/scripts/rocpd_summary.py pmc $(find $1 -name "*.db" | head -1) | grep -v "rocclr\|at::native\|rocprim" > $2; }
for n in 1000 16000; do
  MOCAP_BA_NO_PREARM=1 timeout 200 rocprofv3 --kernel-trace -d $O/k$n -o p -- python $R/scripts/prof_ba.py $n > $O/ba_$n.log 2>&1
  stats $O/k$n $O/ba_kernel_stats_$n.csv; tail -1 $O/ba_$n.log
done
timeout 300 rocprofv3 --kernel-trace -d $O/p4 -o p -- python $R/bench.py --workload 4x4 --steps 5 --warmup 2 > $O/bench_4x4.log 2>&1
stats $O/p4 $O/kernel_stats_4x4.csv; grep '^{"metric"' $O/bench_4x4.log > $O/bench_line_4x4.json
timeout 900 rocprofv3 --kernel-trace -d $O/p64 -o p -- python $R/bench.py --workload 64x256 --frames 12500 --steps 2 --warmup 1 > $O/bench_64x256.log 2>&1
stats $O/p64 $O/kernel_stats_64x256_12500frames.csv; grep '^{"metric"' $O/bench_64x256.log > $O/bench_line_64x256_12500frames.json
W="python $R/bench.py --workload 64x256 --frames 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-ba --no-blobs --no-latency"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU -d $O/w1 -o p -- $W > $O/w1.log 2>&1; pmc $O/w1 $O/wide_pmc_sq.csv; stats $O/w1 $O/wide_pmc_sq_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVE_CYCLES -d $O/w2 -o p -- $W > $O/w2.log 2>&1; pmc $O/w2 $O/wide_pmc_lds.csv
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/w3 -o p -- $W > $O/w3.log 2>&1; pmc $O/w3 $O/wide_pmc_fetch.csv
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w4 -o p -- $W > $O/w4.log 2>&1; pmc $O/w4 $O/wide_pmc_write.csv
find $O -name "*.db" -delete
head -3 $O/kernel_stats_4x4.csv $O/kernel_stats_64x256_12500frames.csv $O/ba_kernel_stats_1000.csv $O/ba_kernel_stats_16000.csv
