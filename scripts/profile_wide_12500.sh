#!/bin/bash
# Wide variant at BASELINE configs[4]'s per-GPU share (12 500 frames of 64 x 256 per launch): counter passes, one set per run.
# Summary -> gpurun_out/r04/wide12500_pmc.csv
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/w12; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
W="python $R/bench.py --workload 64x256 --frames 12500 --steps 2 --warmup 1 --no-cpu-baseline --no-ba --no-blobs --no-latency"
pmc() { python $R/scripts/rocpd_summary.py pmc $(find $1 -name "*.db" | head -1) | grep "frame_kernel<1024"; find $1 -name "*.db" -delete; }
{
echo "# wide kernel, bench.py --workload 64x256 --frames 12500 (scripts/profile_wide_12500.sh): per launch of 12 500 frames"
echo "kernel,counter,dispatches,sum,avg_per_dispatch"
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU -d $O/a -o p -- $W > $O/a.log 2>&1
python $R/scripts/rocpd_summary.py stats $(find $O/a -name "*.db" | head -1) | grep "frame_kernel<1024" | sed 's/^/# kernel stats of this pass: /'
pmc $O/a
[ "${WIDE_SKIP_B:-0}" = 1 ] || timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVE_CYCLES -d $O/b -o p -- $W > $O/b.log 2>&1; [ "${WIDE_SKIP_B:-0}" = 1 ] || pmc $O/b
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/c -o p -- $W > $O/c.log 2>&1; pmc $O/c
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/d -o p -- $W > $O/d.log 2>&1; pmc $O/d
} > $R/gpurun_out/r04/wide12500_pmc.csv
cat $R/gpurun_out/r04/wide12500_pmc.csv | cut -c1-160
