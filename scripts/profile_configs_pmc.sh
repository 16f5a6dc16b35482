#!/bin/bash
# GPU box: kernel stats + counter passes (FP64 instruction mix, lane cycles, FETCH_SIZE, WRITE_SIZE: one set per run, never
# combined with other trace domains) of BASELINE configs[1] (4 x 4, 1 M frames) and configs[4] (64 x 256, 12 500 frames), and
# the JSON summaries bench.py's sub-records read.   usage: profile_configs_pmc.sh <git head> [tag]
set -u
HEAD=${1:-unknown}; TAG=${2:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG/cfgpmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
summ() {  # <dir> <name>
  DB=$(find $1 -name "*.db" | head -1)
  python $R/scripts/rocpd_summary.py stats $DB | grep -v "rocclr\|at::native\|rocprim" > $O/$2_kernel_stats.csv
  python $R/scripts/rocpd_summary.py pmc $DB | grep -v "rocclr\|at::native\|rocprim" > $O/$2_pmc.csv
  find $1 -name "*.db" -delete
}
for cfg in "4x4:" "64x256:--frames 12500"; do
  name=${cfg%%:*}; extra=${cfg#*:}
  CMD="python $R/bench.py --workload $name $extra --steps 2 --warmup 1 --no-cpu-baseline"
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/st -o p -- $CMD > $O/${name}_stats.log 2>&1; summ $O/st ${name}; mv $O/${name}_pmc.csv $O/${name}_unused.csv
  grep '^{"metric"' $O/${name}_stats.log | tail -1 > $O/${name}_bench_line.json
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES -d $O/mx -o p -- $CMD > $O/${name}_mix.log 2>&1
  DB=$(find $O/mx -name "*.db" | head -1); python $R/scripts/rocpd_summary.py pmc $DB | grep -v "rocclr\|at::native\|rocprim" > $O/${name}_mix_pmc.csv; find $O/mx -name "*.db" -delete
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU -d $O/is -o p -- $CMD > $O/${name}_issue.log 2>&1
  DB=$(find $O/is -name "*.db" | head -1); python $R/scripts/rocpd_summary.py pmc $DB | grep -v "rocclr\|at::native\|rocprim" > $O/${name}_issue_pmc.csv; find $O/is -name "*.db" -delete
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fe -o p -- $CMD > $O/${name}_fetch.log 2>&1
  DB=$(find $O/fe -name "*.db" | head -1); python $R/scripts/rocpd_summary.py pmc $DB | grep -v "rocclr\|at::native\|rocprim" > $O/${name}_fetch_pmc.csv; find $O/fe -name "*.db" -delete
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/wr -o p -- $CMD > $O/${name}_write.log 2>&1
  DB=$(find $O/wr -name "*.db" | head -1); python $R/scripts/rocpd_summary.py pmc $DB | grep -v "rocclr\|at::native\|rocprim" > $O/${name}_write_pmc.csv; find $O/wr -name "*.db" -delete
  python $R/scripts/make_config_pmc_json.py $O $name 3 $O/$TAG $HEAD
done
