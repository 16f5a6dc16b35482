#!/bin/bash
# round 6, call E: heavy_enum_kernel with the contribution table + shared prefix sums: tests, stress timing, kernel trace
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06e; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_wide_adversarial.py tests/test_gpu_track.py tests/test_gpu_bench_scale.py tests/test_gpu_blobs.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python $R/bench.py --workload 64x256 --frames 12500 --steps 3 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1
DB=$(find $O/prof -name "*.db" | head -1); python $R/scripts/rocpd_summary.py stats $DB | grep "mocap::" | head -8 | tee $O/stress_kernel_stats.csv
find $O/prof -name "*.db" -delete
grep '^{"metric"' $O/prof.log | python -c "import json,sys; l=json.loads(sys.stdin.read()); c=l['config']; print(l['ms_per_step'], c['frames_per_s'], c['overflow_frames'], c['flagged_by_first_pass'], l['parity']['corr_bit_exact'])"
