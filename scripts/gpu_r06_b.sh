#!/bin/bash
# round 6, call B: re-submit changes (scratch sizing, FINAL / INTRACTABLE bits, heavy_enum_kernel) -- their tests, then the
# stress shape at 12 500 frames with and without the enumeration over the whole GPU
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06b; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_wide_adversarial.py tests/test_gpu_track.py tests/test_gpu_bench_scale.py tests/test_gpu_parity.py tests/test_gpu_multirank.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest.log
for v in default noenum; do
  [ $v = noenum ] && export MOCAP_NO_HEAVY_ENUM=1 || unset MOCAP_NO_HEAVY_ENUM
  timeout 600 python bench.py --workload 64x256 --frames 12500 --steps 3 --warmup 1 --no-cpu-baseline > $O/stress_$v.log 2>&1
  grep '^{"metric"' $O/stress_$v.log | python -c "import json,sys; l=json.loads(sys.stdin.read()); c=l['config']; print('$v', l['ms_per_step'], c['frames_per_s'], c['overflow_frames'], c['flagged_by_first_pass'], c['overflow_by_cap'].get('roots_K_max'), c['overflow_by_cap'].get('candidates_G_cap'), l['parity'])"
done
unset MOCAP_NO_HEAVY_ENUM
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python $R/bench.py --workload 64x256 --frames 12500 --steps 3 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1
DB=$(find $O/prof -name "*.db" | head -1); python $R/scripts/rocpd_summary.py stats $DB | grep -v "rocclr\|at::native\|rocprim" | head -12 | tee $O/stress_kernel_stats.csv
find $O/prof -name "*.db" -delete
