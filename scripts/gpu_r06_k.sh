#!/bin/bash
# round 6, call K: heavy_enum_kernel with the two-level suffix: kernel trace at the stress shape + its tests
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06k; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_wide_adversarial.py -m gpu -x -q -k "heavy_root or behind" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python $R/bench.py --workload 64x256 --frames 12500 --steps 3 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1
DB=$(find $O/prof -name "*.db" | head -1); python $R/scripts/rocpd_summary.py stats $DB | grep "mocap::" | head -6
find $O/prof -name "*.db" -delete
grep '^{"metric"' $O/prof.log | python -c "import json,sys; l=json.loads(sys.stdin.read()); c=l['config']; print(l['ms_per_step'], c['frames_per_s'], c['overflow_frames'], c['flagged_by_first_pass'], l['parity']['corr_bit_exact'])"
