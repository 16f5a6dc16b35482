#!/bin/bash
# round 6, call Q: compiler-flag variants of csrc/frame_kernel.hip: the stress shape (wide variant) and 4 x 4 (one-wave workgroups)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash scripts/gpu_wide_ab.sh 12500 base "$@" 2>&1 | grep "^==" | cut -c1-90
for v in base "$@"; do
  [ $v = base ] && unset MOCAP_CORE_LIB || export MOCAP_CORE_LIB=$R/low-cost-mocap_amd/lib/libmocap_core_$v.so
  echo "4x4 $v: $(timeout 300 python bench.py --workload 4x4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "import json,sys; l=json.loads(sys.stdin.read()); print(l['ms_per_step'], l['value'])")"
done
