#!/bin/bash
# round 6, call Z: library variants at the stress shape, the WHOLE hot-path call (first pass + re-submit: heavy-root search, enumeration)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for i in 1 2; do
for v in base "$@"; do
  [ $v = base ] && unset MOCAP_CORE_LIB || export MOCAP_CORE_LIB=$R/low-cost-mocap_amd/lib/libmocap_core_$v.so
  echo "== $v: $(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pz_$v$i -o p -- python $R/bench.py --workload 64x256 --frames 12500 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "import json,sys; l=json.loads(sys.stdin.read()); print(l['ms_per_step'])") $(python $R/scripts/rocpd_summary.py stats $(find /tmp/pz_$v$i -name '*.db' | head -1) | grep 'heavy' | awk -F, '{printf "%s=%.3f ", substr($1,14,18), $4/1e6}')"
done
done
