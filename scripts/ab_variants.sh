#!/bin/bash
# A/B of kernel variants on the GPU box: one bench.py run per library under lib/variants/ (or the
# paths given), same workload, prints kernel_ms / parity per variant.  Usage (via gpurun):
#   bash scripts/ab_variants.sh [frames] [lib ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
FRAMES=${1:-100000}; shift
LIBS=${@:-$(ls $R/low-cost-mocap_amd/lib/variants/*.so)}
for L in $LIBS; do
  MOCAP_CORE_LIB=$L python $R/bench.py --steps 5 --warmup 2 --frames $FRAMES --no-cpu-baseline --no-ba 2>&1 | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $L)', 'kernel_ms=%.3f'%d['roofline']['kernel_ms'], 'ms_per_step=%.3f'%d['ms_per_step'], d['parity'])"
done
