#!/bin/bash
# One box: wide-variant timing of library variants (lib/libmocap_core_<tag>.so; "base" = product) at N frames of 64 x 256.
#   usage: scripts/gpu_wide_ab.sh <frames> base tag1 tag2 ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06; mkdir -p $O
cd $R
N=$1; shift
python scripts/time_wide.py $N 1 > /dev/null 2>&1
{
for i in 1 2; do
  for v in "$@"; do
    [ $v = base ] && unset MOCAP_CORE_LIB || export MOCAP_CORE_LIB=$R/low-cost-mocap_amd/lib/libmocap_core_$v.so
    echo "== $v: $(timeout 200 python scripts/time_wide.py $N 5 2>&1 | tail -1 | cut -c1-200)"
  done
done
} 2>&1 | tee $O/wide_ab_$(date +%H%M%S).log
