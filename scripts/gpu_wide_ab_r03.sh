#!/bin/bash
# One box: wide-variant timing of library variants (lib/libmocap_core_<tag>.so; "base" = product) at 1 024 frames of 64 x 256,
# optionally the wide parity tests (last argument "parity").
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python scripts/time_wide.py 1024 1 > /dev/null 2>&1
{
for i in 1 2; do
  for v in "$@"; do
    [ $v = parity ] && continue
    [ $v = base ] && unset MOCAP_CORE_LIB || export MOCAP_CORE_LIB=$R/low-cost-mocap_amd/lib/libmocap_core_$v.so
    echo "== $v: $(timeout 200 python scripts/time_wide.py 1024 5 2>&1 | tail -1 | cut -c1-200)"
  done
done
} 2>&1 | tee $O/wide_ab_$(date +%H%M%S).log
unset MOCAP_CORE_LIB
for v in "$@"; do
  [ $v = parity ] && timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "wide or medium or stress or scheduling or non_uniform" 2>&1 | tail -5 | tee $O/wide_parity.log
done
