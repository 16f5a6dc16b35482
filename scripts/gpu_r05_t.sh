#!/bin/bash
# quick timing of the library in the tree (+ optional variants): gpu_r05_t.sh [suffix ...]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r05t; mkdir -p $O; rm -f $O/time_frame.log
for v in "" "$@" "" "$@"; do
  MOCAP_CORE_LIB=low-cost-mocap_amd/lib/libmocap_core$v.so timeout 300 python scripts/time_frame.py 100000 9 >> $O/time_frame.log 2>&1
done
grep -v amdgpu.ids $O/time_frame.log | sed 's/frame_bb_kernel<CW=1> //; s/cands.frame [0-9.]* //'
