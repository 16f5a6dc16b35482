#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for pl in 8 10 12 14 16 20; do
  MOCAP_BB_PL=$pl timeout 300 python scripts/time_frame.py 100000 7 2>&1 | grep -v amdgpu | sed 's/frame_bb_kernel<CW=1> //; s/cands.frame [0-9.]* //'
done
for fl in 128 512; do
  MOCAP_BB_FLUSH=$fl timeout 300 python scripts/time_frame.py 100000 7 2>&1 | grep -v amdgpu | sed 's/frame_bb_kernel<CW=1> //; s/cands.frame [0-9.]* //'
done
