#!/bin/bash
# round 6, call R: compiler-flag variants of csrc/frame_bb.hip (the headline kernel) at the bench's 100 k frames of 8 x 16
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python scripts/time_frame.py 100000 1 > /dev/null 2>&1
for i in 1 2; do
  for v in base "$@"; do
    [ $v = base ] && unset MOCAP_CORE_LIB || export MOCAP_CORE_LIB=$R/low-cost-mocap_amd/lib/libmocap_core_$v.so
    echo "== $v: $(timeout 200 python scripts/time_frame.py 100000 7 2>&1 | tail -1 | cut -c1-110)"
  done
done
