#!/bin/bash
# round 6, call X: phase-priority variants of the blob stage's mask kernel (lib/libmocap_core_<tag>.so), 8 192 images
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python scripts/bench_blobs.py --frames 1024 --steps 2 > /dev/null 2>&1
for i in 1 2; do
  for v in base "$@"; do
    [ $v = base ] && unset MOCAP_CORE_LIB || export MOCAP_CORE_LIB=$R/low-cost-mocap_amd/lib/libmocap_core_$v.so
    echo "== $v: $(timeout 200 python scripts/bench_blobs.py --frames 1024 --steps 7 2>&1 | tail -1 | cut -c1-200)"
  done
done
