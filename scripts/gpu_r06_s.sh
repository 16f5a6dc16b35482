#!/bin/bash
# round 6, call S: the wide variant's speculation threshold (builds) and heavy-frame slicing (runtime knobs) at the stress shape
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash scripts/gpu_wide_ab.sh 12500 base t16 t48 2>&1 | grep "^==" | cut -c1-90
unset MOCAP_CORE_LIB
for kv in "MOCAP_HEAVY_THRESHOLD=16384" "MOCAP_HEAVY_THRESHOLD=65536" "MOCAP_HEAVY_THRESHOLD=8192 MOCAP_SLICE_SIZE=4096" "MOCAP_SLICE_SIZE=4096" "MOCAP_SLICE_SIZE=16384" "MOCAP_HEAVY_THRESHOLD=0"; do
  echo "== $kv: $(env $kv timeout 200 python scripts/time_wide.py 12500 5 2>&1 | tail -1 | cut -c1-60)"
done
