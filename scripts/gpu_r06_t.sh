#!/bin/bash
# round 6, call T: heavy-frame slicing of the wide variant at the stress shape (runtime knobs; every slice re-does the frame's matching),
# on two streams (TW_SEED)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for seed in 1 7; do
export TW_SEED=$seed
python scripts/time_wide.py 12500 1 > /dev/null 2>&1
for kv in "MOCAP_SLICE_SIZE=8192 MOCAP_HEAVY_THRESHOLD=32768" "MOCAP_SLICE_SIZE=20480 MOCAP_HEAVY_THRESHOLD=40960" "MOCAP_SLICE_SIZE=24576 MOCAP_HEAVY_THRESHOLD=49152" "MOCAP_SLICE_SIZE=28672 MOCAP_HEAVY_THRESHOLD=57344" "MOCAP_SLICE_SIZE=24576 MOCAP_HEAVY_THRESHOLD=32768" "MOCAP_SLICE_SIZE=24576 MOCAP_HEAVY_THRESHOLD=65536" "MOCAP_SLICE_SIZE=32768 MOCAP_HEAVY_THRESHOLD=49152" "MOCAP_SLICE_SIZE=16384 MOCAP_HEAVY_THRESHOLD=49152"; do
  echo "== seed $seed $kv: $(env $kv timeout 200 python scripts/time_wide.py 12500 5 2>&1 | tail -1 | cut -c1-60)"
done
done
