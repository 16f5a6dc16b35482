#!/bin/bash
# round 6, call W: the headline kernel's run-time knobs (block size, flush level) swept again under the phase priorities
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python scripts/time_frame.py 100000 1 > /dev/null 2>&1
for i in 1 2; do
  for kv in "X=0" MOCAP_BB_PL=8 MOCAP_BB_PL=12 MOCAP_BB_PL=24 MOCAP_BB_PL=32 MOCAP_BB_FLUSH=128 MOCAP_BB_FLUSH=192 MOCAP_BB_FLUSH=320 MOCAP_BB_FLUSH=512; do
    echo "== $kv: $(env $kv timeout 200 python scripts/time_frame.py 100000 7 2>&1 | tail -1 | cut -c50-130)"
  done
done
