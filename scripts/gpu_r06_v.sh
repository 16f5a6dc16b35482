#!/bin/bash
# round 6, call V: where the waves of the headline kernel's workgroups land (scripts/probe_wave_placement.hip) and the
# wave-renumbering variants of csrc/frame_bb.hip (MOCAP_BB_ROT) at the bench's 100 k frames of 8 x 16
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
[ "${PROBE:-0}" = 1 ] && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w -o /tmp/probe_wp scripts/probe_wave_placement.hip && /tmp/probe_wp 1024
python scripts/time_frame.py 100000 1 > /dev/null 2>&1
for i in 1 2; do
  for v in base "$@"; do
    [ $v = base ] && unset MOCAP_CORE_LIB || export MOCAP_CORE_LIB=$R/low-cost-mocap_amd/lib/libmocap_core_$v.so
    echo "== $v: $(timeout 200 python scripts/time_frame.py 100000 7 2>&1 | tail -1 | cut -c1-150)"
  done
done
