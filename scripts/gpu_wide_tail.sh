#!/bin/bash
# wide variant: where the time goes at 1 024 (typical frames) and 12 500 frames (with the heavy tail), geometry on / off,
# heavy-frame slicing thresholds
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
L=$R/low-cost-mocap_amd/lib
run() { echo "== $1: $(timeout 300 python scripts/time_wide.py $2 3 2>&1 | tail -1 | cut -c1-150)"; }
python scripts/time_wide.py 1024 1 > /dev/null 2>&1; python scripts/time_wide.py 12500 1 > /dev/null 2>&1
run base 1024; run base 12500
MOCAP_CORE_LIB=$L/libmocap_core_skip4.so run skip4 1024; MOCAP_CORE_LIB=$L/libmocap_core_skip4.so run skip4 12500
MOCAP_CORE_LIB=$L/libmocap_core_skip2.so run skip2 1024
for h in 0 8192 16384 65536 262144; do MOCAP_HEAVY_THRESHOLD=$h run heavy$h 12500; done
for sl in 16384 65536; do MOCAP_HEAVY_THRESHOLD=32768 MOCAP_SLICE_SIZE=$sl run slice$sl 12500; done
