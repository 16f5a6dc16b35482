#!/bin/bash
# round 5: where the headline kernel's time goes INSIDE the candidate evaluation -- parts run twice (same results), the
# difference to the shipped kernel is the part's cost in the mix.  gpurun_out/r05c/.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r05c; mkdir -p $O
for v in "" _d1 _d2 _d4 _dr ""; do
  MOCAP_CORE_LIB=low-cost-mocap_amd/lib/libmocap_core$v.so timeout 300 python scripts/time_frame.py 100000 9 >> $O/time_frame.log 2>&1
done
grep -v amdgpu.ids $O/time_frame.log
