#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r05d; mkdir -p $O; rm -f $O/time_frame.log
for v in "" _prev "" _prev; do
  MOCAP_CORE_LIB=low-cost-mocap_amd/lib/libmocap_core$v.so timeout 300 python scripts/time_frame.py 100000 9 >> $O/time_frame.log 2>&1
done
grep -v amdgpu.ids $O/time_frame.log | sed 's/frame_bb_kernel<CW=1> //; s/cands.frame [0-9.]* //'
timeout 900 python -m pytest tests/test_gpu_bb_adversarial.py tests/test_gpu_parity.py tests/test_gpu_boundary.py -m gpu -q --timeout 300 -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
