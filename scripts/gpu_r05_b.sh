#!/bin/bash
# round 5, second GPU call: the whole -m gpu suite on the new headline kernel, A/B of its two changes (correction-free block
# decode, record of zeros for absent cameras), block-size sweep, the default bench line.  Everything under gpurun_out/r05b/.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r05b; mkdir -p $O
for v in "" _old _td _zs "" _old; do
  MOCAP_CORE_LIB=low-cost-mocap_amd/lib/libmocap_core$v.so timeout 300 python scripts/time_frame.py 100000 9 >> $O/time_frame.log 2>&1
done
for pl in 12 20 24; do
  MOCAP_BB_PL=$pl timeout 300 python scripts/time_frame.py 100000 7 >> $O/time_frame.log 2>&1
done
grep -v amdgpu.ids $O/time_frame.log
timeout 1200 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; head -c 600 $O/bench.json; tail -3 $O/bench.err
