#!/bin/bash
# round 5, first GPU call: the whole -m gpu suite, headline-kernel A/B (base / 3 waves per SIMD / with the re-submit path),
# the default bench line and the stress shape.  Everything under gpurun_out/r05a/.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
for v in "" _w3; do
  for auto in 0 1; do
    MOCAP_CORE_LIB=low-cost-mocap_amd/lib/libmocap_core$v.so TF_AUTO=$auto timeout 300 python scripts/time_frame.py 100000 7 >> $O/time_frame.log 2>&1
  done
done
cat $O/time_frame.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; head -c 1500 $O/bench.json; tail -3 $O/bench.err
timeout 600 python bench.py --workload 64x256 --frames 12500 --steps 3 --warmup 1 > $O/bench_64x256.json 2> $O/bench_64x256.err; echo "bench64 rc $?"; head -c 1500 $O/bench_64x256.json; tail -3 $O/bench_64x256.err
