#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r05i; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_wide_adversarial.py -m gpu -q --timeout 600 -x 2>&1 | tail -3
for ncap in 16384 65536; do
MOCAP_HEAVY_NCAP=$ncap MOCAP_HEAVY_DEBUG=1 timeout 900 python bench.py --workload 64x256 --frames 12500 --steps 2 --warmup 1 > $O/bench_$ncap.json 2> $O/bench.err
grep -h HEAVY $O/bench_$ncap.json $O/bench.err | sort | uniq > $O/heavy_$ncap.txt; echo "ncap $ncap roots $(wc -l < $O/heavy_$ncap.txt) gave up $(grep -c 'give_up 1' $O/heavy_$ncap.txt)"
grep -v HEAVY $O/bench_$ncap.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print('ms_per_step', d['ms_per_step'], 'overflow', c['overflow_frames'], 'frames/s', c['frames_per_s'])"
done
