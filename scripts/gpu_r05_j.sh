#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r05j; mkdir -p $O
for g in 16384 4096; do
MOCAP_BENCH_G_CAP=$g timeout 900 python bench.py --workload 64x256 --frames 12500 --steps 3 --warmup 1 > $O/bench_$g.json 2> $O/bench.err
grep -o '{"metric".*' $O/bench_$g.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print('G_cap $g ms_per_step', d['ms_per_step'], 'overflow', c['overflow_frames'], 'flagged', c['flagged_by_first_pass'], 'frames/s', c['frames_per_s'], d['parity'])"
done
