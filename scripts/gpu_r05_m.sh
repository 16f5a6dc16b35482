#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r05m; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_wide_adversarial.py -m gpu -q --timeout 600 -x 2>&1 | tail -2
timeout 900 python bench.py --workload 64x256 --frames 12500 --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
grep -o '{"metric".*' $O/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print('ms_per_step', d['ms_per_step'], 'overflow', c['overflow_frames'], 'flagged', c['flagged_by_first_pass'], 'frames/s', c['frames_per_s'], c['first_pass_G_cap'])"
