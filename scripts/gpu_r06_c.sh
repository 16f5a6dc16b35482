#!/bin/bash
# round 6, call C: the re-submit tests again, wide-kernel A/B (scalar-cache pre-test), BA default mode with / without the sub-records
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06c; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_wide_adversarial.py tests/test_gpu_track.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
bash scripts/gpu_wide_ab.sh 12500 base sload 2>&1 | grep "^=="
for v in noconf full; do
  [ $v = noconf ] && X="--no-configs --no-full-parity" || X=""
  timeout 600 python bench.py --no-cpu-baseline --no-blobs --no-latency $X > $O/bench_$v.log 2>&1
  grep '^{"metric"' $O/bench_$v.log | python -c "import json,sys; l=json.loads(sys.stdin.read()); d=l['ba']['default_mode']; print('$v', l['value'], d['wall_s'], d['inside_core_calls_s'], d['runs_s'], l['ba']['value']); c=l.get('configs',{}).get('64x256'); print(c and {k:c.get(k) for k in ('ms_per_step','frames_per_s','overflow_frames','flagged_by_first_pass','bounded_resubmit','error')})"
done
