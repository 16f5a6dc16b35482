#!/bin/bash
# round 6, call J2: the wide variant with 2-bit hit codes, 512 lanes per frame / two frames per CU by default: timing vs the
# 1 024-lane plan, then every wide / stress / multirank test
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python scripts/time_wide.py 12500 1 > /dev/null 2>&1
for i in 1 2; do
echo "T1024 K384: $(MOCAP_WIDE_THREADS=1024 python scripts/time_wide.py 12500 5 2>&1 | tail -1 | cut -c1-150)"
echo "T512  K384: $(python scripts/time_wide.py 12500 5 2>&1 | tail -1 | cut -c1-150)"
done
timeout 900 python -m pytest tests/test_gpu_wide_adversarial.py tests/test_gpu_bench_scale.py tests/test_gpu_track.py tests/test_gpu_multirank.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r06j_pytest.log 2>&1; tail -3 gpurun_out/r06j_pytest.log
timeout 400 python bench.py --workload 64x256 --frames 12500 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep '^{"metric"' | python -c "import json,sys; l=json.loads(sys.stdin.read()); c=l['config']; print(l['ms_per_step'], c['frames_per_s'], c['overflow_frames'], l['roofline']['kernel'], l['parity'])"
