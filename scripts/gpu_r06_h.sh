#!/bin/bash
# round 6, call H: first-pass candidate cap at the stress shape now that the re-submit enumerates fast (12 500 frames)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06h; mkdir -p $O
cd $R
for G in 1024 4096 16384 65536 1048576; do
  MOCAP_BENCH_G_CAP=$G timeout 400 python bench.py --workload 64x256 --frames 12500 --steps 3 --warmup 1 --no-cpu-baseline > $O/g_$G.log 2>&1
  grep '^{"metric"' $O/g_$G.log | python -c "import json,sys; l=json.loads(sys.stdin.read()); c=l['config']; print('G_cap $G', round(l['ms_per_step'],2), c['overflow_frames'], c['flagged_by_first_pass'], c['overflow_by_cap']['intractable_roots_over_2^24_groups'], l['parity']['corr_bit_exact'])"
done
