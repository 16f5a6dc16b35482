#!/bin/bash
# round 6, call U: the caller's first-pass cap on candidate groups per root (MOCAP_BENCH_G_CAP) at the stress shape: what the whole
# hot-path call costs when more roots go to the re-submit's heavy-root search / enumeration instead of being enumerated in the first pass
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for cap in 1048576 131072 32768 16384; do
  echo "== G_cap $cap: $(MOCAP_BENCH_G_CAP=$cap timeout 400 python bench.py --workload 64x256 --frames 12500 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "import json,sys; l=json.loads(sys.stdin.read()); c=l['config']; print(round(l['ms_per_step'],2), 'ms', round(c['frames_per_s']), 'frames/s overflow', c['overflow_frames'], 'flagged', c['flagged_by_first_pass'], l['parity'].get('prefix_vs_oracle', l['parity']) if isinstance(l.get('parity'), dict) else None)" | cut -c1-300)"
done
