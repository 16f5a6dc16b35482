#!/bin/bash
# round 5: HBM traffic counters on the final tree for the two figures DESIGN quotes "as in round 4 / round 2":
# the wide first pass at 12 500 frames of 64 x 256 and the blob stage.  FETCH_SIZE and WRITE_SIZE in separate passes.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05v; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
W="python $R/bench.py --workload 64x256 --frames 12500 --steps 2 --warmup 1 --no-cpu-baseline --no-ba --no-blobs --no-latency"
B="python $R/scripts/bench_blobs.py --frames 1024 --steps 2"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O/w_$c -o p -- $W > $O/w_$c.log 2>&1
  python $R/scripts/rocpd_summary.py pmc $(find $O/w_$c -name "*.db" | head -1) | grep -E "^kernel|mocap::" > $O/wide_$c.csv
  find $O/w_$c -name "*.db" -delete
  timeout 120 rocprofv3 --kernel-trace --pmc $c -d $O/b_$c -o p -- $B > $O/b_$c.log 2>&1
  python $R/scripts/rocpd_summary.py pmc $(find $O/b_$c -name "*.db" | head -1) | grep -E "^kernel|mocap::" > $O/blob_$c.csv
  find $O/b_$c -name "*.db" -delete
done
cut -c1-200 $O/*.csv
