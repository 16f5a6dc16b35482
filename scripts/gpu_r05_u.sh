#!/bin/bash
# round 5, last call on the final tree: the whole GPU suite, smoke(), the driver's own bench command, and a kernel trace of
# the blob stage (with / without the dark-tile early-out) for DESIGN 7.4.  Output: gpurun_out/r05u/
set -u
O=gpurun_out/r05u
mkdir -p $O
export TMPDIR=/tmp
( timeout 420 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15 ) > $O/suite.txt
( timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) > $O/smoke.txt
( timeout 300 python bench.py 2>$O/bench.err | tail -1 ) > $O/bench_line.json
R=$PWD
for v in skip noskip; do
  f=""; [ $v = noskip ] && f="--no-skip"
  ( cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d $R/$O/blob_$v -o blob -- python $R/scripts/bench_blobs.py --frames 1024 --steps 5 $f ) > $O/blob_$v.log 2>&1
done
ls -R $O | head -40
tail -3 $O/suite.txt; cat $O/smoke.txt; cut -c1-300 $O/bench_line.json
