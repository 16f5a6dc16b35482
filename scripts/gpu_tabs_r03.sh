#!/bin/bash
# One box: the search kernel with its LDS layout fixed at compile time (8 x 16, 48 / 64 root slots) against the runtime
# layout (MOCAP_BB_FIXED_LAYOUT=0), then the parity suites of the search kernel on the product build.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python scripts/time_frame.py 100000 2 > /dev/null 2>&1   # stream cache + page-in
t() { echo "== K=$1 fixed=${MOCAP_BB_FIXED_LAYOUT:-1}: $(timeout 120 python scripts/time_frame.py 100000 7 $1 2>&1 | tail -1 | cut -c1-150)"; }
{
  for i in 1 2; do
    t 48; t 64
    export MOCAP_BB_FIXED_LAYOUT=0; t 48; t 64; unset MOCAP_BB_FIXED_LAYOUT
  done
} 2>&1 | tee $O/layout_ab.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bb_adversarial.py -x -q 2>&1 | tail -5 | tee $O/layout_parity.log
