#!/bin/bash
# round 6, call O: the wide variant's chain over cameras 2 .. C-1 as one speculative pass (MOCAP_WIDE_SPEC=1, the product build):
# timing, then every wide / stress / multirank / parity test
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06o; mkdir -p $O
cd $R
bash scripts/gpu_wide_ab.sh 12500 base "$@" 2>&1 | grep "^==" | cut -c1-150
timeout 1200 python -m pytest tests/test_gpu_wide_adversarial.py tests/test_gpu_bench_scale.py tests/test_gpu_track.py tests/test_gpu_multirank.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
