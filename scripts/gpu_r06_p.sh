#!/bin/bash
# round 6, call P: A/B timing of library variants at the stress shape, then the wide tests on ONE of them ($1)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06p; mkdir -p $O
cd $R
T=$1; shift
bash scripts/gpu_wide_ab.sh 12500 base $T "$@" 2>&1 | grep "^==" | cut -c1-150
MOCAP_CORE_LIB=$R/low-cost-mocap_amd/lib/libmocap_core_$T.so timeout 1200 python -m pytest tests/test_gpu_wide_adversarial.py tests/test_gpu_bench_scale.py tests/test_gpu_parity.py tests/test_gpu_multirank.py -m gpu -x -q -k "not self_check and not pretest" > $O/pytest.log 2>&1; echo "pytest($T) rc=$?"; tail -3 $O/pytest.log
