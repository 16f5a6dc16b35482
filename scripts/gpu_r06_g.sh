#!/bin/bash
# round 6, call G: wide pre-test without the per-step branch (MOCAP_WIDE_ACC=2): A/B timing + the wide tests on the variant
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06g; mkdir -p $O
cd $R
bash scripts/gpu_wide_ab.sh 12500 base acc2 2>&1 | grep "^==" | cut -c1-140
MOCAP_CORE_LIB=$R/low-cost-mocap_amd/lib/libmocap_core_acc2.so timeout 600 python -m pytest tests/test_gpu_wide_adversarial.py tests/test_gpu_parity.py -m gpu -x -q -k "not self_check and not pretest" > $O/pytest.log 2>&1; echo "pytest(acc2) rc=$?"; tail -3 $O/pytest.log
