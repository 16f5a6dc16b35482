#!/bin/bash
# round 6, call M: wide pre-test with keys (MOCAP_WIDE_ACC=3: v_pk_fma + v_and_or + v_min + v_med3 per blob, no compare / branch):
# A/B timing of the whole first pass and of the camera-0 pass alone (SKIP=6), then the wide tests on the variant
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06m; mkdir -p $O
cd $R
bash scripts/gpu_wide_ab.sh 12500 base acc3 skip6 skip6acc3 2>&1 | grep "^==" | cut -c1-140
MOCAP_CORE_LIB=$R/low-cost-mocap_amd/lib/libmocap_core_acc3.so timeout 900 python -m pytest tests/test_gpu_wide_adversarial.py tests/test_gpu_parity.py -m gpu -x -q -k "not self_check and not pretest" > $O/pytest.log 2>&1; echo "pytest(acc3) rc=$?"; tail -3 $O/pytest.log
