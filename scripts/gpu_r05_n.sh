#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
bash scripts/gpu_wide_ab.sh 12500 base ns 2>&1 | tail -4 | cut -c1-150
timeout 900 python -m pytest tests/test_gpu_wide_adversarial.py tests/test_gpu_multirank.py tests/test_gpu_parity.py -m gpu -q --timeout 600 -x 2>&1 | tail -15 | cut -c1-300
