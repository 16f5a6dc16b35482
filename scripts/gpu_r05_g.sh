#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r05g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wide_adversarial.py -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -30 $O/pytest.log | cut -c1-220
