#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r05e; mkdir -p $O
timeout 300 python - > $O/tr_device_bench.txt 2>&1 <<'PY'
import sys; sys.path.insert(0, "low-cost-mocap_amd")
from mocap_core import capi
core = capi.MocapCore()
for reps in (8, 200, 1000):
    print(reps, core.tr_device_bench(reps=reps))
PY
cat $O/tr_device_bench.txt | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_gpu_ba.py -m gpu -q --timeout 300 -x 2>&1 | tail -3
