set -u
O=gpurun_out/r03; mkdir -p $O
L=low-cost-mocap_amd/lib
python scripts/time_frame.py 100000 5 2>&1 | tail -1
for v in nospec w5 skip1 skip3 skip7 skip15; do MOCAP_CORE_LIB=$L/libmocap_core_$v.so python scripts/time_frame.py 100000 5 2>&1 | tail -1 | sed "s/^/$v: /"; done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bb_adversarial.py -q -m gpu > $O/pytest_bb2.log 2>&1; tail -12 $O/pytest_bb2.log
