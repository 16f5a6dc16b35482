set -u
O=gpurun_out/r03; mkdir -p $O
python scripts/time_frame.py 100000 5 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bb_adversarial.py -q -m gpu -x > $O/pytest_bb4.log 2>&1; tail -5 $O/pytest_bb4.log
