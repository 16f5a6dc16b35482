set -u
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bb_adversarial.py -x -q -m gpu > $O/pytest_bb.log 2>&1; tail -15 $O/pytest_bb.log
python scripts/time_frame.py 100000 5 > $O/time_frame_bb.log 2>&1; tail -2 $O/time_frame_bb.log
MOCAP_EVAL_BB=0 python scripts/time_frame.py 100000 3 > $O/time_frame_exh.log 2>&1; tail -1 $O/time_frame_exh.log
timeout 600 python -m pytest tests/test_gpu_ba.py tests/test_gpu_boundary.py -x -q -m gpu > $O/pytest_ba.log 2>&1; tail -8 $O/pytest_ba.log
