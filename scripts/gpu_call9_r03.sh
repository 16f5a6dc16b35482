set -u
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bb_adversarial.py -q -m gpu -x > $O/pytest_w.log 2>&1; tail -4 $O/pytest_w.log
python scripts/time_wide.py 1024 3 2>&1 | tail -1
timeout 900 python bench.py --workload 64x256 --frames 12500 --steps 2 --warmup 1 2>&1 | grep "^{" > $O/bench_64x256_12500.json; cut -c1-1200 $O/bench_64x256_12500.json
