set -u
O=gpurun_out/r03; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu_full.log 2>&1; tail -6 $O/pytest_gpu_full.log
timeout 900 python bench.py > $O/bench_full.log 2>&1; tail -c 3000 $O/bench_full.log
