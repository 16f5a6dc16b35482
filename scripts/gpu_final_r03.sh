set -u
O=gpurun_out/r03; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu_full.log 2>&1; tail -6 $O/pytest_gpu_full.log
timeout 900 python bench.py > $O/bench_full.log 2>&1; grep '^{"metric"' $O/bench_full.log > $O/bench_line.json; cut -c1-600 $O/bench_line.json
bash scripts/profile_frame_pmc.sh 881e3b8 r03 > $O/profile_frame.log 2>&1; tail -3 $O/profile_frame.log
bash scripts/profile_r03.sh > $O/profile_cfg.log 2>&1; tail -14 $O/profile_cfg.log
