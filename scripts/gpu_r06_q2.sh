#!/bin/bash
# round 6, last session: the second pass's own work queues (no queue fills per call) -- kernel trace of the default workload,
# bench lines without / with the exchange path, then the whole GPU suite
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06/q2; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/p -o p -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-full-parity --no-ba --no-blobs --no-latency --no-configs > $O/trace_8x16.log 2>&1
DB=$(find $O/p -name "*.db" | head -1); python $R/scripts/rocpd_summary.py stats $DB > $O/kernel_stats_8x16.csv; find $O/p -name "*.db" -delete
head -9 $O/kernel_stats_8x16.csv | cut -c1-140
cd $R
for i in 1 2; do
echo "8x16: $(timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ba --no-blobs --no-latency --no-configs 2>/dev/null | grep '^{"metric"' | python -c "import json,sys; l=json.loads(sys.stdin.read()); print(l['ms_per_step'], l['roofline']['kernel_ms'], l['parity']['full_batch_vs_exhaustive_bit_exact'])")"
echo "8x16 exchange: $(MOCAP_BENCH_EXCHANGE=1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ba --no-blobs --no-latency --no-configs 2>/dev/null | grep '^{"metric"' | python -c "import json,sys; l=json.loads(sys.stdin.read()); print(l['ms_per_step'], l['roofline']['kernel_ms'], l['parity']['full_batch_vs_exhaustive_bit_exact'])")"
done
echo "64x256: $(timeout 300 python bench.py --workload 64x256 --frames 12500 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "import json,sys; l=json.loads(sys.stdin.read()); print(l['ms_per_step'], l['config']['overflow_frames'])")"
echo "4x4: $(timeout 300 python bench.py --workload 4x4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "import json,sys; l=json.loads(sys.stdin.read()); print(l['ms_per_step'])")"
timeout 1200 python -m pytest tests -m gpu -q > $O/gpu_suite.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/gpu_suite.txt | tail -3
