#!/bin/bash
# round 6: sub-batches per step of the exchange code path on one GPU (MOCAP_BENCH_EXCHANGE=1), headline workload and stress shape
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r06
L=gpurun_out/r06/chunks_$(date +%H%M%S).log
for rep in 1 2; do
for c in 1 2 4; do
  echo "8x16 chunks=$c steps=20: $(MOCAP_BENCH_EXCHANGE=1 timeout 300 python bench.py --chunks $c --steps 20 --warmup 3 --no-cpu-baseline --no-full-parity 2>/dev/null | grep '^{"metric"' | python -c "import json,sys; l=json.loads(sys.stdin.read()); print(l['ms_per_step'], l['roofline']['kernel_ms'], l['config']['exchange']['exposed_ms'])")" | tee -a $L
done
echo "8x16 no exchange: $(timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-full-parity 2>/dev/null | grep '^{"metric"' | python -c "import json,sys; l=json.loads(sys.stdin.read()); print(l['ms_per_step'], l['roofline']['kernel_ms'])")" | tee -a $L
done
for c in 1 2; do
  echo "64x256 chunks=$c steps=3: $(MOCAP_BENCH_EXCHANGE=1 timeout 300 python bench.py --workload 64x256 --frames 12500 --chunks $c --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "import json,sys; l=json.loads(sys.stdin.read()); print(l['ms_per_step'], l['roofline']['kernel_ms'], l['config']['exchange']['exposed_ms'])")" | tee -a $L
done
