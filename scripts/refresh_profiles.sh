#!/bin/bash
# Local orchestration (needs gpurun): re-take the headline kernel's counter summaries and the bench lines on the CURRENT
# sources and copy them into profiles/ under the round's tag -- to be run after any change under low-cost-mocap_amd/csrc,
# include/ or the Makefile (bench.py marks figures derived from older summaries `stale`).   usage: refresh_profiles.sh r04
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r04}
HEAD=$(git rev-parse --short HEAD)
gpurun --timeout 1200 -- "bash scripts/profile_frame_pmc.sh $HEAD $TAG 2>&1 | tail -3"
G=gpurun_out/$TAG; P=profiles
cp $G/prof/bench_kernel_stats.csv $P/${TAG}_kernel_stats.csv
cp $G/prof/frame_mix_pmc.csv $P/${TAG}_pmc_frame_kernel_fp64_mix.csv
cp $G/prof/frame_issue_pmc.csv $P/${TAG}_pmc_frame_kernel_issue_mix.csv
cat $G/prof/frame_fetch_pmc.csv $G/prof/frame_write_pmc.csv > $P/${TAG}_pmc_frame_kernel_hbm.csv
cp $G/prof/${TAG}_fp64_mix.json $G/prof/${TAG}_hbm_traffic.json $P/
git add -A profiles && git commit -qm "profiles: $TAG counters of the headline kernel on the current sources" || true
gpurun --timeout 1500 -- "mkdir -p gpurun_out/$TAG; timeout 900 python bench.py > gpurun_out/$TAG/bench_final.log 2>&1; grep '^{\"metric\"' gpurun_out/$TAG/bench_final.log > gpurun_out/$TAG/bench_line_final.json; timeout 300 python bench.py --workload 64x256 --frames 12500 --steps 3 --warmup 1 > gpurun_out/$TAG/bench_64x256.log 2>&1; grep '^{\"metric\"' gpurun_out/$TAG/bench_64x256.log > gpurun_out/$TAG/bench_line_64x256.json; MOCAP_BENCH_EXCHANGE=1 timeout 300 python bench.py --workload 64x256 --frames 12500 --steps 3 --warmup 1 > gpurun_out/$TAG/bench_64x256_exchange.log 2>&1; grep '^{\"metric\"' gpurun_out/$TAG/bench_64x256_exchange.log > gpurun_out/$TAG/bench_line_64x256_exchange.json; tail -c 600 gpurun_out/$TAG/bench_final.log"
cp $G/bench_line_final.json $P/${TAG}_bench_line.json
cp $G/bench_line_64x256.json $P/${TAG}_bench_line_64x256_12500frames.json
cp $G/bench_line_64x256_exchange.json $P/${TAG}_bench_line_64x256_exchange_1gpu.json
git add -A profiles && git commit -qm "profiles: $TAG bench lines of the current tree" || true
