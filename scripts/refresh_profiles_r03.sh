#!/bin/bash
# Local orchestration (needs gpurun): re-take the frame kernel's counter summaries and the bench line on the CURRENT
# sources and copy them into profiles/ -- to be run after any change under low-cost-mocap_amd/csrc, include/ or the
# Makefile (bench.py marks figures derived from older summaries `stale`).
set -e
cd "$(dirname "$0")/.."
HEAD=$(git rev-parse --short HEAD)
gpurun --timeout 1800 -- "bash scripts/profile_frame_pmc.sh $HEAD r03 2>&1 | tail -3"
G=gpurun_out/r03; P=profiles
cp $G/prof/bench_kernel_stats.csv $P/r03_kernel_stats.csv
cp $G/prof/frame_mix_pmc.csv $P/r03_pmc_frame_kernel_fp64_mix.csv
cp $G/prof/frame_issue_pmc.csv $P/r03_pmc_frame_kernel_issue_mix.csv
cat $G/prof/frame_fetch_pmc.csv $G/prof/frame_write_pmc.csv > $P/r03_pmc_frame_kernel_hbm.csv
cp $G/prof/r03_fp64_mix.json $G/prof/r03_hbm_traffic.json $P/
git add -A profiles && git commit -qm "profiles: counters on the current sources" || true
gpurun --timeout 1500 -- 'mkdir -p gpurun_out/r03; timeout 900 python bench.py > gpurun_out/r03/bench_final.log 2>&1; grep "^{\"metric\"" gpurun_out/r03/bench_final.log > gpurun_out/r03/bench_line_final.json'
cp $G/bench_line_final.json $P/r03_bench_line.json
git add -A profiles && git commit -qm "profiles: bench line of the current tree" || true
