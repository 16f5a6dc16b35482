#!/bin/bash
# Local orchestration (needs gpurun): re-take the round's tracked profiles on the CURRENT sources in ONE GPU call
# (scripts/gpu_refresh_r06.sh) and copy the summaries into profiles/ under the round's tag -- to be run after any change under
# low-cost-mocap_amd/csrc, include/ or the Makefile (bench.py marks figures derived from older summaries `stale`).
#   usage: refresh_profiles.sh        (tag r06)
set -e
cd "$(dirname "$0")/.."
TAG=r06
HEAD=$(git rev-parse --short HEAD)
[ "${SKIP_GPU:-0}" = 1 ] || gpurun --timeout 3600 -- "bash scripts/gpu_refresh_r06.sh $HEAD"
G=gpurun_out/$TAG; P=profiles
cp $G/prof/bench_kernel_stats.csv $P/${TAG}_kernel_stats.csv
cp $G/prof/frame_mix_pmc.csv $P/${TAG}_pmc_frame_kernel_fp64_mix.csv
cp $G/prof/frame_issue_pmc.csv $P/${TAG}_pmc_frame_kernel_issue_mix.csv
cat $G/prof/frame_fetch_pmc.csv $G/prof/frame_write_pmc.csv > $P/${TAG}_pmc_frame_kernel_hbm.csv
cp $G/prof/${TAG}_fp64_mix.json $G/prof/${TAG}_hbm_traffic.json $P/
cp $G/cfgpmc/${TAG}_fp64_mix_4x4.json $G/cfgpmc/${TAG}_fp64_mix_64x256.json $P/
cp $G/cfgpmc/64x256_kernel_stats.csv $P/${TAG}_kernel_stats_64x256_12500frames.csv
cp $G/cfgpmc/4x4_kernel_stats.csv $P/${TAG}_kernel_stats_4x4.csv
cat $G/cfgpmc/64x256_fetch_pmc.csv $G/cfgpmc/64x256_write_pmc.csv > $P/${TAG}_wide_pmc_traffic_12500frames.csv
cat $G/cfgpmc/64x256_mix_pmc.csv $G/cfgpmc/64x256_issue_pmc.csv > $P/${TAG}_wide_pmc_instruction_mix_12500frames.csv
cp $G/cfgpmc/4x4_bench_line.json $P/${TAG}_bench_line_4x4.json
cp $G/final/bench_line_final.json $P/${TAG}_bench_line.json
cp $G/final/bench_line_64x256.json $P/${TAG}_bench_line_64x256_12500frames.json
cp $G/final/bench_line_64x256_exchange.json $P/${TAG}_bench_line_64x256_exchange_1gpu.json
cp $G/final/bench_line_8x16_exchange.json $P/${TAG}_bench_line_8x16_exchange_1gpu.json
cp $G/final/blob_kernel_stats_skip.csv $P/${TAG}_blob_kernel_stats_prepass.csv
cp $G/final/blob_kernel_stats_fold.csv $P/${TAG}_blob_kernel_stats_folded.csv
cp $G/final/blob_pmc_traffic_skip.csv $P/${TAG}_blob_pmc_traffic_prepass.csv
cp $G/final/blob_pmc_traffic_fold.csv $P/${TAG}_blob_pmc_traffic_folded.csv
grep "passed\|failed" $G/final/gpu_suite.txt | tail -2 > $P/${TAG}_gpu_suite_final.txt
ls $P/${TAG}_*
