set -u
O=gpurun_out/r03; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu_full.log 2>&1; tail -3 $O/pytest_gpu_full.log
bash scripts/profile_frame_pmc.sh efee7f2 r03 > $O/profile_frame.log 2>&1; tail -3 $O/profile_frame.log
bash scripts/profile_r03.sh > $O/profile_cfg.log 2>&1; tail -6 $O/profile_cfg.log
for v in "" "--no-skip"; do (cd /tmp && TMPDIR=/tmp timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES -d $GRAFT_REPO_ROOT/$O/blobpmc$v -o p -- python $GRAFT_REPO_ROOT/scripts/bench_blobs.py --frames 1024 --steps 2 $v > $GRAFT_REPO_ROOT/$O/blobpmc$v.log 2>&1); DB=$(find $O/blobpmc$v -name "*.db" | head -1); { echo "# bench_blobs.py --frames 1024 --steps 2 $v"; python scripts/rocpd_summary.py pmc $DB | grep -v "rocclr\|at::native\|rocprim"; python scripts/rocpd_summary.py stats $DB | grep -v "rocclr\|at::native\|rocprim"; } > $O/blob_pmc_sq$v.csv; find $O/blobpmc$v -name "*.db" -delete; done
