set -u
mkdir -p gpurun_out/r03
python scripts/diag_ba_stall.py 16000 20 > gpurun_out/r03/ba_stall_16k.log 2> gpurun_out/r03/ba_stall_16k.err
python scripts/diag_ba_stall.py 1000 12 > gpurun_out/r03/ba_stall_1k.log 2> gpurun_out/r03/ba_stall_1k.err
MOCAP_BA_UNFUSED=1 python scripts/diag_ba_stall.py 16000 8 > gpurun_out/r03/ba_stall_16k_unfused.log 2> gpurun_out/r03/ba_stall_16k_unfused.err
python scripts/time_frame.py 100000 5 > gpurun_out/r03/time_frame_head.log 2>&1
bash scripts/profile_r03_wide.sh 1024 > gpurun_out/r03/wide_profile.log 2>&1
cat gpurun_out/r03/ba_stall_16k.log; tail -25 gpurun_out/r03/ba_stall_16k.err; cat gpurun_out/r03/time_frame_head.log
