# BA stall experiment: which runtime operation inside a solve triggers the one-off 60-90 ms?
set -u
O=gpurun_out/r03; mkdir -p $O
export DIAG_NO_BURST=1
run() { tag=$1; shift; for N in 1000 16000; do env "$@" python scripts/diag_ba_stall.py $N 14 > $O/stall_${tag}_$N.log 2> $O/stall_${tag}_$N.err; grep "cold+warm" $O/stall_${tag}_$N.log; grep "ba_setup" $O/stall_${tag}_$N.err | head -8; done; }
run stage2 MOCAP_BA_STAGE=2
run stage2b MOCAP_BA_STAGE=2
run stage1 MOCAP_BA_STAGE=1
run stage0 MOCAP_BA_STAGE=0
run stage0_nosdma MOCAP_BA_STAGE=0 HSA_ENABLE_SDMA=0
run stage2_noprearm MOCAP_BA_STAGE=2 MOCAP_BA_NO_PREARM=1
run stage0_noprearm MOCAP_BA_STAGE=0 MOCAP_BA_NO_PREARM=1
