set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export DIAG_NO_BURST=1 MOCAP_BA_NO_PREARM=1
for tag in a b c; do
mkdir -p $O/gap_$tag
timeout 60 rocprofv3 --kernel-trace -d $O/gap_$tag -o p -- python $R/scripts/diag_ba_stall.py 1000 14 > $O/gap_$tag.log 2>&1
DB=$(find $O/gap_$tag -name "*.db" | head -1)
grep "cold+warm" $O/gap_$tag.log | cut -c1-300
grep -i "fault" $O/gap_$tag.log | head -3
python $R/scripts/rocpd_gap.py $DB 15 > $O/gap_$tag.txt 2>&1
find $O/gap_$tag -name "*.db" -delete
head -c 3000 $O/gap_$tag.txt
done
