#!/bin/bash
# A/B timing of library variants on one box: gpu_ab_r03.sh "tag[:K_max]" ...  (lib/libmocap_core_<tag>.so; tag "base" = the product build)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python scripts/time_frame.py 100000 2 > /dev/null 2>&1   # stream cache + page-in
for i in 1 2; do
  for spec in "$@"; do
    v=${spec%%:*}; K=48; [ "$v" != "$spec" ] && K=${spec#*:}
    [ $v = base ] && unset MOCAP_CORE_LIB || export MOCAP_CORE_LIB=$R/low-cost-mocap_amd/lib/libmocap_core_$v.so
    echo "== $v K=$K: $(timeout 120 python scripts/time_frame.py 100000 7 $K 2>&1 | tail -1 | cut -c1-140)"
  done
done 2>&1 | tee $O/ab_$(date +%H%M%S).log
