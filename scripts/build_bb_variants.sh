#!/bin/bash
# Variants of the branch-and-bound kernel for A/B timing on the GPU box: lib/libmocap_core_<tag>.so, same objects as the
# product library except csrc/frame_bb.hip compiled with the given flags.  usage: build_bb_variants.sh tag=flags ...
# (the product's FRAME_BB_FLAGS unless BBFLAGS says otherwise, BBFLAGS= for none)
#   e.g. scripts/build_bb_variants.sh "skip1=-DMOCAP_BB_DEBUG_SKIP=1" "w5=-DMOCAP_BB_WAVES_PER_EU=5"
set -e
cd "$(dirname "$0")/../low-cost-mocap_amd"
make -j8 lib/libmocap_core.so >/dev/null
for spec in "$@"; do
  tag=${spec%%=*}; flags=${spec#*=}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function ${BBFLAGS--mllvm -disable-machine-licm} $flags -c csrc/frame_bb.hip -o build/frame_bb_$tag.o &
done
wait
for spec in "$@"; do
  tag=${spec%%=*}
  objs=$(ls build/*.o | grep -v "frame_bb" | grep -v "_v_\|pretest" | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libmocap_core_$tag.so $objs build/frame_bb_$tag.o
  echo "built lib/libmocap_core_$tag.so ($spec)"
done
