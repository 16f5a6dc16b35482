#!/bin/bash
# Variants of the general frame kernel (csrc/frame_kernel.hip: the wide variant among others) for A/B timing on the GPU box:
# lib/libmocap_core_<tag>.so = the product library's objects with frame_kernel.hip compiled with the given flags.
#   usage: scripts/build_frame_variants.sh "skip1=-DMOCAP_WIDE_DEBUG_SKIP=1" "b4=-DMOCAP_WIDE_BATCH=4"
set -e
cd "$(dirname "$0")/../low-cost-mocap_amd"
make -j8 lib/libmocap_core.so >/dev/null
for spec in "$@"; do
  tag=${spec%%=*}; flags=${spec#*=}
  # (both translation units of the frame kernel: the small-frame kernels, and the wide variant with the product's FRAME_WIDE_FLAGS
  # unless the variant says NOWIDEFLAGS=1)
  WF="-mllvm -amdgpu-schedule-relaxed-occupancy=true -mllvm -greedy-regclass-priority-trumps-globalness=1 -mllvm -enable-post-misched=false -mllvm -disable-machine-licm"
  [ "${NOWIDEFLAGS:-0}" = 1 ] && WF=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $flags -c csrc/frame_kernel.hip -o build/frame_kernel_v_$tag.o &
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $WF $flags -c csrc/frame_kernel_wide.hip -o build/frame_kernel_wide_v_$tag.o &
done
wait
for spec in "$@"; do
  tag=${spec%%=*}
  objs=$(ls build/*.o | grep -v "frame_kernel" | grep -v "eigcheck\|frame_bb_" | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libmocap_core_$tag.so $objs build/frame_kernel_v_$tag.o build/frame_kernel_wide_v_$tag.o
  echo "built lib/libmocap_core_$tag.so ($spec)"
done
