// frame_kernel_wide.hip -- the wide variant's instantiations of csrc/frame_kernel.hip (64 cameras x 256 blobs: BASELINE
// configs[4]; helpers.py:339-421 for frames whose state exceeds LDS) as a translation unit of their own, so that the build can
// give this one very large kernel body its own scheduler / register-allocator options (Makefile: FRAME_WIDE_FLAGS) without
// touching the small-frame kernels.  Everything is in frame_kernel.hip; this file only selects which half is emitted.
#define MOCAP_FRAME_TU_WIDE 1
#include "frame_kernel.hip"
