// frame_kernel_wide.hip -- the wide variant's instantiations of csrc/frame_kernel.hip (64 cameras x 256 blobs: BASELINE
// configs[4]; helpers.py:339-421 for frames whose state exceeds LDS) as a translation unit of their own, so that the build can
// give this one very large kernel body its own scheduler / register-allocator options (Makefile: FRAME_WIDE_FLAGS) without
// touching the small-frame kernels.  Everything is in frame_kernel.hip; this file only selects which half is emitted.
#define MOCAP_FRAME_TU_WIDE 1
// Issue priority of the phases (s_setprio, see csrc/frame_bb.hip): the chain over the cameras with its single-wave stretches and
// the output above the camera-0 pass and the candidate evaluation of the CU's other frame: 24.1 -> 23.8 ms per 12 500 stress frames.
#ifndef MOCAP_FRAME_PRIO_CHAIN
#define MOCAP_FRAME_PRIO_CHAIN 1
#endif
#ifndef MOCAP_FRAME_PRIO_OUT
#define MOCAP_FRAME_PRIO_OUT 1
#endif
// (the wide variant is compiled without machine LICM instead of taking the lane's number afresh per phase: Makefile, FRAME_WIDE_FLAGS)
#ifndef MOCAP_FRAME_FRESH_TID
#define MOCAP_FRAME_FRESH_TID 0
#endif
#include "frame_kernel.hip"
