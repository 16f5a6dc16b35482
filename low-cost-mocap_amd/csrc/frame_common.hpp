// frame_common.hpp -- small device helpers shared by the two frame-path kernels (frame_kernel.hip: the general case;
// frame_bb.hip: identical plain intrinsics with the exact branch-and-bound selection).
#pragma once
#include "mocap_device.hpp"
#include "kernels.hpp"

namespace mocap {

constexpr uint16_t kNone = 0xFFFF;

// Exact quotient/remainder for rem < 2^24, 1 <= n <= 2^16: float(rem) is exact and the float
// quotient (1-ulp v_rcp_f32, one rounded multiply, truncation) is within [-2, +1] of the true one
// (|error| <= 3/n, exact for n = 1, 2), so two correction steps per direction make it exact at about
// half the instructions of a 32-bit integer division.  Candidate indices per root are < 2^24 (G_cap).
__device__ __forceinline__ void divmod_small(uint32_t rem, uint32_t n, uint32_t& q, uint32_t& r) {
  const float inv = __builtin_amdgcn_rcpf((float)n);
  q = (uint32_t)((float)rem * inv);
  int32_t rr = (int32_t)(rem - q * n);
#pragma unroll
  for (int k = 0; k < 2; k++) {
    if (rr < 0) { rr += (int32_t)n; q--; }
    if (rr >= (int32_t)n) { rr -= (int32_t)n; q++; }
  }
  r = (uint32_t)rr;
}

// The same for rem < 2^13, 1 <= n <= 64 (frame_bb.hip: a candidate's offset inside its block against a hit count) without
// correction steps: the float quotient rem * rcp(n) is within 2^-9.4 (v_rcp_f32: 1 ulp) + 2^-11 (the fma's rounding at
// magnitude < 2^13) = 0.0020 of rem / n, and rem / n is an integer or at least 1/64 away from one, so after a bias of 2^-8 the
// truncation is the true quotient: integer k -> (k + 0.0019, k + 0.0059), otherwise the fraction stays inside
// (1/64 + 0.0019, 63/64 + 0.0059).  7 instructions instead of ~28 (tests/test_host_cpu.py checks every (rem, n) with the
// reciprocal off by an ulp either way).
__device__ __forceinline__ void divmod_tiny(uint32_t rem, uint32_t n, uint32_t& q, uint32_t& r) {
  const float inv = __builtin_amdgcn_rcpf((float)n);
  q = (uint32_t)fmaf((float)rem, inv, 0x1p-8f);
  r = rem - q * n;
}

// queue words shared between workgroups inside one launch (MODE_ALL): agent-scope relaxed accesses (sc1)
__device__ __forceinline__ int q_load(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void q_store(int32_t* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int q_add(int32_t* p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class Tp>
__device__ __forceinline__ Tp q_ld(const Tp* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class Tp>
__device__ __forceinline__ void q_st(Tp* p, Tp v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Frames of the batch: FrameArgs::n_frames, or fewer when the count lives on the device (the re-submit pass).  Read where
// it is used (the queue pull of one lane), never hoisted: nothing of it stays in registers across a frame.
__device__ __forceinline__ int64_t frame_count(const FrameArgs& p) {
  if (p.n_frames_dev) {
    const int64_t v = q_load(p.n_frames_dev);
    return v < p.n_frames ? v : p.n_frames;
  }
  return p.n_frames;
}

// per-root epipolar line record in LDS: a, b, c, sqrt(a^2+b^2), its reciprocal, pad
constexpr int kLineStride = 6;

// LDS accesses of one wave execute in order; this only stops the compiler from moving them across
// the point where lanes of the same wave exchange data through LDS (no s_barrier needed)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// misc[] slots of a workgroup's frame state
enum { MI_NROOTS = 0, MI_STATUS = 1, MI_NOUT = 2, MI_G = 3, MI_ITEM = 4, MI_DEFER = 5, MI_KIND = 6,
       MI_OMAX = 7 /* bit pattern of the largest |coordinate| among the frame's blobs (float >= 0) */,
       MI_BBCTR = 8 /* frame_bb.hip: queued blocks | their candidates << 10 */, MI_NEXT = 9 /* frame_bb.hip: the next frame */,
       MI_BBCTR2 = 10 /* frame_bb.hip: MI_BBCTR's partner (the two take turns) */,
       MI_HEAVY_N = 11 /* wide frames: heavy roots of this frame (FrameArgs::heavy_bb) */, MI_HEAVY_SLOT = 12,
       MI_SPEC_N = 13 /* wide frames, speculative chain: provisional roots (-1: too many), then 1 = a claim beyond the closest hit */,
       MI_SPEC_OVER = 14 /* ... provisional roots with a hit list over the cap, two words */,
       MI_HEAVY_R0 = 16 /* ... their root numbers, kMaxHeavyPerFrame entries */ };
constexpr int kMaxHeavyPerFrame = 48;

// inclusive prefix sum over the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = (uint32_t)__shfl_up((int)v, d);
    if (lane >= d) v += y;
  }
  return v;
}

// Index of the last element <= key of a non-decreasing sequence seq(0 .. n-1) with seq(0) <= every key, for the 64
// CONSECUTIVE keys kw + lane of a wave (every lane must call; lanes with nothing to look up pass any key >= kw).
// One ballot round finds the wave's first element, a few wave-uniform reads step the lanes forward; a per-lane binary
// search (log2 n dependent LDS round trips, what this replaces) is only the fallback for runs of very short segments.
template <class Seq>
__device__ __forceinline__ int coop_last_le(Seq&& seq, int n, uint32_t kw, uint32_t key, int lane) {
  int c = 0;
  for (int j = 0; j < n; j += 64) {
    const bool le = j + lane < n && seq(j + lane) <= kw;
    c += __popcll(__ballot(le));
  }
  const int first = c - 1;
  int lo = first;
  uint32_t s_last = 0xFFFFFFFFu;
#pragma unroll
  for (int t = 1; t <= 6; t++) {
    s_last = first + t < n ? seq(first + t) : 0xFFFFFFFFu;  // wave-uniform address: one broadcast read
    lo += s_last <= key ? 1 : 0;
  }
  if (s_last <= kw + 63u) {  // wave-uniform: more than 6 segment starts inside these 64 keys
    if (s_last <= key) {
      int a = first + 6, b = n - 1;
      while (a < b) {
        const int mid = (a + b + 1) >> 1;
        if (seq(mid) <= key) a = mid; else b = mid - 1;
      }
      lo = a;
    }
  }
  return lo;
}

// A kept point leaves the kernel: plain, or through the world-coordinate epilogue of the frame loop
// (helpers.py:96-103) fused into the store:  p' = diag(-1,-1,1) p ; h = W [p'; 1] ; q = h[:3] / h[3] ; swap y <-> z
__device__ __forceinline__ void store_point(const FrameArgs& p, size_t o, const double (&X)[3]) {
  if (p.world) {
    ctab_t W = as_ctab(p.world);
    const double x = -X[0], y = -X[1], z = X[2];
    const double h0 = W[0] * x + W[1] * y + W[2] * z + W[3];
    const double h1 = W[4] * x + W[5] * y + W[6] * z + W[7];
    const double h2 = W[8] * x + W[9] * y + W[10] * z + W[11];
    const double h3 = W[12] * x + W[13] * y + W[14] * z + W[15];
    p.xyz[o * 3 + 0] = h0 / h3;
    p.xyz[o * 3 + 1] = h2 / h3;
    p.xyz[o * 3 + 2] = h1 / h3;
  } else {
    p.xyz[o * 3 + 0] = X[0];
    p.xyz[o * 3 + 1] = X[1];
    p.xyz[o * 3 + 2] = X[2];
  }
}

}  // namespace mocap
