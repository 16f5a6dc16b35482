// frame_kernel.hip -- per-frame epipolar correspondence search + candidate triangulation +
// per-root selection.  Replaces
//   find_point_correspondance_and_object_points   (reference computer_code/api/helpers.py:339-421)
// for a batch of independent frames.  One workgroup works on one frame at a time; every
// intermediate (blobs, roots, sorted hit lists, candidate bookkeeping) lives in LDS; HBM sees only
// the blob arrays in and the kept points out (~2 KB per 8x16 frame).
//
// Phases per frame (T threads):
//   A  coalesced load of the frame's blobs/counts into LDS
//   B  for camera i = 1..C-1 (sequential dependency, helpers.py:359-406):
//        B1 one lane per root: epipolar line (helpers.py:362-364) -> LDS
//        B2 one lane per (root, blob): point-line distance (helpers.py:373)
//        B3 rank-by-counting sort of the gated hits (helpers.py:375-385), stable (distance, index)
//        B4 mark blobs equal (by value) to a root's closest hit (helpers.py:391)
//        B5 ballot-compaction of the unclaimed blobs into new roots (helpers.py:402-406)
//      (wide frames -- state beyond LDS, 1 024 lanes -- take match_wide instead: camera-0 roots against all cameras
//      in one barrier-free pass, then a chain over the cameras that involves only the roots created on the way;
//      realistic rigs with identical plain intrinsics never come here: csrc/frame_bb.hip)
//   C  per-root candidate counts (Cartesian product sizes, helpers.py:394-400), offsets
//   D  a range [g_lo, g_hi) of the flat candidate space is split into T contiguous runs; each lane
//      walks its run with a mixed-radix odometer (only the digits that change are re-read), keeps
//      the group's observations in its own LDS column, triangulates and scores every group
//      (csrc/mocap_device.hpp) and keeps an (error, index) first-minimum per (lane, root) segment
//      -- at most T + R segments, stored in LDS, no atomics
//   E  one lane per kept root scans its segments (np.argmin first-minimum, helpers.py:418), decodes
//      the winning group's blob indices and writes xyz / err / corr
//
// Scheduling.  The number of candidate groups per frame is heavy-tailed (8x16: median 2 k, mean
// 3.4 k, p99 24 k, max > 200 k), so a static frame->workgroup map leaves most of the chip idle behind
// a few monster frames.  Three launches of the same code, all fed by device-side atomic queues:
//   MODE_MAIN   persistent workgroups pull frames from a global counter; a frame whose candidate
//               count exceeds `heavy_threshold` is not evaluated but appended to a heavy list, cut
//               into slices of the candidate space;
//   MODE_SLICE  workgroups pull (heavy frame, slice) items, redo the cheap phases A-C (bit-identical
//               by construction) and evaluate only their slice, writing per-root partial winners;
//   MODE_MERGE  one workgroup per heavy frame redoes A-C and picks, per root, the first minimum
//               over the slices in slice order -> same result as the single-workgroup evaluation.
// MODE_ALL (the default) is the same schedule in ONE launch: a workgroup pulls frames while there are any, then
// takes slice tickets; a ticket whose slice has not been published yet waits for it (producers never wait, so
// this cannot deadlock) until every frame has been matched; the workgroup that finishes the LAST slice of a heavy
// frame merges the slices with the frame state it already holds in LDS (no third pass over phases A-C, no third
// launch).  The three-launch form stays selectable (MOCAP_FRAME_LAUNCHES=3) for A/B runs; results are identical.
#include "mocap_device.hpp"
#include "kernels.hpp"
#include "frame_common.hpp"

#ifndef MOCAP_WIDE_BATCH
#define MOCAP_WIDE_BATCH 8  // wide frames: observations fetched ahead per pass of the candidate evaluation (4: 214.6 k, 8: 216.7 k frames/s)
#endif

namespace mocap {

// register budget: 4 waves per SIMD (<= 128 VGPRs); the LDS footprint at 8 x 16 allows 4 workgroups
// of 256 lanes per CU, so both limits meet at 16 waves per CU
// timing experiments only (results invalid): wide frames without the camera-0 pairs (1) / the chain's pairs (2) / the geometry of the candidate evaluation (4: groups are still loaded, walked and merged)
#ifndef MOCAP_WIDE_DEBUG_SKIP
#define MOCAP_WIDE_DEBUG_SKIP 0
#endif
#ifndef MOCAP_WIDE_ACC
#define MOCAP_WIDE_ACC 1  // (1: 32.4 -> 30.5 ms per 12 500 stress frames)
#endif
#ifndef MOCAP_WIDE_DEPTH_CUT
#define MOCAP_WIDE_DEPTH_CUT 0
#endif
#ifndef MOCAP_WIDE_CHAIN_T
#define MOCAP_WIDE_CHAIN_T 24  // wide frames: a chain step with fewer new roots than this keeps the blobs in registers and broadcasts the roots (swept: 8 +2 %, 0 +34 %, never = 24)
#endif
#ifndef MOCAP_WIDE_SPEC
#define MOCAP_WIDE_SPEC 1  // wide frames: the rest of the chain over the cameras matched speculatively in one pass once few blobs are left unclaimed (0: camera by camera)
#endif
#ifndef MOCAP_WIDE_PAIR2
#define MOCAP_WIDE_PAIR2 1  // wide frames: a (root, camera) pair with two candidates at different positions of a step is decided by its own lane
#endif
#ifndef MOCAP_WIDE_CAM1
#define MOCAP_WIDE_CAM1 1  // wide frames: camera 1 first, its roots then ride with the camera-0 roots through cameras 2 .. C-1 (0: a chain step for them)
#endif
#ifndef MOCAP_WIDE_SPEC_T
#define MOCAP_WIDE_SPEC_T 32  // ... at most this many (every provisional root costs one broadcast round per camera: a marker no earlier camera saw -- some 60 unclaimed blobs -- is cheaper as one sequential step)
#endif
#ifndef MOCAP_FRAME_WAVES_PER_EU
#define MOCAP_FRAME_WAVES_PER_EU 4
#endif

// Where the per-frame state lives.  Narrow frames (the realistic rigs: 8 cameras x 16 markers needs
// 39 KB) keep everything in LDS.  Wide frames (up to 64 cameras x 256 blobs: the blobs alone are
// 131 KB) keep the small, hot tables in LDS and move the big ones -- hit lists, per-lane group
// columns -- to a per-workgroup workspace in HBM that stays L2-resident; the blobs are then read in
// place from the input batch.  Same code either way: the arrays are reached through pointers.
struct FrameLayout {
  // byte offsets, computed identically on host (sizes) and device (carving)
  size_t line, dist, seg_e, seg_x, seg_g, goff, gcnt, outslot, root_blob, root_cam, claimed, claimw, nact,
      cnt, misc, rbound, h0, nhc, wbl, wsrow, lds_total;    // always LDS
  size_t bxy, cxy, hits, dig, nh, act;               // narrow: LDS.  wide: hits / nh (exact counts of the multi-hit pairs) in the workspace, the rest unused
  size_t bt;                                          // table mode: DLT contribution per (camera, blob)
  size_t ws_total;
  int Hs;  // hit-list capacity per (root, camera): M when narrow (no cap), H when wide
  __host__ __device__ static size_t align(size_t x, size_t a) { return (x + a - 1) / a * a; }
  // table: identical intrinsics and narrow -> the per-lane group column holds blob INDICES (1 byte per camera)
  // and the DLT contribution of every (camera, blob) is tabulated once per frame: [C][M][10] doubles
  __host__ __device__ FrameLayout(int C, int M, int R, int T, int H, bool wide, bool table) {
    Hs = wide ? H : M;
    size_t o = 0;
    line = o;      o += wide ? 0 : sizeof(double) * kLineStride * R;   // wide: the lines stay in registers (match_pairs_wide)
    seg_e = o;     o += sizeof(double) * (T + R);
    seg_x = o;     o += sizeof(double) * 3 * (T + R);
    // dist (phase B scratch) and the segment arrays (phase D/E) are never live together
    dist = line + (wide ? 0 : sizeof(double) * kLineStride * R);
    const size_t dist_n = wide ? 0 : ((size_t)R * M > (size_t)T ? (size_t)R * M : (size_t)T);
    const size_t dist_end = dist + sizeof(double) * dist_n;
    if (dist_end > o) o = dist_end;
    rbound = o;    o += sizeof(double) * R;   // best error seen so far per root (any lane), phase D
    seg_g = o;     o += sizeof(uint32_t) * (T + R);
    goff = o;      o += sizeof(uint32_t) * (R + 1);
    gcnt = o;      o += sizeof(uint32_t) * R;
    outslot = o;   o += sizeof(int32_t) * R;
    cnt = o;       o += sizeof(int32_t) * C;
    misc = o;      o += sizeof(int32_t) * 64;
    root_blob = o; o += sizeof(uint16_t) * R;
    wsrow = o;     o += wide ? sizeof(uint16_t) * R : 0;  // wide: the workspace row that holds root r's multi-hit lists (its row at the time they were written)
    root_cam = o;  o += R;
    claimed = o;   o += wide ? 0 : M;
    nact = o;      o += wide ? 0 : R;
    o = align(o, 16);
    // wide: blobs claimed so far per camera, 64 per word (match_wide)
    o = align(o, 8);
    claimw = o;    o += wide ? sizeof(unsigned long long) * (size_t)C * ((M + 63) / 64) : 0;
    // wide: THE state of a frame's matching, LDS-resident: per (root, camera) the closest hit's blob index and the number
    // of gated hits (saturating at 255).  Almost every pair of a wide frame has exactly one hit (the marker's own
    // blob), so these two bytes are all a candidate group is ever decoded from; the full hit lists of the rare
    // multi-hit pairs (and their exact counts) sit in the HBM workspace and are touched only for those pairs.
    // (round 6: the count as two BITS per (root, camera) -- "has a hit", "has several: the exact count is in the workspace" --
    // kept as two 64-bit camera masks per root, [R][2]: 6 KB instead of 24.5 KB of bytes at 384 roots x 64 cameras, which
    // is what lets TWO frames share a CU's LDS; a group's decode loads its root's two masks once and tests a bit per camera)
    h0 = o;        o += wide ? (size_t)R * C : 0;
    o = align(o, 8);
    nhc = o;       o += wide ? 16 * (size_t)R : 0;
    o = align(o, 16);
    // wide: one camera's blobs per wave, staged for match_roots_wide (read back as broadcasts): matching scratch, dead before the
    // segment arrays of phase D / E come alive -- it lies over them (seg_e + seg_x = 32 (T + R) bytes >= 32 KB at T = 1024)
    wbl = seg_e;
    size_t w = wide ? 0 : o;  // the movable arrays continue in LDS, or start a workspace
    cxy = w;       w += wide ? 0 : (table ? (size_t)C * T : sizeof(float2) * (size_t)C * T);
    w = align(w, 16);
    bt = w;        w += table ? sizeof(double) * 10 * (size_t)C * M : 0;
    bxy = w;       w += wide ? 0 : sizeof(float2) * (size_t)C * M;
    nh = w;        w += sizeof(uint16_t) * (size_t)R * C;
    hits = w;      w += (size_t)R * C * Hs;
    w = align(w, 8);  // (the branch-and-bound variant keeps doubles here)
    dig = w;       w += wide ? 0 : (size_t)C * T;
    act = w;       w += wide ? 0 : (size_t)R * C;
    w = align(w, 256);
    lds_total = wide ? o : align(w, 16);
    ws_total = wide ? w : 0;
  }
};

#ifndef MOCAP_FRAME_TU_WIDE
size_t frame_lds_bytes(int C, int M, int R, int T, int H, bool wide, bool table) { return FrameLayout(C, M, R, T, H, wide, table && !wide).lds_total; }
size_t frame_ws_bytes(int C, int M, int R, int T, int H, bool wide, bool table) { return FrameLayout(C, M, R, T, H, wide, table && !wide).ws_total; }
#endif

// The kernel arguments arrive through one s_load_dwordx16; a value that lives in a slice of those sixteen registers is spilled and
// reloaded as the whole block (16 v_readlane per reload: the camera tables' pointers were reloaded that way inside the candidate
// evaluation of the wide variant, at 17 places of its loops).  own_sgprs makes each pointer a 64-bit scalar value of its own.
// (wide variant only: 25.4 -> 24.1 ms per 12 500 stress frames; the one-wave kernels of small frames measure the same either way)
#ifndef MOCAP_SPLIT_CV
#ifdef MOCAP_FRAME_TU_WIDE
#define MOCAP_SPLIT_CV 1
#else
#define MOCAP_SPLIT_CV 0
#endif
#endif
__device__ __forceinline__ uint32_t opaque_v(uint32_t x) {
  uint32_t r;
  asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(x));
  return r;
}
template <class Tp>
__device__ __forceinline__ Tp* own_sgprs(Tp* ptr) {
  // through a vector register and back: a scalar value DEFINED by v_readfirstlane (an empty asm is a copy the register coalescer
  // folds back into the slice of the block)
  const unsigned long long u = (unsigned long long)(uintptr_t)ptr;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)opaque_v((uint32_t)u));
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)opaque_v((uint32_t)(u >> 32)));
  return (Tp*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ CamView split_cam_view(const CamView& v) {
  CamView c = v;
  c.Pq = own_sgprs(v.Pq);
  c.RT = own_sgprs(v.RT);
  c.K4 = own_sgprs(v.K4);
  c.F = own_sgprs(v.F);
  return c;
}

// HEAVY (wide variant, re-submit pass only): roots over the candidate cap are exported to the heavy-root search instead of
// flagging their frames -- a separate instantiation, so that the code does not weigh on the registers of the first pass
// Every phase below is __forceinline__: the frame kernels are ONE body per instantiation.  (Round 6: a self-check build had
// write_point outlined as a real function -- calls, a stack in scratch, and a barrier the ISA test could no longer prove safe.)
#ifndef MOCAP_FRAME_FRESH_TID
#define MOCAP_FRAME_FRESH_TID 1
#endif
#ifndef MOCAP_FRAME_PRIO_MATCH
#define MOCAP_FRAME_PRIO_MATCH 0
#endif
#ifndef MOCAP_FRAME_PRIO_CHAIN
#define MOCAP_FRAME_PRIO_CHAIN 0
#endif
#ifndef MOCAP_FRAME_PRIO_SERIAL
#define MOCAP_FRAME_PRIO_SERIAL MOCAP_FRAME_PRIO_CHAIN
#endif
#ifndef MOCAP_FRAME_PRIO_EVAL
#define MOCAP_FRAME_PRIO_EVAL 0
#endif
#ifndef MOCAP_FRAME_PRIO_OUT
#define MOCAP_FRAME_PRIO_OUT 0
#endif
#define MOCAP_FRAME_PRIO_ANY (MOCAP_FRAME_PRIO_MATCH | MOCAP_FRAME_PRIO_CHAIN | MOCAP_FRAME_PRIO_EVAL | MOCAP_FRAME_PRIO_OUT)
template <int P>
__device__ __forceinline__ void frame_prio() {
#if MOCAP_FRAME_PRIO_ANY
  __builtin_amdgcn_s_setprio(P);
#endif
}
template <int T, bool UNIFORM_K, bool F32R, bool WIDE, bool HEAVY = false>
struct FrameState {
  const FrameArgs& p;
#if MOCAP_SPLIT_CV
  const CamView cv;  // a copy whose table pointers are scalar values of their own (split_cam_view), not slices of the 16-dword kernel-argument load
#else
  const CamView& cv;
#endif
  const int C, M, R;
  int tid;
  // The lane's number taken afresh (an empty asm the optimiser cannot see through; csrc/frame_bb.hip has the measurements): what a
  // phase derives from it is computed in that phase instead of once before the frame loop -- where it would sit in scratch
  __device__ __forceinline__ void fresh_tid() {
#if MOCAP_FRAME_FRESH_TID
    int t = tid;
    asm volatile("" : "+v"(t));
    tid = t;
#endif
  }
  int Hs;  // hit-list stride per (root, camera)
  double *line, *dist, *seg_e, *seg_x;
  unsigned long long* rbound;  // [R] bit pattern of the smallest error any lane has found for the root (+inf at start)
  uint32_t *seg_g, *goff, *gcnt;
  int32_t *outslot, *cnt, *misc;
  float2 *bxy;  // [C][M]  the frame's blobs
  float2 *cxy;  // [C][T]  this lane's current group: observation per camera (NaN = none)   (!TABLE)
  uint8_t *cix; // [C][T]  this lane's current group: blob index per camera (0xFF = none)    (TABLE)
  double *bt;   // [C][M][10] DLT contribution of every blob                                   (TABLE)
  static constexpr bool TABLE = UNIFORM_K && !WIDE;
  uint16_t *nh, *root_blob;
  uint8_t *hits;  // [R][C][M] blob indices of the gated hits, ascending distance (M <= 256)
  uint8_t *dig;   // [C][T]    this lane's odometer digits
  uint8_t *root_cam, *claimed, *act, *nact;  // act [R][C]: cameras of root r with >= 2 hits
  unsigned long long* claimw;  // wide: [C][ceil(M / 64)] blobs claimed so far (match_wide)
  uint8_t* h0;                 // wide: [R][C] blob index of the closest gated hit (the root's own blob at its camera)   (LDS)
  unsigned long long* nhc;     // wide: [R][2] camera masks of root r: bit c of [0] = a gated hit in camera c (the root's own camera included), of [1] = several (LDS)
  uint16_t* wsrow;             // wide: [R] workspace row of root r's exact counts / hit lists (a provisional root keeps them where they were written when it moves down: spec_finish)
  float2* wbl;                 // wide: [T / 64][kMaxBlobs] a wave's copy of the camera it is matching against               (LDS)
  int spec_base = -1;          // wide: first row of the provisional roots while they are matched speculatively (spec_begin / spec_finish), else -1

  __device__ FrameState(const FrameArgs& p_, unsigned char* smem)
#if MOCAP_SPLIT_CV
      : p(p_), cv(split_cam_view(p_.cv)), C(p_.cv.C), M(p_.M), R(p_.K_max), tid(threadIdx.x) {
#else
      : p(p_), cv(p_.cv), C(p_.cv.C), M(p_.M), R(p_.K_max), tid(threadIdx.x) {
#endif
    const FrameLayout L(C, M, R, T, p_.H, WIDE, TABLE);
    Hs = L.Hs;
    line = (double*)(smem + L.line);
    dist = (double*)(smem + L.dist);
    seg_e = (double*)(smem + L.seg_e);
    seg_x = (double*)(smem + L.seg_x);
    rbound = (unsigned long long*)(smem + L.rbound);
    seg_g = (uint32_t*)(smem + L.seg_g);
    goff = (uint32_t*)(smem + L.goff);
    gcnt = (uint32_t*)(smem + L.gcnt);
    outslot = (int32_t*)(smem + L.outslot);
    cnt = (int32_t*)(smem + L.cnt);
    misc = (int32_t*)(smem + L.misc);
    root_blob = (uint16_t*)(smem + L.root_blob);
    root_cam = (uint8_t*)(smem + L.root_cam);
    claimed = (uint8_t*)(smem + L.claimed);
    nact = (uint8_t*)(smem + L.nact);
    // the movable arrays: LDS, or this workgroup's slice of the HBM workspace
    unsigned char* big = WIDE ? p_.ws + (size_t)blockIdx.x * p_.ws_stride : smem;
    bxy = (float2*)(big + L.bxy);  // wide: re-pointed at the input frame in match()
    cxy = (float2*)(big + L.cxy) + tid;
    cix = (uint8_t*)(big + L.cxy) + tid;
    bt = (double*)(big + L.bt);
    hits = (uint8_t*)(big + L.hits);
    dig = (uint8_t*)(big + L.dig) + tid;
    nh = (uint16_t*)(big + L.nh);
    act = (uint8_t*)(big + L.act);
    claimw = (unsigned long long*)(smem + L.claimw);
    h0 = smem + L.h0;
    nhc = (unsigned long long*)(smem + L.nhc);
    wbl = (float2*)(smem + L.wbl);
    wsrow = (uint16_t*)(smem + L.wsrow);
  }

  // gated hits of root r in camera c (wide: the LDS byte, or the exact count from the workspace when it saturated)
  __device__ __forceinline__ uint32_t nhits(int r, int c) const {
    if constexpr (WIDE) {
      if (!((nhc[2 * (size_t)r] >> c) & 1ull)) return 0u;
      return ((nhc[2 * (size_t)r + 1] >> c) & 1ull) ? (uint32_t)nh[(size_t)wsrow[r] * C + c] : 1u;  // (several: resolve_pair left the exact count in the workspace)
    } else {
      return nh[(size_t)r * C + c];
    }
  }
  // wide: record that root r has one or several gated hits in camera c.  The masks start at zero for every frame (match_wide)
  // and every (root, camera) pair is decided once -- by the wave that owns camera c, other cameras' waves set other bits of
  // the same word at the same time: LDS atomics.
  __device__ __forceinline__ void set_hit_code(int r, int c, int n_hits) {
    if (n_hits <= 0) return;
    atomicOr(&nhc[2 * (size_t)r], 1ull << c);
    if (n_hits >= 2) atomicOr(&nhc[2 * (size_t)r + 1], 1ull << c);
  }
  // blob index of root r's d-th closest hit in camera c (d < nhits(r, c))
  __device__ __forceinline__ uint32_t hit_at(int r, int c, uint32_t d) const {
    if constexpr (WIDE) {
      return d == 0 ? (uint32_t)h0[(size_t)r * C + c] : (uint32_t)hits[((size_t)wsrow[r] * C + c) * Hs + d];
    } else {
      return hits[((size_t)r * C + c) * Hs + d];
    }
  }

  // ---------------------------------------------------------------- phases A-C
  // Leaves roots / hit lists / candidate offsets in LDS; returns with all lanes synchronised.  (Narrow frames; wide
  // frames take match_wide.)
  __device__ __forceinline__ void match(int64_t frame, int skip = 0 /* timing experiments: 1 = no table build, 2 = no camera loop */) {
    fresh_tid();
    {
      const float2* src = (const float2*)(p.blobs + (size_t)frame * C * M * 2);
      for (int i = tid; i < C * M; i += T) bxy[i] = src[i];
      if (tid < C) {
        int n = p.counts[(size_t)frame * C + tid];
        cnt[tid] = n < 0 ? 0 : (n > M ? M : n);
      }
      if (tid == 0) {
        misc[MI_STATUS] = 0;
        misc[MI_OMAX] = 0;
      }
    }
    __syncthreads();
    if (p.p3max2 > 0.0) {  // EigCut's float32 allowance scales with the largest coordinate (inf: the cut-off never fires)
      float om = 0.0f;
      for (int i = tid; i < C * M; i += T) {
        const int c = i / M, k = i - c * M;
        if (k < cnt[c]) {
          const float2 v = bxy[i];
          om = fmaxf(om, fmaxf(fabsf(v.x), fabsf(v.y)));
        }
      }
      if (om > 0.0f) atomicMax(&misc[MI_OMAX], __float_as_int(om));
    }
    if (TABLE && !(skip & 1)) {
      // DLT contribution of every blob, once per frame: a candidate group then ADDS ten doubles per view
      // instead of rebuilding two rows of A and their outer products (the Cartesian product revisits every
      // blob thousands of times)
      for (int i = tid; i < C * M; i += T) {
        const int c = i / M, k = i - c * M;
        if (k < cnt[c]) {
          const float2 v = bxy[i];
          double Bc[10];
          dlt_contribution(Bc, as_ctab(cv.Pq + 12 * c), (double)v.x, (double)v.y);
#pragma unroll
          for (int e = 0; e < 10; e++) bt[(size_t)i * 10 + e] = Bc[e];
        }
      }
    }
    {  // roots from camera 0 (helpers.py:349,357)
      const int n0 = cnt[0];
      for (int r = tid; r < n0 && r < R; r += T) {
        root_cam[r] = 0;
        root_blob[r] = (uint16_t)r;
      }
      if (tid == 0) {
        misc[MI_NROOTS] = n0 < R ? n0 : R;
        if (n0 > R) misc[MI_STATUS] |= MOCAP_ST_ROOT_OVERFLOW_;
      }
    }
    __syncthreads();

    // group geometry of the fused B2-B4 step: the blobs of one root occupy GS = 2^gs_shift >= M
    // consecutive lanes; when that fits a wave (M <= 64) gate, order and claim need no LDS round
    // trips through the whole workgroup, only wave-level ballots
    int gs_shift = 0;
    while ((1 << gs_shift) < M) gs_shift++;
    const bool fused = gs_shift <= 6;

    for (int i = 1; i < ((skip & 2) ? 1 : C); i++) {
      const int Mi = cnt[i];
      const float2* pts = bxy + (size_t)i * M;
      // B1 (wave 0, which also owns B5: no workgroup barrier between B5 of camera i-1 and this):
      // epipolar line of every root in camera i.  cv.computeCorrespondEpilines on a float32
      // point: double math, scale by 1/sqrt(a^2+b^2), float32 result (helpers.py:363-364).
      if (tid < 64) {
        const int nr = misc[MI_NROOTS];
        for (int r = tid; r < nr; r += 64) {
          const int rc = root_cam[r], rb = root_blob[r];
          ctab_t Fm = as_ctab(cv.F + 9 * ((size_t)rc * C + i));
          const float2 rp = bxy[(size_t)rc * M + rb];
          const double x = (double)rp.x, y = (double)rp.y;
          double a = Fm[0] * x + Fm[1] * y + Fm[2];
          double b = Fm[3] * x + Fm[4] * y + Fm[5];
          double c = Fm[6] * x + Fm[7] * y + Fm[8];
          double nu = a * a + b * b;
          nu = nu != 0.0 ? 1.0 / sqrt(nu) : 1.0;
          a *= nu;
          b *= nu;
          c *= nu;
          if (F32R) {
            a = (double)(float)a;
            b = (double)(float)b;
            c = (double)(float)c;
          }
          const double den = sqrt(a * a + b * b);  // helpers.py:373 divides by it again
          line[kLineStride * r + 0] = a;
          line[kLineStride * r + 1] = b;
          line[kLineStride * r + 2] = c;
          line[kLineStride * r + 3] = den;
          line[kLineStride * r + 4] = recip_refined(den);  // shared by the M quotients of B2
          nh[(size_t)r * C + i] = 0;
        }
        for (int k = tid; k < M; k += 64) claimed[k] = 0;
      }
      __syncthreads();
      const int nroots = misc[MI_NROOTS];
      if (fused) {
        // B2-B4, one lane per (root, blob): |a x + b y + c| / sqrt(a^2 + b^2) (helpers.py:373), gate
        // (strict <, helpers.py:375,383), rank among the root's hits by (distance, index), and the
        // closest hit's removal *by value* from the unmatched set (helpers.py:391)
        const int GS = 1 << gs_shift;
        const int k = tid & (GS - 1);
        const int gl0 = (tid & 63) & ~(GS - 1);  // first lane of the group inside its wave
        const unsigned long long gall = gs_shift == 6 ? ~0ull : ((1ull << GS) - 1ull);
        const int roots_per_pass = T >> gs_shift;
        for (int base = 0; base < nroots; base += roots_per_pass) {  // uniform trip count
          const int r = base + (tid >> gs_shift);
          const bool valid = r < nroots && k < Mi;
          double d = 0.0;
          bool hit = false;
          if (valid) {
            const double* ln = line + kLineStride * r;
            const double px = (double)pts[k].x, py = (double)pts[k].y;
            const double num = fabs(ln[0] * px + ln[1] * py + ln[2]);
            d = div_by(num, ln[3], ln[4]);
            hit = d < p.gate_px;
          }
          dist[tid] = d;
          const unsigned long long gm = (__ballot(hit) >> gl0) & gall;
          wave_lds_sync();
          int rank = -1;
          if (hit) {
            rank = 0;
            unsigned long long mm = gm & ~(1ull << k);
            while (mm) {
              const int j = __ffsll(mm) - 1;
              mm &= mm - 1;
              const double d2 = dist[tid - k + j];
              rank += (d2 < d || (d2 == d && j < k)) ? 1 : 0;
            }
            hits[((size_t)r * C + i) * Hs + rank] = (uint8_t)k;
          }
          const unsigned long long g0 = (__ballot(rank == 0) >> gl0) & gall;
          if (valid) {
            if (k == 0) nh[(size_t)r * C + i] = (uint16_t)__popcll(gm);
            if (g0) {
              const int k0 = __ffsll(g0) - 1;
              if (pts[k].x == pts[k0].x && pts[k].y == pts[k0].y) claimed[k] = 1;
            }
          }
          wave_lds_sync();  // dist[] is reused by the next pass
        }
      } else {
        // generic path (a root's blobs span several waves): same steps through LDS + barriers
        const int npairs = nroots * Mi;
        for (int idx = tid; idx < npairs; idx += T) {
          const int r = idx / Mi, k = idx - r * Mi;
          const double* ln = line + kLineStride * r;
          const double px = (double)pts[k].x, py = (double)pts[k].y;
          dist[(size_t)r * M + k] = fabs(ln[0] * px + ln[1] * py + ln[2]) / ln[3];
        }
        __syncthreads();
        for (int idx = tid; idx < npairs; idx += T) {
          const int r = idx / Mi, k = idx - r * Mi;
          const double* dr = dist + (size_t)r * M;
          const double d = dr[k];
          if (d < p.gate_px) {
            int rank = 0;
            for (int k2 = 0; k2 < Mi; k2++) {
              const double d2 = dr[k2];
              rank += (d2 < d || (d2 == d && k2 < k)) ? 1 : 0;
            }
            hits[((size_t)r * C + i) * Hs + rank] = (uint8_t)k;
          }
          if (k == 0) {
            int n = 0;
            for (int k2 = 0; k2 < Mi; k2++) n += dr[k2] < p.gate_px ? 1 : 0;
            nh[(size_t)r * C + i] = (uint16_t)n;
          }
        }
        __syncthreads();
        for (int idx = tid; idx < npairs; idx += T) {
          const int r = idx / Mi, k = idx - r * Mi;
          if (nh[(size_t)r * C + i] > 0) {
            const int k0 = hits[((size_t)r * C + i) * Hs];
            if (pts[k].x == pts[k0].x && pts[k].y == pts[k0].y) claimed[k] = 1;
          }
        }
      }
      __syncthreads();
      // B5: unclaimed blobs become new roots, in blob order (helpers.py:402-406); wave 0 compacts
      if (tid < 64) {
        int base_root = nroots;
        for (int k0 = 0; k0 < Mi; k0 += 64) {
          const int k = k0 + tid;
          const bool flag = k < Mi && !claimed[k];
          const unsigned long long mask = __ballot(flag);
          const int pos = __popcll(mask & ((1ull << tid) - 1ull));
          if (flag) {
            const int rr = base_root + pos;
            if (rr < R) {
              root_cam[rr] = (uint8_t)i;
              root_blob[rr] = (uint16_t)k;
            }
          }
          base_root += __popcll(mask);
        }
        if (tid == 0) {
          if (base_root > R) {
            misc[MI_STATUS] |= MOCAP_ST_ROOT_OVERFLOW_;
            base_root = R;
          }
          misc[MI_NROOTS] = base_root;
        }
        wave_lds_sync();  // B1 of the next camera runs in this wave
      }
    }
    __syncthreads();

    count_candidates();
  }

  // One (root, camera) pair with several candidate blobs, the rare case of a wide frame: the wave holds the camera's blobs
  // four per lane (bl), pc marks the ones to decide, lane q holds the root's line.  Decision of helpers.py:373,375 in
  // double; the hits ranked by (distance, blob index) -- stable where NumPy's default argsort is not (helpers.py:384;
  // documented deviation) -- among the lanes that hold them: every hit's (d, k) is broadcast once and compared with the
  // (at most four) hits of every lane.  A list of two or more goes to the workspace in HBM; the closest hit's
  // coordinates claim every blob that has them (helpers.py:391).  Result in lane q's my_nh / my_k0.  Wave-uniform call.
  __device__ __forceinline__ void resolve_pair(int q, int rq, int i, const float2 (&bl)[4], const bool (&pc)[4], double la, double lb,
                                               double lc, double lden, double lrden, int& my_nh, int& my_k0) {
    const int lane = tid & 63;
    const int H = p.H, MW = (M + 63) / 64;
    auto bcast = [&](double v, int l) {
      const long long bits = __double_as_longlong(v);
      return __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(bits >> 32), l) << 32) |
                                  (unsigned int)__builtin_amdgcn_readlane((int)bits, l));
    };
    const double qa = bcast(la, q), qb = bcast(lb, q), qc = bcast(lc, q), qden = bcast(lden, q), qrden = bcast(lrden, q);
    double dd[4] = {0.0, 0.0, 0.0, 0.0};
    bool hit[4] = {false, false, false, false};
    unsigned long long hmask[4] = {0ull, 0ull, 0ull, 0ull};
    int nhits = 0;
#pragma unroll
    for (int sg = 0; sg < 4; sg++) {
      if (!__ballot(pc[sg])) continue;  // wave-uniform: usually one segment of the four has a candidate
      if (pc[sg]) {
        dd[sg] = div_by(fabs(qa * (double)bl[sg].x + qb * (double)bl[sg].y + qc), qden, qrden);
        hit[sg] = dd[sg] < p.gate_px;  // strict <, helpers.py:375,383
      }
      hmask[sg] = __ballot(hit[sg]);
      nhits += __popcll(hmask[sg]);
    }
    if (!nhits) return;
    const bool spec = spec_base >= 0;  // a provisional root: it claims nothing yet, and its flags count only if it turns out real
    if (nhits > H) {
      if (spec)
        atomicOr(&misc[MI_SPEC_OVER + ((rq - spec_base) >> 5)], 1 << ((rq - spec_base) & 31));
      else
        atomicOr(&misc[MI_STATUS], MOCAP_ST_HIT_OVERFLOW_);
    }
    if (nhits == 1) {
      // the one blob inside the gate is the closest hit and claims itself (helpers.py:391)
      int k1 = 0;
#pragma unroll
      for (int sg = 0; sg < 4; sg++)
        if (hmask[sg]) k1 = 64 * sg + (__ffsll((long long)hmask[sg]) - 1);
      if (!spec) {
#pragma unroll
        for (int sg = 0; sg < 4; sg++)
          if (hit[sg]) atomicOr(&claimw[(size_t)i * MW + sg], 1ull << lane);
      }
      if (lane == q) {
        my_nh = 1;
        my_k0 = k1;
      }
      return;
    }
    int rank[4] = {0, 0, 0, 0};
#pragma unroll
    for (int s2 = 0; s2 < 4; s2++) {
      unsigned long long mm = hmask[s2];
      while (mm) {  // wave-uniform
        const int l2 = __ffsll((long long)mm) - 1;
        mm &= mm - 1;
        const double d2 = bcast(dd[s2], l2);
        const int k2 = 64 * s2 + l2;
#pragma unroll
        for (int sg = 0; sg < 4; sg++) rank[sg] += (hit[sg] && (d2 < dd[sg] || (d2 == dd[sg] && k2 < 64 * sg + lane))) ? 1 : 0;
      }
    }
    uint8_t* hl = hits + ((size_t)rq * C + i) * Hs;
    int k0 = 0;
#pragma unroll
    for (int sg = 0; sg < 4; sg++) {
      if (hit[sg] && rank[sg] < H) hl[rank[sg]] = (uint8_t)(64 * sg + lane);
      const unsigned long long first = __ballot(hit[sg] && rank[sg] == 0);
      if (first) k0 = 64 * sg + (__ffsll((long long)first) - 1);  // wave-uniform
    }
    if (lane == q) {
      my_nh = nhits < H ? nhits : H;
      my_k0 = k0;
      nh[(size_t)rq * C + i] = (uint16_t)my_nh;  // the exact count (the LDS byte saturates at 255)
    }
    float p0x = 0.f, p0y = 0.f;
#pragma unroll
    for (int sg = 0; sg < 4; sg++)
      if ((k0 >> 6) == sg) {
        p0x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bl[sg].x), k0 & 63));
        p0y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bl[sg].y), k0 & 63));
      }
    int nclaim = 0;
#pragma unroll
    for (int sg = 0; sg < 4; sg++) {
      const unsigned long long cm = __ballot(hit[sg] && bl[sg].x == p0x && bl[sg].y == p0y);
      if (cm && lane == 0 && !spec) atomicOr(&claimw[(size_t)i * MW + sg], cm);
      nclaim += __popcll(cm);
    }
    // speculative pass: the claims are replayed later from the closest hits alone (spec_begin / spec_finish); hits that share the
    // closest hit's coordinates are claimed with it (helpers.py:391) -- such a frame goes to the sequential chain instead
    if (spec && nclaim != 1 && lane == 0) misc[MI_SPEC_N] = 1;
  }

  // Epipolar line of a root in camera i: cv.computeCorrespondEpilines on a float32 point -- double math, scale by
  // 1/sqrt(a^2+b^2), float32 result (helpers.py:363-364) -- plus what the distance of helpers.py:373 divides by.
  template <class Tab>
  __device__ __forceinline__ void epiline_wide(Tab Fm, float2 rp, double& la, double& lb, double& lc, double& lden, double& lrden) {
    const double x = (double)rp.x, y = (double)rp.y;
    double a = Fm[0] * x + Fm[1] * y + Fm[2];
    double bb = Fm[3] * x + Fm[4] * y + Fm[5];
    double c = Fm[6] * x + Fm[7] * y + Fm[8];
    double nu = a * a + bb * bb;
    nu = nu != 0.0 ? 1.0 / sqrt(nu) : 1.0;
    a *= nu;
    bb *= nu;
    c *= nu;
    if (F32R) {
      a = (double)(float)a;
      bb = (double)(float)bb;
      c = (double)(float)c;
    }
    la = a;
    lb = bb;
    lc = c;
    lden = sqrt(a * a + bb * bb);  // helpers.py:373 divides by it again
    lrden = recip_refined(lden);
  }

  // ---------------------------------------------------------------- many roots against the cameras after them (round 4)
  // One LANE per root, the camera's blobs broadcast from the wave's LDS copy.  What a (root, camera) pair costs is the
  // float32 pre-test of the camera's blobs against the root's line -- two FMAs and a compare per blob for the 64 roots of a
  // batch at once -- and little else: the usual outcome (exactly one blob passes, the marker's own) is simply what the
  // root's lane has in its registers when the loop ends (blob index + count; the bookkeeping instructions run only for
  // the blobs that pass for some root of the batch, a scalar branch).  The exact decision of helpers.py:373,375
  // follows for the 64 roots at once; a root with several candidates goes through resolve_pair.  Round 3 kept the blobs in
  // registers and broadcast the roots instead: each pair then paid ~30 scalar / cross-lane instructions to hand the one
  // passing blob to the lane that holds the root's line (55 issue slots per pair, measured).  CAM01: every root of the
  // call comes from camera 0 or camera 1 (the two fundamental matrices of a camera come over the scalar cache, a select per
  // lane).  Cameras [clo, chi): a wave takes every (T / 64)-th; `split`: every wave of the first four takes EVERY camera and
  // one batch of each group instead (the call for camera 1 alone, on which everything after it waits).
  template <bool CAM01>
  __device__ __forceinline__ void match_roots_wide(int rlo, int rhi, int clo, int chi, bool split) {
    fresh_tid();
    constexpr int W = T / 64;
    const int lane = tid & 63, wave = tid >> 6;
    if (split && wave >= 4) return;  // (no workgroup barrier inside)
    const int MW = (M + 63) / 64;
    const double om = (double)__int_as_float(misc[MI_OMAX]);
    // the wave's copy of the camera's blobs, x and y apart: four consecutive x (or y) are one broadcast read
    float* wx = reinterpret_cast<float*>(wbl) + (size_t)wave * 2 * kMaxBlobs;
    float* wy = wx + kMaxBlobs;
    const float finf = __int_as_float(0x7f800000);
    for (int g0 = rlo; g0 < rhi; g0 += 256) {
      // the roots' own points: once per group of four batches, not once per camera
      float2 rp[4];
      int rcv[4];
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const int r = g0 + 64 * b + lane;
        const bool okr = r < rhi;
        rcv[b] = okr ? root_cam[r] : 0;
        rp[b] = bxy[(size_t)rcv[b] * M + (okr ? root_blob[r] : 0)];
      }
      for (int i = clo + (split ? 0 : wave); i < chi; i += (split ? 1 : W)) {  // wave-uniform
        const int Mi = __builtin_amdgcn_readfirstlane(cnt[i]);  // (uniform anyway: tells the compiler so -- scalar loop control below)
        const int M4 = (Mi + 3) & ~3;
        const float2* row = bxy + (size_t)i * M;
        wave_lds_sync();  // the previous camera's reads are done
#pragma unroll
        for (int sg = 0; sg < 4; sg++) {
          const int k = 64 * sg + lane;
          // slots beyond the camera's count hold +inf: |fl(a x + b y + c)| <= thr is false for them (inf or NaN)
          if (k < M4) {
            const float2 v = k < Mi ? row[k] : make_float2(finf, finf);
            wx[k] = v.x;
            wy[k] = v.y;
          }
        }
        wave_lds_sync();
        for (int b = split ? wave : 0; b < 4 && g0 + 64 * b < rhi; b += split ? 4 : 1) {  // wave-uniform; ONE copy of the body (not unrolled)
          const int r = g0 + 64 * b + lane;
          const bool have = r < rhi;
          // this batch's root point: selected from registers (b is wave-uniform), never an indexed array
          const float2 rpb = b == 0 ? rp[0] : (b == 1 ? rp[1] : (b == 2 ? rp[2] : rp[3]));
          const int rcb = b == 0 ? rcv[0] : (b == 1 ? rcv[1] : (b == 2 ? rcv[2] : rcv[3]));
          double la = 0, lb = 0, lc = 0, lden = 1, lrden = 1;
          if (have) {
            if constexpr (CAM01) {
              struct F01 {  // F[0][i] or F[1][i], by the root's camera: two scalar loads and a select per entry
                ctab_t f0, f1;
                bool one;
                __device__ __forceinline__ double operator[](int k) const {
                  const double a = f0[k], b = f1[k];
                  return one ? b : a;
                }
              };
              epiline_wide(F01{as_ctab(cv.F + 9 * (size_t)i), as_ctab(cv.F + 9 * ((size_t)C + i)), rcb != 0}, rpb, la, lb, lc, lden, lrden);
            } else {
              epiline_wide(cv.F + 9 * ((size_t)rcb * C + i), rpb, la, lb, lc, lden, lrden);
            }
          }
          // pre-test threshold, rounded up (a rigorous bound on the float32 evaluation, see match_pairs_wide); +inf --
          // everything goes to the exact test -- without the float32 line; idle lanes never pass
          const float a32 = (float)la, b32 = (float)lb, c32 = (float)lc;
          float thr = finf;
          if (F32R) thr = __double2float_ru(p.gate_px * lden * (1.0 + 1e-12) + (2.5 * 0x1p-24) * (2.0 * om + fabs(lc)) * (1.0 + 1e-6));
          if (!have) thr = -1.0f;
          int np = 0, kk = 0;
#if MOCAP_WIDE_ACC
          // bookkeeping as four accumulators, one per position of a step: += 0x10000 + k0 when the blob passes, i.e. the
          // count in the high half and the index (sum) in the low half -- a select and an add per blob
          uint32_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
#endif
          for (int k0 = 0; k0 < M4; k0 += 4) {  // scalar loop; four blobs per step, two broadcast reads
            const float4 X = *reinterpret_cast<const float4*>(wx + k0);
            const float4 Y = *reinterpret_cast<const float4*>(wy + k0);
            const float t0 = fmaf(a32, X.x, fmaf(b32, Y.x, c32)), t1 = fmaf(a32, X.y, fmaf(b32, Y.y, c32));
            const float t2 = fmaf(a32, X.z, fmaf(b32, Y.z, c32)), t3 = fmaf(a32, X.w, fmaf(b32, Y.w, c32));
            const unsigned long long m0 = __ballot(fabsf(t0) <= thr), m1 = __ballot(fabsf(t1) <= thr);
            const unsigned long long m2 = __ballot(fabsf(t2) <= thr), m3 = __ballot(fabsf(t3) <= thr);
#if MOCAP_WIDE_ACC
            if ((m0 | m1) | (m2 | m3)) {
              const uint32_t kb = 0x10000u + (uint32_t)k0;
              acc0 += fabsf(t0) <= thr ? kb : 0u;
              acc1 += fabsf(t1) <= thr ? kb : 0u;
              acc2 += fabsf(t2) <= thr ? kb : 0u;
              acc3 += fabsf(t3) <= thr ? kb : 0u;
            }
#else
            if ((m0 | m1) | (m2 | m3)) {  // some root of the batch has a candidate among these four (scalar branch)
              if (m0) {
                const bool ps = fabsf(t0) <= thr;
                np += ps ? 1 : 0;
                kk = ps ? k0 : kk;
              }
              if (m1) {
                const bool ps = fabsf(t1) <= thr;
                np += ps ? 1 : 0;
                kk = ps ? k0 + 1 : kk;
              }
              if (m2) {
                const bool ps = fabsf(t2) <= thr;
                np += ps ? 1 : 0;
                kk = ps ? k0 + 2 : kk;
              }
              if (m3) {
                const bool ps = fabsf(t3) <= thr;
                np += ps ? 1 : 0;
                kk = ps ? k0 + 3 : kk;
              }
            }
#endif
          }
#if MOCAP_WIDE_ACC
          np = (int)((acc0 >> 16) + (acc1 >> 16) + (acc2 >> 16) + (acc3 >> 16));
          kk = acc0 ? (int)(acc0 & 0xffffu) : (acc1 ? (int)(acc1 & 0xffffu) + 1 : (acc2 ? (int)(acc2 & 0xffffu) + 2 : (int)(acc3 & 0xffffu) + 3));  // (meaningful when np == 1)
#endif
#ifdef MOCAP_DEBUG_PRETEST  // self-check build: the exact decision (helpers.py:373,375) for EVERY blob the float32 pre-test
          if (have) {          // rejected; a blob inside the gate among them is a false negative (must never happen)
            int fneg = 0;
            for (int k = 0; k < Mi; k++) {
              const float2 v = make_float2(wx[k], wy[k]);
              const bool pass = fabsf(fmaf(a32, v.x, fmaf(b32, v.y, c32))) <= thr;
              if (!pass && div_by(fabs(la * (double)v.x + lb * (double)v.y + lc), lden, lrden) < p.gate_px) fneg++;
            }
            atomicAdd(&p.status[p.n_frames], 1);
            if (fneg) atomicAdd(&p.status[p.n_frames + 1], fneg);  // (no printf here: it costs the build 300 spilled registers)
          }
#endif
          int my_nh = 0, my_k0 = 0;
          if (have && np == 1) {  // the usual outcome: one candidate; helpers.py:373 in double, strict < (helpers.py:375,383)
            const float2 v = make_float2(wx[kk], wy[kk]);
            if (kk < Mi && div_by(fabs(la * (double)v.x + lb * (double)v.y + lc), lden, lrden) < p.gate_px) {
              atomicOr(&claimw[(size_t)i * MW + (kk >> 6)], 1ull << (kk & 63));  // a lone hit claims itself (helpers.py:391)
              my_nh = 1;
              my_k0 = kk;
            }
          }
          // Two candidates at different positions of a step (three quarters of the pairs with several candidates: ~300 such pairs
          // per stress frame, one in most (batch, camera) rounds): each accumulator holds ONE index, so the pair is decided right
          // here, for all such lanes at once -- both distances in double (helpers.py:373), strict gate (helpers.py:375,383), the hits
          // in (distance, index) order, the closest claims by value (helpers.py:391) -- exactly resolve_pair's outcome without its
          // wave-serial round (broadcast of the line, distances of all 256 blobs, ranking by broadcast: ~250 instructions per pair).
          bool two = false;
#if MOCAP_WIDE_ACC && MOCAP_WIDE_PAIR2
          two = have && np == 2 && p.H >= 2 && ((acc0 | acc1 | acc2 | acc3) >> 17) == 0u;  // (no accumulator counts two)
          if (__ballot(two)) {  // wave-uniform
            if (two) {
              const int c0 = (int)(acc0 >> 16), c1 = (int)(acc1 >> 16), c2 = (int)(acc2 >> 16), c3 = (int)(acc3 >> 16);
              const int k0c = (int)(acc0 & 0xffffu), k1c = (int)(acc1 & 0xffffu) + 1, k2c = (int)(acc2 & 0xffffu) + 2, k3c = (int)(acc3 & 0xffffu) + 3;
              const int lo = c0 ? k0c : (c1 ? k1c : k2c), hi = c3 ? k3c : (c2 ? k2c : k1c);  // the lowest / highest position that counted
              const int kA = lo < hi ? lo : hi, kB = lo < hi ? hi : lo;                      // by blob index
              const float2 vA = make_float2(wx[kA], wy[kA]), vB = make_float2(wx[kB], wy[kB]);
              const double dA = div_by(fabs(la * (double)vA.x + lb * (double)vA.y + lc), lden, lrden);
              const double dB = div_by(fabs(la * (double)vB.x + lb * (double)vB.y + lc), lden, lrden);
              const bool hitA = kA < Mi && dA < p.gate_px, hitB = kB < Mi && dB < p.gate_px;
              if (hitA && hitB) {
                const bool bfirst = dB < dA;  // equal distances: the lower blob index first (stable order, see resolve_pair)
                const int k1 = bfirst ? kB : kA, k2 = bfirst ? kA : kB;
                uint8_t* hl = hits + ((size_t)r * C + i) * Hs;
                hl[0] = (uint8_t)k1;
                hl[1] = (uint8_t)k2;
                nh[(size_t)r * C + i] = (uint16_t)2;  // the exact count (the LDS code only says "several")
                atomicOr(&claimw[(size_t)i * MW + (k1 >> 6)], 1ull << (k1 & 63));
                if (vA.x == vB.x && vA.y == vB.y) atomicOr(&claimw[(size_t)i * MW + (k2 >> 6)], 1ull << (k2 & 63));  // the same coordinates: claimed with it
                my_nh = 2;
                my_k0 = k1;
              } else if (hitA || hitB) {  // one blob inside the gate after all: it is the closest hit and claims itself
                const int k1 = hitA ? kA : kB;
                atomicOr(&claimw[(size_t)i * MW + (k1 >> 6)], 1ull << (k1 & 63));
                my_nh = 1;
                my_k0 = k1;
              }
            }
          }
#endif
          unsigned long long multi = __ballot(have && np >= 2 && !two);
          if (multi) {  // rare: roots with several candidates, one at a time, the camera's blobs four per lane
            float2 bl[4];
            bool pc[4];
#pragma unroll
            for (int sg = 0; sg < 4; sg++) {
              const int k = 64 * sg + lane;
              pc[sg] = k < Mi;
              bl[sg] = pc[sg] ? make_float2(wx[k], wy[k]) : make_float2(finf, finf);
            }
            while (multi) {
              const int q = __ffsll((long long)multi) - 1;
              multi &= multi - 1;
              resolve_pair(q, g0 + 64 * b + q, i, bl, pc, la, lb, lc, lden, lrden, my_nh, my_k0);
            }
          }
          if (have) {
            h0[(size_t)r * C + i] = (uint8_t)my_k0;
            set_hit_code(r, i, my_nh);
          }
        }
      }
    }
  }

  // ---------------------------------------------------------------- phases A-B, wide frames (round 3)
  // The reference matches camera after camera (helpers.py:359-406) because a blob no root claims becomes a new root
  // for the cameras after it; only THAT is sequential.  A root's lines, gates, orders and claims in all the cameras
  // after its own are independent of everything else, so they are computed the moment the root exists:
  //   B0   camera-0 roots x cameras 1 .. C-1, all waves, no barrier inside;
  //   B1   for camera j = 1 .. C-1: wave 0 turns the unclaimed blobs of camera j into roots (helpers.py:402-406);
  //        if there are any, all waves match THEM against cameras j+1 .. C-1.
  // (Round 2 ran 63 x [lines | barrier | 4.3 M distances spread as (root, blob) lanes + LDS appends | barrier | one
  // lane per root sorts | barrier | new roots]: 1.83 ms per 64 x 256 frame, VALU 31 % busy, the rest barrier waits.)
  // Work unit = one camera i (a wave takes cameras clo + wave, + W, ...): its blobs are read ONCE, coalesced, four per
  // lane, and stay in registers while the roots [rlo, rhi) are walked.  The roots' epipolar lines in camera i are
  // computed 64 at a time (one per lane, the expressions of match() phase B1 in the same order) and broadcast one
  // after the other; a (root, camera) pair then costs ~25 wave instructions for its 256 distances: float32 pre-test
  //     |fl(a x + b y + c)| <= gate den + E,   E = 2.5 * 2^-24 (2 omax + |c|)  (a rigorous bound on the float32
  // evaluation: with F32R a, b, c ARE float32 values, helpers.py:364, x, y are float32 input, only the two fused
  // multiply-adds round; omax = largest coordinate of the frame), and whatever passes is decided by the exact double
  // expression of helpers.py:373.  The rare hits go through a 64-entry list of the wave in LDS and come out in
  // (distance, index) order.
  __device__ __forceinline__ void match_pairs_wide(int rlo, int rhi, int clo) {
    fresh_tid();
    constexpr int W = T / 64;
    // a wave visits cameras clo + wave, + W, ...: at most NV of them (C <= 64); a lane holds ONE (root, visit) pair of the
    // sub-batch: root q = lane / NV of up to NQ roots, visit v = lane % NV.  (Round 6: before, every visit computed the lines
    // of its camera for a batch of 64 roots -- two square roots and a division in double for the ONE or dozen roots of a chain
    // step, eight times per wave; now all the sub-batch's lines, exact decisions and stores are one pass per wave.)
    constexpr int VB = W >= 16 ? 2 : 3, NV = 1 << VB, NQ = 64 >> VB;
    static_assert(W * NV >= 64, "a wave's visits cover every camera");
    const int lane = tid & 63, wave = tid >> 6;
    const int MW = (M + 63) / 64;
    const double om = (double)__int_as_float(misc[MI_OMAX]);
    const float finf = __int_as_float(0x7f800000);
    const int ql = lane >> VB, vl = lane & (NV - 1);
    const int il = clo + wave + W * vl;  // this lane's camera
    for (int sb = rlo; sb < rhi; sb += NQ) {  // workgroup-uniform
      const int r = sb + ql;
      // (a root meets the cameras AFTER its own: always the case for the roots of a chain step -- they were created at
      // camera clo - 1 -- and the test that matters for provisional roots, which come from many cameras)
      const bool have = r < rhi && il < C && (int)root_cam[r < rhi ? r : rlo] < il;
      const unsigned long long havem = __ballot(have);
      if (!havem) continue;  // wave-uniform
      double la = 0, lb = 0, lc = 0, lden = 1, lrden = 1;
      if (have) {
        // cv.computeCorrespondEpilines on a float32 point: double math, scale by 1/sqrt(a^2+b^2), float32 result
        // (helpers.py:363-364)
        const int rc = root_cam[r], rbl = root_blob[r];
        epiline_wide(cv.F + 9 * ((size_t)rc * C + il), bxy[(size_t)rc * M + rbl], la, lb, lc, lden, lrden);
      }
      // pre-test threshold, rounded up; +inf (everything goes to the exact test) without the float32 line
      const float a32 = (float)la, b32 = (float)lb, c32 = (float)lc;
      float thr = finf;
      if (F32R) thr = __double2float_ru(p.gate_px * lden * (1.0 + 1e-12) + (2.5 * 0x1p-24) * (2.0 * om + fabs(lc)) * (1.0 + 1e-6));
      int my_nh = 0, my_k0 = 0;
      // the usual outcome of a (root, camera) pair: ONE blob passes the pre-test (the marker's own blob).  It is handed
      // to the lane that holds the pair's line (its coordinates travel, not the line), and the exact decision is
      // taken for all the pairs at once after the visits.
      float cand_x = 0.f, cand_y = 0.f;
      int cand_k = -1;
      for (int v = 0; v < NV; v++) {  // wave-uniform: the wave's v-th camera
        const int i = clo + wave + W * v;
        if (i >= C) break;
        unsigned long long vm = 0ull;  // lanes of visit v: bits v, v + NV, ...
#pragma unroll
        for (int t = 0; t < NQ; t++) vm |= 1ull << (t * NV);
        unsigned long long m = havem & (vm << v);
        if (!m) continue;
        const int Mi = cnt[i];
        const float2* row = bxy + (size_t)i * M;
        float2 bl[4];
        bool ok[4];
#pragma unroll
        for (int sg = 0; sg < 4; sg++) {
          const int k = 64 * sg + lane;
          ok[sg] = k < Mi;
          // slots beyond the camera's count hold +inf: the pre-test's |fl(a x + b y + c)| <= thr is false for them
          // (inf or NaN), no separate validity test in the inner loop
          bl[sg] = ok[sg] ? row[k] : make_float2(finf, finf);
        }
        while (m) {  // wave-uniform: the pair held by lane q against the camera's blobs
          const int q = __ffsll((long long)m) - 1;
          m &= m - 1;
          const float fa = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a32), q));
          const float fb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b32), q));
          const float fc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c32), q));
          const float ft = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(thr), q));
          bool pm[4];
          unsigned long long pmask[4];
#pragma unroll
          for (int sg = 0; sg < 4; sg++) {
            pm[sg] = fabsf(fmaf(fa, bl[sg].x, fmaf(fb, bl[sg].y, fc))) <= ft;
            if (!F32R) pm[sg] = ok[sg];  // (thr = +inf would let the +inf padding's NaN through as "false": keep it explicit)
            pmask[sg] = __ballot(pm[sg]);
          }
#ifdef MOCAP_DEBUG_PRETEST  // self-check build: the exact decision (helpers.py:373,375) for EVERY blob the float32 pre-test
          {                    // rejected; a blob inside the gate among them is a false negative (must never happen)
            auto bc = [&](double v_) {
              const long long bits = __double_as_longlong(v_);
              return __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(bits >> 32), q) << 32) |
                                          (unsigned int)__builtin_amdgcn_readlane((int)bits, q));
            };
            const double qa = bc(la), qb = bc(lb), qc = bc(lc), qden = bc(lden), qrden = bc(lrden);
            int fneg = 0;
#pragma unroll
            for (int sg = 0; sg < 4; sg++)
              if (ok[sg] && !pm[sg] && div_by(fabs(qa * (double)bl[sg].x + qb * (double)bl[sg].y + qc), qden, qrden) < p.gate_px) fneg++;
            if (lane == 0) atomicAdd(&p.status[p.n_frames], 1);
            if (fneg) atomicAdd(&p.status[p.n_frames + 1], fneg);
          }
#endif
          const unsigned long long any01 = pmask[0] | pmask[1], any23 = pmask[2] | pmask[3];
          if (!(any01 | any23)) continue;
          const int npass = __popcll(pmask[0]) + __popcll(pmask[1]) + __popcll(pmask[2]) + __popcll(pmask[3]);
          if (npass == 1) {
            float cx, cy;
            int ck;
            if (pmask[0]) {
              const int l1 = __ffsll((long long)pmask[0]) - 1;
              cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bl[0].x), l1));
              cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bl[0].y), l1));
              ck = l1;
            } else if (pmask[1]) {
              const int l1 = __ffsll((long long)pmask[1]) - 1;
              cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bl[1].x), l1));
              cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bl[1].y), l1));
              ck = 64 + l1;
            } else if (pmask[2]) {
              const int l1 = __ffsll((long long)pmask[2]) - 1;
              cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bl[2].x), l1));
              cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bl[2].y), l1));
              ck = 128 + l1;
            } else {
              const int l1 = __ffsll((long long)pmask[3]) - 1;
              cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bl[3].x), l1));
              cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bl[3].y), l1));
              ck = 192 + l1;
            }
            if (lane == q) {
              cand_x = cx;
              cand_y = cy;
              cand_k = ck;
            }
            continue;
          }
          // ---- rare: several blobs within reach of the gate
          {
            bool pc[4];
#pragma unroll
            for (int sg = 0; sg < 4; sg++) pc[sg] = pm[sg] && ok[sg];
            resolve_pair(q, sb + (q >> VB), i, bl, pc, la, lb, lc, lden, lrden, my_nh, my_k0);
          }
        }
      }
      if (cand_k >= 0) {  // single candidates: helpers.py:373 in double, strict < (helpers.py:375,383); a lone hit claims itself
        if (div_by(fabs(la * (double)cand_x + lb * (double)cand_y + lc), lden, lrden) < p.gate_px) {
          if (spec_base < 0) atomicOr(&claimw[(size_t)il * MW + (cand_k >> 6)], 1ull << (cand_k & 63));
          my_nh = 1;
          my_k0 = cand_k;
        }
      }
      if (have) {
        h0[(size_t)r * C + il] = (uint8_t)my_k0;
        set_hit_code(r, il, my_nh);
      }
    }
  }

  // ---------------------------------------------------------------- the chain over cameras jlo .. C-1 in one pass (round 6)
  // The sequential chain (match_wide below) pays one round of memory latency per root-creating camera: a blob no root has
  // claimed becomes a root, is matched against the cameras after it, claims its closest hits there, and only then is the
  // next camera's unclaimed set known -- 9 to 16 such rounds per stress frame, nearly all for ONE spurious root each.
  // But a root's lines, gates, orders and hit lists do not depend on the claims at all; only its EXISTENCE does.  So: every
  // blob of cameras jlo .. C-1 that is unclaimed NOW (by the roots of cameras 0 .. jlo-1) is a provisional root -- the only
  // blobs that can still become roots, a dozen per stress frame -- and all of them are matched at once, one lane per
  // provisional root, against the cameras after their own, claiming nothing.  Then one wave replays the reference's order
  // (helpers.py:402-406): camera by camera, a provisional root whose blob is still unclaimed is real and its closest hits
  // claim their blobs in the later cameras (helpers.py:391); the real ones move down to consecutive rows in (camera, blob)
  // order -- the numbering of the sequential chain.  Falls back to that chain (returns false, nothing changed) when there
  // are more than MOCAP_WIDE_SPEC_T provisional roots or no rows for them, or when a hit list holds a second blob with the closest hit's coordinates
  // (claimed with it by value: not replayable from the closest hit alone).  All lanes; synchronised on entry and exit.
  // spec_begin: the provisional roots and their rows; returns their number, 0 (no blob left unclaimed: the chain is complete) or -1.
  __device__ __forceinline__ int spec_begin(int jlo, int n_roots) {
    fresh_tid();
    const int MW = (M + 63) / 64, lane = tid & 63;
    if (tid < 64) frame_prio<MOCAP_FRAME_PRIO_SERIAL>();
    if (tid < 64) {
      int u = 0;
      if (lane >= jlo && lane < C) {
        const int Mj = cnt[lane];
        for (int w = 0; w * 64 < Mj; w++) {
          const unsigned long long valid = Mj - 64 * w >= 64 ? ~0ull : ((1ull << (Mj - 64 * w)) - 1ull);
          u += __popcll(~claimw[(size_t)lane * MW + w] & valid);
        }
      }
      const uint32_t incl = wave_inclusive_scan((uint32_t)u, lane);
      const int total = __builtin_amdgcn_readlane((int)incl, 63);
      const bool fits = total <= MOCAP_WIDE_SPEC_T && n_roots + total <= R;
      if (fits && u) {  // lane = camera: its unclaimed blobs in blob order, behind those of the cameras before it
        int idx = n_roots + (int)incl - u;
        const int Mj = cnt[lane];
        for (int w = 0; w * 64 < Mj; w++) {
          const unsigned long long valid = Mj - 64 * w >= 64 ? ~0ull : ((1ull << (Mj - 64 * w)) - 1ull);
          unsigned long long m = ~claimw[(size_t)lane * MW + w] & valid;
          while (m) {
            const int k = 64 * w + __ffsll((long long)m) - 1;
            m &= m - 1;
            root_cam[idx] = (uint8_t)lane;
            root_blob[idx] = (uint16_t)k;
            wsrow[idx] = (uint16_t)idx;
            idx++;
          }
        }
      }
      if (lane == 0) {
        misc[MI_SPEC_N] = fits ? total : -1;
        misc[MI_SPEC_OVER] = 0;
        misc[MI_SPEC_OVER + 1] = 0;
      }
    }
    frame_prio<MOCAP_FRAME_PRIO_CHAIN>();
    __syncthreads();
    const int nP = misc[MI_SPEC_N];
    __syncthreads();  // (the slot is reused as the fall-back flag below)
    if (nP <= 0) return nP;
    if (tid == 0) misc[MI_SPEC_N] = 0;
    for (int q = tid; q < nP; q += T) {  // a root's own camera counts as one hit: its blob
      const int r = n_roots + q, rc = root_cam[r];
      h0[(size_t)r * C + rc] = (uint8_t)root_blob[r];
      set_hit_code(r, rc, 1);
    }
    __syncthreads();
    return nP;
  }
  // spec_finish: after the provisional roots [n_roots, n_roots + nP) have met their cameras (match_pairs_wide with spec_base set)
  // and a barrier.  True: the chain is complete, n_roots updated.  False: rows clean, the sequential chain takes over.
  __device__ __forceinline__ bool spec_finish(int nP, int& n_roots) {
    fresh_tid();
    const int MW = (M + 63) / 64, lane = tid & 63;
    if (misc[MI_SPEC_N]) {  // workgroup-uniform: the sequential chain starts over from camera jlo with clean rows
      for (int q = tid; q < 2 * nP; q += T) nhc[2 * (size_t)n_roots + q] = 0ull;
      __syncthreads();
      return false;
    }
    if (tid < 64) frame_prio<MOCAP_FRAME_PRIO_SERIAL>();
    if (tid < 64) {
      const int row = n_roots + lane;
      const bool mine = lane < nP;
      const int jp = mine ? (int)root_cam[row] : -1, kp = mine ? (int)root_blob[row] : 0;
      const unsigned long long present = mine ? nhc[2 * (size_t)row] : 0ull;
      auto bcast64 = [&](unsigned long long v, int l) {
        return ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, l);
      };
      unsigned long long realm = 0ull;
      int p0 = 0;
      while (p0 < nP) {  // wave-uniform: the provisional roots are in camera order, one group per camera
        const int j = __builtin_amdgcn_readlane(jp, p0);
        const bool at = jp == j;
        bool real = false;
        if (at) real = !((claimw[(size_t)j * MW + (kp >> 6)] >> (kp & 63)) & 1ull);
        const unsigned long long grp = __ballot(at);
        unsigned long long m = __ballot(real);
        realm |= m;
        while (m) {  // a real root's closest hits claim their blobs in the cameras after its own: lane = camera
          const int q = __ffsll((long long)m) - 1;
          m &= m - 1;
          const unsigned long long pres = bcast64(present, q);
          if (lane > j && lane < C && ((pres >> lane) & 1ull)) {
            const int k = h0[(size_t)(n_roots + q) * C + lane];
            atomicOr(&claimw[(size_t)lane * MW + (k >> 6)], 1ull << (k & 63));
          }
        }
        wave_lds_sync();
        p0 += __popcll(grp);
      }
      // the real ones move down to consecutive rows, in order (a row number only ever decreases: no row is overwritten
      // before it has moved)
      int d = n_roots;
      unsigned long long m = realm;
      while (m) {
        const int q = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int s = n_roots + q;
        if (s != d) {
          const unsigned long long pres = nhc[2 * (size_t)s], sev = nhc[2 * (size_t)s + 1];
          if (lane < C) h0[(size_t)d * C + lane] = h0[(size_t)s * C + lane];  // (the multi-hit lists stay in workspace row s: wsrow)
          if (lane == 0) {
            nhc[2 * (size_t)d] = pres;
            nhc[2 * (size_t)d + 1] = sev;
            root_cam[d] = root_cam[s];
            root_blob[d] = root_blob[s];
            wsrow[d] = (uint16_t)s;
          }
          wave_lds_sync();
        }
        d++;
      }
      const int nreal = __popcll(realm);
      for (int r = n_roots + nreal + lane; r < n_roots + nP; r += 64) {  // rows left behind: no hits (nothing else of a row is read without its masks)
        nhc[2 * (size_t)r] = 0ull;
        nhc[2 * (size_t)r + 1] = 0ull;
      }
      if (lane == 0) {
        const unsigned long long over = ((unsigned long long)(uint32_t)misc[MI_SPEC_OVER + 1] << 32) | (uint32_t)misc[MI_SPEC_OVER];
        if (over & realm) misc[MI_STATUS] |= MOCAP_ST_HIT_OVERFLOW_;
        misc[MI_NROOTS] = n_roots + nreal;
      }
    }
    frame_prio<MOCAP_FRAME_PRIO_CHAIN>();
    __syncthreads();
    n_roots = misc[MI_NROOTS];
    return true;
  }

  // The unclaimed blobs of camera j become roots n_roots .., in blob order (helpers.py:402-406; wave 0 compacts), with their
  // rows up to their own camera.  Returns the new number of roots.  All lanes; synchronised on entry, not on exit (the rows'
  // bytes for the cameras after j belong to the matching pass that follows).
  __device__ __forceinline__ int create_roots(int j, int n_roots) {
    fresh_tid();
    const int MW = (M + 63) / 64;
    if (tid < 64) frame_prio<MOCAP_FRAME_PRIO_SERIAL>();
    if (tid < 64) {
      const int Mj = cnt[j];
      int base_root = n_roots;
      for (int w = 0; w * 64 < Mj; w++) {
        const int k = 64 * w + tid;
        const unsigned long long cl = claimw[(size_t)j * MW + w];
        const bool flag = k < Mj && !((cl >> tid) & 1ull);
        const unsigned long long mask = __ballot(flag);
        const int pos = __popcll(mask & ((1ull << tid) - 1ull));
        if (flag) {
          const int rr = base_root + pos;
          if (rr < R) {
            root_cam[rr] = (uint8_t)j;
            root_blob[rr] = (uint16_t)k;
            wsrow[rr] = (uint16_t)rr;
          }
        }
        base_root += __popcll(mask);
      }
      if (tid == 0) {
        if (base_root > R) {
          misc[MI_STATUS] |= MOCAP_ST_ROOT_OVERFLOW_;
          base_root = R;
        }
        misc[MI_NROOTS] = base_root;
      }
    }
    frame_prio<MOCAP_FRAME_PRIO_CHAIN>();
    __syncthreads();
    const int now = misc[MI_NROOTS];
    if (now > n_roots) {  // workgroup-uniform
      // cameras up to the root's own (the pairs with the cameras after it are written by the matching pass, by other waves at
      // the same time: the two must not touch the same bytes)
      const int ncam = (MOCAP_WIDE_DEBUG_SKIP & 2) ? C : j + 1;
      for (int idx = tid; idx < (now - n_roots) * ncam; idx += T) {
        const int r = n_roots + idx / ncam, c = idx % ncam;
        if (c == j) set_hit_code(r, j, 1);
        h0[(size_t)r * C + c] = (uint8_t)root_blob[r];
      }
    }
    return now;
  }

  __device__ __forceinline__ void match_wide(int64_t frame) {
    fresh_tid();
    const int MW = (M + 63) / 64;
    {
      bxy = const_cast<float2*>((const float2*)(p.blobs + (size_t)frame * C * M * 2));  // read in place, never written
      if (tid < C) {
        int n = p.counts[(size_t)frame * C + tid];
        cnt[tid] = n < 0 ? 0 : (n > M ? M : n);
      }
      for (int i = tid; i < C * MW; i += T) claimw[i] = 0ull;
      for (int i = tid; i < 2 * R; i += T) nhc[i] = 0ull;  // every (root, camera) pair: no hit yet
      if (tid == 0) {
        misc[MI_STATUS] = 0;
        misc[MI_OMAX] = 0;
        misc[MI_HEAVY_N] = 0;
      }
    }
    __syncthreads();
    {  // largest coordinate: the float32 pre-test's error bound and EigCut's allowance scale with it
      float om = 0.0f;
      for (int i = tid; i < C * M; i += T) {
        const int c = i / M, k = i - c * M;
        if (k < cnt[c]) {
          const float2 v = bxy[i];
          om = fmaxf(om, fmaxf(fabsf(v.x), fabsf(v.y)));
        }
      }
      if (om > 0.0f) atomicMax(&misc[MI_OMAX], __float_as_int(om));
    }
    const int n0 = cnt[0] < R ? cnt[0] : R;
    {  // roots from camera 0 (helpers.py:349,357)
      for (int r = tid; r < n0; r += T) {
        root_cam[r] = 0;
        root_blob[r] = (uint16_t)r;
        wsrow[r] = (uint16_t)r;
      }
      // a root's own camera counts as one "hit" (its blob), the cameras before it as none: a candidate group is decoded
      // from these two bytes per camera alone
      const int ncam0 = (MOCAP_WIDE_DEBUG_SKIP & 1) ? C : 1;
      for (int idx = tid; idx < n0 * ncam0; idx += T) {
        const int r = idx / ncam0, c = idx - r * ncam0;
        if (c == 0) set_hit_code(r, 0, 1);
        h0[(size_t)r * C + c] = (uint8_t)r;
      }
      if (tid == 0) {
        misc[MI_NROOTS] = n0;
        if (cnt[0] > R) misc[MI_STATUS] |= MOCAP_ST_ROOT_OVERFLOW_;
      }
    }
    __syncthreads();
    // The camera-0 roots meet camera 1 first (one batch per wave: everything else waits for it), camera 1's unclaimed blobs
    // become roots, and BOTH sets meet cameras 2 .. C-1 in one pass, a lane per root: the dozen roots of camera 1 ride in the
    // free lanes of the last batch instead of paying a chain step of their own (2.6 of 29 ms per 12 500 stress frames).
    int n_roots = n0, n_tot = n0, n_main = n0;
    bool pre1 = false;
#if !(MOCAP_WIDE_DEBUG_SKIP & 1)
#pragma nounroll
    for (int ph = 0; ph < 2; ph++) {
      const int clo = 1 + ph, chi = ph ? C : (C < 2 ? C : 2), hi = ph ? n_main : n0;
      if (clo < chi && hi > 0) match_roots_wide<true>(0, hi, clo, chi, ph == 0);
      __syncthreads();
      if (ph == 0 && C > 2 && MOCAP_WIDE_CAM1 && !(MOCAP_WIDE_DEBUG_SKIP & 2)) {
        n_tot = create_roots(1, n0);
        pre1 = true;
        // ... unless they would open a new group of 256 lanes for a handful of roots: those meet their cameras as a chain step
        n_main = n_tot;
        if (n_tot > 256 && (n_tot & 255) < MOCAP_WIDE_CHAIN_T) n_main = n_tot - (n_tot & 255);
      }
    }
#else
    __syncthreads();
#endif
    frame_prio<MOCAP_FRAME_PRIO_CHAIN>();
    bool spec_ok = MOCAP_WIDE_SPEC && !(MOCAP_WIDE_DEBUG_SKIP & 2) && p.wide != 2;
    for (int j = 1; j < C; j++) {
      // cameras j .. C-1 in one speculative pass as soon as few blobs are left unclaimed (spec_begin), else camera j alone
      int nP = -1;
      if ((MOCAP_WIDE_DEBUG_SKIP & 32) && j == 2) break;
      int now = n_roots;
      if (j == 1 && pre1) {
        n_roots = n_main;  // camera 1's roots exist; those from n_main on have not met the cameras after it yet
        now = n_tot;
      } else {
        if (spec_ok) {
          nP = spec_begin(j, n_roots);
          if (nP == 0) break;
        }
        now = nP < 0 ? create_roots(j, n_roots) : n_roots + nP;
      }
      if (now > n_roots) {  // workgroup-uniform
        if (j + 1 < C && !(MOCAP_WIDE_DEBUG_SKIP & 2)) {
          // few new roots (the usual chain step), or provisional ones: the blobs stay in registers and the roots are broadcast; many: one lane per root
          spec_base = nP > 0 ? n_roots : -1;
          if (nP > 0 || now - n_roots < MOCAP_WIDE_CHAIN_T) match_pairs_wide(n_roots, now, j + 1); else match_roots_wide<false>(n_roots, now, j + 1, C, false);
          spec_base = -1;
        }
        __syncthreads();
      }
      if (nP > 0) {
        if (MOCAP_WIDE_DEBUG_SKIP & 64) break;
        if (spec_finish(nP, n_roots)) break;
        spec_ok = false;  // (rare: see spec_finish) camera j again, and the rest of the frame, sequentially
        j--;
        continue;
      }
      n_roots = now;
    }
    count_candidates();
    if constexpr (WIDE && HEAVY) {
      if (misc[MI_HEAVY_N]) export_heavy(frame);
    }
  }

  // Heavy roots (FrameArgs::heavy_bb): the root's hit counts and hit lists leave for the heavy-root search (csrc/heavy_bb.hip).
  // All lanes; the workgroup is synchronised on entry and on exit.
  __device__ __forceinline__ void export_heavy(int64_t frame) {
    const int nhv = misc[MI_HEAVY_N] < kMaxHeavyPerFrame ? misc[MI_HEAVY_N] : kMaxHeavyPerFrame;
    for (int h = 0; h < nhv; h++) {
      const int r = misc[MI_HEAVY_R0 + h];
      if (tid == 0) misc[MI_HEAVY_SLOT] = atomicAdd(p.heavy_count, 1);
      __syncthreads();
      const int slot = misc[MI_HEAVY_SLOT];
      if (slot < p.heavy_cap) {
        unsigned char* rec = p.heavy_recs + (size_t)slot * p.heavy_stride;
        if (tid == 0) {
          HeavyRecHdr hd;
          hd.frame = (int32_t)frame;
          hd.root = r;
          hd.outslot = outslot[r];
          hd.rc = root_cam[r];
          hd.rb = root_blob[r];
          hd.omax_bits = misc[MI_OMAX];
          int views = 0;
          for (int c = 0; c < C; c++) views += nhits(r, c) ? 1 : 0;
          hd.views = views;
          hd.Hs = Hs;
          *reinterpret_cast<HeavyRecHdr*>(rec) = hd;
        }
        uint16_t* nc = reinterpret_cast<uint16_t*>(rec + heavy_rec_counts_off());
        for (int c = tid; c < C; c += T) nc[c] = (uint16_t)nhits(r, c);
        uint8_t* hl = rec + heavy_rec_hits_off(C);
        for (int idx = tid; idx < C * Hs; idx += T) {
          const int c = idx / Hs, d = idx - c * Hs;
          hl[idx] = (uint32_t)d < nhits(r, c) ? (uint8_t)hit_at(r, c, (uint32_t)d) : (uint8_t)0;
        }
      } else if (tid == 0) {
        misc[MI_STATUS] |= MOCAP_ST_CAND_OVERFLOW_;  // no room for the record: the frame stays flagged
      }
      __syncthreads();
    }
  }

  // C: candidate counts per root (all lanes; the roots and hit lists are in place and the workgroup is synchronised)
  __device__ __forceinline__ void count_candidates() {
    fresh_tid();
    const int nroots = misc[MI_NROOTS];
    for (int r = tid; r < nroots; r += T) {
      const int rc = root_cam[r];
      unsigned long long total = 1;
      int views = 1, na = 0;
      bool over = false;
      for (int c = rc + 1; c < C; c++) {
        const unsigned n = nhits(r, c);
        if constexpr (!WIDE) {
          if (n > 1) act[(size_t)r * C + na++] = (uint8_t)c;  // the digits an odometer step can change
        }
        if (n) {
          views++;
          total *= n;
          if (total > (unsigned long long)p.G_cap) {
            over = true;
            total = 1;
          }
        }
      }
      bool heavy = false;
      if constexpr (WIDE && HEAVY) {
        // re-submit pass: the root is handed to the heavy-root search; here it counts ONE candidate, group 0 (the closest hit
        // in every camera), whose error is the bound that search starts from
        if (over && views > 1) {
          const int idx = atomicAdd(&misc[MI_HEAVY_N], 1);
          if (idx < kMaxHeavyPerFrame) {
            misc[MI_HEAVY_R0 + idx] = r;
            heavy = true;
          }
        }
      }
      if (over && !heavy) atomicOr(&misc[MI_STATUS], MOCAP_ST_CAND_OVERFLOW_);
      if constexpr (!WIDE) nact[r] = (uint8_t)na;
      rbound[r] = 0x7ff0000000000000ull;
      gcnt[r] = views > 1 ? (heavy ? 1u : (over ? 0u : (uint32_t)total)) : 0u;  // helpers.py:413-414 drops 1-view roots
    }
    __syncthreads();
    if (tid < 64) frame_prio<MOCAP_FRAME_PRIO_SERIAL>();
    if (tid < 64) {  // candidate offsets and output slots: wave scans over the roots, 64 at a time
      const int lane = tid;
      unsigned long long carry = 0;
      int slots = 0;
      for (int base = 0; base < nroots; base += 64) {
        const int r = base + lane;
        const uint32_t g = r < nroots ? gcnt[r] : 0u;
        const uint32_t incl = wave_inclusive_scan(g, lane);  // (64 x 2^24 < 2^32: a chunk cannot overflow)
        const unsigned long long nz = __ballot(g != 0u);
        if (r < nroots) {
          goff[r] = (uint32_t)carry + incl - g;
          outslot[r] = g ? slots + __popcll(nz & ((1ull << lane) - 1ull)) : -1;
        }
        carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        slots += __popcll(nz);
      }
      if (lane == 0) {
        if (carry >> 32) misc[MI_STATUS] |= MOCAP_ST_CAND_OVERFLOW_;  // 32-bit candidate offsets per frame
        goff[nroots] = (uint32_t)carry;
        misc[MI_NOUT] = slots;
        misc[MI_G] = misc[MI_STATUS] ? 0 : (int32_t)(uint32_t)carry;
      }
    }
    frame_prio<MOCAP_FRAME_PRIO_CHAIN>();
    __syncthreads();
  }

  // ---------------------------------------------------------------- phase D
  // Group index -> observations.  The reference materialises every group as a nested list
  // (copy.deepcopy, helpers.py:394-400); here a group is the mixed-radix number gl whose digit for
  // camera c (> root camera, >= 1 hit) picks the gl_c-th closest hit; camera rc+1 is the fastest
  // digit (probed against the reference).  Each lane keeps the digits and the observations of ITS
  // current group in its own LDS column (dig / cxy, [C][T]: conflict-free), so stepping to the next
  // group rewrites only the digits that change instead of decoding all C of them.
  template <bool FIRST>  // FIRST: gl == 0, every digit is 0 (root crossing inside a lane's run)
  __device__ void load_group(int r, uint32_t gl) {
    const int rc = root_cam[r];
    const uint16_t rb = root_blob[r];
    const uint16_t* nhr = nh + (size_t)r * C;
    const uint8_t* hr = hits + (size_t)r * C * Hs;
    const float qn = __int_as_float(0x7fc00000);
    uint32_t rem = gl;
    for (int c = 0; c < C; c++) {
      uint16_t s = kNone;
      uint32_t dgt = 0;
      if (c == rc) {
        s = rb;
      } else if (c > rc) {
        const uint32_t n = nhr[c];
        if (n) {
          if (!FIRST) {
            uint32_t qd;
            divmod_small(rem, n, qd, dgt);
            rem = qd;
          }
          s = hr[(size_t)c * Hs + dgt];
        }
      }
      dig[(size_t)c * T] = (uint8_t)dgt;
      if (TABLE)
        cix[(size_t)c * T] = s == kNone ? (uint8_t)0xFF : (uint8_t)s;
      else
        cxy[(size_t)c * T] = s == kNone ? make_float2(qn, qn) : bxy[(size_t)c * M + s];
    }
  }

  __device__ void next_group(int r) {
    const uint8_t* a = act + (size_t)r * C;
    const int na = nact[r];
    const uint16_t* nhr = nh + (size_t)r * C;
    const uint8_t* hr = hits + (size_t)r * C * Hs;
    for (int k = 0; k < na; k++) {
      const int c = a[k];
      uint32_t d = (uint32_t)dig[(size_t)c * T] + 1u;
      const bool wrap = d >= nhr[c];
      if (wrap) d = 0;
      dig[(size_t)c * T] = (uint8_t)d;
      if (TABLE)
        cix[(size_t)c * T] = hr[(size_t)c * Hs + d];
      else
        cxy[(size_t)c * T] = bxy[(size_t)c * M + hr[(size_t)c * Hs + d]];
      if (!wrap) break;
    }
  }

  // Evaluate candidates [g_lo, g_hi); per (lane, root) segment winners land in seg_* .
  __device__ __forceinline__ void evaluate(uint32_t g_lo, uint32_t g_hi) {
    fresh_tid();
    const int nroots = misc[MI_NROOTS];
    const uint32_t q = (g_hi - g_lo + T - 1) / T;
    uint32_t g = g_lo + (uint32_t)tid * q;
    const uint32_t g_end = (g + q < g_hi) ? g + q : g_hi;
    if (g < g_end) {
      int lo = 0, hi = nroots - 1;  // last root whose offset is <= g (empty roots share offsets)
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (goff[mid] <= g) lo = mid; else hi = mid - 1;
      }
      int r = lo;
      uint32_t r_beg = goff[r], r_end = goff[r + 1];
      if constexpr (!WIDE) load_group<false>(r, g - r_beg);
      double best_e = __builtin_huge_val(), best_X[3] = {0, 0, 0};
      uint32_t best_g = 0;
      bool have = false;
      // a blob never has NaN coordinates inside a multi-view group (NaN fails the gate), so NaN
      // marks "camera not in the group"
      struct ColumnObs {
        const float2* col;  // this lane's column: camera c at col[c * T]
        __device__ __forceinline__ unsigned long long raw(int c) const {
          return *reinterpret_cast<const unsigned long long*>(col + (size_t)c * T);
        }
        __device__ __forceinline__ bool decode(unsigned long long w, double& x, double& y) const {
          const float fx = __uint_as_float((uint32_t)w), fy = __uint_as_float((uint32_t)(w >> 32));
          if (fx != fx) return false;
          x = (double)fx;
          y = (double)fy;
          return true;
        }
        __device__ __forceinline__ bool operator()(int c, double& x, double& y) const { return decode(raw(c), x, y); }
      };
      const ColumnObs obs{cxy};
      // Wide frames: no per-lane group column at all.  A candidate group is DECODED while it is evaluated: camera c of
      // group gl of root r is the root's only hit there (one LDS byte -- almost every pair), or, for the few cameras with
      // several hits, the digit of gl in the mixed radix of those cameras (helpers.py:394-400 order: the first camera
      // after the root's is the fastest digit), peeled off by one small division as the pass walks the cameras in
      // ascending order.  Each pass (DLT, depths, reprojection) starts at camera 0, which resets the remainder.  The
      // blobs are read in place from the input batch (L2); nothing per candidate is written anywhere.
      // CONTRACT of raw(): a pass calls it for the cameras in ascending order, each exactly once, starting at camera 0 --
      // the remainder is state.  The two places of mocap_device.hpp that bend this are harmless by construction and rely
      // on it staying so: (1) score_point skips a block of four cameras for a lane whose running sum is over the limit;
      // what such a lane decodes afterwards (the tail cameras, C % 4) is a wrong but in-range blob (the digit is still
      // `rem % n` < n) of a group whose error is already +inf; (2) the padded batches of triangulate_and_score /
      // solve_and_score call raw(C - 1) again for the slots past the last camera: extra digits are peeled AFTER the last
      // real one and the values are discarded (`c0 + u < C`).  A pass that needs anything else must decode statelessly.
      struct WideObs {
        const uint8_t* h0r;     // LDS [C]: closest hit per camera of the current root
        unsigned long long present, several;  // the current root's camera masks (FrameState::nhc)
        const uint16_t* nh16r;  // workspace [C]: exact counts (read only where the mask says "several")
        const uint8_t* hitsr;   // workspace [C][Hs]: full hit lists of the multi-hit pairs
        const float2* blobs;    // the frame's blobs [C][M]
        int M, Hs;
        uint32_t gl;
        uint32_t rem;
        __device__ __forceinline__ unsigned long long raw(int c) {
          if (c == 0) rem = gl;
          unsigned long long w = 0x7fc000007fc00000ull;  // NaN, NaN: camera not in the group
          if ((present >> c) & 1ull) {
            uint32_t k = h0r[c];
            if ((several >> c) & 1ull) {
              const uint32_t n = nh16r[c];
              uint32_t qd, dg;
              divmod_small(rem, n, qd, dg);
              rem = qd;
              if (dg) k = hitsr[(size_t)c * Hs + dg];
            }
            w = *reinterpret_cast<const unsigned long long*>(blobs + (size_t)c * M + k);
          }
          return w;
        }
        __device__ __forceinline__ bool decode(unsigned long long w, double& x, double& y) const {
          const float fx = __uint_as_float((uint32_t)w), fy = __uint_as_float((uint32_t)(w >> 32));
          if (fx != fx) return false;
          x = (double)fx;
          y = (double)fy;
          return true;
        }
        __device__ __forceinline__ bool operator()(int c, double& x, double& y) { return decode(raw(c), x, y); }
      };
      const size_t wr0 = WIDE ? (size_t)wsrow[r] : (size_t)r;
      WideObs wobs{h0 + (size_t)r * C, nhc[2 * (size_t)r], nhc[2 * (size_t)r + 1], nh + wr0 * C, hits + wr0 * C * Hs, bxy, M, Hs, g - r_beg, 0u};
      // table mode: the column holds blob indices; 0xFF marks "camera not in the group"
      auto contrib = [&](int c, double (&B)[10]) -> bool {
        const uint32_t k = cix[(size_t)c * T];
        if (k == 0xFFu) return false;
        const double* t = bt + ((size_t)c * M + k) * 10;
#pragma unroll
        for (int e = 0; e < 10; e++) B[e] = B[e] + t[e];
        return true;
      };
      auto obs_ix = [&](int c, double& x, double& y) -> bool {
        const uint32_t k = cix[(size_t)c * T];
        if (k == 0xFFu) return false;
        const float2 v = bxy[(size_t)c * M + k];
        x = (double)v.x;
        y = (double)v.y;
        return true;
      };
      // Selection only asks whether a group beats the best one of its root, so the reprojection of a group stops
      // once its running sum of squares is out of reach of the smallest error ANY lane has found for the root so
      // far (rbound, LDS, 64-bit unsigned min on the bit pattern of a non-negative double).  The global minimum
      // and its ties are never cut short (their partial sums stay below every bound), so the result -- first
      // minimum in candidate order -- does not depend on which lane got where first; only the amount of work does.
      const double inf = __builtin_huge_val();
      const bool prune = p.prune != 0;
      EigCut ec;
      if (prune && p.p3max2 > 0.0) {
        const double om = (double)__int_as_float(misc[MI_OMAX]);
        ec.p3max2 = p.p3max2;
        ec.o2slack = (1100.0 * 0x1p-46) * (om * om);
      }
      while (true) {
        double X[3], e;
        const double bound = prune ? __longlong_as_double((long long)rbound[r]) : inf;
        if constexpr (WIDE && (MOCAP_WIDE_DEBUG_SKIP & 4)) {
          e = 1.0 + (double)(g & 7u);
          X[0] = X[1] = X[2] = 0.0;
        } else if constexpr (TABLE)
          triangulate_and_score_tab<true, F32R>(cv, contrib, obs_ix, X, e, bound, ec);
        else if constexpr (WIDE)
          // (no depth form of the eigenvalue bound here: a wide frame's rival groups differ from the right one in one or two
          // of ~60 views by a blob within the gate of the line -- near-winners no bound separates -- so that pass over the
          // cameras, a sixth of a group's instructions, cut nothing; exactness is unaffected: a cut-off only ever skips work)
          triangulate_and_score<UNIFORM_K, true, F32R, MOCAP_WIDE_BATCH, (MOCAP_WIDE_DEPTH_CUT != 0)>(cv, wobs, wobs, X, e, bound, ec);
        else
          triangulate_and_score<UNIFORM_K, true, F32R, 1>(cv, obs, obs, X, e, bound, ec);
#ifdef MOCAP_DEBUG_EIGCHECK  // self-check of the cut-offs: a group that was cut must not beat the bound it was cut against
        if (!(e < inf)) {
          double X2[3], e2;
          if constexpr (TABLE)
            triangulate_and_score_tab<true, F32R>(cv, contrib, obs_ix, X2, e2);
          else if constexpr (WIDE)
            triangulate_and_score<UNIFORM_K, true, F32R>(cv, wobs, wobs, X2, e2);
          else
            triangulate_and_score<UNIFORM_K, true, F32R>(cv, obs, obs, X2, e2);
          if (e2 <= bound)
            printf("EIGCHECK frame-item tid %d root %d g %u bound %.17g true %.17g omax %g\n", tid, r, g - r_beg, bound, e2,
                   (double)__int_as_float(misc[MI_OMAX]));
        }
#endif
        if (e < best_e) {  // strict <: first minimum within the lane's ascending run (best_e starts at +inf; a
          best_e = e;      // group that was cut short or whose error is not finite never enters)
          best_g = g - r_beg;
          best_X[0] = X[0];
          best_X[1] = X[1];
          best_X[2] = X[2];
          if (prune) atomicMin(&rbound[r], (unsigned long long)__double_as_longlong(e));
        } else if (!have && !(e < inf)) {  // first group of the segment without a finite error (cut short, or a
          best_e = e;                      // degenerate point): it stands unless something smaller follows; a NaN
          best_g = g - r_beg;              // sticks, as np.argmin keeps the first NaN (helpers.py:418)
          best_X[0] = X[0];
          best_X[1] = X[1];
          best_X[2] = X[2];
        }
        have = true;
        if (++g >= g_end) break;
        if (g >= r_end) {  // leaving root r: flush the (lane, root) segment
          const int s = tid + outslot[r];
          seg_e[s] = best_e;
          seg_g[s] = best_g;
          seg_x[3 * s + 0] = best_X[0];
          seg_x[3 * s + 1] = best_X[1];
          seg_x[3 * s + 2] = best_X[2];
          have = false;
          best_e = inf;
          do { r++; } while (goff[r + 1] <= g);
          r_beg = goff[r];
          r_end = goff[r + 1];
          if constexpr (WIDE) {
            wobs.h0r = h0 + (size_t)r * C;
            wobs.present = nhc[2 * (size_t)r];
            wobs.several = nhc[2 * (size_t)r + 1];
            wobs.nh16r = nh + (size_t)wsrow[r] * C;
            wobs.hitsr = hits + (size_t)wsrow[r] * C * Hs;
            wobs.gl = 0;
          } else {
            load_group<true>(r, 0);
          }
        } else {
          if constexpr (WIDE) wobs.gl = g - r_beg; else next_group(r);
        }
      }
      const int s = tid + outslot[r];
      seg_e[s] = best_e;
      seg_g[s] = best_g;
      seg_x[3 * s + 0] = best_X[0];
      seg_x[3 * s + 1] = best_X[1];
      seg_x[3 * s + 2] = best_X[2];
    }
    __syncthreads();
  }

  // first minimum over the (lane, root) segments of root r inside [g_lo, g_hi); false if the
  // root has no candidate in the range
  __device__ __forceinline__ bool root_winner(int r, uint32_t g_lo, uint32_t g_hi, double& eb, uint32_t& gb, double (&Xb)[3]) {
    const uint32_t a = goff[r] > g_lo ? goff[r] : g_lo;
    const uint32_t b = goff[r + 1] < g_hi ? goff[r + 1] : g_hi;
    if (a >= b) return false;
    const uint32_t q = (g_hi - g_lo + T - 1) / T;
    const int k = outslot[r];
    const uint32_t t0 = (a - g_lo) / q, t1 = (b - 1 - g_lo) / q;
    int sbest = (int)t0 + k;
    eb = seg_e[sbest];
    for (uint32_t t = t0 + 1; t <= t1; t++) {
      const int s = (int)t + k;
      const double e = seg_e[s];
      if (e < eb) {  // earlier segment wins ties (np.argmin, helpers.py:418)
        eb = e;
        sbest = s;
      }
    }
    gb = seg_g[sbest];
    Xb[0] = seg_x[3 * sbest + 0];
    Xb[1] = seg_x[3 * sbest + 1];
    Xb[2] = seg_x[3 * sbest + 2];
    return true;
  }

  // ---------------------------------------------------------------- phase E
  __device__ __forceinline__ void write_point(int64_t frame, int r, double e, uint32_t gl, const double (&X)[3]) {
    const size_t o = (size_t)frame * R + outslot[r];
    store_point(p, o, X);  // incl. the fused world-coordinate epilogue (helpers.py:96-103)
    p.err[o] = e;
    uint32_t rem = gl;  // decode the winning group
    const int rc = root_cam[r];
    int16_t* co = p.corr + o * C;
    for (int c = 0; c < C; c++) {
      int16_t s = -1;
      if (c == rc) {
        s = (int16_t)root_blob[r];
      } else if (c > rc) {
        const uint32_t n = nhits(r, c);
        if (n) {
          uint32_t qd, dgt;
          divmod_small(rem, n, qd, dgt);
          s = (int16_t)hit_at(r, c, dgt);
          rem = qd;
        }
      }
      co[c] = s;
    }
  }

  __device__ __forceinline__ void write_frame_header(int64_t frame) {
    if (tid == 0) {
      const int status = misc[MI_STATUS];
      p.n_out[frame] = status ? 0 : misc[MI_NOUT];
      p.status[frame] = status;
      if (p.n_cand) p.n_cand[frame] = misc[MI_G];
    }
  }
};

template <int T, bool UNIFORM_K, bool F32R, bool WIDE, int MODE, bool HEAVY = false>
__global__ __launch_bounds__(T, MOCAP_FRAME_WAVES_PER_EU) void frame_kernel(FrameArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  FrameState<T, UNIFORM_K, F32R, WIDE, HEAVY> st(p, smem);
  const int tid = threadIdx.x;
  const FrameQueues& q = p.q;
  const int R = p.K_max;

  int chunk_next = 0, chunk_end = 0;  // lane 0 only: frames of the chunk it pulled last
  int done_local = 0;                 // lane 0 only: frames finished since the last count (MODE_ALL)
  if constexpr (MODE == MODE_ALL) {
    bool frames_left = true;  // lane 0 only
    while (true) {
      // ---------------------------------------------------------- pull a frame, else a slice ticket
      if (tid == 0) {
        int item = -1, kind = 0;
        if (frames_left) {
          if (chunk_next >= chunk_end) {
            chunk_next = q_add(&q.counters[QC_NEXT_FRAME], q.frame_chunk);
            chunk_end = chunk_next + q.frame_chunk;
          }
          item = chunk_next++;
          if (item >= frame_count(p)) {
            item = -1;
            frames_left = false;
            if (done_local) {  // the frames of a chunk that reached past the end of the batch
              wait_own_stores();
              q_add(&q.counters[QC_FRAMES_DONE], done_local);
              done_local = 0;
            }
          } else {
            kind = 1;
          }
        }
        if (!kind && q.heavy_threshold) {
          const int s = q_add(&q.counters[QC_NEXT_SLICE], 1);
          while (s < q.W_cap) {
            if (q_load(&q.slice_gen[s]) == q.gen) {
              kind = 2;
              item = s;
              break;
            }
            // a producer publishes its slices BEFORE it counts its frame as done: once every frame is done, an
            // unpublished ticket will never be published (and neither will any later one)
            if (q_load(&q.counters[QC_FRAMES_DONE]) >= (int)frame_count(p)) {
              if (q_load(&q.slice_gen[s]) == q.gen) {
                kind = 2;
                item = s;
              }
              break;
            }
            __builtin_amdgcn_s_sleep(8);
          }
        }
        st.misc[MI_ITEM] = item;
        st.misc[MI_KIND] = kind;
      }
      __syncthreads();
      const int item = st.misc[MI_ITEM], kind = st.misc[MI_KIND];
      __syncthreads();  // every lane has read the slots before lane 0 can overwrite them
      if (!kind) {
        // the last workgroup to leave puts the queue counters back to zero: the next launch needs no memset
        if (tid == 0 && q_add(&q.counters[QC_EXITED], 1) == (int)gridDim.x - 1)
          for (int c = 0; c < QC_COUNT; c++) q_store(&q.counters[c], 0);
        break;
      }
      // one code path for both kinds of item (a single inlined copy of match() and evaluate(): frame and slice
      // workgroups share the CU's instruction cache)
      int64_t frame = item;
      int h = -1, base = 0, S = 1, sl = 0;
      if (kind == 2) {
        h = q_load(&q.slice_heavy[item]);
        frame = q_load(&q.heavy[4 * h + 0]);
        base = q_load(&q.heavy[4 * h + 1]);
        S = q_load(&q.heavy[4 * h + 2]);
        sl = item - base;
      }
      frame_prio<MOCAP_FRAME_PRIO_MATCH>();
      if constexpr (WIDE) st.match_wide(frame); else st.match(frame);
      const uint32_t G = (uint32_t)st.misc[MI_G];
      if (kind == 1) {
        if (tid == 0) {
          int defer = 0;
          if (q.heavy_threshold && G > q.heavy_threshold) {
            uint32_t Sn = (G + q.slice_size - 1) / q.slice_size;
            if (Sn > 64) Sn = 64;
            const int hn = q_add(&q.counters[QC_N_HEAVY], 1);
            if (hn < q.H_cap) {
              const int bn = q_add(&q.counters[QC_N_SLICES], (int)Sn);
              if (bn + (int)Sn <= q.W_cap) {
                q_store(&q.heavy[4 * hn + 0], (int32_t)frame);
                q_store(&q.heavy[4 * hn + 1], bn);
                q_store(&q.heavy[4 * hn + 2], (int32_t)Sn);
                q_store(&q.heavy[4 * hn + 3], 0);
                for (uint32_t s = 0; s < Sn; s++) q_store(&q.slice_heavy[bn + s], hn);
                wait_own_stores();  // the entries are in memory before their tickets become valid
                for (uint32_t s = 0; s < Sn; s++) q_store(&q.slice_gen[bn + s], q.gen);
                defer = 1;
              }
            }
          }
          st.misc[MI_DEFER] = defer;
        }
        __syncthreads();
        st.write_frame_header(frame);  // n_out / status / n_cand are known after the match, whoever evaluates
      }
      // candidate range of this item: the whole frame, nothing (deferred to its slices), or one slice
      uint32_t g_lo = 0, g_hi = G;
      if (kind == 1) {
        if (st.misc[MI_DEFER]) g_hi = 0;
      } else {
        g_lo = (uint32_t)((uint64_t)G * (uint64_t)sl / S);
        g_hi = (uint32_t)((uint64_t)G * (uint64_t)(sl + 1) / S);
      }
      frame_prio<MOCAP_FRAME_PRIO_EVAL>();
      if (g_hi > g_lo) st.evaluate(g_lo, g_hi);
      frame_prio<MOCAP_FRAME_PRIO_OUT>();
      const int nroots = st.misc[MI_NROOTS];
      bool merge = false;
      for (int r = tid; r < nroots; r += T) {
        const int k = st.outslot[r];
        if (k < 0) continue;
        double e = __longlong_as_double(0x7ff0000000000000ll), X[3] = {0, 0, 0};  // +inf: no candidate here
        uint32_t gl = 0;
        bool won = false;
        if (g_hi > g_lo) won = st.root_winner(r, g_lo, g_hi, e, gl, X);
        if (kind == 1) {
          if (won) st.write_point(frame, r, e, gl, X);
        } else {
          const size_t o = (size_t)item * R + k;
          q_st(&q.part_e[o], e);
          q_st(&q.part_g[o], gl);
          q_st(&q.part_x[3 * o + 0], X[0]);
          q_st(&q.part_x[3 * o + 1], X[1]);
          q_st(&q.part_x[3 * o + 2], X[2]);
        }
      }
      if (kind == 1) {
        // finished frames are counted per chunk, not per frame: one device-scope atomic per frame on a single address
        // tops out near 90 M/s, which small frames (4 x 4) exceed.  (Lane 0 published this frame's slices itself, so
        // its own stores are the ones that have to be out before the count; the count is only read by workgroups
        // waiting for tickets, after they have run out of frames.)
        if (tid == 0) {
          done_local++;
          if (chunk_next >= chunk_end) {
            wait_own_stores();
            q_add(&q.counters[QC_FRAMES_DONE], done_local);
            done_local = 0;
          }
        }
      } else {
        // the workgroup that finishes a heavy frame's LAST slice merges them: it already holds the frame's roots, hit
        // lists and output slots in LDS
        wait_own_stores();  // every wave's partials have left the CU
        __syncthreads();
        if (tid == 0) st.misc[MI_DEFER] = q_add(&q.heavy[4 * h + 3], 1) == S - 1;
        __syncthreads();
        merge = st.misc[MI_DEFER] != 0;
        if (merge) {
          for (int r = tid; r < nroots; r += T) {
            const int k = st.outslot[r];
            if (k < 0) continue;
            size_t ob = (size_t)base * R + k;
            double eb = q_ld(&q.part_e[ob]);
            for (int s = 1; s < S; s++) {
              const size_t o = (size_t)(base + s) * R + k;
              const double e = q_ld(&q.part_e[o]);
              if (e < eb) {  // strict <: the earliest slice wins ties, slices ascend in candidate index
                eb = e;
                ob = o;
              }
            }
            const double X[3] = {q_ld(&q.part_x[3 * ob + 0]), q_ld(&q.part_x[3 * ob + 1]), q_ld(&q.part_x[3 * ob + 2])};
            st.write_point(frame, r, eb, q_ld(&q.part_g[ob]), X);
          }
        }
        __syncthreads();
      }
    }
    return;
  }
  while (true) {
    // ------------------------------------------------------------ pull a work item
    if (tid == 0) {
      int item;
      if (MODE == MODE_MAIN) {
        // frames are pulled `frame_chunk` at a time: one device-scope atomic per frame on a single
        // address tops out near 90 M/s, which small frames (4 x 4: a few candidates each) exceed
        if (chunk_next >= chunk_end) {
          chunk_next = atomicAdd(&q.counters[QC_NEXT_FRAME], q.frame_chunk);
          chunk_end = chunk_next + q.frame_chunk;
        }
        item = chunk_next++;
        if (item >= frame_count(p)) item = -1;
      } else if (MODE == MODE_SLICE) {
        int n = q.counters[QC_N_SLICES];
        if (n > q.W_cap) n = q.W_cap;
        item = atomicAdd(&q.counters[QC_NEXT_SLICE], 1);
        if (item >= n) item = -1;
      } else {
        int n = q.counters[QC_N_HEAVY];
        if (n > q.H_cap) n = q.H_cap;
        item = atomicAdd(&q.counters[QC_NEXT_MERGE], 1);
        if (item >= n) item = -1;
      }
      st.misc[MI_ITEM] = item;
    }
    __syncthreads();
    const int item = st.misc[MI_ITEM];
    __syncthreads();  // every lane has read the slot before lane 0 can overwrite it
    if (item < 0) break;

    if (MODE == MODE_MAIN) {
      const int64_t frame = item;
      if constexpr (WIDE) st.match_wide(frame); else st.match(frame);
      const uint32_t G = (uint32_t)st.misc[MI_G];
      // heavy frame: hand its candidate space to the slice pass instead of evaluating here
      if (tid == 0) {
        int defer = 0;
        if (q.heavy_threshold && G > q.heavy_threshold) {
          uint32_t S = (G + q.slice_size - 1) / q.slice_size;
          if (S > 64) S = 64;
          const int h = atomicAdd(&q.counters[QC_N_HEAVY], 1);
          if (h < q.H_cap) {
            const int base = atomicAdd(&q.counters[QC_N_SLICES], (int)S);
            if (base + (int)S <= q.W_cap) {
              q.heavy[4 * h + 0] = (int32_t)frame;
              q.heavy[4 * h + 1] = base;
              q.heavy[4 * h + 2] = (int32_t)S;
              for (uint32_t s = 0; s < S; s++) q.slice_heavy[base + s] = h;
              defer = 1;
            } else {
              q.heavy[4 * h + 0] = -1;  // no room for its slices: evaluated in place below
            }
          }
        }
        st.misc[MI_DEFER] = defer;
      }
      __syncthreads();
      if (st.misc[MI_DEFER]) continue;  // uniform
      st.write_frame_header(frame);
      if (G) {
        st.evaluate(0, G);
        const int nroots = st.misc[MI_NROOTS];
        for (int r = tid; r < nroots; r += T) {
          if (st.outslot[r] < 0) continue;
          double e, X[3];
          uint32_t gl;
          if (st.root_winner(r, 0, G, e, gl, X)) st.write_point(frame, r, e, gl, X);
        }
      }
      __syncthreads();
    } else if (MODE == MODE_SLICE) {
      const int h = q.slice_heavy[item];
      if (h < 0 || q.heavy[4 * h + 0] < 0) continue;  // uniform (same value for all lanes)
      const int64_t frame = q.heavy[4 * h + 0];
      const int base = q.heavy[4 * h + 1], S = q.heavy[4 * h + 2];
      const int sl = item - base;
      if constexpr (WIDE) st.match_wide(frame); else st.match(frame);
      const uint64_t G = (uint32_t)st.misc[MI_G];
      const uint32_t g_lo = (uint32_t)(G * (uint64_t)sl / S), g_hi = (uint32_t)(G * (uint64_t)(sl + 1) / S);
      if (g_hi > g_lo) st.evaluate(g_lo, g_hi);
      const int nroots = st.misc[MI_NROOTS];
      for (int r = tid; r < nroots; r += T) {
        const int k = st.outslot[r];
        if (k < 0) continue;
        double e = __longlong_as_double(0x7ff0000000000000ll), X[3] = {0, 0, 0};  // +inf: no candidate here
        uint32_t gl = 0;
        if (g_hi > g_lo) st.root_winner(r, g_lo, g_hi, e, gl, X);
        const size_t o = (size_t)item * R + k;
        q.part_e[o] = e;
        q.part_g[o] = gl;
        q.part_x[3 * o + 0] = X[0];
        q.part_x[3 * o + 1] = X[1];
        q.part_x[3 * o + 2] = X[2];
      }
      __syncthreads();
    } else {  // MODE_MERGE
      const int h = item;
      if (q.heavy[4 * h + 0] < 0) continue;
      const int64_t frame = q.heavy[4 * h + 0];
      const int base = q.heavy[4 * h + 1], S = q.heavy[4 * h + 2];
      if constexpr (WIDE) st.match_wide(frame); else st.match(frame);
      st.write_frame_header(frame);
      const int nroots = st.misc[MI_NROOTS];
      for (int r = tid; r < nroots; r += T) {
        const int k = st.outslot[r];
        if (k < 0) continue;
        size_t ob = (size_t)base * R + k;
        double eb = q.part_e[ob];
        for (int s = 1; s < S; s++) {
          const size_t o = (size_t)(base + s) * R + k;
          const double e = q.part_e[o];
          if (e < eb) {  // strict <: the earliest slice wins ties, slices ascend in candidate index
            eb = e;
            ob = o;
          }
        }
        const double X[3] = {q.part_x[3 * ob + 0], q.part_x[3 * ob + 1], q.part_x[3 * ob + 2]};
        st.write_point(frame, r, eb, q.part_g[ob], X);
      }
      __syncthreads();
    }
  }
}

template <int T, bool WIDE, int MODE>
static hipError_t launch_TM(const FrameArgs& a, int grid, size_t lds, hipStream_t stream) {
  void (*k)(FrameArgs);
  if (a.cv.f32_rounding)
    k = a.cv.uniformK ? frame_kernel<T, true, true, WIDE, MODE> : frame_kernel<T, false, true, WIDE, MODE>;
  else
    k = a.cv.uniformK ? frame_kernel<T, true, false, WIDE, MODE> : frame_kernel<T, false, false, WIDE, MODE>;
  if constexpr (WIDE && MODE == MODE_ALL) {
    if (a.heavy_bb) {  // (identical intrinsics only: the host asks for it nowhere else)
      if (!a.cv.uniformK) return hipErrorInvalidValue;
      k = a.cv.f32_rounding ? frame_kernel<T, true, true, true, MODE_ALL, true> : frame_kernel<T, true, false, true, MODE_ALL, true>;
    }
  } else if (a.heavy_bb) {
    return hipErrorInvalidValue;
  }
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(T), lds, stream, a);
  return hipGetLastError();
}

template <int T, bool WIDE>
static hipError_t launch_T(const FrameArgs& a, int mode, int grid, size_t lds, hipStream_t stream) {
  switch (mode) {
    case MODE_MAIN: return launch_TM<T, WIDE, MODE_MAIN>(a, grid, lds, stream);
    case MODE_SLICE: return launch_TM<T, WIDE, MODE_SLICE>(a, grid, lds, stream);
    case MODE_MERGE: return launch_TM<T, WIDE, MODE_MERGE>(a, grid, lds, stream);
    default: return launch_TM<T, WIDE, MODE_ALL>(a, grid, lds, stream);
  }
}

// The wide variant's instantiations are a translation unit of their own (csrc/frame_kernel_wide.hip = this file with
// MOCAP_FRAME_TU_WIDE defined): its 1 400-line kernel body is compiled with scheduler / allocator options that cost the one-wave
// kernels of small frames 4 % (Makefile: FRAME_WIDE_FLAGS; 28.0 -> 27.0 ms per 12 500 stress frames).
#ifdef MOCAP_FRAME_TU_WIDE
hipError_t launch_frame_kernel_wide(const FrameArgs& a, int mode, int threads, int grid, size_t lds, hipStream_t stream) {
  if (threads == kWideThreads) return launch_T<kWideThreads, true>(a, mode, grid, lds, stream);
  if (threads == 512 && mode == MODE_ALL) return launch_TM<512, true, MODE_ALL>(a, grid, lds, stream);  // two frames per CU (capi.hip plan_frame)
  return hipErrorInvalidValue;
}
#else
hipError_t launch_frame_kernel_wide(const FrameArgs& a, int mode, int threads, int grid, size_t lds, hipStream_t stream);

hipError_t launch_frame_kernel(const FrameArgs& a, int mode, int threads, int grid, hipStream_t stream) {
  const size_t lds = frame_lds_bytes(a.cv.C, a.M, a.K_max, threads, a.H, a.wide != 0, a.cv.uniformK != 0);
  if (a.wide) return launch_frame_kernel_wide(a, mode, threads, grid, lds, stream);
  switch (threads) {
    case 64: return launch_T<64, false>(a, mode, grid, lds, stream);
    case 128: return launch_T<128, false>(a, mode, grid, lds, stream);
    case 256: return launch_T<256, false>(a, mode, grid, lds, stream);
    default: return hipErrorInvalidValue;
  }
}
#endif

}  // namespace mocap
