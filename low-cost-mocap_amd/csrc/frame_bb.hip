// frame_bb.hip -- the frame path for the realistic rigs (identical plain intrinsics, <= 16 cameras, <= 64 blobs per
// camera, <= 255 roots): epipolar correspondence search + EXACT branch-and-bound selection, one 256-lane workgroup per
// frame, persistent, everything between the blob arrays in and the kept points out in LDS.  Replaces
//   find_point_correspondance_and_object_points   (reference computer_code/api/helpers.py:339-421)
// like frame_kernel.hip does (which keeps the general case: per-camera intrinsics, wide frames, tiny frames, and
// the exhaustive walk every result of this file is tested against, MOCAP_EVAL_BB=0).
//
// What is different from frame_kernel.hip (measurements: DESIGN.md 3.1, docs/HISTORY.md):
//  * its own kernel and LDS layout: C <= 16 (blob indices of a group in one or two 64-bit words), any K_max <= 255 that fits
//    LDS -- the reference seam's default K_max = min(C M, 64) and the re-submit's C M included.
//  * phase B without the per-camera barrier chain.  The reference matches camera after camera (helpers.py:359-406) because a
//    blob no root claims becomes a new root for the cameras after it.  Only THAT is sequential: the camera-0 roots' lines,
//    gates, orders and claims in all other cameras are independent -> one barrier-free pass over (root, camera) pairs, then
//    the same pass for the provisional roots, then a chain over the cameras that is bookkeeping only.
//  * written for instruction issue, which is what binds it: one lane per (root, camera) pair with the camera's blobs walked
//    serially; the camera count (8) and, for 8 x 16 with K_max <= 48 / <= 64, the whole LDS layout at compile time; the
//    blocks' bounds and blob indices cached between seed and test pass; the camera tables behind one base pointer.
//  * a software pipeline over the frames: frame k + 1 is pulled from the queue and fetched straight into a spare LDS
//    buffer (global_load_lds, no registers) while frame k is searched.
//  * one evaluation path: a frame below the search threshold (MOCAP_BB_MIN_G, default 0 = never) queues all its blocks
//    unconditionally and takes the same evaluation rounds -- no second copy of the geometry core in the kernel.
//
// Phases per frame (256 lanes = 4 waves):
//   A   the prefetched frame becomes the current one (buffer swap)
//   B0  roots of camera 0; every (camera-0 root, camera) pair on its own lane: epipolar line (helpers.py:362-364),
//       the camera's blobs walked serially -- point-line distance (helpers.py:373), 5 px gate (helpers.py:375,383), hits
//       as a bit mask, written in stable (distance, index) order by repeated minimum (helpers.py:384), claim of the
//       closest hit BY VALUE (helpers.py:391) as a mask; speculative lines of every blob that might become a root
//   B1  wave 0: for camera i = 1 .. C-1: the roots created at cameras < i against camera i (lanes = (root, blob) groups
//       of 2^ceil(log2 M), ballots), then the unclaimed blobs of camera i become roots (helpers.py:402-406); meanwhile
//       waves 1-3 tabulate the DLT contribution of every blob (mocap_device.hpp dlt_contribution), one lane pulls the
//       next frame from the queue
//   C   per-root candidate counts (helpers.py:394-400), offsets, output slots (wave scans)
//   D   branch and bound over blocks of the Cartesian product: seeds, block tests, evaluation of the
//       survivors' candidates spread over all lanes; per (wave, root) slot = lexicographic minimum of (error bits,
//       candidate index) = np.argmin's first minimum (helpers.py:418) whatever the evaluation order
//   E   one lane per kept root: merge the four waves' slots, decode the winning group, write xyz / err / corr
#include "mocap_device.hpp"
#include "kernels.hpp"
#include <cstdlib>
#include "frame_common.hpp"
// (Round 6: the timing-only switches -- phases compiled out, parts run twice -- and the probe scheme, measured slower in
// round 5, left this file; they are in the history at 7c94a55 and in docs/HISTORY.md.  What remains switchable is what a
// test uses: -DMOCAP_DEBUG_EIGCHECK, the self-check build of tests/test_gpu_bb_adversarial.py.)

namespace mocap {

constexpr int kBBThreads = 256;
constexpr int kBBWaves = kBBThreads / 64;
constexpr int kBBRecs = kBBThreads + 64;  // surviving blocks queued between two evaluation rounds (flush above 64)

// LDS carving, identical on host (size) and device (pointers): the frame's persistent state (blobs, per-blob DLT table,
// roots, hit lists), the search's arrays (block records, result slots -- dead while matching, which keeps its speculative
// lines there), the spare blob buffer of the frame pipeline and the cache of block bounds.
// Workgroups (= waves per SIMD) per CU an instantiation is built for: its register budget (512 / n per lane) and the occupancy
// step its LDS layout sizes the block cache for.  Four everywhere (128 VGPRs; 33-40 KB of LDS) except the 8-camera, 16-blob,
// 48-slot layout -- the bench's -- whose 29.2 KB leave room for FIVE frames per CU: 96 VGPRs (52 spilled instead of 22, +6 % per
// frame) and a block cache of 161 entries instead of 385, and still 4.50 -> 4.31 ms per 100 k frames (profiles/r06_wide_experiments.txt, (15)).
#ifndef MOCAP_BB_WAVES_PER_EU
#define MOCAP_BB_WAVES_PER_EU 4
#endif
#ifndef MOCAP_BB_WAVES_PER_EU_48
#define MOCAP_BB_WAVES_PER_EU_48 5
#endif
constexpr int bb_wg_per_cu(int RL) { return RL == 48 ? MOCAP_BB_WAVES_PER_EU_48 : MOCAP_BB_WAVES_PER_EU; }
#ifndef MOCAP_BB_LDS_SLACK
#define MOCAP_BB_LDS_SLACK 1024  // (measured: 512 keeps the occupancy step as well, 0 does not)
#endif
struct BBLayout {
  size_t bxy, bxy_nx, cnt_nx, bt, rbound, seedkey, slot_key, claimw, recs, rpk, scr, scr_bytes, goff, gcnt, outslot, boff, bnb, seedgh, slot_g, cnt,
      misc, bpl, nh, hits, act, root_blob, root_cam, nact, bnl, bv, bcache, bpk, total;
  int ncache;
  __host__ __device__ static size_t al(size_t x, size_t a) { return (x + a - 1) / a * a; }
  __host__ __device__ BBLayout(int C, int M, int R, int CW, int wg_per_cu) {
    size_t o = 0;
    auto take = [&](size_t bytes, size_t a) {
      o = al(o, a);
      const size_t at = o;
      o += bytes;
      return at;
    };
    // the two blob buffers come first: global_load_lds addresses LDS through M0, keep its targets in the low 64 KB
    bxy = take(sizeof(float2) * (size_t)C * M, 16);
    bxy_nx = take(sizeof(float2) * (size_t)C * M, 16);  // the NEXT frame's blobs land here while this one is searched
    cnt_nx = take(4 * (size_t)C, 4);
    bt = take(sizeof(double) * 10 * ((size_t)C * M + 1), 16);  // DLT contribution per (camera, blob): five b128 reads; + one record of zeros ("camera not in the group")
    rbound = take(8 * (size_t)R, 8);
    claimw = take(8 * (size_t)C, 8);
    // the search's block records and result slots: dead while matching -> phase B keeps the speculative epipolar
    // lines here (match(): lines of every blob that MIGHT become a root, computed off the critical path)
    scr = seedkey = take(8 * (size_t)R, 8);
    slot_key = take(8 * (size_t)kBBWaves * R, 8);
    recs = take(8 * (size_t)kBBRecs, 8);
    rpk = take(8 * (size_t)kBBRecs * CW, 8);
    seedgh = take(4 * (size_t)R, 4);
    slot_g = take(4 * (size_t)kBBWaves * R, 4);
    scr_bytes = o - scr;
    goff = take(4 * (size_t)(R + 1), 4);
    gcnt = take(4 * (size_t)R, 4);
    outslot = take(4 * (size_t)R, 4);
    boff = take(4 * (size_t)(R + 1), 4);
    bnb = take(4 * (size_t)R, 4);
    cnt = take(4 * (size_t)C, 4);
    misc = take(4 * 16, 4);
    bpl = take(2 * (size_t)R, 2);
    nh = take((size_t)R * C, 1);
    hits = take((size_t)R * C * M, 1);
    act = take((size_t)R * C, 1);
    root_blob = take((size_t)R, 1);
    root_cam = take((size_t)R, 1);
    nact = take((size_t)R, 1);
    bnl = take((size_t)R, 1);
    bv = take((size_t)R, 1);
    // cache of the blocks' bounds (seed pass -> test pass: {s1, trace} rounded UP to float, 8 bytes per block): as many
    // entries as fit below the next occupancy step of the 160 KB LDS (5, 4, 3, ... workgroups per CU), at most 1024
    // (round 5: + the block's blob indices, 8 CW bytes: a surviving block is queued from the cache alone -- no second decode)
    bcache = take(0, 8);
    ncache = 0;
    const size_t per_entry = 8 + 8 * (size_t)CW;
    for (int per_cu = wg_per_cu; per_cu >= 1; per_cu--) {
      const size_t lim = ((size_t)160 * 1024 / per_cu - MOCAP_BB_LDS_SLACK) / 256 * 256;  // (slack: allocation granule, other LDS users)
      if (lim >= o + per_entry * 64) {
        const size_t n = (lim - o) / per_entry;
        ncache = (int)(n > 1024 ? 1024 : n);
        break;
      }
    }
    o += 8 * (size_t)ncache;
    bpk = o;
    o += 8 * (size_t)CW * ncache;
    total = al(o, 16);
  }
};

// Root slots the launch lays out for a frame shape: the 8-camera, 16-blob shape has instantiations with the whole
// layout fixed at compile time (48 or 64 slots, 0 = none applies); every other shape is laid out for K_max itself.
static int bb_fixed_slots(int C, int M, int R) {
  const char* e = getenv("MOCAP_BB_FIXED_LAYOUT");  // 0: the runtime-layout instantiation for every shape (tests, A/B timing)
  if (e && e[0] == '0') return 0;
  if (C == 8 && M == 16 && R <= 48) return 48;
  if (C == 8 && M == 16 && R <= 64) return 64;
  return 0;
}
static int frame_bb_root_slots(int C, int M, int R) {
  const int f = bb_fixed_slots(C, M, R);
  return f ? f : R;
}
size_t frame_bb_lds_bytes(int C, int M, int R) { return BBLayout(C, M, frame_bb_root_slots(C, M, R), C <= 8 ? 1 : 2, bb_wg_per_cu(bb_fixed_slots(C, M, R))).total; }
static size_t frame_bb_lds_bytes_min(int C, int M, int R) {
  const BBLayout L(C, M, frame_bb_root_slots(C, M, R), C <= 8 ? 1 : 2, bb_wg_per_cu(bb_fixed_slots(C, M, R)));
  return L.total - (8 + 8 * (size_t)(C <= 8 ? 1 : 2)) * (size_t)L.ncache;
}
bool frame_bb_fits(int C, int M, int R) {
  // blob indices and root numbers are bytes (0xFF = none); a (root, blob) group is at most one wave; the expanded
  // candidate list of one evaluation round is counted in 22 bits (FrameArgs::bb_pl is lowered by the host if needed)
  return C >= 2 && C <= 16 && M >= 1 && M <= 64 && R >= 1 && R <= 255 && frame_bb_lds_bytes_min(C, M, R) <= (size_t)128 * 1024;
}

// blob indices of one (partial) group: one byte per camera, 0xFF = the camera is open or not in the group
template <int CW>
struct Packed {
  unsigned long long w[CW];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int k = 0; k < CW; k++) w[k] = ~0ull;
  }
  __device__ __forceinline__ void set(int c, uint32_t idx) {  // the camera's byte must still be 0xFF
    if (CW == 1)
      w[0] ^= (unsigned long long)(idx ^ 0xFFu) << (8 * c);
    else
      w[c >> 3] ^= (unsigned long long)(idx ^ 0xFFu) << (8 * (c & 7));
  }
  __device__ __forceinline__ uint32_t get(int c) const {
    return CW == 1 ? (uint32_t)(w[0] >> (8 * c)) & 0xFFu : (uint32_t)(w[c >> 3] >> (8 * (c & 7))) & 0xFFu;
  }
};

// CT: the camera count when it is known at compile time (8: the headline rig -- camera loops unroll, their control
// and address arithmetic leave the scalar unit, which the issue-bound kernel shares with the vector work), 0 = runtime
// The camera tables as ONE base pointer plus offsets fixed by the camera count (mocap_set_cameras lays its block out as
// Pq | RT | K4 | F, capi.hip; identical intrinsics: Pq is [C][12]; launch_frame_bb checks the pointers against this).
// Through a CamView the compiler re-read the four table pointers from the kernel arguments before every table read --
// per reprojected view: arguments -> RT -> arguments -> K4, four dependent scalar-cache round trips around 42 vector
// instructions.  The base is kept in registers (its value is hidden from the optimiser once, at the start of the
// kernel), so a view costs one round trip, and K4 (one entry: identical intrinsics) is read once per candidate.
template <int CT>
struct BBCamTables {
  const double* base;
  static constexpr int C = CT;
  __device__ __forceinline__ ctab_t pq(size_t off) const { return as_ctab(base + off); }
  __device__ __forceinline__ ctab_t rt(size_t off) const { return as_ctab(base + 12 * CT + off); }
  __device__ __forceinline__ ctab_t k4(size_t off) const { return as_ctab(base + 24 * CT + off); }
  __device__ __forceinline__ const double* f() const { return base + 28 * CT; }
};
template <>
struct BBCamTables<0> {
  const double* base;
  int C;
  __device__ __forceinline__ ctab_t pq(size_t off) const { return as_ctab(base + off); }
  __device__ __forceinline__ ctab_t rt(size_t off) const { return as_ctab(base + 12 * C + off); }
  __device__ __forceinline__ ctab_t k4(size_t off) const { return as_ctab(base + 24 * C + off); }
  __device__ __forceinline__ const double* f() const { return base + 28 * C; }
};

// Issue priority of the frame's phases (s_setprio: arbitration between the waves of one SIMD -- here the four frames
// of a CU, each with one wave per SIMD, each in a phase of its own).  The candidate evaluation is the bulk of the work and
// runs at 0; everything that holds the other waves of its workgroup at a barrier runs above it: the single-wave stretches
// (chain bookkeeping, scans) at 3, the rest of the matching and the output at 2, the seed pass at 1.  A frame's serial
// depth then costs what it costs alone, not what it costs sharing its SIMD's issue slots with three evaluations:
// 4.70 -> 4.50 ms per 100 k frames of 8 x 16 (profiles/r06_wide_experiments.txt, (14); MOCAP_BB_PRIO=0: without).
#ifndef MOCAP_BB_FRESH_TID
#define MOCAP_BB_FRESH_TID 1
#endif
#ifndef MOCAP_BB_PRIO
#define MOCAP_BB_PRIO 1
#endif
enum { kPrioEval = 0, kPrioSeed = 1, kPrioPhase = 2, kPrioSerial = 3 };
template <int P>
__device__ __forceinline__ void bb_prio() {
#if MOCAP_BB_PRIO
  __builtin_amdgcn_s_setprio(P);
#endif
}

// ML, RL: the layout's blobs per camera and root slots when they are known at compile time (with CT: every LDS array
// sits at a constant address, which takes the base registers, their spills to vector lanes and the address arithmetic
// out of every phase -- 5.72 -> 5.39 ms per 100 k frames of 8 x 16), 0 = runtime.  ML is the frame's M_max itself (the
// blob buffers are copied flat); RL only has to hold K_max roots: the slot arrays are laid out for RL, the root limit
// and the output stride stay K_max.
template <bool F32R, int CW, int CT, int ML, int RL>
struct BBState {
  static constexpr int T = kBBThreads, W = kBBWaves;
  const FrameArgs& p;
  BBCamTables<CT> cv;
  const int C_, M, R, RS;  // R = K_max (root limit, output stride), RS = root slots of the layout
  int tid, lane, wave;
  // The lane's number taken afresh (an empty asm the optimiser cannot see through): what a phase derives from it -- LDS addresses,
  // (root, camera) pair indices, masks -- is computed in that phase instead of once before the frame loop, where it would sit in a
  // register for the whole frame, i.e. in scratch (MOCAP_BB_FRESH_TID=0: without)
  __device__ __forceinline__ void fresh_tid() {
#if MOCAP_BB_FRESH_TID
    int t = tid;
    asm volatile("" : "+v"(t));
    tid = t;
    lane = t & 63;
    wave = t >> 6;
#endif
  }
  __device__ __forceinline__ int cn() const { return CT > 0 ? CT : C_; }
  double* bt;
  float2 *bxy, *bxy_nx;
  int32_t* cnt_nx;
  unsigned long long *rbound, *seedkey, *slot_key, *claimw, *rpk;
  struct BRec { uint32_t gh, rs; };  // surviving block gh of root (rs & 0xFF); its candidates start at rs >> 8 of the expanded list
  BRec* recs;
  uint32_t *goff, *gcnt, *boff, *bnb, *seedgh, *slot_g;
  int32_t *outslot, *cnt, *misc;
  uint16_t* bpl;
  uint8_t *nh, *hits, *act, *root_blob, *root_cam, *nact, *bnl, *bv;
  float2* bcache;      // [ncache] {s1, trace} of the first blocks, rounded up (BBLayout::bcache)
  unsigned long long* bpk;  // [ncache][CW] ... and their partial groups' blob indices
  int ncache;
  unsigned char* scr;  // phase B scratch = the search's records and slots (BBLayout::scr)
  size_t scr_bytes;
  static constexpr unsigned long long kInfBits = 0x7ff0000000000000ull;

  __device__ BBState(const FrameArgs& p_, unsigned char* smem)
      : p(p_), C_(p_.cv.C), M(ML > 0 ? ML : p_.M), R(p_.K_max), RS(RL > 0 ? RL : p_.K_max), tid(threadIdx.x), lane(threadIdx.x & 63), wave(threadIdx.x >> 6) {
    const int C = cn();
    {
      const unsigned long long ta = (unsigned long long)(uintptr_t)p_.cv.Pq;
      uint32_t tlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ta);
      uint32_t thi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ta >> 32));
      asm volatile("" : "+s"(tlo), "+s"(thi));
      cv.base = (const double*)(uintptr_t)(((unsigned long long)thi << 32) | tlo);
      if constexpr (CT == 0) cv.C = C;
    }
    const BBLayout L(C, M, RS, CW, bb_wg_per_cu(RL));
    bt = (double*)(smem + L.bt);
    bxy = (float2*)(smem + L.bxy);
    bxy_nx = (float2*)(smem + L.bxy_nx);
    cnt_nx = (int32_t*)(smem + L.cnt_nx);
    rbound = (unsigned long long*)(smem + L.rbound);
    seedkey = (unsigned long long*)(smem + L.seedkey);
    slot_key = (unsigned long long*)(smem + L.slot_key);
    claimw = (unsigned long long*)(smem + L.claimw);
    recs = (BRec*)(smem + L.recs);
    rpk = (unsigned long long*)(smem + L.rpk);
    goff = (uint32_t*)(smem + L.goff);
    gcnt = (uint32_t*)(smem + L.gcnt);
    outslot = (int32_t*)(smem + L.outslot);
    boff = (uint32_t*)(smem + L.boff);
    bnb = (uint32_t*)(smem + L.bnb);
    seedgh = (uint32_t*)(smem + L.seedgh);
    slot_g = (uint32_t*)(smem + L.slot_g);
    cnt = (int32_t*)(smem + L.cnt);
    misc = (int32_t*)(smem + L.misc);
    bpl = (uint16_t*)(smem + L.bpl);
    nh = (uint8_t*)(smem + L.nh);
    hits = (uint8_t*)(smem + L.hits);
    act = (uint8_t*)(smem + L.act);
    root_blob = (uint8_t*)(smem + L.root_blob);
    root_cam = (uint8_t*)(smem + L.root_cam);
    nact = (uint8_t*)(smem + L.nact);
    bnl = (uint8_t*)(smem + L.bnl);
    bv = (uint8_t*)(smem + L.bv);
    scr = smem + L.scr;
    scr_bytes = L.scr_bytes;
    bcache = (float2*)(smem + L.bcache);
    bpk = (unsigned long long*)(smem + L.bpk);
    ncache = L.ncache;
  }

  // ---------------------------------------------------------------- phase B building blocks
  // Epipolar line of root r in camera i: cv.computeCorrespondEpilines on a float32 point -- double math, scale by
  // 1/sqrt(a^2+b^2), float32 result (helpers.py:363-364); den = sqrt(a^2+b^2) of the ROUNDED line, by which
  // helpers.py:373 divides again, and its reciprocal for the quotients.  The same expressions, in the same order, as
  // frame_kernel.hip phase B1 (bit-identical by construction; tested against it).
  struct Line { double a, b, c, den, rden; };
  __device__ __forceinline__ Line epiline(int r, int i) const { return epiline_of(root_cam[r], root_blob[r], i); }
  __device__ __forceinline__ Line epiline_of(int rc, int rb, int i) const {
    const int C = cn();
    const double* Fm = cv.f() + 9 * ((size_t)rc * C + i);  // (per-lane camera pair: vector loads, L1/L2-resident table)
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
    Line L;
    L.a = a;
    L.b = b;
    L.c = c;
    L.den = sqrt(a * a + b * b);
    L.rden = recip_refined(L.den);
    return L;
  }

  // Gate / order / claim of up to 64 (root, camera) pairs whose lines sit in the lanes of this wave (lane l holds the
  // line of pair l; pair_r / pair_i = its root and camera, n_pairs valid).  Lanes regroup as (pair slot, blob):
  // GS = 2^gs_shift >= M lanes per pair, 64 / GS pairs per sub-pass.
  // CHAIN: every pair is in camera `cam` (wave-uniform, Mi blobs); the blobs the pairs claim come back as a mask
  // (wave-uniform) instead of going to claimw[] -- the chain over the cameras keeps its state in registers.
  template <bool CHAIN>
  __device__ __forceinline__ unsigned long long match_pairs(const Line& mine, int pair_r, int pair_i, int n_pairs, int gs_shift,
                                                            int cam = 0, int Mi_chain = 0) {
    const int C = cn();
    const int GS = 1 << gs_shift, PPS = 64 >> gs_shift;
    const int k = lane & (GS - 1), q = lane >> gs_shift;
    const int gl0 = lane & ~(GS - 1);
    const unsigned long long gall = gs_shift == 6 ? ~0ull : ((1ull << GS) - 1ull);
    unsigned long long claims = 0ull;
    for (int s0 = 0; s0 < n_pairs; s0 += PPS) {  // wave-uniform trip count
      const int src = s0 + q;                    // the lane that holds this group's line
      const bool has = src < n_pairs;
      const int sl = has ? src : 0;
      const double la = __shfl(mine.a, sl), lb = __shfl(mine.b, sl), lc = __shfl(mine.c, sl);
      const double lden = __shfl(mine.den, sl), lrden = __shfl(mine.rden, sl);
      const int r = __shfl(pair_r, sl);
      const int i = CHAIN ? cam : __shfl(pair_i, sl);
      const int Mi = has ? (CHAIN ? Mi_chain : cnt[i]) : 0;
      const bool valid = k < Mi;
      float2 pt = make_float2(0.f, 0.f);
      double d = 0.0;
      bool hit = false;
      if (valid) {
        pt = bxy[(size_t)i * M + k];
        const double num = fabs(la * (double)pt.x + lb * (double)pt.y + lc);  // helpers.py:373
        d = div_by(num, lden, lrden);
        hit = d < p.gate_px;  // strict <, helpers.py:375,383
      }
      const unsigned long long wm = __ballot(hit);
      const unsigned long long gm = (wm >> gl0) & gall;
      // rank among the group's hits by (distance, blob index): a stable order where NumPy's default argsort is not
      // (helpers.py:384; documented deviation).  The loop runs over the hits of the whole wave (a handful), each
      // broadcast from its lane; a lane counts the ones of its own group that precede it.
      int rank = 0;
      const unsigned int dlo = (unsigned int)__double_as_longlong(d), dhi = (unsigned int)(__double_as_longlong(d) >> 32);
      for (unsigned long long mm = wm; mm;) {
        const int j = __ffsll((long long)mm) - 1;  // wave-uniform
        mm &= mm - 1;
        const double dj = __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)dhi, j) << 32) |
                                               (unsigned int)__builtin_amdgcn_readlane((int)dlo, j));
        const bool mine_grp = (j & ~(GS - 1)) == gl0 && j != lane;
        rank += (hit && mine_grp && (dj < d || (dj == d && j < lane))) ? 1 : 0;
      }
      if (hit) hits[((size_t)r * C + i) * M + rank] = (uint8_t)k;
      const unsigned long long g0 = (__ballot(hit && rank == 0) >> gl0) & gall;
      // removal by value (helpers.py:391): every blob with the closest hit's coordinates is claimed
      bool same = false;
      if (valid && g0) {
        const float2 p0 = bxy[(size_t)i * M + (__ffsll((long long)g0) - 1)];
        same = pt.x == p0.x && pt.y == p0.y;
      }
      const unsigned long long cw = __ballot(same);
      if (CHAIN) {
        for (int sh = 0; sh < 64; sh += GS) claims |= (cw >> sh) & gall;  // lane -> blob: fold the groups onto each other
      }
      if (has && k == 0) {
        nh[(size_t)r * C + i] = (uint8_t)__popcll(gm);
        if (!CHAIN) {
          const unsigned long long cm = (cw >> gl0) & gall;
          if (cm) atomicOr(&claimw[i], cm);
        }
      }
    }
    return claims;
  }

  // The next frame's blobs and counts travel from HBM straight into the spare LDS buffer while the current frame is
  // searched (global_load_lds: no VGPR is held -- a register prefetch was spilled to scratch by the compiler, i.e. went
  // to HBM and back): neither the queue atomic nor the first touch of a frame sits on the per-frame critical path.
  // The loads complete before the frame's last barrier (wait_own_stores: vmcnt(0) counts them).
  __device__ __forceinline__ void prefetch_lds(int64_t frame) const {
    const int C = cn();
    const float* src = p.blobs + (size_t)frame * C * M * 2;
    const int n_dw = C * M * 2;
    for (int base = wave * 64; base < n_dw; base += T)  // a wave writes 64 consecutive dwords at its LDS base + 4 lane
      if (base + lane < n_dw)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + base + lane),
                                         (__attribute__((address_space(3))) void*)((float*)bxy_nx + base), 4, 0, 0);
    if (wave == 0 && lane < C)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.counts + (size_t)frame * C + lane),
                                       (__attribute__((address_space(3))) void*)cnt_nx, 4, 0, 0);
  }
  // the prefetched frame becomes the current one (buffer swap)
  __device__ __forceinline__ void stage(int64_t frame) {
    fresh_tid();
    const int C = cn();
    float2* t = bxy;
    bxy = bxy_nx;
    bxy_nx = t;
    if (tid < C) {
      const int n = cnt_nx[tid];
      cnt[tid] = n < 0 ? 0 : (n > M ? M : n);
      claimw[tid] = 0ull;
    }
    if (tid == 0) {
      misc[MI_STATUS] = 0;
      misc[MI_OMAX] = 0;
    }
  }

  // ---------------------------------------------------------------- phases B-C (the frame is staged)
  // Also pulls the NEXT frame from the queue into misc[MI_NEXT] (one lane of wave 1, while wave 0 walks the cameras: the
  // atomic's round trip to L2 is off everybody's critical path), so that every lane can prefetch that frame afterwards.
  __device__ void match() {
    fresh_tid();
    const int C = cn();
    __syncthreads();
    int gs_shift = 0;
    while ((1 << gs_shift) < M) gs_shift++;
    const int n0 = cnt[0] < R ? cnt[0] : R;
    {  // roots from camera 0 (helpers.py:349,357)
      for (int r = tid; r < n0; r += T) {
        root_cam[r] = 0;
        root_blob[r] = (uint8_t)r;
      }
      if (tid == 0 && cnt[0] > R) misc[MI_STATUS] |= MOCAP_ST_ROOT_OVERFLOW_;
    }
    __syncthreads();
    // B0: the (camera-0 root, camera) pairs, ONE LANE PER PAIR, the camera's blobs walked serially (the kernel is issue
    // bound: ~10 wave instructions per pair instead of ~250 per 4 pairs with a lane per (pair, blob)).  Pass 1 collects the
    // gated blobs of the pair as a bit mask (M <= 64); pass 2 extracts them in (distance, index) order by repeated minimum
    // (hit lists are a handful long; distances are recomputed: same expression, same bits).
    // Stage 1 of the same pass: the PROVISIONAL roots -- the blobs of cameras 1 .. C-1 the camera-0 roots left unclaimed
    // (only they can become roots, helpers.py:402-406) -- against the cameras after them, into staging rows (the search's
    // dead arrays); the chain over the cameras below then only decides which provisional roots are real.  A frame whose
    // provisional roots do not fit the staging rows (or one wave) takes the chain with its own matching, as before.
    auto U_of = [&](int j) {  // provisional blobs of camera j (valid behind stage 0's barrier)
      const int n = cnt[j];
      return ~claimw[j] & (n >= 64 ? ~0ull : ((1ull << n) - 1ull));
    };
    int n_prov = 0, n_ppairs = 0;
    bool pre = false;
    uint8_t* nh_s = nullptr;
    uint8_t* hits_s = nullptr;
    unsigned long long* pclaim = (unsigned long long*)scr;
    {
      int Mmax = 0;
      for (int c = 1; c < C; c++) Mmax = cnt[c] > Mmax ? cnt[c] : Mmax;  // wave-uniform
#pragma nounroll
      for (int stage_ = 0; stage_ < 2; stage_++) {
        const int stage = __builtin_amdgcn_readfirstlane(stage_);  // (opaque: one copy of the pair code)
        int NP = n0 * (C - 1);
        if (stage == 1) {
          // (explicit: the compiler emitted THIS barrier without the s_waitcnt lgkmcnt(0) every other one has -- a wave could
          // read the claim words while another wave's ds_or of stage 0 was still queued: provisional sets that differed between
          // waves, one wrong frame in ~10^5)
          block_sync_lds();  // the camera-0 roots' claims are complete
          for (int j = 1; j < C; j++) {
            const int np = __popcll(U_of(j));
            n_prov += np;
            n_ppairs += np * (C - 1 - j);
          }
          pre = n_prov <= 64 && (size_t)n_prov * C * (8 + 1 + (size_t)M) <= scr_bytes;
          if (!pre) break;
          nh_s = (uint8_t*)(pclaim + (size_t)n_prov * C);
          hits_s = nh_s + (size_t)n_prov * C;
          NP = n_ppairs;
        }
        for (int base = wave * 64; base < NP; base += W * 64) {
          const int pi = base + lane;
          const bool have = pi < NP;
          int r = 0, i = 1, Mi = 0;   // r: row the pair writes (stage 0: the root; stage 1: the provisional root's staging row)
          Line L = {0, 0, 0, 1, 1};
          if (have) {
            if (stage == 0) {
              r = pi / (C - 1);
              i = 1 + (pi - r * (C - 1));
              L = epiline(r, i);
            } else {
              int j = 1, acc = 0, pb = 0, span = C - 2;
              unsigned long long U = 0ull;
              for (; j <= C - 2; j++) {
                span = C - 1 - j;
                U = U_of(j);
                const int nj = __popcll(U) * span;
                if (pi < acc + nj) break;
                acc += nj;
                pb += __popcll(U);
              }
              uint32_t ord, off;
              divmod_tiny((uint32_t)(pi - acc), (uint32_t)span, ord, off);  // (< 64 x 15, span <= 14)
              for (uint32_t t = 0; t < ord; t++) U &= U - 1;
              const int k = __ffsll((long long)U) - 1;
              i = j + 1 + (int)off;
              r = pb + (int)ord;
              L = epiline_of(j, k, i);
            }
            Mi = cnt[i];
          }
          const float2* pts = bxy + (size_t)i * M;
          auto dist = [&](int k) {
            const float2 pt = pts[k];
            return div_by(fabs(L.a * (double)pt.x + L.b * (double)pt.y + L.c), L.den, L.rden);  // helpers.py:373
          };
          unsigned long long hm = 0ull;
#pragma unroll 4
          for (int k = 0; k < Mmax; k++)  // (independent iterations: unrolled, their latencies overlap -- the stage runs on one or two waves)
            if (k < Mi && dist(k) < p.gate_px) hm |= 1ull << k;  // strict <, helpers.py:375,383
          uint8_t* nhp = stage == 0 ? nh : nh_s;
          uint8_t* hl = (stage == 0 ? hits : hits_s) + ((size_t)r * C + i) * M;
          if (have) nhp[(size_t)r * C + i] = (uint8_t)__popcll(hm);
          // order by (distance, blob index): a stable order where NumPy's default argsort is not (helpers.py:384;
          // documented deviation); the closest hit's coordinates claim every blob that has them (helpers.py:391)
          unsigned long long rem = hm, claim = 0ull;
          int pos = 0;
          float2 p0 = make_float2(0.f, 0.f);
          while (__ballot(rem != 0ull)) {
            if (rem) {
              double bd = __builtin_huge_val();
              int bk = 0;
              for (unsigned long long t = rem; t; t &= t - 1) {  // ascending index, strict <: ties keep the smaller index
                const int k = __ffsll((long long)t) - 1;
                const double d = dist(k);
                if (d < bd) {
                  bd = d;
                  bk = k;
                }
              }
              hl[pos] = (uint8_t)bk;
              if (pos == 0) {
                p0 = pts[bk];
                for (unsigned long long t = hm; t; t &= t - 1) {
                  const int k = __ffsll((long long)t) - 1;
                  const float2 q = pts[k];
                  if (q.x == p0.x && q.y == p0.y) claim |= 1ull << k;
                }
              }
              rem &= ~(1ull << bk);
              pos++;
            }
          }
          if (stage == 0) {
            if (claim) atomicOr(&claimw[i], claim);
          } else if (have) {
            pclaim[(size_t)r * C + i] = claim;
          }
        }
      }
    }
    // Speculative lines (only for frames that take the old chain): the epipolar lines of the blobs that might become roots
    // depend on nothing, so all waves compute them off the chain's critical path.  Table = the search's dead arrays: line
    // (j, k) -> i at sp_base(j) + k (C-1-j) + (i-j-1), float32 a, b, c (F32R: that is what they are, helpers.py:364) + den.
    const int NL = M * ((C - 1) * (C - 2) / 2);
    const bool spec = F32R && (size_t)NL * 20 + 8 <= scr_bytes;
    double* sp_den = (double*)scr;
    float* sp_abc = (float*)(sp_den + NL);
    auto sp_base = [&](int j) { return M * ((j - 1) * (C - 1) - (j - 1) * j / 2); };  // lines of cameras 1 .. j-1
    // ... of the blobs the camera-0 roots have NOT claimed only (a claimed blob never becomes a root): one lane per
    // (unclaimed blob, later camera) pair, ~40 lines per frame.
    if (spec && !pre) {  // (behind stage 0's barrier: the claim words are complete)
      int tot = 0;
      for (int j = 1; j <= C - 2; j++) {
        const int n = cnt[j];
        const unsigned long long U = ~claimw[j] & (n >= 64 ? ~0ull : ((1ull << n) - 1ull));
        tot += __popcll(U) * (C - 1 - j);
      }
      for (int l = tid; l < tot; l += T) {
        int j = 1, acc = 0, span = C - 2;
        unsigned long long U = 0ull;
        for (; j <= C - 2; j++) {
          const int n = cnt[j];
          U = ~claimw[j] & (n >= 64 ? ~0ull : ((1ull << n) - 1ull));
          span = C - 1 - j;
          const int nj = __popcll(U) * span;
          if (l < acc + nj) break;
          acc += nj;
        }
        uint32_t ord, off;
        divmod_tiny((uint32_t)(l - acc), (uint32_t)span, ord, off);  // (< 64 x 15, span <= 14)
        for (uint32_t t = 0; t < ord; t++) U &= U - 1;
        const int k = __ffsll((long long)U) - 1, i = j + 1 + (int)off;
        const int ll = sp_base(j) + k * span + (int)off;
        const Line L = epiline_of(j, k, i);
        sp_den[ll] = L.den;
        sp_abc[3 * ll + 0] = (float)L.a;
        sp_abc[3 * ll + 1] = (float)L.b;
        sp_abc[3 * ll + 2] = (float)L.c;
      }
    }
    __syncthreads();
    fresh_tid();
    if (wave == 0) bb_prio<kPrioSerial>();
    if (wave == 0 && pre) {
      // B1, pre-matched: lane p <-> provisional root p = the p-th unclaimed blob of cameras 1 .. C-1 in (camera, blob) order.
      // A provisional root is real when no root created at an earlier camera claims its blob (helpers.py:391,402-406).
      int jp = 0, kp = 0;
      if (lane < n_prov) {
        int acc = 0;
        unsigned long long U = 0ull;
        for (jp = 1; jp < C; jp++) {
          U = U_of(jp);
          const int np = __popcll(U);
          if (lane < acc + np) break;
          acc += np;
        }
        for (int t = 0; t < lane - acc; t++) U &= U - 1;
        kp = __ffsll((long long)U) - 1;
      }
      // Camera by camera: the provisional roots of camera i look their blob up in the camera's claim word; the real ones add
      // their own claims to the words of the cameras after them (the claim words end up as the reference's "claimed" sets).
      // One LDS read and a handful of atomics per camera -- a wave-wide OR by shuffles was 12 dependent cross-lane operations per camera.
      bool real = false;
      for (int i = 1; i < C; i++) {
        wave_lds_sync();
        if (lane < n_prov && jp == i) {
          real = !((claimw[i] >> kp) & 1ull);
          if (real)
            for (int c = i + 1; c < C; c++) {
              const unsigned long long m = pclaim[(size_t)lane * C + c];
              if (m) atomicOr(&claimw[c], m);
            }
        }
      }
      const unsigned long long rm = __ballot(real);
      const int rank = __popcll(rm & ((1ull << lane) - 1ull));
      int n_roots = n0 + __popcll(rm);
      if (real && n0 + rank < R) {
        root_cam[n0 + rank] = (uint8_t)jp;
        root_blob[n0 + rank] = (uint8_t)kp;
        bnl[rank] = (uint8_t)lane;  // (bnl is dead until phase C: the staging row of the rank-th new root)
      }
      if (lane == 0) {
        if (n_roots > R) {
          misc[MI_STATUS] |= MOCAP_ST_ROOT_OVERFLOW_;
          n_roots = R;
        }
        misc[MI_NROOTS] = n_roots;
      }
    } else if (wave == 0) {
      // B1: the chain over the cameras -- only the roots created on the way take part (camera-0 roots are done).
      // State in registers: lane l <-> camera l (blob count, blobs claimed so far), lane l <-> the l-th new root.
      const int cntL = lane < C ? cnt[lane] : 0;
      const unsigned long long clmL = lane < C ? claimw[lane] : 0ull;  // the camera-0 roots' claims (B0 is complete)
      int nr_cam = 0, nr_blob = 0;
      int n_roots = n0;
      bool over = false;
      for (int i = 1; i < C; i++) {
        const int Mi = __builtin_amdgcn_readlane(cntL, i);
        unsigned long long claimed = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(clmL >> 32), i) << 32) |
                                     (unsigned int)__builtin_amdgcn_readlane((int)clmL, i);
        const int n_new = n_roots - n0;
        for (int base = 0; base < n_new; base += 64) {
          const int l = base + lane;
          const bool have = l < n_new;  // created at an earlier camera
          int j = nr_cam, k = nr_blob;
          if (base) {                   // (beyond the 64 kept in registers: from the root arrays this wave wrote)
            wave_lds_sync();
            j = have ? root_cam[n0 + l] : 0;
            k = have ? root_blob[n0 + l] : 0;
          }
          Line L = {0, 0, 0, 1, 1};
          if (have) {
            if (spec) {
              const int ll = sp_base(j) + k * (C - 1 - j) + (i - j - 1);
              L.a = (double)sp_abc[3 * ll + 0];
              L.b = (double)sp_abc[3 * ll + 1];
              L.c = (double)sp_abc[3 * ll + 2];
              L.den = sp_den[ll];
              L.rden = recip_refined(L.den);
            } else {
              L = epiline_of(j, k, i);
            }
          }
          claimed |= match_pairs<true>(L, n0 + l, i, (n_new - base < 64) ? n_new - base : 64, gs_shift, i, Mi);
        }
        // unclaimed blobs become new roots, in blob order (helpers.py:402-406)
        const bool flag = lane < Mi && !((claimed >> lane) & 1ull);
        const unsigned long long mask = __ballot(flag);
        const int pos = __popcll(mask & ((1ull << lane) - 1ull));
        if (flag) {
          const int rr = n_roots + pos;
          if (rr < R) {
            root_cam[rr] = (uint8_t)i;
            root_blob[rr] = (uint8_t)lane;
          }
        }
        int slot = n_new;
        for (unsigned long long mm = mask; mm; slot++) {  // wave-uniform: the new roots' registers
          const int jb = __ffsll((long long)mm) - 1;
          mm &= mm - 1;
          if (lane == slot) {
            nr_cam = i;
            nr_blob = jb;
          }
        }
        n_roots += __popcll(mask);
        if (n_roots > R) {
          over = true;
          n_roots = R;
        }
      }
      if (lane == 0) {
        if (over) misc[MI_STATUS] |= MOCAP_ST_ROOT_OVERFLOW_;
        misc[MI_NROOTS] = n_roots;
      }
    } else {
      // meanwhile: DLT contribution of every blob, once per frame (a candidate group then ADDS ten doubles per view),
      // and the largest coordinate (float32 allowance of the bounds)
      float om = 0.0f;
      for (int i = tid - 64; i < C * M; i += T - 64) {
        const int c = i / M, k = i - c * M;
        if (k < cnt[c]) {
          const float2 v = bxy[i];
          om = fmaxf(om, fmaxf(fabsf(v.x), fabsf(v.y)));
          double Bc[10];
          dlt_contribution(Bc, cv.pq(12 * c), (double)v.x, (double)v.y);
#pragma unroll
          for (int e = 0; e < 10; e++) bt[(size_t)i * 10 + e] = Bc[e];
        }
      }
      if (om > 0.0f) atomicMax(&misc[MI_OMAX], __float_as_int(om));
      if (tid == 64) {
        const int it = q_add(&p.q.counters[QC_NEXT_FRAME], 1);
        misc[MI_NEXT] = it < frame_count(p) ? it : -1;
      }
    }
    bb_prio<kPrioPhase>();
    __syncthreads();
    if (pre) {
      // the real provisional roots' rows leave the staging area for their final place: one lane per (new root, camera)
      const int n_new = misc[MI_NROOTS] - n0;
      for (int idx = tid; idx < n_new * C; idx += T) {
        const int q = idx / C, c = idx - q * C;
        const int r = n0 + q, ps = bnl[q];
        if (c > (int)root_cam[r]) {
          const int n = nh_s[(size_t)ps * C + c];
          nh[(size_t)r * C + c] = (uint8_t)n;
          const uint8_t* src = hits_s + ((size_t)ps * C + c) * M;
          uint8_t* dst = hits + ((size_t)r * C + c) * M;
          for (int t = 0; t < n; t++) dst[t] = src[t];
        }
      }
      __syncthreads();  // (the staging rows are the search's arrays: phase C initialises them next)
    }

    // C: candidate counts per root
    fresh_tid();
    const int nroots = misc[MI_NROOTS];
    for (int r = tid; r < nroots; r += T) {
      const int rc = root_cam[r];
      unsigned long long total = 1;
      int views = 1, na = 0;
      bool over = false;
      for (int c = rc + 1; c < C; c++) {
        const unsigned n = nh[(size_t)r * C + c];
        if (n > 1) act[(size_t)r * C + na++] = (uint8_t)c;  // multi-hit cameras = the digits of the candidate index
        if (n) {
          views++;
          total *= n;
          if (total > (unsigned long long)p.G_cap) {
            over = true;
            total = 1;
          }
        }
      }
      if (over) atomicOr(&misc[MI_STATUS], MOCAP_ST_CAND_OVERFLOW_);
      nact[r] = (uint8_t)na;
      bv[r] = (uint8_t)views;
      rbound[r] = kInfBits;
      const uint32_t g = (views > 1 && !over) ? (uint32_t)total : 0u;  // helpers.py:413-414 drops 1-view roots
      gcnt[r] = g;
      // the search's blocks of this root (phase D): the nl fastest digits stay open (pl >= bb_pl candidates per block), one
      // block per value of the others.  Here rather than at the start of the search: the same lanes, no extra barrier pair.
      {
        const uint8_t* a = act + (size_t)r * C;
        uint32_t pl = 1, nb = 1;
        int nl = 0;
        while (nl < na && pl < (uint32_t)p.bb_pl) pl *= nh[(size_t)r * C + a[nl++]];
        for (int k = nl; k < na; k++) nb *= nh[(size_t)r * C + a[k]];
        bpl[r] = (uint16_t)pl;
        bnl[r] = (uint8_t)nl;
        bnb[r] = g ? nb : 0u;
        seedkey[r] = 0ull;  // (the speculative lines that lived here are dead: the chain is behind the barrier above)
        seedgh[r] = 0xFFFFFFFFu;
      }
    }
    for (int s = tid; s < W * RS; s += T) {
      slot_key[s] = ~0ull;
      slot_g[s] = 0xFFFFFFFFu;
    }
    if (tid == 0) misc[MI_BBCTR] = misc[MI_BBCTR2] = 0;
    __syncthreads();
    if (wave == 0) bb_prio<kPrioSerial>();
    if (wave == 0) {  // candidate offsets, block offsets and output slots: scans over the roots, 64 at a time (sums stay below 2^32: 255 x 2^24)
      uint32_t carry = 0, bcarry = 0;
      int slots = 0;
      for (int base = 0; base < nroots; base += 64) {
        const int r = base + lane;
        const uint32_t g = r < nroots ? gcnt[r] : 0u;
        const uint32_t nb = r < nroots ? bnb[r] : 0u;
        const uint32_t incl = wave_inclusive_scan(g, lane);
        const uint32_t bincl = wave_inclusive_scan(nb, lane);
        const unsigned long long nz = __ballot(g != 0u);
        if (r < nroots) {
          goff[r] = carry + incl - g;
          boff[r] = bcarry + bincl - nb;
          outslot[r] = g ? slots + __popcll(nz & ((1ull << lane) - 1ull)) : -1;
        }
        carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        bcarry += (uint32_t)__builtin_amdgcn_readlane((int)bincl, 63);
        slots += __popcll(nz);
      }
      if (lane == 0) {
        goff[nroots] = carry;
        boff[nroots] = bcarry;
        misc[MI_NOUT] = slots;
        misc[MI_G] = misc[MI_STATUS] ? 0 : (int32_t)carry;
      }
    }
    bb_prio<kPrioPhase>();
    __syncthreads();
  }

  // ---------------------------------------------------------------- phase D
  // DLT matrix of candidate `rem` of root r with the first `skip` multi-hit cameras left open (skip = 0: the whole
  // group, rem = candidate index; skip = bnl[r]: the block's partial group, rem = block index).  Returns the views.
  template <bool WITH_B>
  __device__ __forceinline__ int group_matrix(int r, uint32_t rem, int skip, double (&B)[10], Packed<CW>& pk) const {
    const int C = cn();
    const int rc = root_cam[r];
    const uint8_t* nhr = nh + (size_t)r * C;
    const uint8_t* hr = hits + (size_t)r * C * M;
    pk.clear();
    int v = 0, ka = 0;
    if (WITH_B) {
#pragma unroll
      for (int e = 0; e < 10; e++) B[e] = 0.0;
    }
    for (int c = 0; c < C; c++) {
      uint32_t k = 0xFFu;
      if (c == rc) {
        k = root_blob[r];
      } else if (c > rc) {
        const uint32_t n = nhr[c];
        if (n == 1) {
          k = hr[(size_t)c * M];
        } else if (n > 1) {
          if (ka >= skip) {
            uint32_t qd, dgt;
            if (rem < 8192u) divmod_tiny(rem, n, qd, dgt); else  // (the usual case for every lane of the wave: one path runs)
            divmod_small(rem, n, qd, dgt);
            rem = qd;
            k = hr[(size_t)c * M + dgt];
          }
          ka++;
        }
      }
      if (k != 0xFFu) {
        if (WITH_B) {
          const double* t = bt + ((size_t)c * M + k) * 10;
#pragma unroll
          for (int e = 0; e < 10; e++) B[e] = B[e] + t[e];
        }
        v++;
        pk.set(c, k);
      }
    }
    return v;
  }

  // Record r of the queue and candidate i of the expanded list -> the candidate's root, index, blob indices and DLT matrix
  // (cameras in ascending order: the one canonical rounding of B; a camera that is not in the group adds the table's record
  // of zeros: x + (+0.0) = x for every x the sum can hold -- it starts at +0.0, so it is never -0.0 -- the same bits
  // without a save-exec / branch / restore around every camera).  Returns the views (the root's: every candidate of a root
  // has the same cameras).
  __device__ __forceinline__ int fetch_candidate(uint32_t lo, uint32_t i, int& r, uint32_t& gl, Packed<CW>& pk, double (&B)[10]) const {
    const int C = cn();
    const BRec rec = recs[lo];
    r = (int)(rec.rs & 0xFFu);
#pragma unroll
    for (int k = 0; k < CW; k++) pk.w[k] = rpk[(size_t)lo * CW + k];
    if (rec.rs >> 31) {
      gl = rec.gh;  // a single candidate: its blob indices are complete
    } else {
      uint32_t rem = i - (rec.rs >> 8);
      gl = rec.gh * (uint32_t)bpl[r] + rem;
      const uint8_t* a = act + (size_t)r * C;
      const int nl = bnl[r];
      for (int k = 0; k < nl; k++) {  // the block's open digits (rem < pl < 2^13, hit counts <= 64: divmod_tiny is exact)
        const int c = a[k];
        uint32_t qd, dgt;
        divmod_tiny(rem, nh[(size_t)r * C + c], qd, dgt);
        rem = qd;
        pk.set(c, hits[((size_t)r * C + c) * M + dgt]);
      }
    }
#pragma unroll
    for (int ee = 0; ee < 10; ee++) B[ee] = 0.0;
#pragma unroll CT > 0 ? CT : 1
    for (int c = 0; c < C; c++) {
      const uint32_t k = pk.get(c);
      const double* t = bt + (size_t)(k != 0xFFu ? (uint32_t)c * (uint32_t)M + k : (uint32_t)C * (uint32_t)M) * 10;
#pragma unroll
      for (int ee = 0; ee < 10; ee++) B[ee] = B[ee] + t[ee];
    }
    return bv[r];
  }

  __device__ void search(bool bound_tests) {
    fresh_tid();
    const int C = cn();
    (void)C;
    const int nroots = misc[MI_NROOTS];
    // queued records | their candidates << 10.  Two words that take turns: a flush moves on to the other one (zero since the
    // flush before), and lane 0 zeroes the one just used behind the round's barrier -- nobody touches it until the next
    // flush has passed its own barrier: one barrier per evaluation round instead of two
    int32_t* ctr = &misc[MI_BBCTR];
    int32_t* ctr_other = &misc[MI_BBCTR2];
    const double inf = __builtin_huge_val();
    EigCut ec;
    {
      const double om = (double)__int_as_float(misc[MI_OMAX]);
      ec.p3max2 = p.p3max2;
      ec.o2slack = (1100.0 * 0x1p-46) * (om * om);
    }
    const double c0[3] = {p.bb_c0[0], p.bb_c0[1], p.bb_c0[2]};  // origin of the frame the block bounds are taken in
    // (blocks per root, their offsets, the result slots and the queue counters were set up with the candidate counts: match(), phase C)
    const uint32_t nblocks = boff[nroots];
    // root of block b0 + tid (consecutive blocks per wave): last root whose first block is <= b
    auto root_of_block = [&](uint32_t b_first_of_wave, uint32_t b) {
      return coop_last_le([&](int r) { return boff[r]; }, nroots, b_first_of_wave, b, lane);
    };
    auto push_block = [&](int r, uint32_t gh, const Packed<CW>& pk) {
      const uint32_t old = (uint32_t)atomicAdd(ctr, (int32_t)(((uint32_t)bpl[r] << 10) | 1u));
      const uint32_t slot = old & 0x3FFu;
      BRec rec;
      rec.gh = gh;
      rec.rs = (uint32_t)r | ((old >> 10) << 8);
      recs[slot] = rec;
#pragma unroll
      for (int k = 0; k < CW; k++) rpk[(size_t)slot * CW + k] = pk.w[k];
    };
    auto rec_start = [&](int k) { return (recs[k].rs & 0x7FFFFFFFu) >> 8; };
    // EigCut's first test of a (partial) group of root r against the best error of the root so far
    auto dropped = [&](int r, double s1, double tr) {
      const int vf = bv[r];
      const double bound = __longlong_as_double((long long)rbound[r]);
      const double limit = bound * (double)(2 * vf) * (1.0 + 0x1p-40);
      const double limit_adj = fma(1.002, limit, (double)(2 * vf) * ec.o2slack);
      return s1 * fma(2e-12, tr, p.p3max2c * limit_adj) < 1.0;
    };
    if (bound_tests) {
      bb_prio<kPrioSeed>();
      // ---- 1. seeds: s1 of every block (cached for the tests); per root the block with the largest s1 (smallest
      // bound) almost always holds the winner
      // (a frame of at most T blocks -- the usual one -- takes one trip through this loop: the lane that turns out to hold its
      // root's seed block queues it itself afterwards, with the blob indices it has, instead of one lane per root decoding them again)
      int my_r = -1;
      uint32_t my_gh = 0;
      unsigned long long my_key = 0ull;
      Packed<CW> my_pk;
      my_pk.clear();
      for (uint32_t s0 = 0; s0 < nblocks; s0 += T) {
        const uint32_t b = s0 + (uint32_t)tid;
        if (s0 + (uint32_t)(wave * 64) >= nblocks) continue;  // wave-uniform
        const int r = root_of_block(s0 + (uint32_t)(wave * 64), b < nblocks ? b : nblocks - 1);
        if (b < nblocks) {
          const uint32_t gh = b - boff[r];
          double B[10], tr = 0.0;
          Packed<CW> pk;
          const int v = group_matrix<true>(r, gh, bnl[r], B, pk);
          double s1d = __builtin_huge_val();  // a one-view partial group carries no information: never dropped, any seed
          float s1 = 0.0f;
          if (v >= 2) {
            s1d = eigcut_s1_shifted(B, c0, tr);
            s1 = (float)fmin(s1d, 3e38);
          }
          if (b < (uint32_t)ncache) {
            bcache[b] = make_float2(__double2float_ru(s1d), __double2float_ru(tr));
#pragma unroll
            for (int k = 0; k < CW; k++) bpk[(size_t)b * CW + k] = pk.w[k];
          }
          my_key = ((unsigned long long)__float_as_uint(s1) << 32) | (unsigned long long)(0xFFFFFFFFu - gh);
          my_r = r;
          my_gh = gh;
          my_pk = pk;
          atomicMax(&seedkey[r], my_key);
        }
      }
      __syncthreads();
      if (nblocks <= (uint32_t)T) {
        if (my_r >= 0 && seedkey[my_r] == my_key) {  // (keys are unique inside a root: exactly one lane per root with blocks)
          seedgh[my_r] = my_gh;
          push_block(my_r, my_gh, my_pk);
        }
      } else
      for (int r = tid; r < nroots; r += T) {
        if (bnb[r]) {
          const uint32_t gh = 0xFFFFFFFFu - (uint32_t)seedkey[r];
          seedgh[r] = gh;
          seedkey[r] = 0ull;
          double B[10];
          Packed<CW> pk;
          group_matrix<false>(r, gh, bnl[r], B, pk);
          push_block(r, gh, pk);
        }
      }
      bb_prio<kPrioEval>();
      __syncthreads();
    }
    // ---- 2. the queued records' candidates (spread over all lanes, whatever root they belong to), then the next
    // blocks' tests, until nothing is left
    uint32_t b0 = 0;
    fresh_tid();
    while (true) {
      const uint32_t cv_ = (uint32_t)*ctr;
      const uint32_t ns = cv_ & 0x3FFu, ne = cv_ >> 10;
      const bool blocks_left = b0 < nblocks;
      const bool full = ns > (uint32_t)(kBBRecs - T);
      if (ns && (full || !blocks_left || ne >= (uint32_t)p.bb_flush)) {
        for (uint32_t i0 = 0; i0 < ne; i0 += T) {
          const uint32_t i = i0 + (uint32_t)tid;
          const bool have = i < ne;
          double e = inf, X[3] = {0, 0, 0};
          int r = 0;
          uint32_t gl = 0;
          // the record that candidate i of the expanded list belongs to: last one starting at or before i
          uint32_t lo = 0;
          if (i0 + (uint32_t)(wave * 64) < ne)  // wave-uniform
            lo = (uint32_t)coop_last_le(rec_start, (int)ns, i0 + (uint32_t)(wave * 64), have ? i : ne - 1, lane);
          if (have) {
            Packed<CW> pk;
            double B[10];
            const int v = fetch_candidate(lo, i, r, gl, pk, B);
            auto obs_p = [&](int c, double& x, double& y) -> bool {
              const uint32_t k = pk.get(c);
              if (k == 0xFFu) return false;
              const float2 w = bxy[(size_t)c * M + k];
              x = (double)w.x;
              y = (double)w.y;
              return true;
            };
            const double bound = __longlong_as_double((long long)rbound[r]);
            solve_and_score<true, true, F32R, false>(cv, B, v, obs_p, X, e, bound * (double)(2 * v) * (1.0 + 0x1p-40), ec);
#ifdef MOCAP_DEBUG_EIGCHECK  // self-check build: a candidate whose evaluation was cut short must not beat the bound it was cut against
            if (!(e < inf)) {
              double B2[10], X2[3], e2;
#pragma unroll
              for (int ee = 0; ee < 10; ee++) B2[ee] = 0.0;
              for (int c = 0; c < C; c++) {
                const uint32_t k = pk.get(c);
                if (k != 0xFFu) {
                  const double* t = bt + ((size_t)c * M + k) * 10;
#pragma unroll
                  for (int ee = 0; ee < 10; ee++) B2[ee] = B2[ee] + t[ee];
                }
              }
              solve_and_score<true, true, F32R>(cv, B2, v, obs_p, X2, e2);
              atomicAdd(&p.status[p.n_frames], 1);
              if (e2 <= bound) printf("EIGCHECK candidate: root %d g %u bound %.17g true %.17g\n", r, gl, bound, e2);
            }
#endif
          }
          // deliver: lexicographic minimum of (error bits, candidate index) in the (wave, root) slot; an error that is
          // not finite counts as +inf (such a candidate only ever stands when the root has no finite error at all);
          // the index word carries "the error was NaN" in its lowest bit (the error itself is not stored)
          const bool fin = e < inf;
          const unsigned long long key = fin ? (unsigned long long)__double_as_longlong(e) : kInfBits;
          const uint32_t gword = (gl << 1) | ((e != e) ? 1u : 0u);
          const int ss = wave * RS + r;
          if (fin) atomicMin(&rbound[r], key);
          const bool want = have && (key < slot_key[ss] || (key == slot_key[ss] && gword < slot_g[ss]));
          if (__ballot(want)) {
            unsigned long long old = 0;
            if (want) old = atomicMin(&slot_key[ss], key);
            wave_lds_sync();
            const bool holder = want && slot_key[ss] == key;
            if (holder && old > key) slot_g[ss] = 0xFFFFFFFFu;  // the error went down in this round: any index is better
            wave_lds_sync();
            if (holder) atomicMin(&slot_g[ss], gword);
            wave_lds_sync();
          }
        }
        __syncthreads();  // every lane is done with the records: new ones may be queued (through the other counter)
        if (tid == 0) *ctr = 0;
        {
          int32_t* t = ctr;
          ctr = ctr_other;
          ctr_other = t;
        }
        continue;
      }
      if (!blocks_left) break;
      const uint32_t b = b0 + (uint32_t)tid;
      const uint32_t bw = b0 + (uint32_t)(wave * 64);
      b0 += T;
      int r = 0;
      if (bw < nblocks) r = root_of_block(bw, b < nblocks ? b : nblocks - 1);  // wave-uniform branch
      // The tests first, the pushes behind a barrier.  Every lane read the queue counter at the top of this iteration; a wave that
      // is through its tests early must not bump it before the slowest wave has read it -- that wave would see a different queue,
      // take the other branch of the flush decision and the workgroup would fall apart (a window of a few dozen instructions
      // when the cached blocks need no decode: found in round 5 as 1 wrong frame in ~10^4 with 2-candidate blocks).
      bool do_push = false;
      uint32_t gh = 0;
      Packed<CW> pk;
      pk.clear();
      if (b < nblocks) {
        gh = b - boff[r];
        if (gh != seedgh[r]) {
          double B[10], tr;
          bool survive = true;
          if (bound_tests && b < (uint32_t)ncache) {
            // {s1, trace} from the seed pass, rounded up: a larger s1 or trace only ever keeps a block (safe side)
            const float2 sc = bcache[b];
            survive = !dropped(r, (double)sc.x, (double)sc.y);
#pragma unroll
            for (int k = 0; k < CW; k++) pk.w[k] = bpk[(size_t)b * CW + k];
          } else if (bound_tests) {
            const int v = group_matrix<true>(r, gh, bnl[r], B, pk);
            if (v >= 2) {
              const double s1 = eigcut_s1_shifted(B, c0, tr);
              survive = !dropped(r, s1, tr);
            }
          } else {
            group_matrix<false>(r, gh, bnl[r], B, pk);
          }
          do_push = survive;
#ifdef MOCAP_DEBUG_EIGCHECK  // self-check build: EVERY candidate of a dropped block is evaluated in full against the bound it was dropped on
          if (!survive) {
            const double bound = __longlong_as_double((long long)rbound[r]);
            const uint32_t pl = bpl[r];
            for (uint32_t l = 0; l < pl; l++) {
              double B2[10], X2[3], e2 = inf;
              Packed<CW> pk2;
              const int v2 = group_matrix<true>(r, gh * pl + l, 0, B2, pk2);
              auto obs2 = [&](int c, double& x, double& y) -> bool {
                const uint32_t k = pk2.get(c);
                if (k == 0xFFu) return false;
                const float2 w = bxy[(size_t)c * M + k];
                x = (double)w.x;
                y = (double)w.y;
                return true;
              };
              solve_and_score<true, true, F32R>(cv, B2, v2, obs2, X2, e2);
              atomicAdd(&p.status[p.n_frames + 1], 1);
              if (e2 <= bound) printf("EIGCHECK block: root %d block %u candidate %u bound %.17g true %.17g\n", r, gh, l, bound, e2);
            }
          }
#endif
        }
      }
      __syncthreads();
      if (do_push) push_block(r, gh, pk);
      __syncthreads();
    }
    __syncthreads();
  }

  // winner of root r: the (wave, root) slots merged
  __device__ bool root_winner(int r, double& eb, uint32_t& gb) const {
    if (!gcnt[r]) return false;
    unsigned long long kb = ~0ull;
    uint32_t gw = 0xFFFFFFFFu;
    for (int w = 0; w < W; w++) {
      const int s = w * RS + r;
      const unsigned long long k = slot_key[s];
      const uint32_t g = slot_g[s];
      if (k < kb || (k == kb && g < gw)) {
        kb = k;
        gw = g;
      }
    }
    if (kb == ~0ull) return false;
    gb = gw >> 1;
    eb = kb != kInfBits ? __longlong_as_double((long long)kb)
                        : ((gw & 1u) ? __longlong_as_double(0x7ff8000000000000ll) : __builtin_huge_val());
    return true;
  }

  // ---------------------------------------------------------------- phase E
  // The winner's point is not kept while the search runs (12 doubles per root and wave: 4.5 of the frame's 33.8 KB of LDS,
  // the difference between four and five frames per CU): the group is decoded here anyway, its DLT matrix is the canonical
  // sum again (cameras ascending, fetch_candidate's bits) and the point is solve_and_score's own arithmetic on it -- nothing
  // of which depends on the bound the evaluation was cut against.
  __device__ void write_point(int64_t frame, int r, double e, uint32_t gl) const {
    const int C = cn();
    const size_t o = (size_t)frame * R + outslot[r];
    p.err[o] = e;
    double B[10];
#pragma unroll
    for (int ee = 0; ee < 10; ee++) B[ee] = 0.0;
    uint32_t rem = gl;  // decode the winning group
    const int rc = root_cam[r];
    int16_t* co = p.corr + o * C;
    for (int c = 0; c < C; c++) {
      int16_t s = -1;
      if (c == rc) {
        s = (int16_t)root_blob[r];
      } else if (c > rc) {
        const uint32_t n = nh[(size_t)r * C + c];
        if (n) {
          uint32_t qd = rem, dgt = 0;
          if (n > 1) {  // (a single hit is digit 0 of radix 1: nothing to divide)
            if (rem < 8192u) divmod_tiny(rem, n, qd, dgt); else
            divmod_small(rem, n, qd, dgt);
          }
          s = (int16_t)hits[((size_t)r * C + c) * M + dgt];
          rem = qd;
        }
      }
      co[c] = s;
      if (s >= 0) {
        const double* t = bt + ((size_t)c * M + (uint32_t)s) * 10;
#pragma unroll
        for (int ee = 0; ee < 10; ee++) B[ee] = B[ee] + t[ee];
      }
    }
    double X[3];
    solve_point(B, X);
    store_point(p, o, X);  // incl. the fused world-coordinate epilogue (helpers.py:96-103)
  }
};

template <bool F32R, int CW, int CT, int ML, int RL>
__global__ __launch_bounds__(kBBThreads, bb_wg_per_cu(RL)) void frame_bb_kernel(FrameArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  BBState<F32R, CW, CT, ML, RL> st(p, smem);
  const int tid = threadIdx.x;
  const FrameQueues& q = p.q;
  // software pipeline over the frames: while frame k is searched, frame k + 1 has been pulled from the queue (one frame
  // per pull: frames of this size are never cheap enough for the queue atomic to matter) and is on its way from HBM
  // into the spare LDS buffer
  if (tid == 0) {
    const int it = q_add(&q.counters[QC_NEXT_FRAME], 1);
    st.misc[MI_ITEM] = it < frame_count(p) ? it : -1;
  }
  if (tid < 10) st.bt[(size_t)st.cn() * st.M * 10 + tid] = 0.0;  // the table's record of zeros (never overwritten)
  __syncthreads();
  int item = st.misc[MI_ITEM];
  if (item >= 0) st.prefetch_lds(item);
  wait_own_stores();
  __syncthreads();
  while (item >= 0) {
    const int64_t frame = item;
    bb_prio<kPrioPhase>();
    st.stage(frame);
    st.match();
    bb_prio<kPrioEval>();
    const int next = st.misc[MI_NEXT];
    if (next >= 0) st.prefetch_lds(next);  // in flight during the search
    if (tid == 0) {
      const int status = st.misc[MI_STATUS];
      p.n_out[frame] = status ? 0 : st.misc[MI_NOUT];
      p.status[frame] = status;
      if (p.n_cand) p.n_cand[frame] = st.misc[MI_G];
    }
    const uint32_t G = (uint32_t)st.misc[MI_G];
    if (G) {
      // the bound tests pay their fixed cost (seed pass + a test per block) only on frames with enough candidates;
      // smaller frames queue every block -- same evaluation rounds, same result
      st.search(G >= (uint32_t)p.bb_min_g);
      bb_prio<kPrioPhase>();
      const int nroots = st.misc[MI_NROOTS];
      st.fresh_tid();
      for (int r = st.tid; r < nroots; r += kBBThreads) {
        if (st.outslot[r] < 0) continue;
        double e;
        uint32_t gl;
        if (st.root_winner(r, e, gl)) st.write_point(frame, r, e, gl);
      }
    }
    wait_own_stores();  // ... and loads: the next frame's blobs are in LDS
    bb_prio<kPrioEval>();
    __syncthreads();  // the frame's LDS state is dead: the next one may be staged
    item = next;
  }
  // the last workgroup to leave puts the queue counters back to zero: the next launch needs no memset
  if (tid == 0 && q_add(&q.counters[QC_EXITED], 1) == (int)gridDim.x - 1)
    for (int c = 0; c < QC_COUNT; c++) q_store(&q.counters[c], 0);
}

int frame_bb_wg_per_cu_cap(int C, int M, int R) { return bb_wg_per_cu(bb_fixed_slots(C, M, R)); }
size_t frame_bb_ws_bytes(int) { return 0; }  // (no per-workgroup HBM workspace: everything between input and output lives in LDS)

hipError_t launch_frame_bb(const FrameArgs& a, int grid, hipStream_t stream) {
  const size_t lds = frame_bb_lds_bytes(a.cv.C, a.M, a.K_max);
  // the kernel addresses the tables from one base (BBCamTables): identical intrinsics, the block layout of mocap_set_cameras
  if (!a.cv.uniformK || a.cv.RT != a.cv.Pq + 12 * a.cv.C || a.cv.K4 != a.cv.Pq + 24 * a.cv.C || a.cv.F != a.cv.Pq + 28 * a.cv.C)
    return hipErrorInvalidValue;
  void (*k)(FrameArgs);
  const bool f32 = a.cv.f32_rounding != 0;
  const int fixed = bb_fixed_slots(a.cv.C, a.M, a.K_max);
  if (fixed == 48)
    k = f32 ? frame_bb_kernel<true, 1, 8, 16, 48> : frame_bb_kernel<false, 1, 8, 16, 48>;
  else if (fixed == 64)
    k = f32 ? frame_bb_kernel<true, 1, 8, 16, 64> : frame_bb_kernel<false, 1, 8, 16, 64>;
  else if (fixed)
    return hipErrorInvalidValue;
  else if (a.cv.C == 8)
    k = f32 ? frame_bb_kernel<true, 1, 8, 0, 0> : frame_bb_kernel<false, 1, 8, 0, 0>;
  else if (a.cv.C <= 8)
    k = f32 ? frame_bb_kernel<true, 1, 0, 0, 0> : frame_bb_kernel<false, 1, 0, 0, 0>;
  else
    k = f32 ? frame_bb_kernel<true, 2, 0, 0, 0> : frame_bb_kernel<false, 2, 0, 0, 0>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(kBBThreads), lds, stream, a);
  return hipGetLastError();
}

}  // namespace mocap
