// heavy_bb.hip -- the candidate space of ONE root when no enumeration reaches its end.
//
// The reference expands the Cartesian product of a root's gated hits over the cameras, whatever its size
// (reference computer_code/api/helpers.py:394-400), triangulates every group and keeps the first minimum of the reprojection
// error (helpers.py:408-421).  Two markers that lie behind each other as seen from the root's camera put TWO hits into
// nearly every other camera: at 64 cameras that is 2^60 groups -- the reference would not return either, and the frame
// kernels cap a root at 2^24 groups and flag the frame.  The winner of such a root can still be found exactly:
//
//   * a group's error is bounded from below by the smallest eigenvalue of its DLT matrix (EigCut, mocap_device.hpp), and the
//     matrix of a PARTIAL group -- some multi-hit cameras not yet decided -- bounds every completion (it is a sum of
//     positive semi-definite terms): the same inequality, with the same rounding allowances, csrc/frame_bb.hip drops its
//     blocks on;
//   * the search goes through the multi-hit cameras level by level (breadth first).  A node = the digits chosen so far + the
//     DLT matrix of the single-hit cameras and those digits; its children (one per hit of the next camera) are tested against
//     the error of group 0 -- the closest hit in every camera, evaluated by the frame kernel itself before this kernel runs,
//     almost always the winner -- and the survivors form the next level's frontier;
//   * the leaves that survive the last level are complete groups: they are triangulated and scored by the SAME device
//     function, in the same camera order, as every other group of the path (triangulate_and_score), and the first minimum
//     in candidate order (error bits, then the mixed-radix index compared digit by digit from the slowest camera down)
//     replaces the root's point if it beats group 0.
//
// Nothing is dropped on anything but a rigorous bound, so the result is the one the enumeration would give
// (tests/test_gpu_wide_adversarial.py runs roots both ways where the enumeration is feasible).  A frontier that outgrows the
// workspace (two markers that coincide to within the noise in most cameras: nothing separates the mixtures) flags the frame
// like the cap did.
//
// One 256-lane workgroup per heavy root (records exported by frame_kernel.hip's wide variant in the re-submit pass);
// identical plain intrinsics only (the host does not export otherwise).
#include "mocap_device.hpp"
#include "kernels.hpp"
#include "frame_common.hpp"

namespace mocap {

constexpr int kHvThreads = 1024;  // (one workgroup per CU: the frontier of a hard root is tens of thousands of nodes per level)
constexpr int kHvEnumDigits = 20;  // multi-hit cameras of a root the fall-back enumerates (product <= 2^20, two hits at least each)
constexpr int kHvGlobalEnumDigits = 24;  // ... of a root heavy_enum_kernel enumerates over the whole GPU (product <= 2^24)
constexpr int kHvEnumThreads = 256;
constexpr int kHvEnumSlice = 16 * kHvEnumThreads;  // groups per slice (roots with more hits than the table holds)
constexpr int kHvEnumTab = 256;  // hits of a root heavy_enum_kernel tabulates (contribution + pixel)
constexpr int kHvDigits = 64;  // digit slots of a node (>= multi-hit cameras of a root: < kMaxCameras)

size_t heavy_bb_ws_bytes(int ncap) { return (size_t)2 * ncap * (sizeof(double) * 10 + kHvDigits); }

template <bool F32R>
__global__ __launch_bounds__(kHvThreads) void heavy_bb_kernel(HeavyArgs a) {
  __shared__ uint16_t s_n[kMaxCameras];      // hits per camera (1 at the root's, 0 where the root has none)
  __shared__ uint8_t s_lvl[kMaxCameras];     // level of a multi-hit camera (0xFF otherwise)
  __shared__ uint8_t s_dcam[kMaxCameras];    // camera of level j
  __shared__ double s_Bs[10];                // DLT matrix of the single-hit cameras (root included)
  __shared__ int s_cnt[2];                   // nodes in the current / next frontier
  __shared__ int s_m;
  __shared__ unsigned long long s_best;      // smallest error among the leaves (bit pattern)
  __shared__ int s_nhold, s_hold[64];        // leaves that hold it
  __shared__ int s_win;
  __shared__ double s_Bp[10];                // greedy descent: the matrix of the path so far
  __shared__ unsigned long long s_gkey;      // ... (s1 bits | 0xFF - digit) of the best child of the level
  __shared__ uint8_t s_gd[kHvDigits];        // ... its digits
  __shared__ double s_eg;                    // ... the error of the group it ends in
  __shared__ uint8_t s_dg[kHvEnumDigits][kHvThreads];  // enumeration fall-back: a lane's digits (column per lane)
  const int tid = threadIdx.x;
  const int C = a.cv.C, M = a.M;
  int total = *a.heavy_count;
  if (total > a.cap) total = a.cap;
  unsigned char* ws = a.ws + (size_t)blockIdx.x * a.ws_stride;
  double* nodeB[2] = {(double*)ws, (double*)ws + (size_t)10 * a.ncap};
  uint8_t* nodeD[2] = {(uint8_t*)((double*)ws + (size_t)20 * a.ncap), (uint8_t*)((double*)ws + (size_t)20 * a.ncap) + (size_t)kHvDigits * a.ncap};
  const double c0[3] = {a.bb_c0[0], a.bb_c0[1], a.bb_c0[2]};
  const double inf = __builtin_huge_val();

  for (int h = blockIdx.x; h < total; h += gridDim.x) {
    const unsigned char* rec = a.recs + (size_t)h * a.stride;
    const HeavyRecHdr hd = *reinterpret_cast<const HeavyRecHdr*>(rec);
    const uint16_t* nc = reinterpret_cast<const uint16_t*>(rec + heavy_rec_counts_off());
    const uint8_t* hl = rec + heavy_rec_hits_off(C);
    const int Hs = hd.Hs;
    const float2* fb = (const float2*)(a.blobs + (size_t)hd.frame * C * M * 2);
    const size_t o = (size_t)hd.frame * a.K_big + hd.outslot;
    block_sync_lds();  // (the previous root's shared state is dead; reached from every exit of the previous iteration: the wait is written out)
    if (tid < C) s_n[tid] = nc[tid];
    if (tid == 0) {
      int m = 0;
      for (int c = 0; c < C; c++) {
        s_lvl[c] = 0xFF;
        if (nc[c] > 1) {
          s_lvl[c] = (uint8_t)m;
          s_dcam[m++] = (uint8_t)c;
        }
      }
      s_m = m;
      s_cnt[0] = 1;
      s_cnt[1] = 0;
      // single-hit cameras, ascending (the order only has to be a fixed one: the sum feeds bounds, never a result)
      double B[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int c = 0; c < C; c++)
        if (nc[c] == 1) {
          const float2 w = fb[(size_t)c * M + hl[(size_t)c * Hs]];
          dlt_accumulate(B, a.cv.pq((size_t)12 * c), (double)w.x, (double)w.y);
        }
      for (int e = 0; e < 10; e++) {
        s_Bs[e] = B[e];
        nodeB[0][e] = B[e];
      }
    }
    __syncthreads();
    const int m = s_m;
    // the bound every node is tested against: the error of group 0, which the frame kernel has written to the root's slot
    const double e00 = a.err[o];
    const int vf = hd.views;
    const double om = (double)__int_as_float(hd.omax_bits);
    EigCut ec;
    ec.p3max2 = a.p3max2;
    ec.o2slack = (1100.0 * 0x1p-46) * (om * om);
    // A second, usually better, bound: when the two markers share the root's pixel (one behind the other as seen from the
    // root's camera) the closest hit of a camera is either marker's blob at random and group 0 is a mixture with an error of
    // many pixels -- nothing could be dropped against it.  A greedy descent -- per level the hit that keeps the partial group's
    // smallest eigenvalue smallest -- stays with one marker; the group it ends in is evaluated like any other.
    double eg = inf;
    for (int pass = 0; pass < 2 && m < kHvDigits; pass++) {
      // (second descent: away from the first one's first digit -- the other marker's group, when there are two to choose from)
      const int avoid = pass ? (int)s_gd[0] : -1;
      __syncthreads();
      if (tid < 10) s_Bp[tid] = s_Bs[tid];
      for (int j = 0; j < m; j++) {
        if (tid == 0) s_gkey = 0ull;
        __syncthreads();
        const int cam = s_dcam[j], nj = s_n[cam];
        double B[10];
        if (tid < nj && !(j == 0 && tid == avoid)) {
#pragma unroll
          for (int e = 0; e < 10; e++) B[e] = s_Bp[e];
          const float2 w = fb[(size_t)cam * M + hl[(size_t)cam * Hs + tid]];
          dlt_accumulate(B, a.cv.pq((size_t)12 * cam), (double)w.x, (double)w.y);
          double tr;
          const double s1 = eigcut_s1_shifted(B, c0, tr);
          atomicMax(&s_gkey, ((unsigned long long)__double_as_longlong(fmin(fmax(s1, 0.0), 1e300)) & ~0xFFull) | (unsigned long long)(0xFF - tid));
        }
        __syncthreads();
        const int dbest = 0xFF - (int)(s_gkey & 0xFFull);
        if (tid == dbest) {
#pragma unroll
          for (int e = 0; e < 10; e++) s_Bp[e] = B[e];
          s_gd[j] = (uint8_t)dbest;
        }
        __syncthreads();
      }
      if (tid == 0) {
        auto obs = [&](int c, double& x, double& y) -> bool {
          const int n = s_n[c];
          if (!n) return false;
          const int d = n > 1 ? s_gd[s_lvl[c]] : 0;
          const float2 w = fb[(size_t)c * M + hl[(size_t)c * Hs + d]];
          x = (double)w.x;
          y = (double)w.y;
          return true;
        };
        double X[3], e = inf;
        triangulate_and_score<true, true, F32R, 1, false>(a.cv, obs, obs, X, e);
        s_eg = e;
      }
      __syncthreads();
      if (s_eg < eg) eg = s_eg;
    }
    const double e0 = (eg < e00) ? eg : e00;  // (NaN-safe: a non-finite greedy error leaves group 0's)
    const double limit = e0 * (double)(2 * vf) * (1.0 + 0x1p-40);
    const double limit_adj = fma(1.002, limit, (double)(2 * vf) * ec.o2slack);
    const double lamcut = a.p3max2c * limit_adj;
    bool give_up = !(e0 < inf) || m >= kHvDigits;  // no finite bound to start from: nothing could be dropped
    int cur = 0;
    for (int j = 0; j < m && !give_up; j++) {
      const int cam = s_dcam[j], nj = s_n[cam], ncur = s_cnt[cur];
      const int nxt = cur ^ 1;
      const int64_t work = (int64_t)ncur * nj;
      for (int64_t idx = tid; idx < work; idx += kHvThreads) {
        const int i = (int)(idx / nj), d = (int)(idx - (int64_t)i * nj);
        double B[10];
#pragma unroll
        for (int e = 0; e < 10; e++) B[e] = nodeB[cur][(size_t)i * 10 + e];
        const float2 w = fb[(size_t)cam * M + hl[(size_t)cam * Hs + d]];
        dlt_accumulate(B, a.cv.pq((size_t)12 * cam), (double)w.x, (double)w.y);
        double tr;
        const double s1 = eigcut_s1_shifted(B, c0, tr);
        if (!(s1 * fma(2e-12, tr, lamcut) < 1.0)) {  // not dropped (a one-view node cannot occur: the root's camera is always in)
          const int pos = atomicAdd(&s_cnt[nxt], 1);
          if (pos < a.ncap) {
#pragma unroll
            for (int e = 0; e < 10; e++) nodeB[nxt][(size_t)pos * 10 + e] = B[e];
            uint8_t* dd = nodeD[nxt] + (size_t)pos * kHvDigits;
            const uint8_t* ds = nodeD[cur] + (size_t)i * kHvDigits;
            for (int k = 0; k < j; k++) dd[k] = ds[k];
            dd[j] = (uint8_t)d;
          }
        }
      }
      __threadfence_block();
      __syncthreads();
      if (s_cnt[nxt] > a.ncap) give_up = true;  // (uniform)
      __syncthreads();
      if (tid == 0) s_cnt[cur] = 0;
      cur = nxt;
      __syncthreads();
    }
    if (a.debug && tid == 0)
      printf("HEAVY rec %d frame %d root %d cam %d m %d views %d e_group0 %.6g e_greedy %.6g give_up %d frontier %d\n", h, hd.frame, hd.root, hd.rc, m, vf,
             e00, eg, (int)give_up, s_cnt[cur]);
    if (give_up) {
      // The frontier outgrew the workspace (or there was no finite bound to start from).  A product that is still small
      // enough is simply enumerated here -- every group through the path's own device function, the running best of the
      // workgroup as cut-off, first minimum in candidate order -- so that whatever an enumeration CAN reach is never lost to the
      // search's limits (a root of 2^15 groups with forced wrong views is enumerable and not searchable; HeavyArgs::enum_cap: 2^16 by default -- 2^20 groups in place would hold one CU for tens of milliseconds).
      double prod = 1.0;
      for (int j = 0; j < m; j++) prod *= (double)s_n[s_dcam[j]];
      if (!(prod <= (double)a.enum_cap) || !(prod <= 1048576.0) || m > kHvEnumDigits) {
        if (tid == 0) {
          // at most 2^24 groups: the whole GPU enumerates the root behind this kernel (heavy_enum_kernel) -- the re-submit
          // pass is exact up to there, as the frame kernels' own enumeration is.  Above: no enumeration reaches it (the
          // reference's own included) and the bound has just failed: the frame says so.
          int slot = -1;
          if (prod <= 16777216.0 && m <= kHvGlobalEnumDigits && a.enum_count) {
            slot = atomicAdd(a.enum_count, 1);
            if (slot < a.enum_max) {
              a.enum_list[slot] = h;
              a.enum_slice[slot] = 0;
              a.enum_done[slot] = 0;
              a.enum_bound[slot] = 0x7ff0000000000000ull;
            } else {
              slot = -2;
            }
          }
          if (slot < 0) {
            int flags = MOCAP_ST_CAND_OVERFLOW_;
            if (slot == -1 && !(prod <= 16777216.0)) {
              double l2 = 0.0;
              for (int j = 0; j < m; j++) l2 += log2((double)s_n[s_dcam[j]]);
              int lg = (int)ceil(l2 - 1e-9);
              lg = lg < 25 ? 25 : (lg > 511 ? 511 : lg);
              flags |= MOCAP_ST_INTRACTABLE_;
              // the largest root's log2(groups) in bits 20..28 (several roots of a frame may arrive from different workgroups)
              int old = __hip_atomic_load(&a.status[hd.frame], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              while (true) {
                const int of = (old >> MOCAP_ST_LOG2_GROUPS_SHIFT_) & 0x1FF;
                const int want = ((old | flags) & ~(0x1FF << MOCAP_ST_LOG2_GROUPS_SHIFT_)) | ((of > lg ? of : lg) << MOCAP_ST_LOG2_GROUPS_SHIFT_);
                const int seen = atomicCAS(&a.status[hd.frame], old, want);
                if (seen == old) break;
                old = seen;
              }
            } else {
              atomicOr(&a.status[hd.frame], flags);
            }
            a.n_out[hd.frame] = 0;
          }
        }
        continue;
      }
      const uint32_t G = (uint32_t)prod;
      __syncthreads();
      if (tid == 0) {
        s_best = 0x7ff0000000000000ull;
        s_gkey = ~0ull;  // (reused: the smallest candidate index among the groups that hold the best error)
      }
      __syncthreads();
      double be = inf, bX[3] = {0, 0, 0};
      uint32_t bg = 0;
      for (uint32_t g = (uint32_t)tid; g < G; g += kHvThreads) {
        // digits of g: the first multi-hit camera is the fastest one (helpers.py:394-400 order, as in frame_kernel.hip)
        uint32_t rem = g;
        for (int j = 0; j < m; j++) {
          const uint32_t n = s_n[s_dcam[j]];
          uint32_t qd, d;
          divmod_small(rem, n, qd, d);
          rem = qd;
          s_dg[j][tid] = (uint8_t)d;
        }
        auto obs = [&](int c, double& x, double& y) -> bool {
          const int n = s_n[c];
          if (!n) return false;
          const int d = n > 1 ? s_dg[s_lvl[c]][tid] : 0;
          const float2 w = fb[(size_t)c * M + hl[(size_t)c * Hs + d]];
          x = (double)w.x;
          y = (double)w.y;
          return true;
        };
        double X[3], e = inf;
        const double bound = __longlong_as_double((long long)s_best);
        triangulate_and_score<true, true, F32R, 1, false>(a.cv, obs, obs, X, e, bound, ec);
        if (e < be) {  // strict <: the first minimum of this lane's ascending run
          be = e;
          bg = g;
          bX[0] = X[0]; bX[1] = X[1]; bX[2] = X[2];
          atomicMin(&s_best, (unsigned long long)__double_as_longlong(e));
        }
      }
      __syncthreads();
      const double ebest = __longlong_as_double((long long)s_best);
      if (be == ebest && be < inf) atomicMin(&s_gkey, (unsigned long long)bg);
      __syncthreads();
      if (ebest < e00 && be == ebest && (unsigned long long)bg == s_gkey) {  // (one lane; group 0 stands on a tie: it is the smallest index)
        FrameArgs fa;
        fa.xyz = a.xyz;
        fa.world = a.world;
        store_point(fa, o, bX);
        a.err[o] = ebest;
        uint32_t rem = bg;
        for (int j = 0; j < m; j++) {
          uint32_t qd, d;
          divmod_small(rem, (uint32_t)s_n[s_dcam[j]], qd, d);
          rem = qd;
          s_dg[j][tid] = (uint8_t)d;
        }
        for (int c = 0; c < C; c++) {
          const int n = s_n[c];
          a.corr[o * C + c] = n ? (int16_t)hl[(size_t)c * Hs + (n > 1 ? s_dg[s_lvl[c]][tid] : 0)] : (int16_t)-1;
        }
      }
      continue;
    }
    // ---- leaves: complete groups, evaluated like every other group of the path
    const int nleaf = s_cnt[cur];
    if (tid == 0) {
      s_best = 0x7ff0000000000000ull;
      s_nhold = 0;
      s_win = -1;
    }
    __syncthreads();
    const uint8_t* leafD = nodeD[cur];
    for (int base = 0; base < nleaf; base += kHvThreads) {
      const int i = base + tid;
      if (i < nleaf) {
        const uint8_t* dg = leafD + (size_t)i * kHvDigits;
        auto obs = [&](int c, double& x, double& y) -> bool {
          const int n = s_n[c];
          if (!n) return false;
          const int d = n > 1 ? dg[s_lvl[c]] : 0;
          const float2 w = fb[(size_t)c * M + hl[(size_t)c * Hs + d]];
          x = (double)w.x;
          y = (double)w.y;
          return true;
        };
        double X[3], e = inf;
        triangulate_and_score<true, true, F32R, 1, false>(a.cv, obs, obs, X, e, e0, ec);
        if (e < inf) {
          atomicMin(&s_best, (unsigned long long)__double_as_longlong(e));
          // parked next to the leaf's matrix (dead now): the error and the point
          nodeB[cur][(size_t)i * 10 + 0] = e;
          nodeB[cur][(size_t)i * 10 + 1] = X[0];
          nodeB[cur][(size_t)i * 10 + 2] = X[1];
          nodeB[cur][(size_t)i * 10 + 3] = X[2];
        } else {
          nodeB[cur][(size_t)i * 10 + 0] = inf;
        }
      }
    }
    __threadfence_block();
    __syncthreads();
    const double eb = __longlong_as_double((long long)s_best);
    if (!(eb < e00)) continue;  // (uniform) group 0 stands: it is the smallest index, ties included
    for (int base = 0; base < nleaf; base += kHvThreads) {
      const int i = base + tid;
      if (i < nleaf && nodeB[cur][(size_t)i * 10] == eb) {
        const int k = atomicAdd(&s_nhold, 1);
        if (k < 64) s_hold[k] = i;
      }
    }
    __syncthreads();
    if (tid == 0) {
      // first minimum in candidate order: the mixed-radix index, slowest digit (the last multi-hit camera) first.  (More
      // than 64 leaves with the very same error bits: the first 64 found take part -- not reachable with measured blobs.)
      const int nh = s_nhold < 64 ? s_nhold : 64;
      int win = s_hold[0];
      for (int k = 1; k < nh; k++) {
        const uint8_t* x = leafD + (size_t)s_hold[k] * kHvDigits;
        const uint8_t* y = leafD + (size_t)win * kHvDigits;
        for (int j = m - 1; j >= 0; j--) {
          if (x[j] != y[j]) {
            if (x[j] < y[j]) win = s_hold[k];
            break;
          }
        }
      }
      s_win = win;
    }
    __syncthreads();
    const int win = s_win;
    if (tid == 0) {
      const double X[3] = {nodeB[cur][(size_t)win * 10 + 1], nodeB[cur][(size_t)win * 10 + 2], nodeB[cur][(size_t)win * 10 + 3]};
      FrameArgs fa;  // (store_point only looks at xyz and world)
      fa.xyz = a.xyz;
      fa.world = a.world;
      store_point(fa, o, X);
      a.err[o] = eb;
    }
    if (tid < C) {
      const int n = s_n[tid];
      int16_t s = -1;
      if (n) s = (int16_t)hl[(size_t)tid * Hs + (n > 1 ? leafD[(size_t)win * kHvDigits + s_lvl[tid]] : 0)];
      a.corr[o * C + tid] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// heavy_enum_kernel: the roots the search gave up on with at most 2^24 groups, ENUMERATED -- every group of the Cartesian
// product triangulated and scored by the path's own device function (helpers.py:394-421 as written), the whole GPU on one
// root at a time.  All workgroups walk the queued roots in the same order and pull slices of the root's group range from a
// counter; a lane keeps the first minimum of its ascending run, the smallest error seen anywhere (a global word) is everybody's
// cut-off (a cut-off only ever skips work: mocap_device.hpp); per workgroup the lexicographic minimum of (error bits, group
// index) goes to a partial record, and the last workgroup to report merges the records the same way and writes the root's
// slot -- the first minimum in candidate order (np.argmin, helpers.py:418), whatever ran where.
struct HvPart {
  unsigned long long ebits;
  uint32_t g, pad;
  double X[3];
};
static_assert(sizeof(HvPart) == kHeavyEnumPartBytes, "partial record layout");

template <bool F32R>
__global__ __launch_bounds__(kHvEnumThreads) void heavy_enum_kernel(HeavyArgs a) {
  __shared__ uint16_t s_n[kMaxCameras];
  __shared__ uint8_t s_lvl[kMaxCameras];
  __shared__ uint8_t s_dcam[kMaxCameras];
  __shared__ int s_m, s_slice, s_last;
  __shared__ uint32_t s_G;
  __shared__ unsigned long long s_best, s_gbest;
  __shared__ uint8_t s_dg[kHvGlobalEnumDigits][kHvEnumThreads];
  // the root's hits, tabulated once: DLT contribution (dlt_contribution: the function every path of the core rounds with) and
  // pixel coordinates of hit d of camera c at entry s_off[c] + d
  __shared__ double s_tab[kHvEnumTab][10];
  __shared__ float2 s_xy[kHvEnumTab];
  __shared__ uint16_t s_off[kMaxCameras + 1];
  __shared__ int s_split, s_csplit, s_views;
  __shared__ uint32_t s_Npre, s_Psuf;
  const int tid = threadIdx.x;
  const int C = a.cv.C, M = a.M;
  int n_roots = *a.enum_count;
  if (n_roots > a.enum_max) n_roots = a.enum_max;
  const double inf = __builtin_huge_val();
  for (int s = 0; s < n_roots; s++) {
    const int h = a.enum_list[s];
    const unsigned char* rec = a.recs + (size_t)h * a.stride;
    const HeavyRecHdr hd = *reinterpret_cast<const HeavyRecHdr*>(rec);
    const uint16_t* nc = reinterpret_cast<const uint16_t*>(rec + heavy_rec_counts_off());
    const uint8_t* hl = rec + heavy_rec_hits_off(C);
    const int Hs = hd.Hs;
    const float2* fb = (const float2*)(a.blobs + (size_t)hd.frame * C * M * 2);
    const size_t o = (size_t)hd.frame * a.K_big + hd.outslot;
    block_sync_lds();  // (the previous root's shared state is dead)
    if (tid < C) s_n[tid] = nc[tid];
    if (tid == 0) {
      int m = 0;
      unsigned long long G = 1;
      for (int c = 0; c < C; c++) {
        s_lvl[c] = 0xFF;
        if (nc[c] > 1) {
          s_lvl[c] = (uint8_t)m;
          s_dcam[m++] = (uint8_t)c;
          G *= nc[c];
        }
      }
      s_m = m;
      s_G = (uint32_t)G;  // (<= 2^24: heavy_bb_kernel queued it)
      s_best = 0x7ff0000000000000ull;
      s_gbest = ~0ull;
      int off = 0, views = 0;
      for (int c = 0; c < C; c++) {
        s_off[c] = (uint16_t)(off < 0xFFFF ? off : 0xFFFF);
        off += nc[c];
        views += nc[c] ? 1 : 0;
      }
      s_off[C] = (uint16_t)(off < 0xFFFF ? off : 0xFFFF);
      s_views = views;
      // The digits split into a PREFIX (the multi-hit cameras with the lowest camera numbers = the fastest digits of the
      // candidate index) and a SUFFIX (the last ones, >= 64 combinations where the root has them).  A lane takes one prefix
      // and walks the suffix combinations: B is summed in camera order from zeros (mocap_device.hpp triangulate_and_score), so
      // the partial sum over the cameras before the first suffix camera is the same for all of them -- computed once, the very
      // bits the full left-to-right sum passes through.
      int split = m;
      uint32_t psuf = 1;
      while (split > 0 && psuf < 64u) psuf *= nc[s_dcam[--split]];
      s_split = split;
      s_Psuf = psuf;
      s_Npre = (uint32_t)(G / psuf);
      s_csplit = split < m ? s_dcam[split] : C;
    }
    __syncthreads();
    const int m = s_m;
    const uint32_t G = s_G;
    const bool tabbed = s_off[C] <= kHvEnumTab;  // (uniform) else: every group from its raw observations, as before
    if (tabbed) {
      for (int c = 0; c < C; c++) {
        const int n = s_n[c];
        if (tid < n) {
          const float2 w = fb[(size_t)c * M + hl[(size_t)c * Hs + tid]];
          double Bc[10];
          dlt_contribution(Bc, a.cv.pq((size_t)12 * c), (double)w.x, (double)w.y);
#pragma unroll
          for (int e = 0; e < 10; e++) s_tab[s_off[c] + tid][e] = Bc[e];
          s_xy[s_off[c] + tid] = w;
        }
      }
      __syncthreads();
    }
    const int split = s_split, csplit = s_csplit, views = s_views;
    const uint32_t Npre = s_Npre, Psuf = s_Psuf;
    const uint32_t n_slices = tabbed ? (Npre + kHvEnumThreads - 1) / kHvEnumThreads : (G + kHvEnumSlice - 1) / kHvEnumSlice;
    EigCut ec;
    {
      const double om = (double)__int_as_float(hd.omax_bits);
      ec.p3max2 = a.p3max2;
      ec.o2slack = (1100.0 * 0x1p-46) * (om * om);
    }
    double be = inf, bX[3] = {0, 0, 0};
    uint32_t bg = 0;
    while (true) {
      if (tid == 0) s_slice = atomicAdd(&a.enum_slice[s], 1);
      __syncthreads();
      const uint32_t sl = (uint32_t)s_slice;
      __syncthreads();
      if (sl >= n_slices) break;  // (uniform)
      if (tabbed) {
        const uint32_t q = sl * (uint32_t)kHvEnumThreads + (uint32_t)tid;  // this lane's prefix
        if (q < Npre) {
          uint32_t rem = q;
          for (int j = 0; j < split; j++) {
            uint32_t qd, d;
            divmod_small(rem, (uint32_t)s_n[s_dcam[j]], qd, d);
            rem = qd;
            s_dg[j][tid] = (uint8_t)d;
          }
          double Bp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
          for (int c = 0; c < csplit; c++) {
            const int n = s_n[c];
            if (n) {
              const double* t = s_tab[s_off[c] + (n > 1 ? (int)s_dg[s_lvl[c]][tid] : 0)];
#pragma unroll
              for (int e = 0; e < 10; e++) Bp[e] = Bp[e] + t[e];
            }
          }
          auto obs = [&](int c, double& x, double& y) -> bool {
            const int n = s_n[c];
            if (!n) return false;
            const float2 w = s_xy[s_off[c] + (n > 1 ? (int)s_dg[s_lvl[c]][tid] : 0)];
            x = (double)w.x;
            y = (double)w.y;
            return true;
          };
          for (uint32_t sfx = 0; sfx < Psuf; sfx++) {  // ascending candidate index: g = q + Npre * sfx
            uint32_t r2 = sfx;
            for (int j = split; j < m; j++) {
              uint32_t qd, d;
              divmod_small(r2, (uint32_t)s_n[s_dcam[j]], qd, d);
              r2 = qd;
              s_dg[j][tid] = (uint8_t)d;
            }
            double B[10];
#pragma unroll
            for (int e = 0; e < 10; e++) B[e] = Bp[e];
            for (int c = csplit; c < C; c++) {
              const int n = s_n[c];
              if (n) {
                const double* t = s_tab[s_off[c] + (n > 1 ? (int)s_dg[s_lvl[c]][tid] : 0)];
#pragma unroll
                for (int e = 0; e < 10; e++) B[e] = B[e] + t[e];
              }
            }
            double X[3], e = inf;
            const double bound = __longlong_as_double((long long)__hip_atomic_load(&a.enum_bound[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            // (the call triangulate_and_score ends in, with the B it would have summed)
            solve_and_score<true, true, F32R, false, 1>(a.cv, B, views, obs, X, e, bound * (double)(2 * views) * (1.0 + 0x1p-40), ec);
            if (e < be) {  // strict <: the first minimum of this lane's ascending run
              be = e;
              bg = q + Npre * sfx;
              bX[0] = X[0]; bX[1] = X[1]; bX[2] = X[2];
              atomicMin(&a.enum_bound[s], (unsigned long long)__double_as_longlong(e));
            }
          }
        }
        continue;
      }
      const uint32_t g1 = (sl + 1) * (uint32_t)kHvEnumSlice < G ? (sl + 1) * (uint32_t)kHvEnumSlice : G;
      for (uint32_t g = sl * (uint32_t)kHvEnumSlice + (uint32_t)tid; g < g1; g += kHvEnumThreads) {
        // digits of g: the first multi-hit camera is the fastest one (helpers.py:394-400 order, as in frame_kernel.hip)
        uint32_t rem = g;
        for (int j = 0; j < m; j++) {
          uint32_t qd, d;
          divmod_small(rem, (uint32_t)s_n[s_dcam[j]], qd, d);
          rem = qd;
          s_dg[j][tid] = (uint8_t)d;
        }
        auto obs = [&](int c, double& x, double& y) -> bool {
          const int n = s_n[c];
          if (!n) return false;
          const int d = n > 1 ? s_dg[s_lvl[c]][tid] : 0;
          const float2 w = fb[(size_t)c * M + hl[(size_t)c * Hs + d]];
          x = (double)w.x;
          y = (double)w.y;
          return true;
        };
        double X[3], e = inf;
        const double bound = __longlong_as_double((long long)__hip_atomic_load(&a.enum_bound[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        triangulate_and_score<true, true, F32R, 1, false>(a.cv, obs, obs, X, e, bound, ec);
        if (e < be) {  // strict <: the first minimum of this lane's ascending run
          be = e;
          bg = g;
          bX[0] = X[0]; bX[1] = X[1]; bX[2] = X[2];
          atomicMin(&a.enum_bound[s], (unsigned long long)__double_as_longlong(e));
        }
      }
    }
    // the workgroup's winner: smallest error bits, then smallest group index among its holders
    if (be < inf) atomicMin(&s_best, (unsigned long long)__double_as_longlong(be));
    __syncthreads();
    const unsigned long long wb = s_best;
    if (be < inf && (unsigned long long)__double_as_longlong(be) == wb) atomicMin(&s_gbest, (unsigned long long)bg);
    __syncthreads();
    HvPart* part = reinterpret_cast<HvPart*>(a.enum_part) + (size_t)s * a.enum_grid;
    if (wb == 0x7ff0000000000000ull) {
      if (tid == 0) {
        q_st(&part[blockIdx.x].ebits, wb);  // (agent-scope stores: the merging workgroup may sit on another XCD)
        q_st(&part[blockIdx.x].g, 0xFFFFFFFFu);
      }
    } else if (be < inf && (unsigned long long)__double_as_longlong(be) == wb && (unsigned long long)bg == s_gbest) {  // (one lane)
      q_st(&part[blockIdx.x].ebits, wb);
      q_st(&part[blockIdx.x].g, bg);
      q_st(&part[blockIdx.x].X[0], bX[0]);
      q_st(&part[blockIdx.x].X[1], bX[1]);
      q_st(&part[blockIdx.x].X[2], bX[2]);
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(&a.enum_done[s], 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (!s_last) continue;  // (uniform)
    // ---- the last workgroup to report merges (its loads see the others' records: each was fenced before its count)
    __threadfence();
    if (tid == 0) {
      s_best = 0x7ff0000000000000ull;
      s_gbest = ~0ull;
    }
    __syncthreads();
    for (int w = tid; w < (int)gridDim.x; w += kHvEnumThreads) {
      const unsigned long long eb = __hip_atomic_load(&part[w].ebits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (eb != 0x7ff0000000000000ull) atomicMin(&s_best, eb);
    }
    __syncthreads();
    const unsigned long long fb_ = s_best;
    if (fb_ == 0x7ff0000000000000ull) continue;  // no group with a finite error: the frame kernel's group 0 stands
    for (int w = tid; w < (int)gridDim.x; w += kHvEnumThreads)
      if (__hip_atomic_load(&part[w].ebits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == fb_)
        atomicMin(&s_gbest, (unsigned long long)__hip_atomic_load(&part[w].g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    __syncthreads();
    const uint32_t gw = (uint32_t)s_gbest;
    for (int w = tid; w < (int)gridDim.x; w += kHvEnumThreads) {
      if (__hip_atomic_load(&part[w].ebits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == fb_ &&
          __hip_atomic_load(&part[w].g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gw) {  // (one record: a group lives in one slice)
        const double X[3] = {__hip_atomic_load(&part[w].X[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                             __hip_atomic_load(&part[w].X[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                             __hip_atomic_load(&part[w].X[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)};
        FrameArgs fa;  // (store_point only looks at xyz and world)
        fa.xyz = a.xyz;
        fa.world = a.world;
        store_point(fa, o, X);
        a.err[o] = __longlong_as_double((long long)fb_);
        uint32_t rem = gw;
        uint8_t dg[kHvGlobalEnumDigits];
        for (int j = 0; j < m; j++) {
          uint32_t qd, d;
          divmod_small(rem, (uint32_t)s_n[s_dcam[j]], qd, d);
          rem = qd;
          dg[j] = (uint8_t)d;
        }
        for (int c = 0; c < C; c++) {
          const int n = s_n[c];
          int16_t sidx = -1;
          if (n) {
            int d = 0;
            if (n > 1)
              for (int j = 0; j < m; j++)
                if (s_dcam[j] == c) d = dg[j];
            sidx = (int16_t)hl[(size_t)c * Hs + d];
          }
          a.corr[o * C + c] = sidx;
        }
      }
    }
  }
}

size_t heavy_enum_ws_bytes(int enum_max, int grid) {
  return (size_t)enum_max * (4 + 4 + 4 + 8) + 64 + (size_t)enum_max * grid * kHeavyEnumPartBytes;
}

hipError_t launch_heavy_enum(const HeavyArgs& a, hipStream_t stream) {
  if (a.cv.f32_rounding)
    hipLaunchKernelGGL(heavy_enum_kernel<true>, dim3(a.enum_grid), dim3(kHvEnumThreads), 0, stream, a);
  else
    hipLaunchKernelGGL(heavy_enum_kernel<false>, dim3(a.enum_grid), dim3(kHvEnumThreads), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_heavy_bb(const HeavyArgs& a, int grid, hipStream_t stream) {
  if (a.cv.f32_rounding)
    hipLaunchKernelGGL(heavy_bb_kernel<true>, dim3(grid), dim3(kHvThreads), 0, stream, a);
  else
    hipLaunchKernelGGL(heavy_bb_kernel<false>, dim3(grid), dim3(kHvThreads), 0, stream, a);
  return hipGetLastError();
}

}  // namespace mocap
