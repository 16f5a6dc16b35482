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
//   C  per-root candidate counts (Cartesian product sizes, helpers.py:394-400), offsets
//   D  a range [g_lo, g_hi) of the flat candidate space is split into T contiguous runs; each lane
//      triangulates and scores its run (csrc/mocap_device.hpp) keeping an (error, index)
//      first-minimum per (lane, root) segment -- at most T + R segments, stored in LDS, no atomics
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

struct FrameLds {
  // byte offsets into dynamic LDS, computed identically on host (size) and device (carve)
  size_t line, dist, seg_e, seg_x, seg_g, goff, gcnt, outslot, bx, by, hits, sel, nh, root_blob,
      root_cam, claimed, cnt, misc, total;
  __host__ __device__ static size_t align(size_t x, size_t a) { return (x + a - 1) / a * a; }
  __host__ __device__ FrameLds(int C, int M, int R, int T) {
    size_t o = 0;
    line = o;      o += sizeof(double) * 4 * R;
    seg_e = o;     o += sizeof(double) * (T + R);
    seg_x = o;     o += sizeof(double) * 3 * (T + R);
    // dist (phase B scratch) and the segment arrays (phase D/E) are never live together
    dist = line + sizeof(double) * 4 * R;
    const size_t dist_end = dist + sizeof(double) * (size_t)R * M;
    if (dist_end > o) o = dist_end;
    seg_g = o;     o += sizeof(uint32_t) * (T + R);
    goff = o;      o += sizeof(uint32_t) * (R + 1);
    gcnt = o;      o += sizeof(uint32_t) * R;
    outslot = o;   o += sizeof(int32_t) * R;
    bx = o;        o += sizeof(float) * (size_t)C * M;
    by = o;        o += sizeof(float) * (size_t)C * M;
    cnt = o;       o += sizeof(int32_t) * C;
    misc = o;      o += sizeof(int32_t) * 8;
    hits = o;      o += sizeof(uint16_t) * (size_t)R * C * M;
    sel = o;       o += sizeof(uint16_t) * (size_t)T * C;
    nh = o;        o += sizeof(uint16_t) * (size_t)R * C;
    root_blob = o; o += sizeof(uint16_t) * R;
    root_cam = o;  o += R;
    claimed = o;   o += M;
    total = align(o, 16);
  }
};

size_t frame_lds_bytes(int C, int M, int R, int T) { return FrameLds(C, M, R, T).total; }

// misc[] slots
enum { MI_NROOTS = 0, MI_STATUS = 1, MI_NOUT = 2, MI_G = 3, MI_ITEM = 4, MI_DEFER = 5 };

template <int T, bool UNIFORM_K>
struct FrameState {
  const FrameArgs& p;
  const CamView& cv;
  const int C, M, R, tid;
  double *line, *dist, *seg_e, *seg_x;
  uint32_t *seg_g, *goff, *gcnt;
  int32_t *outslot, *cnt, *misc;
  float *bx, *by;
  uint16_t *hits, *sel, *nh, *root_blob;
  uint8_t *root_cam, *claimed;

  __device__ FrameState(const FrameArgs& p_, unsigned char* smem)
      : p(p_), cv(p_.cv), C(p_.cv.C), M(p_.M), R(p_.K_max), tid(threadIdx.x) {
    const FrameLds L(C, M, R, T);
    line = (double*)(smem + L.line);
    dist = (double*)(smem + L.dist);
    seg_e = (double*)(smem + L.seg_e);
    seg_x = (double*)(smem + L.seg_x);
    seg_g = (uint32_t*)(smem + L.seg_g);
    goff = (uint32_t*)(smem + L.goff);
    gcnt = (uint32_t*)(smem + L.gcnt);
    outslot = (int32_t*)(smem + L.outslot);
    bx = (float*)(smem + L.bx);
    by = (float*)(smem + L.by);
    cnt = (int32_t*)(smem + L.cnt);
    misc = (int32_t*)(smem + L.misc);
    hits = (uint16_t*)(smem + L.hits);
    sel = (uint16_t*)(smem + L.sel) + (size_t)tid * C;
    nh = (uint16_t*)(smem + L.nh);
    root_blob = (uint16_t*)(smem + L.root_blob);
    root_cam = (uint8_t*)(smem + L.root_cam);
    claimed = (uint8_t*)(smem + L.claimed);
  }

  // ---------------------------------------------------------------- phases A-C
  // Leaves roots / hit lists / candidate offsets in LDS; returns with all lanes synchronised.
  __device__ void match(int64_t frame) {
    const bool f32r = cv.f32_rounding != 0;
    {
      const float2* src = (const float2*)(p.blobs + (size_t)frame * C * M * 2);
      for (int i = tid; i < C * M; i += T) {
        const float2 b = src[i];
        bx[i] = b.x;
        by[i] = b.y;
      }
      if (tid < C) {
        int n = p.counts[(size_t)frame * C + tid];
        cnt[tid] = n < 0 ? 0 : (n > M ? M : n);
      }
      if (tid == 0) misc[MI_STATUS] = 0;
    }
    __syncthreads();
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

    for (int i = 1; i < C; i++) {
      const int nroots = misc[MI_NROOTS];
      const int Mi = cnt[i];
      const float* pxs = bx + (size_t)i * M;
      const float* pys = by + (size_t)i * M;
      // B1: epipolar line of every root in camera i.  cv.computeCorrespondEpilines on a float32
      // point: double math, scale by 1/sqrt(a^2+b^2), float32 result (helpers.py:363-364).
      for (int r = tid; r < nroots; r += T) {
        const int rc = root_cam[r], rb = root_blob[r];
        ctab_t Fm = as_ctab(cv.F + 9 * ((size_t)rc * C + i));
        const double x = (double)bx[(size_t)rc * M + rb], y = (double)by[(size_t)rc * M + rb];
        double a = Fm[0] * x + Fm[1] * y + Fm[2];
        double b = Fm[3] * x + Fm[4] * y + Fm[5];
        double c = Fm[6] * x + Fm[7] * y + Fm[8];
        double nu = a * a + b * b;
        nu = nu != 0.0 ? 1.0 / sqrt(nu) : 1.0;
        a *= nu;
        b *= nu;
        c *= nu;
        if (f32r) {
          a = (double)(float)a;
          b = (double)(float)b;
          c = (double)(float)c;
        }
        line[4 * r + 0] = a;
        line[4 * r + 1] = b;
        line[4 * r + 2] = c;
        line[4 * r + 3] = sqrt(a * a + b * b);  // helpers.py:373 divides by it again
        nh[(size_t)r * C + i] = 0;
      }
      for (int k = tid; k < M; k += T) claimed[k] = 0;
      __syncthreads();
      // B2: |a x + b y + c| / sqrt(a^2 + b^2) for every (root, blob) pair (helpers.py:373)
      const int npairs = nroots * Mi;
      for (int idx = tid; idx < npairs; idx += T) {
        const int r = idx / Mi, k = idx - r * Mi;
        const double a = line[4 * r + 0], b = line[4 * r + 1], c = line[4 * r + 2], den = line[4 * r + 3];
        const double px = (double)pxs[k], py = (double)pys[k];
        dist[(size_t)r * M + k] = fabs(a * px + b * py + c) / den;
      }
      __syncthreads();
      // B3: gate (strict <, helpers.py:375,383) and order by (distance, index) via rank counting
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
          hits[((size_t)r * C + i) * M + rank] = (uint16_t)k;
        }
        if (k == 0) {
          int n = 0;
          for (int k2 = 0; k2 < Mi; k2++) n += dr[k2] < p.gate_px ? 1 : 0;
          nh[(size_t)r * C + i] = (uint16_t)n;
        }
      }
      __syncthreads();
      // B4: the closest hit of every matched root is removed *by value* from the unmatched set
      // (helpers.py:391): flag every blob with the same coordinates.
      for (int idx = tid; idx < npairs; idx += T) {
        const int r = idx / Mi, k = idx - r * Mi;
        if (nh[(size_t)r * C + i] > 0) {
          const int k0 = hits[((size_t)r * C + i) * M];
          if (pxs[k] == pxs[k0] && pys[k] == pys[k0]) claimed[k] = 1;
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
      }
      __syncthreads();
    }

    // C: candidate counts per root
    const int nroots = misc[MI_NROOTS];
    for (int r = tid; r < nroots; r += T) {
      const int rc = root_cam[r];
      unsigned long long total = 1;
      int views = 1;
      bool over = false;
      for (int c = rc + 1; c < C; c++) {
        const unsigned n = nh[(size_t)r * C + c];
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
      gcnt[r] = (views > 1 && !over) ? (uint32_t)total : 0u;  // helpers.py:413-414 drops 1-view roots
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t acc = 0;
      int slot = 0;
      for (int r = 0; r < nroots; r++) {
        goff[r] = acc;
        outslot[r] = gcnt[r] ? slot : -1;
        slot += gcnt[r] ? 1 : 0;
        const uint32_t nxt = acc + gcnt[r];
        if (nxt < acc) misc[MI_STATUS] |= MOCAP_ST_CAND_OVERFLOW_;
        acc = nxt;
      }
      goff[nroots] = acc;
      misc[MI_NOUT] = slot;
      misc[MI_G] = misc[MI_STATUS] ? 0 : (int32_t)acc;
    }
    __syncthreads();
  }

  // ---------------------------------------------------------------- phase D
  // Evaluate candidates [g_lo, g_hi); per (lane, root) segment winners land in seg_* .
  __device__ void evaluate(uint32_t g_lo, uint32_t g_hi) {
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
      double best_e = 0.0, best_X[3] = {0, 0, 0};
      uint32_t best_g = 0;
      bool have = false;
      for (; g < g_end; g++) {
        if (g >= goff[r + 1]) {  // leaving root r: flush the (lane, root) segment
          const int s = tid + outslot[r];
          seg_e[s] = best_e;
          seg_g[s] = best_g;
          seg_x[3 * s + 0] = best_X[0];
          seg_x[3 * s + 1] = best_X[1];
          seg_x[3 * s + 2] = best_X[2];
          have = false;
          do { r++; } while (goff[r + 1] <= g);
        }
        const uint32_t gl = g - goff[r];
        const int rc = root_cam[r];
        const uint16_t rb = root_blob[r];
        const uint16_t* nhr = nh + (size_t)r * C;
        const uint16_t* hr = hits + (size_t)r * C * M;
        uint32_t rem = gl;
        // pass 1 decodes the mixed-radix group index (camera rc+1 = fastest digit,
        // helpers.py:394-400) and parks the blob index per camera for pass 2
        auto obs1 = [&](int c, double& x, double& y) -> bool {
          uint16_t s = kNone;
          if (c == rc) {
            s = rb;
          } else if (c > rc) {
            const uint32_t n = nhr[c];
            if (n) {
              uint32_t qd, dgt;
              divmod_small(rem, n, qd, dgt);
              rem = qd;
              s = hr[(size_t)c * M + dgt];
            }
          }
          sel[c] = s;
          if (s == kNone) return false;
          x = (double)bx[(size_t)c * M + s];
          y = (double)by[(size_t)c * M + s];
          return true;
        };
        auto obs2 = [&](int c, double& x, double& y) -> bool {
          const uint16_t s = sel[c];
          if (s == kNone) return false;
          x = (double)bx[(size_t)c * M + s];
          y = (double)by[(size_t)c * M + s];
          return true;
        };
        double X[3], e;
        triangulate_and_score<UNIFORM_K, true>(cv, obs1, obs2, X, e);
        if (!have || e < best_e) {  // strict <: first minimum within the lane's ascending run
          have = true;
          best_e = e;
          best_g = gl;
          best_X[0] = X[0];
          best_X[1] = X[1];
          best_X[2] = X[2];
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
  __device__ bool root_winner(int r, uint32_t g_lo, uint32_t g_hi, double& eb, uint32_t& gb, double (&Xb)[3]) {
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
  __device__ void write_point(int64_t frame, int r, double e, uint32_t gl, const double (&X)[3]) {
    const size_t o = (size_t)frame * R + outslot[r];
    p.xyz[o * 3 + 0] = X[0];
    p.xyz[o * 3 + 1] = X[1];
    p.xyz[o * 3 + 2] = X[2];
    p.err[o] = e;
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
          uint32_t qd, dgt;
          divmod_small(rem, n, qd, dgt);
          s = (int16_t)hits[((size_t)r * C + c) * M + dgt];
          rem = qd;
        }
      }
      co[c] = s;
    }
  }

  __device__ void write_frame_header(int64_t frame) {
    if (tid == 0) {
      const int status = misc[MI_STATUS];
      p.n_out[frame] = status ? 0 : misc[MI_NOUT];
      p.status[frame] = status;
      if (p.n_cand) p.n_cand[frame] = misc[MI_G];
    }
  }
};

template <int T, bool UNIFORM_K, int MODE>
__global__ __launch_bounds__(T) void frame_kernel(FrameArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  FrameState<T, UNIFORM_K> st(p, smem);
  const int tid = threadIdx.x;
  const FrameQueues& q = p.q;
  const int R = p.K_max;

  while (true) {
    // ------------------------------------------------------------ pull a work item
    if (tid == 0) {
      int item;
      if (MODE == MODE_MAIN) {
        item = atomicAdd(&q.counters[QC_NEXT_FRAME], 1);
        if (item >= p.n_frames) item = -1;
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
      st.match(frame);
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
      st.match(frame);
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
      st.match(frame);
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

template <int T, int MODE>
static hipError_t launch_TM(const FrameArgs& a, int grid, size_t lds, hipStream_t stream) {
  auto k = a.cv.uniformK ? frame_kernel<T, true, MODE> : frame_kernel<T, false, MODE>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(T), lds, stream, a);
  return hipGetLastError();
}

template <int T>
static hipError_t launch_T(const FrameArgs& a, int mode, int grid, size_t lds, hipStream_t stream) {
  switch (mode) {
    case MODE_MAIN: return launch_TM<T, MODE_MAIN>(a, grid, lds, stream);
    case MODE_SLICE: return launch_TM<T, MODE_SLICE>(a, grid, lds, stream);
    default: return launch_TM<T, MODE_MERGE>(a, grid, lds, stream);
  }
}

hipError_t launch_frame_kernel(const FrameArgs& a, int mode, int threads, int grid, hipStream_t stream) {
  const size_t lds = frame_lds_bytes(a.cv.C, a.M, a.K_max, threads);
  switch (threads) {
    case 64: return launch_T<64>(a, mode, grid, lds, stream);
    case 128: return launch_T<128>(a, mode, grid, lds, stream);
    case 256: return launch_T<256>(a, mode, grid, lds, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace mocap
