// post_kernels.hip -- the step right after the hot path in the reference's frame loop:
//   locate_objects (reference computer_code/api/helpers.py:424-480): find the 3-LED drone patterns
//   (two points 0.095 m from a lead point and 0.15 m from each other, +-0.025) among a frame's
//   triangulated points, heading from the pair, drone index from the side the lead point is on.
// The world-coordinate epilogue (helpers.py:96-103) is fused into the frame kernel's store
// (frame_kernel.hip, write_point).
//
// One lane per frame: the reference's scan is sequential (`already_matched_points`, first valid pair
// wins) and K is a few dozen points, so the parallelism is across the frame batch.  Distances are
// recomputed where the reference reads its K x K matrix (same expression, same bits).
#include "kernels.hpp"

namespace mocap {

__device__ __forceinline__ double dist3(const double* a, const double* b) {
  const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return sqrt((dx * dx + dy * dy) + dz * dz);  // np.sqrt(np.sum(d**2)), helpers.py:434
}

__global__ __launch_bounds__(64) void locate_objects_kernel(LocateArgs a) {
  const double dist1 = 0.095, dist2 = 0.15, tol = 0.025;  // helpers.py:425-426,441,448
  const double pi = 3.141592653589793, half_pi = 1.5707963267948966;
  for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < a.n_frames;
       f += (int64_t)gridDim.x * blockDim.x) {
    const double* P = a.xyz + (size_t)f * a.K_max * 3;
    const double* E = a.err + (size_t)f * a.K_max;
    int K = a.n_pts[f];
    K = (K < 0 || K > a.K_max) ? 0 : K;  // > K_max: a re-submitted frame that needs more slots than the caller gave (nothing was written)
    unsigned long long matched[4] = {0, 0, 0, 0};  // already_matched_points, K <= 256
    int no = 0;
    for (int i = 0; i < K; i++) {
      if (matched[i >> 6] >> (i & 63) & 1ull) continue;  // helpers.py:437-438
      int nm = 0;
      for (int j = 0; j < K; j++) nm += fabs(dist3(P + 3 * i, P + 3 * j) - dist1) < tol ? 1 : 0;
      if (nm < 2) continue;  // helpers.py:443
      bool done = false;
      // cartesian_product(matches, matches), first index slow (helpers.py:444,532-533)
      for (int p1 = 0; p1 < K && !done; p1++) {
        if (!(fabs(dist3(P + 3 * i, P + 3 * p1) - dist1) < tol)) continue;
        for (int p2 = 0; p2 < K; p2++) {
          if (!(fabs(dist3(P + 3 * i, P + 3 * p2) - dist1) < tol)) continue;
          const double pd = dist3(P + 3 * p1, P + 3 * p2);
          if (fabs(pd - dist2) > tol) continue;  // helpers.py:448-449
          matched[i >> 6] |= 1ull << (i & 63);
          matched[p1 >> 6] |= 1ull << (p1 & 63);
          matched[p2 >> 6] |= 1ull << (p2 & 63);
          const double* A = P + 3 * p1;
          const double* B = P + 3 * p2;
          const double loc[3] = {(A[0] + B[0]) / 2, (A[1] + B[1]) / 2, (A[2] + B[2]) / 2};
          const double error = ((E[i] + E[p1]) + E[p2]) / 3.0;  // np.mean of three (helpers.py:459)
          double hx = A[0] - B[0], hy = A[1] - B[1], hz = A[2] - B[2];
          const double nrm = sqrt((hx * hx + hy * hy) + hz * hz);  // linalg.norm (helpers.py:462)
          hx /= nrm;
          hy /= nrm;
          double heading = atan2(hy, hx);
          heading = heading > half_pi ? heading - pi : heading;   // helpers.py:465
          heading = heading < -half_pi ? heading + pi : heading;  // helpers.py:466
          const int drone = (P[3 * i + 1] - loc[1]) > 0 ? 0 : 1;   // helpers.py:469
          if (no < a.O_max) {
            const size_t o = (size_t)f * a.O_max + no;
            a.obj_pos[3 * o + 0] = loc[0];
            a.obj_pos[3 * o + 1] = loc[1];
            a.obj_pos[3 * o + 2] = loc[2];
            a.obj_heading[o] = -heading;
            a.obj_err[o] = error;
            a.obj_drone[o] = drone;
            if (a.obj_lead) a.obj_lead[o] = i;
          }
          no++;
          done = true;  // `break` leaves the pair loop (helpers.py:478)
          break;
        }
      }
    }
    a.n_obj[f] = no;  // may exceed O_max: the caller sees that objects were dropped
  }
}

hipError_t launch_locate_objects(const LocateArgs& a, hipStream_t stream) {
  if (a.n_frames <= 0) return hipSuccess;
  int64_t blocks = (a.n_frames + 63) / 64;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(locate_objects_kernel, dim3((unsigned)blocks), dim3(64), 0, stream, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------- live path: one wave per frame + export
// The same scan for the live loop (one or a few frames per call, helpers.py:94-133), where one lane walking K^2 distances
// is all latency: one WAVE per frame.  The points sit in LDS; for a lead point i the lanes test all j at once
// (`distance_deltas < 0.025`, helpers.py:441 -> one ballot per 64 points), the pair search walks p1 over the set bits in
// ascending order and tests every p2 of the set in one ballot (first set bit = the first pair in cartesian_product's
// order, helpers.py:444-449).  Same expressions, same bits as locate_objects_kernel (tested against it and the golden).
// The kernel also EXPORTS the frame path's outputs: the match kernel wrote them to device memory, this wave copies the
// valid slots (and the objects) to the caller-visible buffers -- pinned host memory in mocap_track_frame, so that one
// event wait delivers the whole `object-points` payload.
__global__ __launch_bounds__(64) void track_export_kernel(LocateArgs a, TrackExportArgs e) {
  const double dist1 = 0.095, dist2 = 0.15, tol = 0.025;
  const double pi = 3.141592653589793, half_pi = 1.5707963267948966;
  __shared__ double P[256 * 3];
  __shared__ double E[256];
  const int lane = threadIdx.x;
  for (int64_t f = blockIdx.x; f < a.n_frames; f += gridDim.x) {
    int K = a.n_pts[f];
    K = (K < 0 || K > a.K_max) ? 0 : K;  // > K_max: a re-submitted frame that needs more slots than the caller gave (nothing was written)
    const double* gP = a.xyz + (size_t)f * a.K_max * 3;
    const double* gE = a.err + (size_t)f * a.K_max;
    const bool search = a.n_obj != nullptr && K <= 256;  // (the LDS copy holds 256 points: the hosts refuse more with the search on)
    __syncthreads();
    if (search) {
      for (int j = lane; j < 3 * K; j += 64) P[j] = gP[j];
      for (int j = lane; j < K; j += 64) E[j] = gE[j];
    }
    __syncthreads();
    // ---- export of the frame path's own outputs (valid slots only; the caller's fill stays beyond): straight from device
    // memory, any K_max (without the object search a frame may hold up to 1 024 points)
    if (e.out_xyz) {
      double* oP = e.out_xyz + (size_t)f * a.K_max * 3;
      for (int j = lane; j < 3 * K; j += 64) oP[j] = gP[j];
      double* oE = e.out_err + (size_t)f * a.K_max;
      for (int j = lane; j < K; j += 64) oE[j] = gE[j];
      if (e.out_corr) {
        const int16_t* gc = e.corr + (size_t)f * a.K_max * e.C;
        int16_t* oc = e.out_corr + (size_t)f * a.K_max * e.C;
        for (int j = lane; j < K * e.C; j += 64) oc[j] = gc[j];
      }
      if (lane == 0) {
        e.out_n_pts[f] = a.n_pts[f];
        e.out_status[f] = e.status[f];
        if (e.out_n_cand) e.out_n_cand[f] = e.n_cand ? e.n_cand[f] : 0;
      }
    }
    if (e.blobs) {
      const int nc = e.C;
      const int32_t* gc = e.counts + (size_t)f * nc;
      for (int c = lane; c < nc; c += 64) {
        e.out_counts[(size_t)f * nc + c] = gc[c];
        e.out_blob_status[(size_t)f * nc + c] = e.blob_status[(size_t)f * nc + c];
      }
      const size_t w = (size_t)nc * e.M * 2;
      const float* gb = e.blobs + (size_t)f * w;
      float* ob = e.out_blobs + (size_t)f * w;
      for (int j = lane; j < (int)w; j += 64) {
        const int c = j / (2 * e.M), k = (j - c * 2 * e.M) >> 1;
        if (k < gc[c]) ob[j] = gb[j];
      }
    }
    if (!a.n_obj) continue;  // is_locating_objects off (helpers.py:107)
    int no = 0;
    if (search) {
      const int chunks = (K + 63) >> 6;
      unsigned long long matched[4] = {0, 0, 0, 0};
      for (int i = 0; i < K; i++) {
        if (matched[i >> 6] >> (i & 63) & 1ull) continue;
        unsigned long long m[4] = {0, 0, 0, 0};
        int nm = 0;
        for (int ch = 0; ch < chunks; ch++) {
          const int j = ch * 64 + lane;
          const bool ok = j < K && fabs(dist3(P + 3 * i, P + 3 * j) - dist1) < tol;
          m[ch] = __ballot(ok);
          nm += __popcll(m[ch]);
        }
        if (nm < 2) continue;
        int p1f = -1, p2f = -1;
        for (int c1 = 0; c1 < chunks && p1f < 0; c1++) {
          unsigned long long rest = m[c1];
          while (rest && p1f < 0) {
            const int p1 = c1 * 64 + __builtin_ctzll(rest);
            rest &= rest - 1;
            for (int ch = 0; ch < chunks; ch++) {
              const int j = ch * 64 + lane;
              bool acc = false;
              if (j < K && (m[ch] >> lane & 1ull)) acc = !(fabs(dist3(P + 3 * p1, P + 3 * j) - dist2) > tol);
              const unsigned long long b = __ballot(acc);
              if (b) {
                p1f = p1;
                p2f = ch * 64 + __builtin_ctzll(b);
                break;
              }
            }
          }
        }
        if (p1f < 0) continue;
        matched[i >> 6] |= 1ull << (i & 63);
        matched[p1f >> 6] |= 1ull << (p1f & 63);
        matched[p2f >> 6] |= 1ull << (p2f & 63);
        if (lane == 0 && no < a.O_max) {
          const double* A = P + 3 * p1f;
          const double* B = P + 3 * p2f;
          const double loc[3] = {(A[0] + B[0]) / 2, (A[1] + B[1]) / 2, (A[2] + B[2]) / 2};
          const double error = ((E[i] + E[p1f]) + E[p2f]) / 3.0;
          double hx = A[0] - B[0], hy = A[1] - B[1], hz = A[2] - B[2];
          const double nrm = sqrt((hx * hx + hy * hy) + hz * hz);
          hx /= nrm;
          hy /= nrm;
          double heading = atan2(hy, hx);
          heading = heading > half_pi ? heading - pi : heading;
          heading = heading < -half_pi ? heading + pi : heading;
          const int drone = (P[3 * i + 1] - loc[1]) > 0 ? 0 : 1;
          const size_t o = (size_t)f * a.O_max + no;
          a.obj_pos[3 * o + 0] = loc[0];
          a.obj_pos[3 * o + 1] = loc[1];
          a.obj_pos[3 * o + 2] = loc[2];
          a.obj_heading[o] = -heading;
          a.obj_err[o] = error;
          a.obj_drone[o] = drone;
          if (a.obj_lead) a.obj_lead[o] = i;
        }
        no++;
      }
    }
    if (lane == 0) a.n_obj[f] = no;
  }
}

hipError_t launch_track_export(const LocateArgs& a, const TrackExportArgs& e, hipStream_t stream) {
  if (a.n_frames <= 0) return hipSuccess;
  const int64_t blocks = a.n_frames > 4096 ? 4096 : a.n_frames;
  hipLaunchKernelGGL(track_export_kernel, dim3((unsigned)blocks), dim3(64), 0, stream, a, e);
  return hipGetLastError();
}

// ---------------------------------------------------------------- track compaction for the exchange step
// The frame kernel writes fixed-capacity outputs ([F][K_max] slots, n_out of them valid): fine for consumers on
// the same GPU, twice the bytes the valid points need when a frame shard's results travel to the gathering rank
// (RCCL over xGMI).  compact = exclusive prefix sum over n_out (per-block scan, scan of the block
// totals, add) + one lane per (frame, slot) copying the valid slots into fixed-stride
// records  { xyz 3 x f64 | err f64 | corr C x i16 | pad to 8 }  in frame order.
constexpr int kScanBlock = 1024;

__global__ __launch_bounds__(kScanBlock) void compact_scan_blocks_kernel(CompactArgs a) {
  __shared__ int64_t sh[kScanBlock / 64];
  const int64_t f = (int64_t)blockIdx.x * kScanBlock + threadIdx.x;
  int64_t v = 0;
  if (f < a.n_frames) {
    const int n = a.n_out[f];
    v = (n < 0 || n > a.K_max) ? 0 : n;  // > K_max: a re-submitted frame that needs more slots (nothing was written)
  }
  // inclusive scan inside the wave, then across the block's 16 waves
  int64_t x = v;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int o = 1; o < 64; o <<= 1) {
    const int64_t y = __shfl_up(x, o);
    if (lane >= o) x += y;
  }
  if (lane == 63) sh[wave] = x;
  __syncthreads();
  int64_t base = 0;
  for (int w = 0; w < wave; w++) base += sh[w];
  if (f < a.n_frames) a.offsets[f] = base + x - v;  // exclusive, relative to the block
  if (threadIdx.x == kScanBlock - 1) a.block_sums[blockIdx.x] = base + x;
}

__global__ __launch_bounds__(kScanBlock) void compact_scan_totals_kernel(CompactArgs a, int n_blocks) {
  // one workgroup: exclusive scan of the block totals in place (n_blocks is a few hundred at most per pass)
  __shared__ int64_t sh[kScanBlock / 64];
  __shared__ int64_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int b0 = 0; b0 < n_blocks; b0 += kScanBlock) {
    const int b = b0 + threadIdx.x;
    const int64_t v = b < n_blocks ? a.block_sums[b] : 0;
    int64_t x = v;
    for (int o = 1; o < 64; o <<= 1) {
      const int64_t y = __shfl_up(x, o);
      if (lane >= o) x += y;
    }
    if (lane == 63) sh[wave] = x;
    __syncthreads();
    int64_t base = carry;
    for (int w = 0; w < wave; w++) base += sh[w];
    if (b < n_blocks) a.block_sums[b] = base + x - v;
    __syncthreads();
    if (threadIdx.x == kScanBlock - 1) carry = base + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    a.offsets[a.n_frames] = carry;  // total number of records
    if (a.total) *a.total = carry;
  }
}

__global__ __launch_bounds__(256) void compact_add_block_offsets_kernel(CompactArgs a) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f < a.n_frames) a.offsets[f] += a.block_sums[f / kScanBlock];
}

__global__ __launch_bounds__(256) void compact_scatter_kernel(CompactArgs a) {
  // one WAVE per frame, its lanes walking the frame's n_out * (stride / 8) record words in order: consecutive lanes write
  // consecutive 8-byte words of the frame's contiguous run of records (full cache lines whatever the record length), only
  // the valid slots are read, and (slot, word) advance by the wave's stride with one carry instead of being divided out of a
  // flat 64-bit lane index.  (Round 2: a lane per record -- strided 160-byte stores at 64 cameras; round 4 .. 6: a lane per
  // (frame, slot, word) of the K_max-padded layout -- half the lanes idle at 23 of 48 slots, two 64-bit divisions per lane:
  // 0.153 ms per 100 k frames of 8 x 16.)
  const int parts = a.stride >> 3;
  const int lane = threadIdx.x & 63;
  const int64_t f = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (f >= a.n_frames) return;
  int n = a.n_out[f];
  n = (n < 0 || n > a.K_max) ? 0 : n;
  const int64_t rec0 = a.offsets[f];
  const int64_t room = a.capacity - rec0;  // records beyond the caller's capacity are dropped (the total still says how many)
  if (room < n) n = room < 0 ? 0 : (int)room;
  const int words = n * parts;
  unsigned long long* dst = (unsigned long long*)(a.records + (size_t)rec0 * a.stride);
  const size_t o0 = (size_t)f * a.K_max;
  const int dq = 64 / parts, dr = 64 - dq * parts;
  int k = lane / parts, part = lane - k * parts;
  for (int t = lane; t < words; t += 64) {
    const size_t o = o0 + k;
    unsigned long long w;
    if (part < 3) {
      w = (unsigned long long)__double_as_longlong(a.xyz[o * 3 + part]);
    } else if (part == 3) {
      w = (unsigned long long)__double_as_longlong(a.err[o]);
    } else {
      const int j0 = (part - 4) * 4;
      const int16_t* src = a.corr + o * a.C;
      if ((a.C & 3) == 0) {  // (j0 < C then: the words beyond the cameras exist only when C is not a multiple of 4)
        w = *(const unsigned long long*)(src + j0);
      } else {
        w = 0;
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (j0 + q < a.C) w |= (unsigned long long)(uint16_t)src[j0 + q] << (16 * q);
      }
    }
    dst[t] = w;
    k += dq;
    part += dr;
    if (part >= parts) {
      part -= parts;
      k++;
    }
  }
}

hipError_t launch_compact_tracks(const CompactArgs& a, hipStream_t stream) {
  if (a.n_frames <= 0) return hipSuccess;
  const int n_blocks = (int)((a.n_frames + kScanBlock - 1) / kScanBlock);
  hipLaunchKernelGGL(compact_scan_blocks_kernel, dim3(n_blocks), dim3(kScanBlock), 0, stream, a);
  hipLaunchKernelGGL(compact_scan_totals_kernel, dim3(1), dim3(kScanBlock), 0, stream, a, n_blocks);
  hipLaunchKernelGGL(compact_add_block_offsets_kernel, dim3((unsigned)((a.n_frames + 255) / 256)), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(compact_scatter_kernel, dim3((unsigned)((a.n_frames + 3) / 4)), dim3(256), 0, stream, a);  // a wave per frame
  return hipGetLastError();
}

// ---------------------------------------------------------------- device-side re-submit (mocap_match_triangulate_dev_auto)
// The reference enumerates the full Cartesian product whatever its size (helpers.py:394-400); the frame kernels work
// under caps and flag the frames that hit one.  Three enqueues repair them without the host ever learning how many
// there were: (1) below, the flagged frames' inputs gathered into a scratch batch + their number, on the device;
// (2) the frame kernel on that batch (FrameArgs::n_frames_dev) with the largest caps; (3) the scatter back.
__global__ __launch_bounds__(256) void resubmit_gather_kernel(ResubmitArgs a) {
  // one wave per 64 consecutive frames: ballot of the flagged ones, one atomic for the wave's slots, then the whole
  // wave copies each flagged frame's blobs (C x M x 2 floats) and counts
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wave == 0 && lane == 0) {
    *a.count_next = 0;
    if (a.heavy_count) *a.heavy_count = 0;
    if (a.enum_count) *a.enum_count = 0;
  }
  const int64_t f0 = wave * 64;
  if (f0 >= a.n_frames) return;
  const int64_t f = f0 + lane;
  const bool flagged = f < a.n_frames && a.status[f] != 0 && !(a.status[f] & MOCAP_ST_FINAL_);  // (FINAL: an earlier re-submit has seen it)
  unsigned long long m = __ballot(flagged);
  if (!m) return;
  int base = 0;
  if (lane == 0) base = atomicAdd(a.count, __popcll(m));
  base = __builtin_amdgcn_readfirstlane(base);
  const int per = a.C * a.M * 2;
  for (int j = base; m; m &= m - 1, j++) {
    if (j >= a.cap) break;  // more flagged frames than the scratch batch holds: they keep their status
    const int64_t src = f0 + (__ffsll((long long)m) - 1);
    if (lane == 0) a.list[j] = (int32_t)src;
    const float* sb = a.blobs + (size_t)src * per;
    float* db = a.b2 + (size_t)j * per;
    for (int i = lane; i < per; i += 64) db[i] = sb[i];
    if (lane < a.C) a.c2[(size_t)j * a.C + lane] = a.counts[(size_t)src * a.C + lane];
  }
}

__global__ __launch_bounds__(256) void resubmit_scatter_kernel(ResubmitArgs a) {
  // one workgroup per re-run frame (grid-stride): header by lane 0, then the valid slots word by word
  int total = *a.count;
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.info) {
    a.info[0] = total;
    a.info[1] = total < a.cap ? total : (int)a.cap;
  }
  if (total > a.cap) total = (int)a.cap;
  for (int j = blockIdx.x; j < total; j += gridDim.x) {
    const int64_t f = a.list[j];
    const int n = a.n2[j], s = a.s2[j];
    if (threadIdx.x == 0) {
      // > K_max: the caller's arrays are too small for this frame -- status says so, n_out how many it needs.  A frame the second
      // pass leaves flagged (s != 0) reports NO valid slot: its xyz / err / corr slots still hold the first pass's data, and the
      // device-side consumers (locate_objects, track export, compaction) gate on n_out alone
      a.n_out[f] = s != 0 ? 0 : n;
      if (a.n_cand) a.n_cand[f] = a.g2[j];
      a.status_out[f] = (s == 0 && n > a.K_max) ? MOCAP_ST_ROOT_OVERFLOW_ : (s ? (s | MOCAP_ST_FINAL_) : 0);
    }
    if (s != 0 || n > a.K_max) continue;  // (uniform)
    const size_t so = (size_t)j * a.K_big, dd = (size_t)f * a.K_max;
    for (int i = threadIdx.x; i < 3 * n; i += blockDim.x) a.xyz[dd * 3 + i] = a.x2[so * 3 + i];
    for (int i = threadIdx.x; i < n; i += blockDim.x) a.err[dd + i] = a.e2[so + i];
    for (int i = threadIdx.x; i < n * a.C; i += blockDim.x) a.corr[dd * a.C + i] = a.r2[so * a.C + i];
  }
}

hipError_t launch_resubmit_gather(const ResubmitArgs& a, hipStream_t stream) {
  if (a.n_frames <= 0) return hipSuccess;
  const int64_t waves = (a.n_frames + 63) / 64;
  hipLaunchKernelGGL(resubmit_gather_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_resubmit_scatter(const ResubmitArgs& a, hipStream_t stream) {
  if (a.n_frames <= 0) return hipSuccess;
  const int64_t blocks = a.cap < 1024 ? (a.cap < 1 ? 1 : a.cap) : 1024;
  hipLaunchKernelGGL(resubmit_scatter_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace mocap
