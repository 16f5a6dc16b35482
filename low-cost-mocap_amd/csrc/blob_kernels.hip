// blob_kernels.hip -- the step right BEFORE the hot path in the reference's frame loop:
// raw camera frames -> the per-camera blob centroids (`image_points`) the frame kernel consumes.
//
//   Cameras._camera_read  reference computer_code/api/helpers.py:68-82   per camera: rot90, make_square
//                                          (helpers.py:507-523), cv.undistort, cv.GaussianBlur 9x9,
//                                          cv.filter2D 5x5 sharpening kernel, cv.cvtColor RGB2BGR
//   Cameras._find_dot     helpers.py:143-163   grey, threshold 255*0.2, cv.findContours RETR_TREE,
//                                          cv.moments per contour, int() centroid if m00 != 0
//
// Every OpenCV stage on 8-bit images is integer / fixed-point arithmetic; it is restated here exactly
// (oracle/cv_image_restate.py documents each formula and its OpenCV source):
//   undistort  = gather through a frame-invariant fixed-point map (built once per camera on the host,
//                blob_capi.hip) with 1/32-px bilinear weights:  (sum w*p + 512) >> 10
//   Gaussian   = separable [4 13 30 51 60 51 30 13 4] / 256, (sum + 32768) >> 16, reflect-101 borders
//   filter2D   = 5x5 integer correlation, clamp to [0, 255], reflect-101 borders
//   grey       = (B*9798 + G*19235 + R*3735 + 16384) >> 15 on the channel-swapped frame, mask = grey > 51
//
// Two kernels:
//   blob_mask_kernel      one workgroup per 64 x 64 output tile: the whole chain for the tile runs out of
//                         LDS (undistorted 76 x 76 x 3 region -> row pass -> column pass -> 5x5 -> grey),
//                         HBM sees the raw frame once (through L2 for the halo) and 1 bit per pixel out.
//   blob_contour_kernel   one workgroup per image: border following WITHOUT the sequential raster scan of
//                         Suzuki-Abe.  Every border (outer or hole) is a cycle of the Moore-tracing step
//                         map; a cycle is identified by the smallest "horizontal pair" it passes
//                         (foreground pixel with background at its left = where the raster scan would
//                         have started an outer border, or at its right = a hole border).  All pairs are
//                         traced in parallel, the pair that finds itself to be its cycle's minimum owns the
//                         contour and accumulates the Green's-theorem sums (exact integers).  Parents come
//                         from the pair met by walking left on the start row (Suzuki's LNBD rule, stated
//                         geometrically), output order = pre-order with siblings in reverse discovery order
//                         (cvInsertNodeIntoTree).  The C oracle (oracle/c) uses the sequential algorithm, so
//                         two independent algorithms cross-check each other in the parity tests.
#include "kernels.hpp"

namespace mocap {

namespace {

constexpr int BT = 64;              // output tile edge
constexpr int HG = 4, HF = 2;       // halos of the 9x9 Gaussian and the 5x5 filter
constexpr int UW = BT + 2 * (HG + HF);  // 76: undistorted region edge
constexpr int UP = 80;              // its padded row stride (bytes)
constexpr int BW = BT + 2 * HF;     // 68: blurred region edge
constexpr int BP = 72;              // its padded row stride (bytes)
constexpr int kBlobThreads = 256;

__device__ __forceinline__ int reflect101(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}

// one pixel of the squared frame (helpers.py:507-523) read straight from the raw frame:
// rows [ay, ay+rows) are the frame, 8 feathered rows above/below are edge rows * (7-i)/8, the rest is 0
__device__ __forceinline__ void squared_px(const uint8_t* __restrict__ raw, int rows, int cols, int ay, int rot,
                                           int Y, int X, int& r, int& g, int& b) {
  r = g = b = 0;
  if ((unsigned)X >= (unsigned)cols || (unsigned)Y >= (unsigned)cols) return;
  int rr = Y - ay, scale = 8;
  if (rr < 0) {
    scale = 8 + rr;  // rr = -1 -> 7/8 ... rr = -8 -> 0
    rr = 0;
  } else if (rr >= rows) {
    scale = 7 - (rr - rows);
    rr = rows - 1;
  }
  if (scale <= 0) return;
  int cc = X;
  if (rot == 2) {
    rr = rows - 1 - rr;
    cc = cols - 1 - cc;
  }
  const uint8_t* p = raw + ((size_t)rr * cols + cc) * 3;
  r = (p[0] * scale) >> 3;
  g = (p[1] * scale) >> 3;
  b = (p[2] * scale) >> 3;
}

}  // namespace

__global__ __launch_bounds__(kBlobThreads) void blob_mask_kernel(BlobArgs a) {
  __shared__ uint8_t U[3][UW * UP];
  __shared__ uint16_t Hh[UW * BW];
  __shared__ uint8_t Bl[BW * BP];
  const int tid = threadIdx.x;
  const int S = a.S;
  const int tiles = (S + BT - 1) / BT;
  const int64_t img = blockIdx.x / (tiles * tiles);
  const int tile = blockIdx.x % (tiles * tiles);
  const int ty0 = (tile / tiles) * BT, tx0 = (tile % tiles) * BT;
  const int cam = (int)(img % a.C);
  const uint8_t* raw = a.raw + (size_t)img * a.rows * a.cols * 3;
  const uint32_t* map = a.map + (size_t)cam * S * S;
  const int rot = a.rot[cam];

  // ---- A: undistorted region, reflect-101 applied while filling so later stages are plain windows
  for (int idx = tid; idx < UW * UW; idx += kBlobThreads) {
    const int vy = idx / UW, vx = idx - vy * UW;
    int Y = reflect101(ty0 - (HG + HF) + vy, S), X = reflect101(tx0 - (HG + HF) + vx, S);
    Y = Y < 0 ? 0 : (Y >= S ? S - 1 : Y);  // partial tiles: positions nobody reads
    X = X < 0 ? 0 : (X >= S ? S - 1 : X);
    const uint32_t m = map[(size_t)Y * S + X];
    int o0 = 0, o1 = 0, o2 = 0;
    const int sxp = (m >> 10) & 2047;
    if (sxp != 2047) {
      const int fx = m & 31, fy = (m >> 5) & 31, sx = sxp - 1, sy = (int)(m >> 21) - 1;
      const int w00 = (32 - fx) * (32 - fy), w01 = fx * (32 - fy), w10 = (32 - fx) * fy, w11 = fx * fy;
      int r0, g0, b0, r1, g1, b1, r2, g2, b2, r3, g3, b3;
      squared_px(raw, a.rows, a.cols, a.ay, rot, sy, sx, r0, g0, b0);
      squared_px(raw, a.rows, a.cols, a.ay, rot, sy, sx + 1, r1, g1, b1);
      squared_px(raw, a.rows, a.cols, a.ay, rot, sy + 1, sx, r2, g2, b2);
      squared_px(raw, a.rows, a.cols, a.ay, rot, sy + 1, sx + 1, r3, g3, b3);
      o0 = (w00 * r0 + w01 * r1 + w10 * r2 + w11 * r3 + 512) >> 10;
      o1 = (w00 * g0 + w01 * g1 + w10 * g2 + w11 * g3 + 512) >> 10;
      o2 = (w00 * b0 + w01 * b1 + w10 * b2 + w11 * b3 + 512) >> 10;
    }
    U[0][vy * UP + vx] = (uint8_t)o0;
    U[1][vy * UP + vx] = (uint8_t)o1;
    U[2][vy * UP + vx] = (uint8_t)o2;
  }
  __syncthreads();

  // lane -> output pixels: x = lane % 64, rows (lane / 64) + 4 j: one wave covers one tile row at a time
  const int px = tid & 63, py0 = tid >> 6;
  int grey[BT / 4];
#pragma unroll
  for (int j = 0; j < BT / 4; j++) grey[j] = 0;
  uint8_t* proc = a.processed ? a.processed + (size_t)img * S * S * 3 : nullptr;

  for (int ch = 0; ch < 3; ch++) {
    const uint8_t* Uc = U[ch];
    // ---- B1: row pass (ufixedpoint16, exact: <= 255 * 256)
    for (int idx = tid; idx < UW * BW; idx += kBlobThreads) {
      const int r = idx / BW, c = idx - r * BW;
      const uint8_t* u = Uc + r * UP + c;
      const int h = 60 * u[4] + 51 * (u[3] + u[5]) + 30 * (u[2] + u[6]) + 13 * (u[1] + u[7]) + 4 * (u[0] + u[8]);
      Hh[idx] = (uint16_t)h;
    }
    __syncthreads();
    // ---- B2: column pass (ufixedpoint32), rounding shift
    for (int idx = tid; idx < BW * BW; idx += kBlobThreads) {
      const int r = idx / BW, c = idx - r * BW;
      const uint16_t* h = Hh + r * BW + c;
      const int v = 60 * h[4 * BW] + 51 * (h[3 * BW] + h[5 * BW]) + 30 * (h[2 * BW] + h[6 * BW]) +
                    13 * (h[1 * BW] + h[7 * BW]) + 4 * (h[0] + h[8 * BW]);
      Bl[r * BP + c] = (uint8_t)((v + 32768) >> 16);
    }
    __syncthreads();
    // ---- B3: 5x5 sharpening kernel (helpers.py:76-80) = -(sum of the 25) - corners + 2 d + 4 e + 5 centre
    const int coef = ch == 0 ? 3735 : (ch == 1 ? 19235 : 9798);  // raw R ends up where RGB2GRAY reads "B"
#pragma unroll
    for (int j = 0; j < BT / 4; j++) {
      const int y = py0 + 4 * j;
      const uint8_t* b = Bl + y * BP + px;
      int s = 0;
#pragma unroll
      for (int i = 0; i < 5; i++)
#pragma unroll
        for (int k = 0; k < 5; k++) s += b[i * BP + k];
      const int corners = b[0] + b[4] + b[4 * BP] + b[4 * BP + 4];
      const int diag = b[BP + 1] + b[BP + 3] + b[3 * BP + 1] + b[3 * BP + 3];
      const int edge = b[BP + 2] + b[2 * BP + 1] + b[2 * BP + 3] + b[3 * BP + 2];
      int f = -s - corners + 2 * diag + 4 * edge + 5 * b[2 * BP + 2];
      f = f < 0 ? 0 : (f > 255 ? 255 : f);
      grey[j] += coef * f;
      if (proc && ty0 + y < S && tx0 + px < S)
        proc[((size_t)(ty0 + y) * S + tx0 + px) * 3 + (2 - ch)] = (uint8_t)f;  // RGB2BGR (helpers.py:82)
    }
    __syncthreads();
  }

  // ---- C: grey, threshold (helpers.py:145-146), one 64-bit word per tile row
  const int words = (S + 63) / 64;
  unsigned long long* mask = a.mask + (size_t)img * S * words;
#pragma unroll
  for (int j = 0; j < BT / 4; j++) {
    const int y = py0 + 4 * j;
    const bool on = ((grey[j] + 16384) >> 15) > 51 && tx0 + px < S;
    const unsigned long long w = __ballot(on);
    if (px == 0 && ty0 + y < S) mask[(size_t)(ty0 + y) * words + tx0 / 64] = w;
  }
}

hipError_t launch_blob_mask(const BlobArgs& a, hipStream_t stream) {
  if (a.n_images <= 0) return hipSuccess;
  const int tiles = (a.S + BT - 1) / BT;
  const int64_t grid = a.n_images * tiles * tiles;
  if (grid > 0x7fffffffll) return hipErrorInvalidValue;
  hipLaunchKernelGGL(blob_mask_kernel, dim3((unsigned)grid), dim3(kBlobThreads), 0, stream, a);
  return hipGetLastError();
}

// =====================================================================================================
// contours
namespace {

struct ContourLds {
  uint32_t* msk;      // [(S+2)][stride] padded binary image, bit x of row y = pixel (y-1, x-1)
  uint32_t* pkey;     // [P_cap] horizontal pairs in raster order: ((y * 1024 + x) << 1) | type
  uint32_t* pmin;     // [P_cap] smallest pair key on the pair's cycle
  uint32_t* ckey;     // [N_cap] start pair of each contour (discovery order)
  long long* ca;      // [N_cap][3] a00, a10, a01
  int* cleft;         // [N_cap] contour met by walking left from the start (-1 = frame)
  int* cpar;          // [N_cap]
  int* cfirst;        // [N_cap + 1] first child (head insertion); slot N_cap = the frame
  int* cnext;         // [N_cap] next sibling
  int* scan;          // [kBlobThreads / 64 + 4] block-scan scratch
};

__device__ __forceinline__ bool mbit(const uint32_t* msk, int stride, int y, int x) {
  return (msk[y * stride + (x >> 5)] >> (x & 31)) & 1u;
}

// 8-neighbourhood of padded pixel (y, x) as a ring mask: bit s = neighbour in direction s
// (0 right, 1 up-right, 2 up, 3 up-left, 4 left, 5 down-left, 6 down, 7 down-right: OpenCV's chain codes)
__device__ __forceinline__ uint32_t ring8(const uint32_t* msk, int stride, int y, int x) {
  const int xb = x - 1, wi = xb >> 5, sh = xb & 31;
  const uint32_t* r = msk + y * stride + wi;
  const unsigned long long up = (((unsigned long long)r[1 - stride] << 32) | r[-stride]) >> sh;
  const unsigned long long mid = (((unsigned long long)r[1] << 32) | r[0]) >> sh;
  const unsigned long long dn = (((unsigned long long)r[1 + stride] << 32) | r[stride]) >> sh;
  return (uint32_t)((mid >> 2) & 1) | (uint32_t)((up >> 2) & 1) << 1 | (uint32_t)((up >> 1) & 1) << 2 |
         (uint32_t)(up & 1) << 3 | (uint32_t)(mid & 1) << 4 | (uint32_t)(dn & 1) << 5 |
         (uint32_t)((dn >> 1) & 1) << 6 | (uint32_t)((dn >> 2) & 1) << 7;
}

__device__ __forceinline__ uint32_t pair_key(int y, int x, int type) { return ((uint32_t)(y * 1024 + x) << 1) | type; }

// Follows the border cycle through the pair (pixel (y0, x0), background neighbour in direction d0 = 4
// left / 0 right).  Returns the smallest horizontal pair on the cycle; with MOMENTS also the polygon sums
// of cv::moments (contourMoments) over the cycle's vertex sequence, in original pixel coordinates.
template <bool MOMENTS>
__device__ uint32_t trace_cycle(const uint32_t* msk, int stride, int y0, int x0, int d0, long long* a) {
  int y = y0, x = x0, d = d0;
  uint32_t mink = 0xffffffffu;
  long long a00 = 0, a10 = 0, a01 = 0;
  for (;;) {
    const uint32_t nb = ring8(msk, stride, y, x);
    // scan s = d+1, d+2, ... : rotate so that bit 0 = direction d+1
    const uint32_t rot = ((nb | (nb << 8)) >> ((d + 1) & 7)) & 0xffu;
    const int k = rot ? __builtin_ctz(rot) : 8;         // background pixels crossed before the next border pixel
    const uint32_t crossed8 = ((1u << k) - 1u) << ((d + 1) & 7);
    const uint32_t crossed = (crossed8 | (crossed8 >> 8)) & 0xffu;
    if (crossed & 0x10u) mink = min(mink, pair_key(y, x, 0));
    if (crossed & 0x01u) mink = min(mink, pair_key(y, x, 1));
    if (y == y0 && x == x0 && ((crossed >> d0) & 1u)) break;  // crossed the starting pair again: closed
    const int s = (d + 1 + k) & 7;
    const int nx = x + (int)((0x21000122u >> (4 * s)) & 0xfu) - 1;
    const int ny = y + (int)((0x22210001u >> (4 * s)) & 0xfu) - 1;
    if (MOMENTS) {
      const int pxo = x - 1, pyo = y - 1, qxo = nx - 1, qyo = ny - 1;
      const int dxy = pxo * qyo - qxo * pyo;
      a00 += dxy;
      a10 += (long long)dxy * (pxo + qxo);
      a01 += (long long)dxy * (pyo + qyo);
    }
    x = nx;
    y = ny;
    d = (s + 4) & 7;
  }
  if (MOMENTS) {
    a[0] = a00;
    a[1] = a10;
    a[2] = a01;
  }
  return mink;
}

__device__ __forceinline__ int lower_bound_u32(const uint32_t* v, int n, uint32_t key) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (v[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// exclusive block scan of one int per lane; returns the lane's offset, total in *total
__device__ int block_scan_excl(int v, int* scratch, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  __syncthreads();
  if (lane == 63) scratch[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < nw; w++) {
    const int t = scratch[w];
    if (w < wave) base += t;
    tot += t;
  }
  *total = tot;
  return base + inc - v;
}

}  // namespace

__global__ __launch_bounds__(kBlobThreads) void blob_contour_kernel(BlobArgs a, int P_cap, int N_cap, int only_overflowed) {
  extern __shared__ unsigned char lds_raw[];
  const int tid = threadIdx.x;
  const int64_t img = blockIdx.x;
  if (only_overflowed && !(a.status[img] & BLOB_ST_CAP_OVERFLOW_)) return;
  const int S = a.S, SP = S + 2;
  const int stride = (SP + 31) / 32 + 1;  // one spare zero word: ring8 reads word + 1
  ContourLds L;
  {
    unsigned char* p = lds_raw;
    L.ca = (long long*)p;       p += sizeof(long long) * 3 * N_cap;
    L.msk = (uint32_t*)p;       p += sizeof(uint32_t) * SP * stride;
    L.pkey = (uint32_t*)p;      p += sizeof(uint32_t) * P_cap;
    L.pmin = (uint32_t*)p;      p += sizeof(uint32_t) * P_cap;
    L.ckey = (uint32_t*)p;      p += sizeof(uint32_t) * N_cap;
    L.cleft = (int*)p;          p += sizeof(int) * N_cap;
    L.cpar = (int*)p;           p += sizeof(int) * N_cap;
    L.cfirst = (int*)p;         p += sizeof(int) * (N_cap + 1);
    L.cnext = (int*)p;          p += sizeof(int) * N_cap;
    L.scan = (int*)p;
  }
  const int words64 = (S + 63) / 64;
  const uint32_t* gm = (const uint32_t*)(a.mask + (size_t)img * S * words64);  // little-endian halves
  const int gw = words64 * 2;

  // ---- load the mask, shifted by one pixel into the zero frame findContours pads with (copyMakeBorder)
  for (int i = tid; i < SP * stride; i += kBlobThreads) {
    const int y = i / stride, w = i - y * stride;
    uint32_t v = 0;
    if (y >= 1 && y <= S) {
      const uint32_t* row = gm + (size_t)(y - 1) * gw;
      const uint32_t cur = w < gw ? row[w] : 0u, prev = (w >= 1 && w - 1 < gw) ? row[w - 1] : 0u;
      v = (cur << 1) | (prev >> 31);
      // bits beyond the image (x > S) are already zero: the mask kernel writes zeros there
    }
    L.msk[i] = v;
  }
  __syncthreads();

  // ---- 1: horizontal pairs in raster order.  Each lane owns a run of consecutive words.
  const int n_words = S * stride;  // rows 1..S
  const int per = (n_words + kBlobThreads - 1) / kBlobThreads;
  const int w_lo = tid * per, w_hi = min(n_words, w_lo + per);
  int cnt = 0;
  for (int i = w_lo; i < w_hi; i++) {
    const int idx = stride + i;  // skip padded row 0
    const int w = i % stride;
    const uint32_t m = L.msk[idx];
    const uint32_t left = (m << 1) | (w ? L.msk[idx - 1] >> 31 : 0u);
    const uint32_t right = (m >> 1) | (w + 1 < stride ? L.msk[idx + 1] << 31 : 0u);
    cnt += __popc(m & ~left) + __popc(m & ~right);
  }
  int n_pairs = 0;
  int off = block_scan_excl(cnt, L.scan, &n_pairs);
  int st = 0;
  if (n_pairs > P_cap) {
    if (tid == 0) {
      a.status[img] = BLOB_ST_CAP_OVERFLOW_;
      a.counts[img] = 0;
    }
    return;
  }
  for (int i = w_lo; i < w_hi; i++) {
    const int idx = stride + i;
    const int y = 1 + i / stride, w = i % stride;
    const uint32_t m = L.msk[idx];
    const uint32_t left = (m << 1) | (w ? L.msk[idx - 1] >> 31 : 0u);
    const uint32_t right = (m >> 1) | (w + 1 < stride ? L.msk[idx + 1] << 31 : 0u);
    const uint32_t lo = m & ~left, ro = m & ~right;
    uint32_t any = lo | ro;
    while (any) {
      const int b = __builtin_ctz(any);
      any &= any - 1;
      const int x = w * 32 + b;
      if ((lo >> b) & 1u) L.pkey[off++] = pair_key(y, x, 0);
      if ((ro >> b) & 1u) L.pkey[off++] = pair_key(y, x, 1);
    }
  }
  __syncthreads();

  // ---- 2: every pair follows its cycle to find the cycle's smallest pair
  for (int i = tid; i < n_pairs; i += kBlobThreads) {
    const uint32_t key = L.pkey[i];
    const int y = (int)(key >> 1) / 1024, x = (int)(key >> 1) % 1024;
    L.pmin[i] = trace_cycle<false>(L.msk, stride, y, x, (key & 1u) ? 0 : 4, nullptr);
  }
  __syncthreads();

  // ---- 3: contours = pairs that are their cycle's minimum, numbered in raster (= discovery) order
  const int pper = (n_pairs + kBlobThreads - 1) / kBlobThreads;
  const int p_lo = min(n_pairs, tid * pper), p_hi = min(n_pairs, p_lo + pper);
  int nst = 0;
  for (int i = p_lo; i < p_hi; i++) nst += L.pmin[i] == L.pkey[i] ? 1 : 0;
  int n_cont = 0;
  int coff = block_scan_excl(nst, L.scan, &n_cont);
  if (n_cont > N_cap) {
    if (tid == 0) {
      a.status[img] = BLOB_ST_CAP_OVERFLOW_;
      a.counts[img] = 0;
    }
    return;
  }
  for (int i = p_lo; i < p_hi; i++)
    if (L.pmin[i] == L.pkey[i]) L.ckey[coff++] = L.pkey[i];
  __syncthreads();

  // ---- 4: per contour: polygon sums, and the border met by walking left on the start row
  for (int c = tid; c < n_cont; c += kBlobThreads) {
    const uint32_t key = L.ckey[c];
    const int hole = key & 1u;
    const int y = (int)(key >> 1) / 1024, x = (int)(key >> 1) % 1024;
    trace_cycle<true>(L.msk, stride, y, x, hole ? 0 : 4, L.ca + 3 * c);
    // outer border: nearest foreground pixel q left of the start (its pair with the background at its
    // right); hole border: left end q of the foreground run the start sits in (pair with its left)
    const uint32_t* row = L.msk + y * stride;
    int q = -1;
    {
      int wi = x >> 5;
      uint32_t mw = (hole ? ~row[wi] : row[wi]) & ((x & 31) ? (0xffffffffu >> (32 - (x & 31))) : 0u);
      for (;;) {
        if (mw) {
          q = wi * 32 + 31 - __builtin_clz(mw);
          break;
        }
        if (--wi < 0) break;
        mw = hole ? ~row[wi] : row[wi];
      }
    }
    int left = -1;
    if (hole) q += 1;  // q was the nearest background pixel; the run starts right of it (column 0 is background)
    if (q >= 0) {
      const uint32_t pk = pair_key(y, q, hole ? 0 : 1);
      const int pi = lower_bound_u32(L.pkey, n_pairs, pk);
      left = lower_bound_u32(L.ckey, n_cont, L.pmin[pi]);
    }
    L.cleft[c] = left;
  }
  __syncthreads();

  // ---- 5: tree and output order (one lane: a few dozen contours)
  if (tid == 0) {
    for (int c = 0; c < n_cont; c++) L.cfirst[c] = -1;
    L.cfirst[N_cap] = -1;
    for (int c = 0; c < n_cont; c++) {
      const int l = L.cleft[c];
      int par = -1;
      if (l >= 0) par = ((L.ckey[l] ^ L.ckey[c]) & 1u) ? l : L.cpar[l];  // Suzuki-Abe's parent table
      L.cpar[c] = par;
      const int slot = par < 0 ? N_cap : par;
      L.cnext[c] = L.cfirst[slot];  // head insertion: siblings end up in reverse discovery order
      L.cfirst[slot] = c;
    }
    float* out = a.blobs + (size_t)img * a.M_max * 2;
    int n = 0;
    int node = L.cfirst[N_cap];
    while (node >= 0) {
      const long long a00 = L.ca[3 * node];
      if (a00 != 0) {  // cv::moments: fabs(a00) > FLT_EPSILON; helpers.py:152
        const double sgn2 = a00 > 0 ? 0.5 : -0.5, sgn6 = a00 > 0 ? 0.16666666666666666666666666666667 : -0.16666666666666666666666666666667;
        const double m00 = (double)a00 * sgn2, m10 = (double)L.ca[3 * node + 1] * sgn6, m01 = (double)L.ca[3 * node + 2] * sgn6;
        if (n < a.M_max) {
          out[2 * n] = (float)(int)(m10 / m00);  // int(): helpers.py:153-154
          out[2 * n + 1] = (float)(int)(m01 / m00);
        }
        n++;
      }
      if (L.cfirst[node] >= 0) {
        node = L.cfirst[node];
      } else {
        while (node >= 0 && L.cnext[node] < 0) node = L.cpar[node];
        if (node >= 0) node = L.cnext[node];
      }
    }
    st |= n > a.M_max ? BLOB_ST_POINT_OVERFLOW_ : 0;
    a.counts[img] = n > a.M_max ? a.M_max : n;
    a.status[img] = st;
    if (a.n_contours) a.n_contours[img] = n_cont;
  }
}

size_t blob_contour_lds_bytes(int S, int P_cap, int N_cap) {
  const int SP = S + 2, stride = (SP + 31) / 32 + 1;
  return sizeof(long long) * 3 * N_cap + sizeof(uint32_t) * SP * stride + sizeof(uint32_t) * 2 * P_cap +
         sizeof(uint32_t) * N_cap + sizeof(int) * (4 * N_cap + 1) + sizeof(int) * 16;
}

hipError_t launch_blob_contours(const BlobArgs& a, int P_cap, int N_cap, int only_overflowed, hipStream_t stream) {
  if (a.n_images <= 0) return hipSuccess;
  if (a.n_images > 0x7fffffffll) return hipErrorInvalidValue;
  const size_t lds = blob_contour_lds_bytes(a.S, P_cap, N_cap);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)blob_contour_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(blob_contour_kernel, dim3((unsigned)a.n_images), dim3(kBlobThreads), lds, stream, a, P_cap, N_cap,
                     only_overflowed);
  return hipGetLastError();
}

}  // namespace mocap
