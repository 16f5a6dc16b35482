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
// (DESIGN.md section 3.5 lists each formula with its OpenCV source file):
//   undistort  = gather through a frame-invariant fixed-point map (built once per camera on the host,
//                blob_capi.hip) with 1/32-px bilinear weights:  (sum w*p + 512) >> 10
//   Gaussian   = separable [4 13 30 51 60 51 30 13 4] / 256, (sum + 32768) >> 16, reflect-101 borders
//   filter2D   = 5x5 integer correlation, clamp to [0, 255], reflect-101 borders
//   grey       = (B*9798 + G*19235 + R*3735 + 16384) >> 15 on the channel-swapped frame, mask = grey > 51
//
// Three kernels:
//   blob_activity_kernel  (only with the dark-tile early-out) min / max byte per (16-row band, 16-byte segment) of
//                         the squared frame, read-only: nothing but the small map is written.
//   blob_mask_kernel      one workgroup per 64 x 64 output tile: the whole chain for the tile runs out of
//                         LDS (table-driven bilinear gather of the 76 x 76 x 3 region STRAIGHT FROM THE RAW
//                         FRAME -- rot90 / make_square live in the table and a short fix-up list -> row pass (dot4) ->
//                         column pass (dot2) -> 5x5 (signed dot4) -> grey), 1 bit per pixel out.
//   blob_contour_kernel   one workgroup per image: border following WITHOUT the sequential raster scan of
//                         Suzuki-Abe.  Every border (outer or hole) is a cycle of the Moore-tracing step
//                         map; a cycle is identified by the smallest "horizontal pair" it passes
//                         (foreground pixel with background at its left = where the raster scan would
//                         have started an outer border, or at its right = a hole border).  Every pair walks
//                         to the next pair of its border (work = total border length), pointer jumping over
//                         these links finds each cycle's minimum in log2(pairs) rounds, and the pairs add
//                         their segments' Green's-theorem sums to their contour (exact integers).  Parents come
//                         from the pair met by walking left on the start row (Suzuki's LNBD rule, stated
//                         geometrically), output order = pre-order with siblings in reverse discovery order
//                         (cvInsertNodeIntoTree).  The parity tests check it against the sequential
//                         raster-scan algorithm: two independent algorithms cross-check each other.
#include "kernels.hpp"

namespace mocap {

namespace {

constexpr int BT = kBlobTile;        // output tile edge
constexpr int HG = 4, HF = 2;       // halos of the 9x9 Gaussian and the 5x5 filter
static_assert(HG + HF == kBlobHalo, "halo");
constexpr int UW = BT + 2 * (HG + HF);  // 76: undistorted region edge
static_assert(UW * UW == kBlobRegion, "region");
constexpr int UP = UW;              // its row stride: region index = LDS offset (the row pass over-reads <= 4 bytes)
constexpr int BW = BT + 2 * HF;     // 68: blurred region edge
constexpr int BP = 68;              // its row stride (bytes): the 5x5 stage reads at most column 67
constexpr int VT = 78;              // stride (u16) of the column-major row-pass result: 76 rows + pad
constexpr int kBlobThreads = 256;
constexpr int kActSlots = kBlobActSlots;  // 16-byte segments per lane per row in the pre-pass


}  // namespace

// packed tap weights (byte 0 = first tap)
constexpr uint32_t pk4(int a, int b, int c, int d) {
  return (uint32_t)(a & 0xff) | (uint32_t)(b & 0xff) << 8 | (uint32_t)(c & 0xff) << 16 | (uint32_t)(d & 0xff) << 24;
}
typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
typedef unsigned char uchar4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t udot2(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_udot2(__builtin_bit_cast(v2u16, a), __builtin_bit_cast(v2u16, b), c, false);
}

// ---- activity pass (only when the dark-tile early-out is on): per 16-row band of the squared frame
// (np.rot90 + make_square, helpers.py:507-523: the frame rows and the 2 x 8 feathered rows, edge row * (7 - i) / 8
// truncated) and per 16-byte segment of a squared row, the min / max of the bytes the squared frame WOULD hold there.
// Nothing but the map is written: the mask kernel gathers from the raw frame itself.
__global__ __launch_bounds__(kBlobThreads) void blob_activity_kernel(BlobArgs a) {
  const int first = a.ay - 8, n_rows = a.rows + 16;  // squared rows [first, first + n_rows)
  const int groups = (n_rows + kSquareRows - 1) / kSquareRows;
  const int64_t img = blockIdx.x / groups;
  const int grp = blockIdx.x % groups;
  const int cam = (int)((a.img_base + img) % a.C);
  const int rot = a.rot[cam];
  const uint8_t* raw = a.raw + (size_t)img * a.rows * a.cols * 3;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row_bytes = a.cols * 3;
  // A lane owns segments lane, lane + 64, ... of a row (rows up to kActSlots * 1024 bytes: the host rejects
  // wider frames), one accumulator pair per segment.
  __shared__ uint32_t act[4][kActSlots * 64];
  uint32_t mn[kActSlots], mx[kActSlots];  // two 16-bit fields, min / max over even and odd bytes alike
#pragma unroll
  for (int j = 0; j < kActSlots; j++) {
    mn[j] = 0x00ff00ffu;
    mx[j] = 0u;
  }
  {
#pragma unroll
    for (int k = 0; k < kSquareRows / 4; k++) {
      const int Y = first + grp * kSquareRows + wave * (kSquareRows / 4) + k;
      if (Y >= first + n_rows) break;
      int r = Y - a.ay, scale = 8;
      if (r < 0) {
        scale = 8 + r;  // r = -1 -> 7/8 ... r = -8 -> 0
        r = 0;
      } else if (r >= a.rows) {
        scale = 7 - (r - a.rows);
        r = a.rows - 1;
      }
      // 16 bytes per lane: rows start 16-byte aligned (cols % 16 == 0 checked by the host).  A camera turned by
      // 180 degrees reads the mirrored row; its segments are mirrored when the map is written below.
      const uint8_t* src = raw + (size_t)(rot ? a.rows - 1 - r : r) * row_bytes;
#pragma unroll
      for (int j = 0; j < kActSlots; j++) {
        const int i = (j * 64 + lane) * 16;
        if (i >= row_bytes) break;
        uint4 v = *(const uint4*)(src + i);
        const uint32_t* w = (const uint32_t*)&v;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          uint32_t e = w[q] & 0x00ff00ffu, o = (w[q] >> 8) & 0x00ff00ffu;
          if (scale != 8) {
            e = (e * (uint32_t)scale) >> 3 & 0x00ff00ffu;
            o = (o * (uint32_t)scale) >> 3 & 0x00ff00ffu;
          }
          mn[j] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(v2u16, mn[j]), __builtin_elementwise_min(__builtin_bit_cast(v2u16, e), __builtin_bit_cast(v2u16, o))));
          mx[j] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(v2u16, mx[j]), __builtin_elementwise_max(__builtin_bit_cast(v2u16, e), __builtin_bit_cast(v2u16, o))));
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < kActSlots; j++) {
    const uint32_t lo = min(mn[j] & 0xffffu, mn[j] >> 16), hi = max(mx[j] & 0xffffu, mx[j] >> 16);
    act[wave][j * 64 + lane] = lo | hi << 8;
  }
  __syncthreads();
  const int segs = row_bytes / 16;
  for (int sg = threadIdx.x; sg < segs; sg += kBlobThreads) {
    // segment sg of the SQUARED row = bytes [16 sg, 16 sg + 16) = pixels X0 .. X1; for a rotated camera those are
    // the raw pixels cols-1-X1 .. cols-1-X0, i.e. a byte range that up to three raw segments cover (a superset of
    // the bytes is a valid bound: the early-out only ever skips less)
    int r0 = sg, r1 = sg;
    if (rot) {
      const int X0 = 16 * sg / 3, X1 = min((16 * sg + 15) / 3, a.cols - 1);
      r0 = 3 * (a.cols - 1 - X1) / 16;
      r1 = (3 * (a.cols - 1 - X0) + 2) / 16;
    }
    uint32_t l = 255u, h = 0u;
    for (int rs = r0; rs <= r1; rs++)
      for (int w2 = 0; w2 < 4; w2++) {
        l = min(l, act[w2][rs] & 0xffu);
        h = max(h, act[w2][rs] >> 8);
      }
    uint8_t* o = a.activity + (((size_t)img * groups + grp) * segs + sg) * 2;
    o[0] = (uint8_t)l;
    o[1] = (uint8_t)h;
  }
}

hipError_t launch_blob_activity(const BlobArgs& a, hipStream_t stream) {
  if (a.n_images <= 0) return hipSuccess;
  const int groups = (a.rows + 16 + kSquareRows - 1) / kSquareRows;
  hipLaunchKernelGGL(blob_activity_kernel, dim3((unsigned)(a.n_images * groups)), dim3(kBlobThreads), 0, stream, a);
  return hipGetLastError();
}

__global__ __launch_bounds__(kBlobThreads) void blob_mask_kernel(BlobArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t U[3][UW * UP + 4];  // undistorted region, row-major (+4: over-read of the last row)
  __shared__ __attribute__((aligned(16))) uint16_t Vt[BW * VT];   // row-pass result, COLUMN-major (8.8 fixed point)
  __shared__ __attribute__((aligned(16))) uint8_t Bl[BW * BP];    // blurred region, row-major, stored as value - 128
  const int tid = threadIdx.x;
  const int S = a.S;
  const int tiles = (S + BT - 1) / BT;
  const int64_t img = blockIdx.x / (tiles * tiles);
  const int tile = blockIdx.x % (tiles * tiles);
  const int ty0 = (tile / tiles) * BT, tx0 = (tile % tiles) * BT;
  const int cam = (int)((a.img_base + img) % a.C);
  // the gather reads the RAW frame: rot90 / make_square (feathered rows, zero padding) are folded into the table
  // (interior taps: plain offsets; the few taps on feathered / zero rows or at the frame's edge: a fix-up list)
  const uint8_t* sq = a.raw + (size_t)img * a.rows * a.cols * 3;
  const size_t lt = (size_t)a.cam_lens[cam] * tiles * tiles + tile;
  const uint32_t* tab = a.gather + lt * kBlobGather;
  const uint8_t* sq_below = sq + (size_t)a.cols * 3;  // the raw row under a tap
  const int words = (S + 63) / 64;
  unsigned long long* mask = a.mask + (size_t)img * S * words;

  // ---- dark-tile early-out (exact).  All bytes this tile can gather lie in [m, M] (activity map of the
  // pre-pass; the zero frame counts as 0).  Bilinear interpolation and the Gaussian are convex combinations
  // with round-to-nearest, so every blurred value stays in [m, M]; the 5x5 kernel sums to 0 with positive
  // weights summing to 20, so its output is at most 20 (M - m); grey is a convex combination of the three
  // channels.  M - m <= 2  =>  grey <= 40 < 52: no mask bit can be set.
  if (a.skip_dark == 1 && !a.processed) {
    const int16_t* box = a.tile_box + lt * 4;
    const int b0 = box[0], b1 = box[1], s0 = box[2], s1 = box[3];
    if (b0 >= 0) {
      const int segs = a.cols * 3 / 16, bands = (a.rows + 16 + kSquareRows - 1) / kSquareRows;
      const uint8_t* act = a.activity + (size_t)img * bands * segs * 2;
      const int nb = b1 - b0 + 1, ns = s1 - s0 + 1;
      uint32_t lo = 255u, hi = 0u;
      for (int i = tid; i < nb * ns; i += kBlobThreads) {
        const int bi = b0 + i / ns, si = s0 + i % ns;
        const uint8_t* e = act + ((size_t)bi * segs + si) * 2;
        lo = min(lo, (uint32_t)e[0]);
        hi = max(hi, (uint32_t)e[1]);
      }
      __shared__ uint32_t red[2][kBlobThreads / 64];
      for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, o));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, o));
      }
      if ((tid & 63) == 0) {
        red[0][tid >> 6] = lo;
        red[1][tid >> 6] = hi;
      }
      __syncthreads();
      lo = min(min(red[0][0], red[0][1]), min(red[0][2], red[0][3]));
      hi = max(max(red[1][0], red[1][1]), max(red[1][2], red[1][3]));
      // a box that reaches into the zero rows / zero frame of the squared layout (host flag) also sees 0
      if (hi <= (a.tile_zero[lt] ? 0u : lo) + 2u) {
        for (int y = tid; y < BT; y += kBlobThreads)
          if (ty0 + y < S) mask[(size_t)(ty0 + y) * words + tx0 / 64] = 0ull;
        return;
      }
    }
  }

  // ---- A: undistorted region = table-driven bilinear gather from the squared frame.  The table already
  // holds the reflect-101 of the region, the tap address and the 1/32-px fractions; the zero frame of the
  // squared layout is cv::remap's BORDER_CONSTANT.  Software-pipelined: all table words of a chunk, then
  // all pixel loads (two unaligned 8-byte loads = 2 x 2 RGB pixels), then the arithmetic.
  constexpr int CH = 8;
#pragma unroll
  for (int base = 0; base < kBlobGather; base += CH * kBlobThreads) {
    constexpr int kIters = kBlobGather / kBlobThreads;
    uint32_t mm[CH];
    unsigned long long tt[CH], bb[CH];
#pragma unroll
    for (int k = 0; k < CH; k++)
      if (base / kBlobThreads + k < kIters) mm[k] = tab[base + k * kBlobThreads + tid];
#pragma unroll
    for (int k = 0; k < CH; k++)
      if (base / kBlobThreads + k < kIters) {
        uint32_t off = mm[k] & 0x3fffffu;  // the same 32-bit lane offset against two uniform row bases
        off = off == 0x3fffffu ? 0u : off;  // "zero or fixed up later": any valid address will do
        __builtin_memcpy(&tt[k], sq + off, 8);
        __builtin_memcpy(&bb[k], sq_below + off, 8);
      }
#pragma unroll
    for (int k = 0; k < CH; k++)
      if (base / kBlobThreads + k < kIters) {
        const int idx = base + k * kBlobThreads + tid;
        // (32-fy) [(32-fx) p00 + fx p01] + fy [(32-fx) p10 + fx p11] = the four-weight sum of cv::remap's
        // BilinearTab_i (weights (32-fx)(32-fy)*32 ... summing to 32768), then (sum*32 + 16384) >> 15.
        // R and B ride in one register as two 16-bit fields (each partial sum < 2^13).
        const uint32_t fx = (mm[k] >> 22) & 31u, fy = mm[k] >> 27, gx = 32u - fx, gy = 32u - fy;
        const uint32_t tl = (uint32_t)tt[k], th = (uint32_t)(tt[k] >> 32), bl = (uint32_t)bb[k], bh = (uint32_t)(bb[k] >> 32);
        const uint32_t t_rb = gx * (tl & 0x00ff00ffu) + fx * __builtin_amdgcn_perm(th, tl, 0x0c050c03u);
        const uint32_t t_g = gx * ((tl >> 8) & 0xffu) + fx * (th & 0xffu);
        const uint32_t b_rb = gx * (bl & 0x00ff00ffu) + fx * __builtin_amdgcn_perm(bh, bl, 0x0c050c03u);
        const uint32_t b_g = gx * ((bl >> 8) & 0xffu) + fx * (bh & 0xffu);
        const bool zero = (mm[k] & 0x3fffffu) == 0x3fffffu;
        const uint32_t o_r = zero ? 0u : (gy * (t_rb & 0xffffu) + fy * (b_rb & 0xffffu) + 512u) >> 10;
        const uint32_t o_g = zero ? 0u : (gy * t_g + fy * b_g + 512u) >> 10;
        const uint32_t o_b = zero ? 0u : (gy * (t_rb >> 16) + fy * (b_rb >> 16) + 512u) >> 10;
        if (idx < kBlobRegion) {
          U[0][idx] = (uint8_t)o_r;
          U[1][idx] = (uint8_t)o_g;
          U[2][idx] = (uint8_t)o_b;
        }
      }
  }
  {
    // fix-ups: region pixels with a tap on a feathered row (edge row * (7 - i) / 8, truncated per byte), on a zero
    // row / outside the frame's columns, or too close to the end of the image for the 8-byte loads above.  Each
    // record names its four taps individually {raw byte offset | scale << 22, scale 0 = zero pixel} and keeps the
    // map's own fractions, so rotation needs no special case here.
    const int n_fix = a.fix_cnt[lt];
    if (n_fix) {
      __syncthreads();
      const uint32_t* rec = a.fix_rec + (size_t)a.fix_off[lt] * 5;
      for (int i = tid; i < n_fix; i += kBlobThreads) {
        const uint32_t w0 = rec[5 * i];
        const uint32_t idx = w0 & 0xffffu, fx = (w0 >> 16) & 31u, fy = (w0 >> 24) & 31u, gx = 32u - fx, gy = 32u - fy;
        uint32_t px[4][3];
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const uint32_t d = rec[5 * i + 1 + t], sc = (d >> 22) & 15u;
          const uint8_t* p = sq + (d & 0x3fffffu);
#pragma unroll
          for (int c = 0; c < 3; c++) px[t][c] = sc ? ((uint32_t)p[c] * sc) >> 3 : 0u;
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const uint32_t top = gx * px[0][c] + fx * px[1][c], bot = gx * px[2][c] + fx * px[3][c];
          U[c][idx] = (uint8_t)((gy * top + fy * bot + 512u) >> 10);
        }
      }
    }
  }
  __syncthreads();
  // ---- dark-tile early-out, folded into this pass (mocap_set_blob_options(2), round 6): the range of the UNDISTORTED region
  // itself, which is in LDS now -- no activity pass, no second read of the image.  Everything below depends on U alone, and the
  // argument above holds with m, M = min / max over U's 3 x 76 x 76 bytes (the Gaussian is a convex combination with
  // round-to-nearest; the 5x5 kernel sums to 0 with positive weights summing to 20; grey is a convex combination of the
  // channels): M - m <= 2  =>  grey <= 40 < 52, no mask bit.  A dark tile pays its gather first: the price, measured.
  if (a.skip_dark == 2 && !a.processed) {
    uint32_t lo4 = 0xffffffffu, hi4 = 0u;  // four byte lanes at a time
    for (int c = 0; c < 3; c++) {
      const uint32_t* w = (const uint32_t*)U[c];
      for (int i = tid; i < UW * UP / 4; i += kBlobThreads) {  // (76 * 76 = 5776 bytes = 1444 words)
        const uint32_t v = w[i];
        lo4 = (uint32_t)__builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(uchar4_t, lo4), __builtin_bit_cast(uchar4_t, v)));
        hi4 = (uint32_t)__builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(uchar4_t, hi4), __builtin_bit_cast(uchar4_t, v)));
      }
    }
    uint32_t lo = min(min(lo4 & 0xffu, (lo4 >> 8) & 0xffu), min((lo4 >> 16) & 0xffu, lo4 >> 24));
    uint32_t hi = max(max(hi4 & 0xffu, (hi4 >> 8) & 0xffu), max((hi4 >> 16) & 0xffu, hi4 >> 24));
    for (int o = 32; o > 0; o >>= 1) {
      lo = min(lo, (uint32_t)__shfl_xor((int)lo, o));
      hi = max(hi, (uint32_t)__shfl_xor((int)hi, o));
    }
    __shared__ uint32_t red2[2][kBlobThreads / 64];
    if ((tid & 63) == 0) {
      red2[0][tid >> 6] = lo;
      red2[1][tid >> 6] = hi;
    }
    __syncthreads();
    lo = min(min(red2[0][0], red2[0][1]), min(red2[0][2], red2[0][3]));
    hi = max(max(red2[1][0], red2[1][1]), max(red2[1][2], red2[1][3]));
    if (hi <= lo + 2u) {  // (uniform)
      for (int y = tid; y < BT; y += kBlobThreads)
        if (ty0 + y < S) mask[(size_t)(ty0 + y) * words + tx0 / 64] = 0ull;
      return;
    }
  }

  // lane -> output pixels: x = lane % 64, rows 16 (lane / 64) + j: one wave covers one tile row at a time
  const int px = tid & 63, wv = tid >> 6;
  int grey[16];
#pragma unroll
  for (int j = 0; j < 16; j++) grey[j] = 0;
  uint8_t* proc = a.processed ? a.processed + (size_t)img * S * S * 3 : nullptr;

  constexpr uint32_t G0 = pk4(4, 13, 30, 51), G1 = pk4(60, 51, 30, 13), G2 = pk4(4, 0, 0, 0);  // Gaussian taps 0..8
  constexpr uint32_t P0 = 4u | 13u << 16, P1 = 30u | 51u << 16, P2 = 60u | 51u << 16, P3 = 30u | 13u << 16, P4 = 4u;
  // rows of the 5x5 sharpening kernel (helpers.py:76-80): rows 0/4, rows 1/3, row 2
  constexpr uint32_t KA0 = pk4(-2, -1, -1, -1), KA1 = pk4(-2, 0, 0, 0);
  constexpr uint32_t KB0 = pk4(-1, 1, 3, 1), KB1 = pk4(-1, 0, 0, 0);
  constexpr uint32_t KC0 = pk4(-1, 3, 4, 3), KC1 = pk4(-1, 0, 0, 0);

  for (int ch = 0; ch < 3; ch++) {
    const uint8_t* Uc = U[ch];
    // ---- B1: row pass, 4 outputs per lane from 16 bytes (v_dot4_u32_u8); exact in 16 bits (<= 255 * 256)
    for (int it = tid; it < UW * (BW / 4); it += kBlobThreads) {
      const int r = it / (BW / 4), c = 4 * (it - r * (BW / 4));
      const uint32_t* u = (const uint32_t*)(Uc + r * UP + c);
      const uint32_t d0 = u[0], d1 = u[1], d2 = u[2], d3 = u[3];
      uint16_t* out = Vt + c * VT + r;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t w0 = j ? __builtin_amdgcn_alignbyte(d1, d0, j) : d0;
        const uint32_t w1 = j ? __builtin_amdgcn_alignbyte(d2, d1, j) : d1;
        const uint32_t w2 = j ? __builtin_amdgcn_alignbyte(d3, d2, j) : d2;
        const uint32_t h = __builtin_amdgcn_udot4(w0, G0, __builtin_amdgcn_udot4(w1, G1, __builtin_amdgcn_udot4(w2, G2, 0u, false), false), false);
        out[j * VT] = (uint16_t)h;
      }
    }
    __syncthreads();
    // ---- B2: column pass, 4 outputs per lane from 14 values (v_dot2_u32_u16).  The accumulator starts at
    // 32768 + (128 << 16): rounding of (sum + 32768) >> 16 and the "- 128" of the signed store in one
    for (int it = tid; it < BW * (BW / 4); it += kBlobThreads) {
      const int m = it / BW, x = it - m * BW;
      const uint32_t* v = (const uint32_t*)(Vt + x * VT + 4 * m);
      uint32_t e[7];
#pragma unroll
      for (int i = 0; i < 7; i++) e[i] = v[i];
      uint32_t q[6];
#pragma unroll
      for (int i = 0; i < 6; i++) q[i] = __builtin_amdgcn_alignbyte(e[i + 1], e[i], 2);
      constexpr uint32_t R0 = 32768u + (128u << 16);
      const uint32_t s0 = udot2(e[0], P0, udot2(e[1], P1, udot2(e[2], P2, udot2(e[3], P3, udot2(e[4], P4, R0)))));
      const uint32_t s1 = udot2(q[0], P0, udot2(q[1], P1, udot2(q[2], P2, udot2(q[3], P3, udot2(q[4], P4, R0)))));
      const uint32_t s2 = udot2(e[1], P0, udot2(e[2], P1, udot2(e[3], P2, udot2(e[4], P3, udot2(e[5], P4, R0)))));
      const uint32_t s3 = udot2(q[1], P0, udot2(q[2], P1, udot2(q[3], P2, udot2(q[4], P3, udot2(q[5], P4, R0)))));
      uint8_t* o = Bl + (4 * m) * BP + x;
      o[0] = (uint8_t)(s0 >> 16);  // value - 128 (mod 256): the 5x5 kernel sums to 0, so the offset cancels
      o[BP] = (uint8_t)(s1 >> 16);
      o[2 * BP] = (uint8_t)(s2 >> 16);
      o[3 * BP] = (uint8_t)(s3 >> 16);
    }
    __syncthreads();
    // ---- B3: 5x5 kernel as three 5-tap row filters per blurred row (v_dot4c_i32_i8), accumulated straight
    // into the output rows they belong to
    int acc[16];
#pragma unroll
    for (int j = 0; j < 16; j++) acc[j] = 0;
    const int sh = px & 3;
#pragma unroll
    for (int t = 0; t < 20; t++) {
      const uint32_t* b = (const uint32_t*)(Bl + (16 * wv + t) * BP + (px & ~3));
      const uint32_t lo = b[0], hi = b[1];
      const int w0 = (int)__builtin_amdgcn_alignbyte(hi, lo, sh), w1 = (int)(hi >> (8 * sh));
      if (t < 16) acc[t] = __builtin_amdgcn_sdot4(w0, (int)KA0, __builtin_amdgcn_sdot4(w1, (int)KA1, acc[t], false), false);
      if (t >= 1 && t - 1 < 16) acc[t - 1] = __builtin_amdgcn_sdot4(w0, (int)KB0, __builtin_amdgcn_sdot4(w1, (int)KB1, acc[t - 1], false), false);
      if (t >= 2 && t - 2 < 16) acc[t - 2] = __builtin_amdgcn_sdot4(w0, (int)KC0, __builtin_amdgcn_sdot4(w1, (int)KC1, acc[t - 2], false), false);
      if (t >= 3 && t - 3 < 16) acc[t - 3] = __builtin_amdgcn_sdot4(w0, (int)KB0, __builtin_amdgcn_sdot4(w1, (int)KB1, acc[t - 3], false), false);
      if (t >= 4) acc[t - 4] = __builtin_amdgcn_sdot4(w0, (int)KA0, __builtin_amdgcn_sdot4(w1, (int)KA1, acc[t - 4], false), false);
    }
    const int coef = ch == 0 ? 3735 : (ch == 1 ? 19235 : 9798);  // raw R ends up where RGB2GRAY reads "B"
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const int f = acc[j] < 0 ? 0 : (acc[j] > 255 ? 255 : acc[j]);
      grey[j] += coef * f;
      const int y = 16 * wv + j;
      if (proc && ty0 + y < S && tx0 + px < S)
        proc[((size_t)(ty0 + y) * S + tx0 + px) * 3 + (2 - ch)] = (uint8_t)f;  // RGB2BGR (helpers.py:82)
    }
    __syncthreads();
  }

  // ---- C: grey, threshold (helpers.py:145-146), one 64-bit word per tile row
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const int y = 16 * wv + j;
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
  uint16_t* pnext[2]; // [P_cap] x 2: index of the next pair on the pair's border (double-buffered pointer jumping)
  uint16_t* pmin[2];  // [P_cap] x 2: smallest pair INDEX seen on the border so far (pairs are sorted: index order = key order)
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

// From the horizontal pair (pixel (y0, x0), background neighbour in direction t0 = 4 left / 0 right) follows
// the border -- the Moore-tracing step map, scanning the 8-neighbourhood counter-clockwise from the pixel it
// came from (OpenCV's icvFetchContour order) -- until it crosses the NEXT horizontal pair, whose key it
// returns.  With MOMENTS also the polygon sums of cv::moments (contourMoments) over the moves made on the
// way, in original pixel coordinates: every move of a border lies between two consecutive pair crossings, so
// the segment sums of a cycle's pairs add up to the contour's sums (exact integers, any order).
template <bool MOMENTS>
__device__ uint32_t trace_segment(const uint32_t* msk, int stride, int y0, int x0, int t0, long long* a) {
  int y = y0, x = x0, d = t0;
  long long a00 = 0, a10 = 0, a01 = 0;
  uint32_t next_key;
  for (;;) {
    const uint32_t nb = ring8(msk, stride, y, x);
    // scan s = d+1, d+2, ... : rotate so that bit 0 = direction d+1
    const uint32_t rot = ((nb | (nb << 8)) >> ((d + 1) & 7)) & 0xffu;
    const int k = rot ? __builtin_ctz(rot) : 8;  // background pixels crossed before the next border pixel
    // scan positions of the left (4) and right (0) neighbour; a pair is crossed when its position is < k.
    // (The pair the segment starts from sits at position 7 of its own first scan: only an isolated pixel,
    // k = 8, comes back to it.)
    const int p4 = (3 - d) & 7, p0 = (7 - d) & 7;
    const int first = p4 < p0 ? p4 : p0;
    if (first < k) {
      next_key = pair_key(y, x, p4 < p0 ? 0 : 1);
      break;
    }
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
  return next_key;
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
    L.pnext[0] = (uint16_t*)p;  p += sizeof(uint16_t) * P_cap;
    L.pnext[1] = (uint16_t*)p;  p += sizeof(uint16_t) * P_cap;
    L.pmin[0] = (uint16_t*)p;   p += sizeof(uint16_t) * P_cap;
    L.pmin[1] = (uint16_t*)p;   p += sizeof(uint16_t) * P_cap;
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
      if (a.n_contours) a.n_contours[img] = -1;
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

  // ---- 2: every pair walks to the NEXT pair of its border: the borders become linked lists (cycles) of pairs.
  // Work is bounded by the total border length, whatever the shapes.
  for (int i = tid; i < n_pairs; i += kBlobThreads) {
    const uint32_t key = L.pkey[i];
    const int y = (int)(key >> 1) / 1024, x = (int)(key >> 1) % 1024;
    const uint32_t nk = trace_segment<false>(L.msk, stride, y, x, (key & 1u) ? 0 : 4, nullptr);
    L.pnext[0][i] = (uint16_t)lower_bound_u32(L.pkey, n_pairs, nk);
    L.pmin[0][i] = (uint16_t)i;
  }
  __syncthreads();
  // ---- 2b: pointer jumping: after r rounds pmin[i] = smallest pair within 2^r steps; a cycle has <= n_pairs
  // pairs, and a minimum does not mind being met twice
  int cur = 0;
  for (int span = 1; span < n_pairs; span <<= 1) {
    const uint16_t* nx = L.pnext[cur];
    const uint16_t* mn = L.pmin[cur];
    uint16_t* nx2 = L.pnext[cur ^ 1];
    uint16_t* mn2 = L.pmin[cur ^ 1];
    for (int i = tid; i < n_pairs; i += kBlobThreads) {
      const int j = nx[i];
      mn2[i] = min(mn[i], mn[j]);
      nx2[i] = nx[j];
    }
    __syncthreads();
    cur ^= 1;
  }
  const uint16_t* pmin = L.pmin[cur];  // index of the border's first pair in raster order = Suzuki-Abe's start

  // ---- 3: contours = pairs that are their border's minimum, numbered in raster (= discovery) order
  const int pper = (n_pairs + kBlobThreads - 1) / kBlobThreads;
  const int p_lo = min(n_pairs, tid * pper), p_hi = min(n_pairs, p_lo + pper);
  int nst = 0;
  for (int i = p_lo; i < p_hi; i++) nst += pmin[i] == i ? 1 : 0;
  int n_cont = 0;
  int coff = block_scan_excl(nst, L.scan, &n_cont);
  if (n_cont > N_cap) {
    if (tid == 0) {
      a.status[img] = BLOB_ST_CAP_OVERFLOW_;
      a.counts[img] = 0;
      if (a.n_contours) a.n_contours[img] = -1;
    }
    return;
  }
  for (int i = p_lo; i < p_hi; i++)
    if (pmin[i] == i) L.ckey[coff++] = L.pkey[i];
  for (int c = tid; c < 3 * n_cont; c += kBlobThreads) L.ca[c] = 0;
  __syncthreads();

  // ---- 4a: polygon sums: every pair adds the sums of its segment to its contour (integer atomics: exact)
  for (int i = tid; i < n_pairs; i += kBlobThreads) {
    const uint32_t key = L.pkey[i];
    const int y = (int)(key >> 1) / 1024, x = (int)(key >> 1) % 1024;
    long long sg[3];
    trace_segment<true>(L.msk, stride, y, x, (key & 1u) ? 0 : 4, sg);
    const int c = lower_bound_u32(L.ckey, n_cont, L.pkey[pmin[i]]);
    unsigned long long* dst = (unsigned long long*)(L.ca + 3 * c);
    if (sg[0]) atomicAdd(dst, (unsigned long long)sg[0]);
    if (sg[1]) atomicAdd(dst + 1, (unsigned long long)sg[1]);
    if (sg[2]) atomicAdd(dst + 2, (unsigned long long)sg[2]);
  }
  // ---- 4b: per contour: the border met by walking left on the start row
  for (int c = tid; c < n_cont; c += kBlobThreads) {
    const uint32_t key = L.ckey[c];
    const int hole = key & 1u;
    const int y = (int)(key >> 1) / 1024, x = (int)(key >> 1) % 1024;
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
      left = lower_bound_u32(L.ckey, n_cont, L.pkey[pmin[pi]]);
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
  return sizeof(long long) * 3 * N_cap + sizeof(uint32_t) * SP * stride + (sizeof(uint32_t) + 4 * sizeof(uint16_t)) * P_cap +
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
