// blob_capi.hip -- C ABI of the blob-extraction stage (include/mocap_core.h, "before the path"):
// mocap_set_image_params builds the frame-invariant undistortion maps once per camera set (the
// reference rebuilds them inside cv.undistort for every frame of every camera, helpers.py:73),
// mocap_find_blobs* run the two kernels of blob_kernels.hip.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/mocap_core.h"
#include "ctx.hpp"

using namespace mocap;

#define HIP_TRY(ctx, expr)                                     \
  do {                                                         \
    hipError_t e__ = (expr);                                   \
    if (e__ != hipSuccess) return (ctx)->hip_fail(e__, #expr); \
  } while (0)

namespace {

// cv::invert of a 3x3 CV_64F matrix (closed form, core/src/lapack.cpp), row-major
void invert3(const double* S, double* t) {
  double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
  d = 1.0 / d;
  t[0] = (S[4] * S[8] - S[5] * S[7]) * d;
  t[1] = (S[2] * S[7] - S[1] * S[8]) * d;
  t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
  t[3] = (S[5] * S[6] - S[3] * S[8]) * d;
  t[4] = (S[0] * S[8] - S[2] * S[6]) * d;
  t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
  t[6] = (S[3] * S[7] - S[4] * S[6]) * d;
  t[7] = (S[1] * S[6] - S[0] * S[7]) * d;
  t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
}

// The CV_16SC2 + CV_16UC1 map cv::undistort builds (imgproc/undistort.dispatch.cpp: stripes of
// (1 << 12) / cols rows with the principal point shifted per stripe, initUndistortRectifyMap's scalar
// loop with R = I, newCameraMatrix = cameraMatrix, 5 distortion coefficients k1 k2 p1 p2 k3),
// packed for the kernel: fx | fy << 5 | (sx + 1) << 10 | (sy + 1) << 21, sx field 2047 = outside.
void build_undistort_map(const double* K, const double* dist, int S, uint32_t* out) {
  const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3], k3 = dist[4];
  const double fx = K[0], fy = K[4], u0 = K[2], v0 = K[5];
  int stripe0 = (1 << 12) / (S > 1 ? S : 1);
  if (stripe0 < 1) stripe0 = 1;
  if (stripe0 > S) stripe0 = S;
  for (int y0 = 0; y0 < S; y0 += stripe0) {
    const int n = stripe0 < S - y0 ? stripe0 : S - y0;
    double Ar[9];
    memcpy(Ar, K, sizeof Ar);
    Ar[5] = v0 - y0;
    double ir[9];
    invert3(Ar, ir);
    for (int i = 0; i < n; i++) {
      double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
      uint32_t* row = out + (size_t)(y0 + i) * S;
      for (int j = 0; j < S; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
        const double w = 1. / _w, x = _x * w, y = _y * w;
        const double x2 = x * x, y2 = y * y;
        const double r2 = x2 + y2, _2xy = 2 * x * y;
        const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((0 * r2 + 0) * r2 + 0) * r2);
        const double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + 0 * r2 + 0 * r2 * r2);
        const double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + 0 * r2 + 0 * r2 * r2);
        const double u = fx * 1.0 * xd + u0;
        const double v = fy * 1.0 * yd + v0;
        // saturate_cast<int>(double) = cvRound = lrint (round half to even); the map stores shorts
        const long iu = lrint(u * 32.0), iv = lrint(v * 32.0);
        long sx = iu >> 5, sy = iv >> 5;
        sx = (long)(short)sx;
        sy = (long)(short)sy;
        const uint32_t fxq = (uint32_t)(iu & 31), fyq = (uint32_t)(iv & 31);
        // BORDER_CONSTANT: every tap outside -> 0 (remapBilinear's early-out); partial overlap is
        // handled tap by tap in the kernel
        const bool outside = sx >= S || sx + 1 < 0 || sy >= S || sy + 1 < 0;
        row[j] = outside ? (2047u << 10) : (fxq | fyq << 5 | (uint32_t)(sx + 1) << 10 | (uint32_t)(sy + 1) << 21);
      }
    }
  }
}

constexpr int kPCapSmall = 1024, kNCapSmall = 128;   // LDS tables of the common case (29 KB at 320 px): 5 workgroups per CU
constexpr int kPCapLarge = 8192, kNCapLarge = 1024;  // re-run of images that overflowed them

int find_blobs_dev_locked(mocap_ctx* ctx, int64_t n_frames, const uint8_t* d_images, int M_max, float* d_blobs,
                          int32_t* d_counts, int32_t* d_status, uint8_t* d_processed, int32_t* d_n_contours) {
  if (!ctx->img_C) return ctx->fail(MOCAP_E_NOCAMS, "mocap_set_image_params has not been called");
  if (n_frames < 0 || M_max < 1) return ctx->fail(MOCAP_E_ARG, "mocap_find_blobs: bad size argument");
  if (n_frames == 0) return MOCAP_OK;
  if (!d_images || !d_blobs || !d_counts || !d_status) return ctx->fail(MOCAP_E_ARG, "mocap_find_blobs: null buffer");
  // frame sets are processed in chunks: a chunk's raw frames (230 KB per image at 240 x 320) are read by the
  // activity pass and then by the mask kernel's gather while they are still in the 256 MB Infinity Cache
  const int C = ctx->img_C, S = ctx->img_S;
  const int64_t chunk_frames = (2048 / C) > 0 ? 2048 / C : 1;
  const int64_t chunk_images = chunk_frames * C;
  const int64_t want = n_frames * C < chunk_images ? n_frames * C : chunk_images;
  const int bands = (ctx->img_rows + 16 + kSquareRows - 1) / kSquareRows, segs = ctx->img_cols * 3 / 16;
  if (ctx->img_act.reserve((size_t)want * bands * segs * 2)) return ctx->fail(MOCAP_E_HIP, "hipMalloc(activity map) failed");
  const size_t words = (size_t)(S + 63) / 64;
  if (ctx->img_mask.reserve((size_t)n_frames * C * S * words * 8)) return ctx->fail(MOCAP_E_HIP, "hipMalloc(mask) failed");
  for (int64_t f0 = 0; f0 < n_frames; f0 += chunk_frames) {
    const int64_t nf = n_frames - f0 < chunk_frames ? n_frames - f0 : chunk_frames;
    const size_t i0 = (size_t)f0 * C;
    BlobArgs a;
    a.n_images = nf * C;
    a.img_base = (int64_t)i0;
    a.C = C;
    a.rows = ctx->img_rows;
    a.cols = ctx->img_cols;
    a.S = S;
    a.ay = ctx->img_ay;
    a.M_max = M_max;
    a.raw = d_images + i0 * ctx->img_rows * ctx->img_cols * 3;
    a.rot = (const int32_t*)ctx->img_rot.ptr;
    a.gather = (const uint32_t*)ctx->img_tiles.ptr;
    a.fix_rec = (const uint32_t*)ctx->img_fix.ptr;
    a.fix_off = (const int32_t*)ctx->img_fixidx.ptr;
    a.fix_cnt = a.fix_off + ctx->img_n_lt;
    a.cam_lens = (const int32_t*)ctx->img_lens.ptr;
    a.activity = (uint8_t*)ctx->img_act.ptr;
    a.tile_box = (const int16_t*)ctx->img_box.ptr;
    a.tile_zero = (const uint8_t*)ctx->img_zero.ptr;
    a.skip_dark = ctx->blob_skip_dark;
    a.mask = (unsigned long long*)ctx->img_mask.ptr + i0 * S * words;
    a.processed = d_processed ? d_processed + i0 * S * S * 3 : nullptr;
    a.blobs = d_blobs + i0 * M_max * 2;
    a.counts = d_counts + i0;
    a.status = d_status + i0;
    a.n_contours = d_n_contours ? d_n_contours + i0 : nullptr;
    if (a.skip_dark == 1 && !a.processed) HIP_TRY(ctx, launch_blob_activity(a, ctx->stream));  // the map's only reader
    HIP_TRY(ctx, launch_blob_mask(a, ctx->stream));
  }
  // contours of the whole batch at once (one workgroup per image, reads only the 1-bit masks)
  BlobArgs a;
  memset(&a, 0, sizeof a);
  a.n_images = n_frames * C;
  a.C = C;
  a.S = S;
  a.M_max = M_max;
  a.mask = (unsigned long long*)ctx->img_mask.ptr;
  a.blobs = d_blobs;
  a.counts = d_counts;
  a.status = d_status;
  a.n_contours = d_n_contours;
  // table sizes that fit LDS next to the padded mask of this frame size (the mask grows with S^2 / 8)
  int p_small = kPCapSmall, n_small = kNCapSmall, p_large = kPCapLarge, n_large = kNCapLarge;
  const size_t lds_cap = 160 * 1024;
  while (blob_contour_lds_bytes(S, p_large, n_large) > lds_cap && p_large > 512) {
    p_large /= 2;
    n_large /= 2;
  }
  while (blob_contour_lds_bytes(S, p_small, n_small) > lds_cap / 2 && p_small > 256) {
    p_small /= 2;
    n_small /= 2;
  }
  if (blob_contour_lds_bytes(S, p_large, n_large) > lds_cap)
    return ctx->fail(MOCAP_E_LIMIT, "frame edge %d: the contour tables do not fit LDS", S);
  HIP_TRY(ctx, launch_blob_contours(a, p_small, n_small, 0, ctx->stream));
  // images whose border tables overflowed run again with the largest tables LDS holds; the launch is
  // a no-op (one status read per workgroup) for every other image
  HIP_TRY(ctx, launch_blob_contours(a, p_large, n_large, 1, ctx->stream));
  return MOCAP_OK;
}

}  // namespace

// internal (ctx.hpp): the blob stage enqueued on device pointers, context lock held by the caller (mocap_track_frame_images)
int mocap_blob_stage_locked(mocap_ctx* ctx, int64_t n_frames, const uint8_t* d_images, int M_max, float* d_blobs,
                            int32_t* d_counts, int32_t* d_status) {
  return find_blobs_dev_locked(ctx, n_frames, d_images, M_max, d_blobs, d_counts, d_status, nullptr, nullptr);
}

extern "C" int mocap_set_image_params(mocap_ctx* ctx, int C, int rows, int cols, const double* K, const double* dist,
                                      const int32_t* rotation) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (C < 1 || C > kMaxCameras || rows < 1 || cols < 1 || !K || !dist)
    return ctx->fail(MOCAP_E_ARG, "mocap_set_image_params: bad argument");
  // make_square (helpers.py:507-523) only works for landscape frames with >= 8 padding rows on both
  // sides (it raises otherwise): same domain here
  const int S = cols > rows ? cols : rows;
  const int ay = (S - rows) / 2;
  if (cols != S || ay < 8 || S - ay - rows < 8)
    return ctx->fail(MOCAP_E_ARG, "frames must be landscape with >= 8 rows of square padding (reference make_square)");
  if (S > kBlobMaxEdge) return ctx->fail(MOCAP_E_LIMIT, "frame edge %d exceeds 832 (contour tables + padded mask must fit 160 KB of LDS)", S);
  if (cols % 16) return ctx->fail(MOCAP_E_ARG, "frame width must be a multiple of 16");
  std::vector<int32_t> rot(C, 0);
  for (int c = 0; c < C; c++) {
    const int r = rotation ? ((rotation[c] % 4) + 4) % 4 : 0;
    if (r != 0 && r != 2)
      return ctx->fail(MOCAP_E_ARG, "camera %d: rotation %d turns a landscape frame to portrait (reference make_square raises)", c, r);
    rot[c] = r;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // a queued batch may still read the old maps
  std::vector<uint32_t> map((size_t)C * S * S);
  for (int c = 0; c < C; c++) {
    // cameras sharing intrinsics share the map (the reference's camera-params.json repeats one entry)
    int same = -1;
    for (int p = 0; p < c && same < 0; p++)
      if (!memcmp(K + 9 * p, K + 9 * c, 9 * sizeof(double)) && !memcmp(dist + 5 * p, dist + 5 * c, 5 * sizeof(double))) same = p;
    if (same >= 0)
      memcpy(map.data() + (size_t)c * S * S, map.data() + (size_t)same * S * S, sizeof(uint32_t) * S * S);
    else
      build_undistort_map(K + 9 * c, dist + 5 * c, S, map.data() + (size_t)c * S * S);
  }
  // gather tables of the mask kernel: per distinct (lens, rotation) and 64 x 64 tile, for each pixel of the tile's
  // 76 x 76 region (reflect-101 of cv::GaussianBlur / cv::filter2D applied here, once) where its four bilinear taps
  // sit IN THE RAW FRAME.  np.rot90 and make_square (helpers.py:71-72, 507-523) are a change of coordinates plus a
  // per-row scale: squared pixel (Y, X) = raw pixel (r, X) [rot 0] or (rows-1-r, cols-1-X) [rot 180], r = Y - ay
  // clamped to the frame, times scale/8 (8 inside the frame, 7..0 on the 8 feathered rows above and below, nothing
  // beyond).  Interior pixels (all four taps inside the frame) become a plain offset + fractions; for a rotated camera
  // the pair starts at the memory-first pixel and the fractions are swapped (32 - f; f = 0 keeps its pixel first).
  const int tiles = (S + kBlobTile - 1) / kBlobTile;
  const int RW = kBlobTile + 2 * kBlobHalo;
  const int row_bytes = cols * 3;
  const int64_t image_bytes = (int64_t)rows * row_bytes;
  std::vector<int32_t> lens(C, 0);
  std::vector<int> lens_cam;  // first camera of each distinct (lens, rotation)
  for (int c = 0; c < C; c++) {
    int same = -1;
    for (size_t l = 0; l < lens_cam.size() && same < 0; l++)
      if (rot[lens_cam[l]] == rot[c] &&
          !memcmp(map.data() + (size_t)lens_cam[l] * S * S, map.data() + (size_t)c * S * S, sizeof(uint32_t) * S * S)) same = (int)l;
    if (same < 0) {
      same = (int)lens_cam.size();
      lens_cam.push_back(c);
    }
    lens[c] = same;
  }
  const size_t n_lt = lens_cam.size() * tiles * tiles;
  constexpr uint32_t kSentinel = 0x3fffffu;
  std::vector<uint32_t> gather(n_lt * (size_t)kBlobGather, kSentinel);
  std::vector<uint32_t> fix;            // 5 words per record
  std::vector<int32_t> fixidx(2 * n_lt, 0);  // [n_lt] first record, [n_lt] count
  std::vector<int16_t> box(n_lt * 4);
  std::vector<uint8_t> zero(n_lt, 0);
  auto refl = [S](int i) {
    i = i < 0 ? -i : i;
    i = i >= S ? 2 * (S - 1) - i : i;
    return i < 0 ? 0 : (i >= S ? S - 1 : i);
  };
  struct Tap {
    bool in_area;  // inside the rows the squared frame fills (frame + feather) and its columns: takes part in the activity box
    int scale;     // 0 = zero pixel
    int64_t off;   // raw byte offset of the pixel
    int r_raw, c_raw;
  };
  for (size_t l = 0; l < lens_cam.size(); l++) {
    const int rotc = rot[lens_cam[l]];
    auto tap_of = [&](int Y, int X) {
      Tap t{false, 0, 0, 0, 0};
      if (X < 0 || X >= S || Y < ay - 8 || Y >= ay + rows + 8) return t;
      t.in_area = true;
      int r = Y - ay, sc = 8;
      if (r < 0) {
        sc = 8 + r;
        r = 0;
      } else if (r >= rows) {
        sc = 7 - (r - rows);
        r = rows - 1;
      }
      t.scale = sc;
      t.r_raw = rotc ? rows - 1 - r : r;
      t.c_raw = rotc ? cols - 1 - X : X;
      t.off = ((int64_t)t.r_raw * cols + t.c_raw) * 3;
      return t;
    };
    for (int t = 0; t < tiles * tiles; t++) {
      const size_t lt = l * tiles * tiles + t;
      const int ty0 = (t / tiles) * kBlobTile, tx0 = (t % tiles) * kBlobTile;
      uint32_t* g = gather.data() + lt * (size_t)kBlobGather;
      fixidx[lt] = (int32_t)(fix.size() / 5);
      int b0 = 1 << 20, b1 = -1, s0 = 1 << 20, s1 = -1;
      bool z = false;
      for (int vy = 0; vy < RW; vy++)
        for (int vx = 0; vx < RW; vx++) {
          const uint32_t m = map[((size_t)lens_cam[l] * S + refl(ty0 - kBlobHalo + vy)) * S + refl(tx0 - kBlobHalo + vx)];
          const int sxp = (m >> 10) & 2047;
          const int idx = vy * RW + vx;
          if (sxp == 2047) {  // every tap outside the squared frame (cv::remap BORDER_CONSTANT 0)
            z = true;
            continue;       // g[idx] stays the sentinel = zero pixel
          }
          const int sx = sxp - 1, sy = (int)(m >> 21) - 1;  // -1 .. S-1
          const uint32_t fx = m & 31u, fy = (m >> 5) & 31u;
          const Tap tp[4] = {tap_of(sy, sx), tap_of(sy, sx + 1), tap_of(sy + 1, sx), tap_of(sy + 1, sx + 1)};
          // activity box of the tile, in the squared frame's bands and segments
          for (int k = 0; k < 4; k++) {
            const int Y = sy + (k >> 1), X = sx + (k & 1);
            if (!tp[k].in_area) {
              z = true;
              continue;
            }
            const int bnd = (Y - (ay - 8)) / kSquareRows, sa = 3 * X / 16, sb = (3 * X + 2) / 16;
            b0 = bnd < b0 ? bnd : b0;
            b1 = bnd > b1 ? bnd : b1;
            s0 = sa < s0 ? sa : s0;
            s1 = sb > s1 ? sb : s1;
          }
          const bool all_zero = !tp[0].scale && !tp[1].scale && !tp[2].scale && !tp[3].scale;
          if (all_zero) continue;  // sentinel
          bool plain = tp[0].scale == 8 && tp[1].scale == 8 && tp[2].scale == 8 && tp[3].scale == 8 &&
                       tp[2].r_raw - tp[0].r_raw == (rotc ? -1 : 1);  // two different frame rows (not a clamped pair)
          uint32_t e = 0;
          if (plain) {
            // memory-first pixel of the pair and memory-upper row of the two
            uint32_t fxe = fx, fye = fy;
            int r_top = tp[0].r_raw, c_first = tp[0].c_raw;
            if (rotc) {
              if (fx) {
                c_first = tp[1].c_raw;  // = cols - 2 - sx: the pair's second pixel comes first in memory
                fxe = 32u - fx;
              }
              if (fy) {
                r_top = tp[2].r_raw;    // = rows - 2 - r
                fye = 32u - fy;
              }
            }
            const int64_t off = ((int64_t)r_top * cols + c_first) * 3;
            // both 8-byte loads (this row and the next one in memory) must stay inside the image
            if (off + row_bytes + 8 > image_bytes || off >= (int64_t)kSentinel) plain = false;
            else e = (uint32_t)off | fxe << 22 | fye << 27;
          }
          if (plain) {
            g[idx] = e;
          } else {
            fix.push_back((uint32_t)idx | fx << 16 | fy << 24);
            for (int k = 0; k < 4; k++) fix.push_back(tp[k].scale ? ((uint32_t)tp[k].off | (uint32_t)tp[k].scale << 22) : 0u);
          }
        }
      fixidx[n_lt + lt] = (int32_t)(fix.size() / 5) - fixidx[lt];
      int16_t* o = box.data() + lt * 4;
      if (b1 < 0) b0 = b1 = s0 = s1 = 0;  // every tap in the zero area: one (any) activity cell, the zero flag decides
      o[0] = (int16_t)b0;
      o[1] = (int16_t)b1;
      o[2] = (int16_t)s0;
      o[3] = (int16_t)s1;
      zero[lt] = z ? 1 : 0;
    }
  }
  if (fix.empty()) fix.assign(5, 0u);
  if (ctx->img_fix.reserve(fix.size() * sizeof(uint32_t)) || ctx->img_fixidx.reserve(fixidx.size() * sizeof(int32_t)))
    return ctx->fail(MOCAP_E_HIP, "hipMalloc(gather fix-up lists) failed");
  HIP_TRY(ctx, hipMemcpyAsync(ctx->img_fix.ptr, fix.data(), fix.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->img_fixidx.ptr, fixidx.data(), fixidx.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  ctx->img_n_lt = (int)n_lt;
  if (ctx->img_map.reserve(map.size() * sizeof(uint32_t)) || ctx->img_rot.reserve(C * sizeof(int32_t)) ||
      ctx->img_tiles.reserve(gather.size() * sizeof(uint32_t)) || ctx->img_lens.reserve(C * sizeof(int32_t)) ||
      ctx->img_box.reserve(box.size() * sizeof(int16_t)) || ctx->img_zero.reserve(zero.size()))
    return ctx->fail(MOCAP_E_HIP, "hipMalloc(undistortion maps) failed");
  HIP_TRY(ctx, hipMemcpyAsync(ctx->img_box.ptr, box.data(), box.size() * sizeof(int16_t), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->img_zero.ptr, zero.data(), zero.size(), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->img_tiles.ptr, gather.data(), gather.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->img_lens.ptr, lens.data(), C * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->img_map.ptr, map.data(), map.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->img_rot.ptr, rot.data(), C * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->img_C = C;
  ctx->img_rows = rows;
  ctx->img_cols = cols;
  ctx->img_S = S;
  ctx->img_ay = ay;
  return MOCAP_OK;
}

extern "C" int mocap_set_blob_options(mocap_ctx* ctx, int skip_dark_tiles) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  ctx->blob_skip_dark = skip_dark_tiles == 2 ? 2 : (skip_dark_tiles ? 1 : 0);
  return MOCAP_OK;
}

extern "C" int mocap_get_undistort_map(mocap_ctx* ctx, int camera, uint32_t* map) {
  if (!ctx || !map) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!ctx->img_C) return ctx->fail(MOCAP_E_NOCAMS, "mocap_set_image_params has not been called");
  if (camera < 0 || camera >= ctx->img_C) return ctx->fail(MOCAP_E_ARG, "camera index out of range");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t n = (size_t)ctx->img_S * ctx->img_S;
  HIP_TRY(ctx, hipMemcpy(map, (const uint32_t*)ctx->img_map.ptr + n * camera, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return MOCAP_OK;
}

extern "C" int mocap_find_blobs_dev(mocap_ctx* ctx, int64_t n_frames, const uint8_t* d_images, int M_max,
                                    float* d_blobs, int32_t* d_counts, int32_t* d_status, uint8_t* d_processed) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int rc = find_blobs_dev_locked(ctx, n_frames, d_images, M_max, d_blobs, d_counts, d_status, d_processed, nullptr);
  return rc ? rc : ctx->mark_enqueued();
}

extern "C" int mocap_find_blobs(mocap_ctx* ctx, int64_t n_frames, const uint8_t* images, int M_max, float* blobs,
                                int32_t* counts, int32_t* status, uint8_t* processed, int32_t* n_contours) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!ctx->img_C) return ctx->fail(MOCAP_E_NOCAMS, "mocap_set_image_params has not been called");
  if (n_frames < 0 || M_max < 1 || (n_frames > 0 && (!images || !blobs || !counts || !status)))
    return ctx->fail(MOCAP_E_ARG, "mocap_find_blobs: bad argument");
  if (n_frames == 0) return MOCAP_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t n_img = (size_t)n_frames * ctx->img_C;
  const size_t b_raw = n_img * ctx->img_rows * ctx->img_cols * 3, b_blobs = n_img * M_max * 2 * sizeof(float),
               b_i32 = n_img * sizeof(int32_t), b_proc = processed ? n_img * ctx->img_S * ctx->img_S * 3 : 0;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  DevBuf& s = ctx->img_stage;
  const size_t total = al(b_raw) + al(b_blobs) + 3 * al(b_i32) + al(b_proc);
  if (s.reserve(total)) return ctx->fail(MOCAP_E_HIP, "hipMalloc(%zu) failed", total);
  unsigned char* p = (unsigned char*)s.ptr;
  uint8_t* d_raw = p;                     p += al(b_raw);
  float* d_blobs = (float*)p;             p += al(b_blobs);
  int32_t* d_counts = (int32_t*)p;        p += al(b_i32);
  int32_t* d_status = (int32_t*)p;        p += al(b_i32);
  int32_t* d_ncont = (int32_t*)p;         p += al(b_i32);
  uint8_t* d_proc = processed ? p : nullptr;
  HIP_TRY(ctx, hipMemcpyAsync(d_raw, images, b_raw, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(d_blobs, 0, b_blobs, ctx->stream));
  int rc = find_blobs_dev_locked(ctx, n_frames, d_raw, M_max, d_blobs, d_counts, d_status, d_proc, d_ncont);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(blobs, d_blobs, b_blobs, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(counts, d_counts, b_i32, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(status, d_status, b_i32, hipMemcpyDeviceToHost, ctx->stream));
  if (n_contours) HIP_TRY(ctx, hipMemcpyAsync(n_contours, d_ncont, b_i32, hipMemcpyDeviceToHost, ctx->stream));
  if (processed) HIP_TRY(ctx, hipMemcpyAsync(processed, d_proc, b_proc, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MOCAP_OK;
}
