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

constexpr int kPCapSmall = 2048, kNCapSmall = 256;   // LDS tables of the common case: 3 workgroups per CU
constexpr int kPCapLarge = 8192, kNCapLarge = 1024;  // re-run of images that overflowed them

int find_blobs_dev_locked(mocap_ctx* ctx, int64_t n_frames, const uint8_t* d_images, int M_max, float* d_blobs,
                          int32_t* d_counts, int32_t* d_status, uint8_t* d_processed, int32_t* d_n_contours) {
  if (!ctx->img_C) return ctx->fail(MOCAP_E_NOCAMS, "mocap_set_image_params has not been called");
  if (n_frames < 0 || M_max < 1) return ctx->fail(MOCAP_E_ARG, "mocap_find_blobs: bad size argument");
  if (n_frames == 0) return MOCAP_OK;
  if (!d_images || !d_blobs || !d_counts || !d_status) return ctx->fail(MOCAP_E_ARG, "mocap_find_blobs: null buffer");
  BlobArgs a;
  a.n_images = n_frames * ctx->img_C;
  a.C = ctx->img_C;
  a.rows = ctx->img_rows;
  a.cols = ctx->img_cols;
  a.S = ctx->img_S;
  a.ay = ctx->img_ay;
  a.M_max = M_max;
  a.raw = d_images;
  a.map = (const uint32_t*)ctx->img_map.ptr;
  a.rot = (const int32_t*)ctx->img_rot.ptr;
  const size_t mask_bytes = (size_t)a.n_images * a.S * ((a.S + 63) / 64) * 8;
  if (ctx->img_mask.reserve(mask_bytes)) return ctx->fail(MOCAP_E_HIP, "hipMalloc(mask, %zu B) failed", mask_bytes);
  a.mask = (unsigned long long*)ctx->img_mask.ptr;
  a.processed = d_processed;
  a.blobs = d_blobs;
  a.counts = d_counts;
  a.status = d_status;
  a.n_contours = d_n_contours;
  HIP_TRY(ctx, launch_blob_mask(a, ctx->stream));
  HIP_TRY(ctx, launch_blob_contours(a, kPCapSmall, kNCapSmall, 0, ctx->stream));
  // images whose border tables overflowed run again with the largest tables LDS holds; the launch is
  // a no-op (one status read per workgroup) for every other image
  HIP_TRY(ctx, launch_blob_contours(a, kPCapLarge, kNCapLarge, 1, ctx->stream));
  return MOCAP_OK;
}

}  // namespace

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
  if (S > 1024) return ctx->fail(MOCAP_E_LIMIT, "frame edge %d exceeds 1024", S);
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
  if (ctx->img_map.reserve(map.size() * sizeof(uint32_t)) || ctx->img_rot.reserve(C * sizeof(int32_t)))
    return ctx->fail(MOCAP_E_HIP, "hipMalloc(undistortion maps) failed");
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
  return find_blobs_dev_locked(ctx, n_frames, d_images, M_max, d_blobs, d_counts, d_status, d_processed, nullptr);
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
