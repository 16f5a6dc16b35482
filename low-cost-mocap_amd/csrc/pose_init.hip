// pose_init.hip -- initial camera poses, the caller of bundle adjustment:
//   calculate_camera_pose   reference computer_code/api/index.py:229-270
//     cv.findFundamentalMat(p1, p2, cv.FM_RANSAC, 1, 0.99999)    index.py:246   RANSAC over 7-point models
//     cv.sfm.essentialFromFundamental / motionFromEssential     index.py:247-248
//     cheirality vote over the four (R, t) candidates            index.py:250-262 (triangulate_points x 4)
//     pose chaining                                              index.py:264-270
//
// OpenCV's RANSAC is a sequential loop whose iteration count shrinks whenever a better model appears.
// Here the hypotheses of a whole batch of iterations are built and scored in parallel, then the loop's
// bookkeeping (strictly-greater inlier count wins, RANSACUpdateNumIters) is replayed over the per-model
// inlier counts in iteration order -- the same model comes out as from the sequential loop.  The random
// subsets come from cv::RNG((uint64)-1) exactly as cv::RANSACPointSetRegistrator draws them (host side: the
// generator is a serial recurrence).
//   seven_point_kernel   one lane per RANSAC iteration: Hartley normalisation, the 7 x 9 system's null
//                        space by Gauss-Jordan with full pivoting (any basis yields the same matrices),
//                        the cubic det(l f1 + (1-l) f2) = 0, de-normalisation; up to 3 models per sample
//   score_kernel         one workgroup per model: symmetric point-line distance of every correspondence
//                        (double arithmetic, float result as in FMEstimatorCallback::computeError), inliers
//                        counted with ballots
// The four candidate poses are triangulated by the existing DLT kernel as four 2-camera sets in one launch.
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

constexpr int kSampleBatch = 128;  // RANSAC iterations evaluated per round trip

// ---------------------------------------------------------------------------------------- device
__device__ int real_cubic_roots(const double* c, double* x) {
  // real roots of c0 x^3 + c1 x^2 + c2 x + c3 (the root set of cv::solveCubic)
  if (c[0] == 0) {
    if (c[1] == 0) {
      if (c[2] == 0) return 0;
      x[0] = -c[3] / c[2];
      return 1;
    }
    double d = c[2] * c[2] - 4 * c[1] * c[3];
    if (d < 0) return 0;
    d = sqrt(d);
    x[0] = (-c[2] + d) / (2 * c[1]);
    x[1] = (-c[2] - d) / (2 * c[1]);
    return d > 0 ? 2 : 1;
  }
  const double a1 = c[1] / c[0], a2 = c[2] / c[0], a3 = c[3] / c[0];
  const double Q = (a1 * a1 - 3 * a2) / 9, R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) / 54;
  const double d = Q * Q * Q - R * R;
  const double pi = 3.14159265358979323846;
  if (d > 0) {
    const double theta = acos(R / sqrt(Q * Q * Q)), t0 = -2 * sqrt(Q);
    x[0] = t0 * cos(theta / 3) - a1 / 3;
    x[1] = t0 * cos((theta + 2 * pi) / 3) - a1 / 3;
    x[2] = t0 * cos((theta + 4 * pi) / 3) - a1 / 3;
    return 3;
  }
  if (d == 0) {
    const double e = -cbrt(R);
    x[0] = 2 * e - a1 / 3;
    x[1] = -e - a1 / 3;
    return 2;
  }
  double e = cbrt(sqrt(-d) + fabs(R));
  if (R > 0) e = -e;
  x[0] = (e + Q / e) - a1 / 3;
  return 1;
}

struct SevenArgs {
  int n_samples;
  const int32_t* idx;  // [n_samples][7]
  const float* p1;     // [N][2]
  const float* p2;
  double* F;           // [n_samples][3][9]
  int32_t* nF;         // [n_samples]
};

__global__ __launch_bounds__(64) void seven_point_kernel(SevenArgs a) {
  __shared__ double A[63 * 64];  // [row * 9 + col][lane]
  __shared__ int perm[9 * 64];
  const int lane = threadIdx.x, s = blockIdx.x * 64 + lane;
  if (s >= a.n_samples) return;
  double* M = A + lane;
  int* pm = perm + lane;
  double x0[7], y0[7], x1[7], y1[7];
  double m1x = 0, m1y = 0, m2x = 0, m2y = 0;
  for (int i = 0; i < 7; i++) {
    const int k = a.idx[s * 7 + i];
    x0[i] = a.p1[2 * k];
    y0[i] = a.p1[2 * k + 1];
    x1[i] = a.p2[2 * k];
    y1[i] = a.p2[2 * k + 1];
    m1x += x0[i];
    m1y += y0[i];
    m2x += x1[i];
    m2y += y1[i];
  }
  const double inv7 = 1. / 7;
  m1x *= inv7;
  m1y *= inv7;
  m2x *= inv7;
  m2y *= inv7;
  double sc1 = 0, sc2 = 0;
  for (int i = 0; i < 7; i++) {
    sc1 += sqrt((x0[i] - m1x) * (x0[i] - m1x) + (y0[i] - m1y) * (y0[i] - m1y));
    sc2 += sqrt((x1[i] - m2x) * (x1[i] - m2x) + (y1[i] - m2y) * (y1[i] - m2y));
  }
  sc1 *= inv7;
  sc2 *= inv7;
  a.nF[s] = 0;
  if (sc1 < 1.1920928955078125e-07 || sc2 < 1.1920928955078125e-07) return;  // FLT_EPSILON
  sc1 = sqrt(2.) / sc1;
  sc2 = sqrt(2.) / sc2;
  for (int i = 0; i < 7; i++) {
    const double u0 = (x0[i] - m1x) * sc1, v0 = (y0[i] - m1y) * sc1, u1 = (x1[i] - m2x) * sc2, v1 = (y1[i] - m2y) * sc2;
    double* r = M + i * 9 * 64;
    r[0] = u1 * u0;
    r[64] = u1 * v0;
    r[128] = u1;
    r[192] = v1 * u0;
    r[256] = v1 * v0;
    r[320] = v1;
    r[384] = u0;
    r[448] = v0;
    r[512] = 1;
  }
  for (int c = 0; c < 9; c++) pm[c * 64] = c;
  // Gauss-Jordan with full pivoting -> [I | B] up to the column permutation
  for (int k = 0; k < 7; k++) {
    int pr = k, pc = k;
    double best = -1;
    for (int r = k; r < 7; r++)
      for (int c = k; c < 9; c++) {
        const double v = fabs(M[(r * 9 + c) * 64]);
        if (v > best) {
          best = v;
          pr = r;
          pc = c;
        }
      }
    if (!(best > 0)) return;  // rank deficient sample: no model (degenerate subsets are filtered by checkSubset)
    if (pr != k)
      for (int c = 0; c < 9; c++) {
        const double t = M[(k * 9 + c) * 64];
        M[(k * 9 + c) * 64] = M[(pr * 9 + c) * 64];
        M[(pr * 9 + c) * 64] = t;
      }
    if (pc != k) {
      for (int r = 0; r < 7; r++) {
        const double t = M[(r * 9 + k) * 64];
        M[(r * 9 + k) * 64] = M[(r * 9 + pc) * 64];
        M[(r * 9 + pc) * 64] = t;
      }
      const int t = pm[k * 64];
      pm[k * 64] = pm[pc * 64];
      pm[pc * 64] = t;
    }
    const double ip = 1.0 / M[(k * 9 + k) * 64];
    for (int c = k; c < 9; c++) M[(k * 9 + c) * 64] *= ip;
    for (int r = 0; r < 7; r++) {
      if (r == k) continue;
      const double f = M[(r * 9 + k) * 64];
      if (f != 0)
        for (int c = k; c < 9; c++) M[(r * 9 + c) * 64] -= f * M[(k * 9 + c) * 64];
    }
  }
  double f1[9], f2[9];
  for (int c = 0; c < 9; c++) f1[c] = f2[c] = 0;
  for (int i = 0; i < 9; i++) {
    const int col = pm[i * 64];
    const double v1 = i < 7 ? -M[(i * 9 + 7) * 64] : (i == 7 ? 1.0 : 0.0);
    const double v2 = i < 7 ? -M[(i * 9 + 8) * 64] : (i == 8 ? 1.0 : 0.0);
#pragma unroll
    for (int c = 0; c < 9; c++)
      if (c == col) {
        f1[c] = v1;
        f2[c] = v2;
      }
  }
  // normalise the basis vectors (scale only: keeps the cubic's coefficients O(1))
  {
    double n1 = 0, n2 = 0;
    for (int c = 0; c < 9; c++) {
      n1 += f1[c] * f1[c];
      n2 += f2[c] * f2[c];
    }
    n1 = 1.0 / sqrt(n1);
    n2 = 1.0 / sqrt(n2);
    for (int c = 0; c < 9; c++) {
      f1[c] *= n1;
      f2[c] *= n2;
    }
  }
  for (int c = 0; c < 9; c++) f1[c] -= f2[c];
  double cf[4];
  {
    double t0 = f2[4] * f2[8] - f2[5] * f2[7], t1 = f2[3] * f2[8] - f2[5] * f2[6], t2 = f2[3] * f2[7] - f2[4] * f2[6];
    cf[3] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2;
    cf[2] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2 - f1[3] * (f2[1] * f2[8] - f2[2] * f2[7]) +
            f1[4] * (f2[0] * f2[8] - f2[2] * f2[6]) - f1[5] * (f2[0] * f2[7] - f2[1] * f2[6]) +
            f1[6] * (f2[1] * f2[5] - f2[2] * f2[4]) - f1[7] * (f2[0] * f2[5] - f2[2] * f2[3]) +
            f1[8] * (f2[0] * f2[4] - f2[1] * f2[3]);
    t0 = f1[4] * f1[8] - f1[5] * f1[7];
    t1 = f1[3] * f1[8] - f1[5] * f1[6];
    t2 = f1[3] * f1[7] - f1[4] * f1[6];
    cf[0] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2;
    cf[1] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2 - f2[3] * (f1[1] * f1[8] - f1[2] * f1[7]) +
            f2[4] * (f1[0] * f1[8] - f1[2] * f1[6]) - f2[5] * (f1[0] * f1[7] - f1[1] * f1[6]) +
            f2[6] * (f1[1] * f1[5] - f1[2] * f1[4]) - f2[7] * (f1[0] * f1[5] - f1[2] * f1[3]) +
            f2[8] * (f1[0] * f1[4] - f1[1] * f1[3]);
  }
  double roots[3];
  const int n = real_cubic_roots(cf, roots);
  double Fm[3][9];
  for (int k = 0; k < n; k++) {
    double lam = roots[k], mu = 1.0, g[9];
    const double sden = f1[8] * lam + f2[8];
    if (fabs(sden) > 2.220446049250313e-16) {
      mu = 1.0 / sden;
      lam *= mu;
      g[8] = 1.0;
    } else {
      g[8] = 0.0;
    }
    for (int i = 0; i < 8; i++) g[i] = f1[i] * lam + f2[i] * mu;
    // F = T2^T g T1,  T = [[s, 0, -s mx], [0, s, -s my], [0, 0, 1]]
    double h[9];  // g T1
    for (int r = 0; r < 3; r++) {
      h[3 * r] = g[3 * r] * sc1;
      h[3 * r + 1] = g[3 * r + 1] * sc1;
      h[3 * r + 2] = g[3 * r + 2] - sc1 * (g[3 * r] * m1x + g[3 * r + 1] * m1y);
    }
    double* Fo = Fm[k];
    for (int c = 0; c < 3; c++) {
      Fo[c] = sc2 * h[c];
      Fo[3 + c] = sc2 * h[3 + c];
      Fo[6 + c] = h[6 + c] - sc2 * (m2x * h[c] + m2y * h[3 + c]);
    }
    if (fabs(Fo[8]) > 1.1920928955078125e-07) {
      const double sc = 1.0 / Fo[8];
      for (int i = 0; i < 9; i++) Fo[i] *= sc;
    }
  }
  // order the models of one sample by F[0][0] (OpenCV's order depends on its SVD's basis of the null space)
  int ord[3] = {0, 1, 2};
  for (int i = 1; i < n; i++)
    for (int j = i; j > 0 && Fm[ord[j]][0] < Fm[ord[j - 1]][0]; j--) {
      const int t = ord[j];
      ord[j] = ord[j - 1];
      ord[j - 1] = t;
    }
  for (int k = 0; k < n; k++)
    for (int i = 0; i < 9; i++) a.F[((size_t)s * 3 + k) * 9 + i] = Fm[ord[k]][i];
  a.nF[s] = n;
}

struct ScoreArgs {
  int64_t N;
  const float* p1;
  const float* p2;
  const double* F;     // [n_models][9]
  const int32_t* nF;   // [n_models / 3] or null (then every model is scored)
  float t;             // (float)(thr * thr)
  int32_t* count;      // [n_models]
  uint8_t* mask;       // [N] or null: inlier mask of model 0
};

__device__ __forceinline__ bool fm_inlier(const double* f, float x1f, float y1f, float x2f, float y2f, float t) {
  const double x1 = x1f, y1 = y1f, x2 = x2f, y2 = y2f;
  double a = f[0] * x1 + f[1] * y1 + f[2], b = f[3] * x1 + f[4] * y1 + f[5], c = f[6] * x1 + f[7] * y1 + f[8];
  const double s2 = 1. / (a * a + b * b), d2 = x2 * a + y2 * b + c;
  a = f[0] * x2 + f[3] * y2 + f[6];
  b = f[1] * x2 + f[4] * y2 + f[7];
  c = f[2] * x2 + f[5] * y2 + f[8];
  const double s1 = 1. / (a * a + b * b), d1 = x1 * a + y1 * b + c;
  const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
  const float err = (float)(e1 > e2 || e2 != e2 ? e1 : e2);  // std::max(e1, e2): returns e1 unless e1 < e2
  return err <= t;
}

__global__ __launch_bounds__(256) void score_kernel(ScoreArgs a) {
  const int m = blockIdx.x;
  if (a.nF && (m % 3) >= a.nF[m / 3]) {
    if (threadIdx.x == 0) a.count[m] = 0;
    return;
  }
  __shared__ int total;
  if (threadIdx.x == 0) total = 0;
  __syncthreads();
  double f[9];
  for (int i = 0; i < 9; i++) f[i] = a.F[(size_t)m * 9 + i];
  int local = 0;
  for (int64_t i = threadIdx.x; i < a.N; i += 256) {
    const bool in = fm_inlier(f, a.p1[2 * i], a.p1[2 * i + 1], a.p2[2 * i], a.p2[2 * i + 1], a.t);
    local += in ? 1 : 0;
    if (a.mask && m == 0) a.mask[i] = in ? 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(&total, local);
  __syncthreads();
  if (threadIdx.x == 0) a.count[m] = total;
}

// ---------------------------------------------------------------------------------------- host
struct CvRng {  // cv::RNG
  uint64_t state;
  explicit CvRng(uint64_t s = 0xffffffffffffffffull) : state(s ? s : 0xffffffffull) {}
  uint32_t next() {
    state = (uint64_t)(uint32_t)state * 4164903690u + (uint32_t)(state >> 32);
    return (uint32_t)state;
  }
  int uniform(int a, int b) { return a == b ? a : (int)(next() % (uint32_t)(b - a) + a); }
};

bool have_collinear(const float* m, const int* idx, int count) {  // fundam.cpp haveCollinearPoints
  const int i = count - 1;
  const double xi = m[2 * idx[i]], yi = m[2 * idx[i] + 1];
  for (int j = 0; j < i; j++) {
    const double dx1 = m[2 * idx[j]] - xi, dy1 = m[2 * idx[j] + 1] - yi;
    for (int k = 0; k < j; k++) {
      const double dx2 = m[2 * idx[k]] - xi, dy2 = m[2 * idx[k] + 1] - yi;
      if (std::fabs(dx2 * dy1 - dy2 * dx1) <= 1.1920928955078125e-07 * (std::fabs(dx1) + std::fabs(dy1) + std::fabs(dx2) + std::fabs(dy2)))
        return true;
    }
  }
  return false;
}

bool get_subset(const float* m1, const float* m2, int count, CvRng& rng, int* idx) {  // ptsetreg.cpp getSubset
  for (int attempt = 0; attempt < 10000; attempt++) {
    for (int i = 0; i < 7; i++) {
      int v;
      bool dup;
      do {
        v = rng.uniform(0, count);
        dup = false;
        for (int j = 0; j < i; j++) dup |= idx[j] == v;
      } while (dup);
      idx[i] = v;
    }
    if (!have_collinear(m1, idx, 7) && !have_collinear(m2, idx, 7)) return true;
  }
  return false;
}

int update_num_iters(double p, double ep, int model_points, int max_iters) {  // RANSACUpdateNumIters
  p = p < 0 ? 0 : (p > 1 ? 1 : p);
  ep = ep < 0 ? 0 : (ep > 1 ? 1 : ep);
  double num = 1. - p < 2.2250738585072014e-308 ? 2.2250738585072014e-308 : 1. - p;
  double denom = 1. - std::pow(1. - ep, model_points);
  if (denom < 2.2250738585072014e-308) return 0;
  num = std::log(num);
  denom = std::log(denom);
  return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)std::lrint(num / denom);
}

// n correspondences (host float [n][2] each) -> best 7-point model; info: inliers, iterations, best iteration
int find_fundamental_locked(mocap_ctx* ctx, int64_t n, const float* p1, const float* p2, double thr, double conf,
                            int max_iters, double* F, uint8_t* mask, int32_t* info) {
  if (n < 15) return ctx->fail(MOCAP_E_ARG, "findFundamentalMat(FM_RANSAC) needs >= 15 points (OpenCV switches to LMedS below)");
  if (n > 0x7fffffff) return ctx->fail(MOCAP_E_LIMIT, "too many correspondences");
  if (thr <= 0) thr = 3;
  if (conf < 2.220446049250313e-16 || conf > 1 - 2.220446049250313e-16) conf = 0.99;
  if (max_iters < 1) max_iters = 1;
  const size_t b_pts = sizeof(float) * 2 * (size_t)n;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  DevBuf& s = ctx->scratch[0];
  const size_t total = 2 * al(b_pts) + al(sizeof(int32_t) * 7 * kSampleBatch) + al(sizeof(double) * 27 * kSampleBatch) +
                       2 * al(sizeof(int32_t) * 3 * kSampleBatch) + al((size_t)n);
  if (s.reserve(total)) return ctx->fail(MOCAP_E_HIP, "hipMalloc(%zu) failed", total);
  unsigned char* p = (unsigned char*)s.ptr;
  float* d_p1 = (float*)p;        p += al(b_pts);
  float* d_p2 = (float*)p;        p += al(b_pts);
  int32_t* d_idx = (int32_t*)p;   p += al(sizeof(int32_t) * 7 * kSampleBatch);
  double* d_F = (double*)p;       p += al(sizeof(double) * 27 * kSampleBatch);
  int32_t* d_nF = (int32_t*)p;    p += al(sizeof(int32_t) * 3 * kSampleBatch);
  int32_t* d_cnt = (int32_t*)p;   p += al(sizeof(int32_t) * 3 * kSampleBatch);
  uint8_t* d_mask = (uint8_t*)p;
  HIP_TRY(ctx, hipMemcpyAsync(d_p1, p1, b_pts, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_p2, p2, b_pts, hipMemcpyHostToDevice, ctx->stream));
  CvRng rng;
  int niters = max_iters, it = 0, max_good = 0, best_iter = -1;
  double best[9] = {0};
  std::vector<int32_t> idx(7 * kSampleBatch), nF(kSampleBatch), cnt(3 * kSampleBatch);
  std::vector<double> Fs(27 * kSampleBatch);
  const float t = (float)(thr * thr);
  bool exhausted = false;
  while (it < niters && !exhausted) {
    // subsets of the next batch of iterations (drawn in order: the generator is a serial recurrence)
    int nb = 0;
    while (nb < kSampleBatch && it + nb < niters) {
      if (!get_subset(p1, p2, (int)n, rng, idx.data() + 7 * nb)) {
        exhausted = true;  // getSubset gave up: the sequential loop stops here (returns false at iteration 0)
        break;
      }
      nb++;
    }
    if (nb == 0) break;
    HIP_TRY(ctx, hipMemcpyAsync(d_idx, idx.data(), sizeof(int32_t) * 7 * nb, hipMemcpyHostToDevice, ctx->stream));
    SevenArgs sa{nb, d_idx, d_p1, d_p2, d_F, d_nF};
    hipLaunchKernelGGL(seven_point_kernel, dim3((nb + 63) / 64), dim3(64), 0, ctx->stream, sa);
    ScoreArgs sc{n, d_p1, d_p2, d_F, d_nF, t, d_cnt, nullptr};
    hipLaunchKernelGGL(score_kernel, dim3(3 * nb), dim3(256), 0, ctx->stream, sc);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(nF.data(), d_nF, sizeof(int32_t) * nb, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(cnt.data(), d_cnt, sizeof(int32_t) * 3 * nb, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(Fs.data(), d_F, sizeof(double) * 27 * nb, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // replay of RANSACPointSetRegistrator::run over this batch
    for (int b = 0; b < nb && it < niters; b++, it++)
      for (int k = 0; k < nF[b]; k++) {
        const int good = cnt[3 * b + k];
        if (good > (max_good > 6 ? max_good : 6)) {
          max_good = good;
          best_iter = it;
          memcpy(best, Fs.data() + (size_t)(3 * b + k) * 9, sizeof best);
          niters = update_num_iters(conf, (double)(n - good) / (double)n, 7, niters);
        }
      }
  }
  if (info) {
    info[0] = max_good;
    info[1] = it;
    info[2] = best_iter;
  }
  if (max_good <= 0) return ctx->fail(MOCAP_E_NOCONV, "findFundamentalMat: no model found");
  memcpy(F, best, sizeof best);
  if (mask) {
    HIP_TRY(ctx, hipMemcpyAsync(d_F, best, sizeof best, hipMemcpyHostToDevice, ctx->stream));
    ScoreArgs sc{n, d_p1, d_p2, d_F, nullptr, t, d_cnt, d_mask};
    hipLaunchKernelGGL(score_kernel, dim3(1), dim3(256), 0, ctx->stream, sc);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(mask, d_mask, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  return MOCAP_OK;
}

// ---- 3x3 helpers (row-major)
void mul33(const double* A, const double* B, double* C) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}
void tr33(const double* A, double* T) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) T[3 * c + r] = A[3 * r + c];
}
double det33(const double* A) {
  return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
}

// SVD of a 3x3 matrix by one-sided Jacobi: E = U diag(w) V^T, w descending; a vanishing singular value gets
// the cross product of the other two left vectors (the essential matrix has w = (a, a, 0))
void svd33(const double* E, double* U, double* w, double* V) {
  double A[9], Vm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  memcpy(A, E, sizeof A);
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double al = 0, be = 0, ga = 0;
        for (int r = 0; r < 3; r++) {
          al += A[3 * r + p] * A[3 * r + p];
          be += A[3 * r + q] * A[3 * r + q];
          ga += A[3 * r + p] * A[3 * r + q];
        }
        off = std::fmax(off, std::fabs(ga) / std::sqrt(std::fmax(al * be, 1e-300)));
        if (ga == 0) continue;
        const double zeta = (be - al) / (2 * ga);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1 + zeta * zeta));
        const double c = 1 / std::sqrt(1 + t * t), s = c * t;
        for (int r = 0; r < 3; r++) {
          const double ap = A[3 * r + p], aq = A[3 * r + q];
          A[3 * r + p] = c * ap - s * aq;
          A[3 * r + q] = s * ap + c * aq;
          const double vp = Vm[3 * r + p], vq = Vm[3 * r + q];
          Vm[3 * r + p] = c * vp - s * vq;
          Vm[3 * r + q] = s * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  double nrm[3];
  int order[3] = {0, 1, 2};
  for (int c = 0; c < 3; c++) nrm[c] = std::sqrt(A[c] * A[c] + A[3 + c] * A[3 + c] + A[6 + c] * A[6 + c]);
  for (int i = 1; i < 3; i++)
    for (int j = i; j > 0 && nrm[order[j]] > nrm[order[j - 1]]; j--) std::swap(order[j], order[j - 1]);
  for (int k = 0; k < 3; k++) {
    const int c = order[k];
    w[k] = nrm[c];
    for (int r = 0; r < 3; r++) {
      V[3 * r + k] = Vm[3 * r + c];
      U[3 * r + k] = nrm[c] > 0 ? A[3 * r + c] / nrm[c] : 0;
    }
  }
  if (w[2] <= 1e-12 * w[0]) {  // rank 2: complete U with u0 x u1
    U[2] = U[3] * U[7] - U[6] * U[4];
    U[5] = U[6] * U[1] - U[0] * U[7];
    U[8] = U[0] * U[4] - U[3] * U[1];
  }
}

}  // namespace

extern "C" int mocap_find_fundamental(mocap_ctx* ctx, int64_t n, const float* p1, const float* p2, double threshold,
                                      double confidence, int max_iters, double* F, uint8_t* mask, int32_t* info) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!p1 || !p2 || !F) return ctx->fail(MOCAP_E_ARG, "mocap_find_fundamental: null buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  return find_fundamental_locked(ctx, n, p1, p2, threshold, confidence, max_iters, F, mask, info);
}

extern "C" int mocap_initial_poses(mocap_ctx* ctx, int C, int64_t N, const double* obs, const double* K,
                                   double threshold, double confidence, int max_iters, double* R, double* t,
                                   int32_t* info) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (C < 2 || C > kMaxCameras || N < 1 || !obs || !K || !R || !t) return ctx->fail(MOCAP_E_ARG, "mocap_initial_poses: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // camera 0 = (I, 0) (index.py:234-237)
  const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  memcpy(R, I3, sizeof I3);
  t[0] = t[1] = t[2] = 0;
  const double* K0 = K;       // the reference takes the intrinsics of cameras 0 and 1 for EVERY pair
  const double* K1 = K + 9;   // (index.py:247) and triangulate_points indexes them by position (helpers.py:305-307)
  const bool uniform = memcmp(K0, K1, 72) == 0;
  std::vector<float> p1, p2;
  std::vector<double> pair_obs, xyz;
  for (int ci = 0; ci + 1 < C; ci++) {
    // correspondences seen by both cameras of the pair, as float32 (index.py:241-244)
    p1.clear();
    p2.clear();
    for (int64_t i = 0; i < N; i++) {
      const double* a = obs + ((size_t)i * C + ci) * 2;
      const double* b = a + 2;
      if (a[0] == a[0] && a[1] == a[1] && b[0] == b[0] && b[1] == b[1]) {
        p1.push_back((float)a[0]);
        p1.push_back((float)a[1]);
        p2.push_back((float)b[0]);
        p2.push_back((float)b[1]);
      }
    }
    const int64_t n = (int64_t)p1.size() / 2;
    double F[9];
    int32_t finfo[3] = {0, 0, 0};
    int rc = find_fundamental_locked(ctx, n, p1.data(), p2.data(), threshold, confidence, max_iters, F, nullptr, finfo);
    if (rc) return rc;
    // E = K1^T F K0 (opencv_contrib sfm essentialFromFundamental = libmv), then libmv's MotionFromEssential
    double K1t[9], tmp[9], E[9];
    tr33(K1, K1t);
    mul33(K1t, F, tmp);
    mul33(tmp, K0, E);
    double U[9], w[3], V[9], Vt[9];
    svd33(E, U, w, V);
    tr33(V, Vt);
    if (det33(U) < 0)
      for (int r = 0; r < 3; r++) U[3 * r + 2] = -U[3 * r + 2];
    if (det33(Vt) < 0)
      for (int c = 0; c < 3; c++) Vt[6 + c] = -Vt[6 + c];
    const double W[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1}, Wt[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
    double Rc[4][9], tc[4][3];
    mul33(U, W, tmp);
    mul33(tmp, Vt, Rc[0]);
    memcpy(Rc[1], Rc[0], 72);
    mul33(U, Wt, tmp);
    mul33(tmp, Vt, Rc[2]);
    memcpy(Rc[3], Rc[2], 72);
    for (int k = 0; k < 4; k++)
      for (int r = 0; r < 3; r++) tc[k][r] = (k & 1) ? -U[3 * r + 2] : U[3 * r + 2];

    // cheirality vote (index.py:250-262): triangulate the pair under [previous GLOBAL pose, candidate]
    // -- the reference passes camera_poses[-1], not the identity -- as four 2-camera sets in one launch
    const double* Rp = R + 9 * ci;
    const double* tp = t + 3 * ci;
    const size_t nPq = uniform ? 24 : 48, nRT = 24;
    std::vector<double> tab(4 * (nPq + nRT) + 8, 0.0);
    auto proj = [](const double* Km, const double* Rm, const double* tm, double* P) {
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) {
          const double r0 = c < 3 ? Rm[c] : tm[0], r1 = c < 3 ? Rm[3 + c] : tm[1], r2 = c < 3 ? Rm[6 + c] : tm[2];
          P[r * 4 + c] = Km[r * 3 + 0] * r0 + Km[r * 3 + 1] * r1 + Km[r * 3 + 2] * r2;
        }
    };
    double* Pq = tab.data();
    double* RT = Pq + 4 * nPq;
    double* K4 = RT + 4 * nRT;
    for (int k = 0; k < 4; k++) {
      double* pq = Pq + k * nPq;
      if (uniform) {
        proj(K0, Rp, tp, pq);
        proj(K0, Rc[k], tc[k], pq + 12);
      } else {  // [j][cam][12]: intrinsics by compacted position j, pose by camera
        proj(K0, Rp, tp, pq);
        proj(K0, Rc[k], tc[k], pq + 12);
        proj(K1, Rc[k], tc[k], pq + 36);
      }
      double* rt = RT + k * nRT;
      memcpy(rt, Rp, 72);
      memcpy(rt + 9, tp, 24);
      memcpy(rt + 12, Rc[k], 72);
      memcpy(rt + 21, tc[k], 24);
    }
    K4[0] = K0[0]; K4[1] = K0[4]; K4[2] = K0[2]; K4[3] = K0[5];
    K4[4] = K1[0]; K4[5] = K1[4]; K4[6] = K1[2]; K4[7] = K1[5];
    pair_obs.resize((size_t)n * 4);
    for (int64_t i = 0; i < n; i++) {
      pair_obs[4 * i] = p1[2 * i];
      pair_obs[4 * i + 1] = p1[2 * i + 1];
      pair_obs[4 * i + 2] = p2[2 * i];
      pair_obs[4 * i + 3] = p2[2 * i + 1];
    }
    DevBuf& s = ctx->scratch[1];
    const size_t b_tab = tab.size() * 8, b_obs = pair_obs.size() * 8, b_xyz = (size_t)n * 4 * 3 * 8;
    if (s.reserve(b_tab + b_obs + b_xyz + 512)) return ctx->fail(MOCAP_E_HIP, "hipMalloc failed");
    double* d_tab = (double*)s.ptr;
    double* d_obs = d_tab + ((tab.size() + 31) & ~(size_t)31);
    double* d_xyz = d_obs + ((pair_obs.size() + 31) & ~(size_t)31);
    HIP_TRY(ctx, hipMemcpyAsync(d_tab, tab.data(), b_tab, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_obs, pair_obs.data(), b_obs, hipMemcpyHostToDevice, ctx->stream));
    TriArgs ta;
    ta.cv.C = 2;
    ta.cv.uniformK = uniform ? 1 : 0;
    ta.cv.f32_rounding = (ctx->flags & MOCAP_OPT_F32_ROUNDING) ? 1 : 0;
    ta.cv._pad = 0;
    ta.cv.Pq = d_tab;
    ta.cv.RT = d_tab + 4 * nPq;
    ta.cv.K4 = d_tab + 4 * nPq + 4 * nRT;
    ta.cv.F = nullptr;
    ta.N = n;
    ta.P = 4;
    ta.stride_Pq = nPq;
    ta.stride_RT = nRT;
    ta.obs = d_obs;
    ta.xyz = d_xyz;
    ta.err = nullptr;
    ta.xyz_in = nullptr;
    HIP_TRY(ctx, launch_triangulate(ta, ctx->stream));
    xyz.resize((size_t)n * 12);
    HIP_TRY(ctx, hipMemcpyAsync(xyz.data(), d_xyz, b_xyz, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    int64_t best_front = 0;
    int best_k = -1;
    for (int k = 0; k < 4; k++) {
      int64_t front = 0;
      const double* X = xyz.data() + (size_t)k * n * 3;
      for (int64_t i = 0; i < n; i++) {
        const double* x = X + 3 * i;
        front += x[2] > 0 ? 1 : 0;                                                   // index.py:256
        front += (Rc[k][2] * x[0] + Rc[k][5] * x[1] + Rc[k][8] * x[2]) > 0 ? 1 : 0;  // (R^T x)[2], index.py:254
      }
      if (front > best_front) {
        best_front = front;
        best_k = k;
      }
    }
    if (best_k < 0) return ctx->fail(MOCAP_E_NOCONV, "camera pair %d: no candidate pose has points in front (the reference raises here)", ci);
    // R = R_cand R_prev ; t = t_prev + R_prev t_cand (index.py:264-265)
    mul33(Rc[best_k], Rp, R + 9 * (ci + 1));
    for (int r = 0; r < 3; r++)
      t[3 * (ci + 1) + r] = tp[r] + Rp[3 * r] * tc[best_k][0] + Rp[3 * r + 1] * tc[best_k][1] + Rp[3 * r + 2] * tc[best_k][2];
    if (info) {
      info[4 * ci] = (int32_t)n;
      info[4 * ci + 1] = finfo[0];
      info[4 * ci + 2] = finfo[1];
      info[4 * ci + 3] = best_k;
    }
  }
  return MOCAP_OK;
}
