// ba_solve.hip -- bundle-adjustment entry points of include/mocap_core.h.
//
// Replaces bundle_adjustment (reference computer_code/api/helpers.py:244-290), i.e.
// scipy.optimize.least_squares(residual_function, x0, loss="cauchy", ftol=1e-2) with its default
// method="trf" (no bounds), 2-point finite-difference Jacobian and exact trust-region solver
// (scipy/optimize/_lsq/trf.py `trf_no_bounds`, _lsq/common.py `solve_lsq_trust_region`,
// `update_tr_radius`, `check_termination`, `evaluate_quadratic`).
//
// Division of labour: every residual evaluation, the (n+1)-way finite-difference batch, the
// robust scaling and the dense J^T J / J^T f contraction (FP64 MFMA) run on the GPU
// (tri_kernel.hip, ba_kernels.hip); the host keeps only the n x n trust-region algebra.
// SciPy factors J with an SVD (J = U S V^T); here the same quantities come from the symmetric
// eigen-decomposition of the Gram matrix: J^T J = V S^2 V^T and S U^T f = V^T (J^T f), which is
// all `solve_lsq_trust_region` uses.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <thread>
#include <cmath>
#include <limits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/mocap_core.h"
#include "ctx.hpp"
#include "tr_host.hpp"

using namespace mocap;

#define HIP_TRY(ctx, expr)                                  \
  do {                                                      \
    hipError_t e__ = (expr);                                \
    if (e__ != hipSuccess) return (ctx)->hip_fail(e__, #expr); \
  } while (0)

namespace {

constexpr double kEps = 2.220446049250313e-16;

// Eigen-decomposition of a symmetric n x n matrix (row-major, destroyed): Householder reduction to
// tridiagonal form, then implicit-shift QL (the classic EISPACK tred2 / tql2 pair, ~4 n^3 flop:
// 0.05 ms at n = 50 where the cyclic Jacobi used at first took 1.2 ms and was 88 % of a BA
// iteration).  Exactly-zero rows/columns (the dead focal parameters, helpers.py:267-270) stay
// decoupled: their Householder step is skipped.  V columns = eigenvectors, d = eigenvalues.
void sym_eig(int n, std::vector<double>& A, std::vector<double>& V, std::vector<double>& d) {
  V = A;
  d.assign(n, 0.0);
  std::vector<double> e(n, 0.0);
  auto v = [&](int i, int j) -> double& { return V[(size_t)i * n + j]; };
  // ---- tridiagonalisation
  for (int j = 0; j < n; j++) d[j] = v(n - 1, j);
  for (int i = n - 1; i > 0; i--) {
    double scale = 0.0, h = 0.0;
    for (int k = 0; k < i; k++) scale += std::fabs(d[k]);
    if (scale == 0.0) {
      e[i] = d[i - 1];
      for (int j = 0; j < i; j++) {
        d[j] = v(i - 1, j);
        v(i, j) = 0.0;
        v(j, i) = 0.0;
      }
    } else {
      for (int k = 0; k < i; k++) {
        d[k] /= scale;
        h += d[k] * d[k];
      }
      double f = d[i - 1];
      double g = std::sqrt(h);
      if (f > 0) g = -g;
      e[i] = scale * g;
      h -= f * g;
      d[i - 1] = f - g;
      for (int j = 0; j < i; j++) e[j] = 0.0;
      for (int j = 0; j < i; j++) {
        f = d[j];
        v(j, i) = f;
        g = e[j] + v(j, j) * f;
        for (int k = j + 1; k <= i - 1; k++) {
          g += v(k, j) * d[k];
          e[k] += v(k, j) * f;
        }
        e[j] = g;
      }
      f = 0.0;
      for (int j = 0; j < i; j++) {
        e[j] /= h;
        f += e[j] * d[j];
      }
      const double hh = f / (h + h);
      for (int j = 0; j < i; j++) e[j] -= hh * d[j];
      for (int j = 0; j < i; j++) {
        f = d[j];
        g = e[j];
        for (int k = j; k <= i - 1; k++) v(k, j) -= (f * e[k] + g * d[k]);
        d[j] = v(i - 1, j);
        v(i, j) = 0.0;
      }
    }
    d[i] = h;
  }
  // ---- accumulate the transformations
  for (int i = 0; i < n - 1; i++) {
    v(n - 1, i) = v(i, i);
    v(i, i) = 1.0;
    const double h = d[i + 1];
    if (h != 0.0) {
      for (int k = 0; k <= i; k++) d[k] = v(k, i + 1) / h;
      for (int j = 0; j <= i; j++) {
        double g = 0.0;
        for (int k = 0; k <= i; k++) g += v(k, i + 1) * v(k, j);
        for (int k = 0; k <= i; k++) v(k, j) -= g * d[k];
      }
    }
    for (int k = 0; k <= i; k++) v(k, i + 1) = 0.0;
  }
  for (int j = 0; j < n; j++) {
    d[j] = v(n - 1, j);
    v(n - 1, j) = 0.0;
  }
  v(n - 1, n - 1) = 1.0;
  e[0] = 0.0;
  // ---- implicit-shift QL on the tridiagonal (d, e)
  for (int i = 1; i < n; i++) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  double f = 0.0, tst1 = 0.0;
  for (int l = 0; l < n; l++) {
    tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
    int m = l;
    while (m < n) {
      if (std::fabs(e[m]) <= kEps * tst1) break;
      m++;
    }
    if (m == n) m = n - 1;
    if (m > l) {
      int iter = 0;
      do {
        iter++;
        double g = d[l];
        double p = (d[l + 1] - g) / (2.0 * e[l]);
        double r = std::hypot(p, 1.0);
        if (p < 0) r = -r;
        d[l] = e[l] / (p + r);
        d[l + 1] = e[l] * (p + r);
        const double dl1 = d[l + 1];
        double h = g - d[l];
        for (int i = l + 2; i < n; i++) d[i] -= h;
        f += h;
        p = d[m];
        double c = 1.0, c2 = c, c3 = c;
        const double el1 = e[l + 1];
        double s = 0.0, s2 = 0.0;
        for (int i = m - 1; i >= l; i--) {
          c3 = c2;
          c2 = c;
          s2 = s;
          g = c * e[i];
          h = c * p;
          r = std::hypot(p, e[i]);
          e[i + 1] = s * r;
          s = e[i] / r;
          c = p / r;
          p = c * d[i] - s * g;
          d[i + 1] = h + s * (c * g + s * d[i]);
          for (int k = 0; k < n; k++) {
            h = v(k, i + 1);
            v(k, i + 1) = s * v(k, i) + c * h;
            v(k, i) = c * v(k, i) - s * h;
          }
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1;
        e[l] = s * p;
        d[l] = c * p;
      } while (std::fabs(e[l]) > kEps * tst1 && iter < 60);
    }
    d[l] = d[l] + f;
    e[l] = 0.0;
  }
}

double norm2(const std::vector<double>& v) {
  double s = 0;
  for (double x : v) s += x * x;
  return std::sqrt(s);
}

// scipy/optimize/_lsq/common.py solve_lsq_trust_region, on (s, suf = s * U^T f, V).
// s sorted descending.  Returns step p, updates alpha.
void solve_tr(int n, int64_t m, const std::vector<double>& suf, const std::vector<double>& s,
              const std::vector<double>& V /* columns = right singular vectors, row-major n x n */, double Delta,
              double& alpha, std::vector<double>& p) {
  auto apply = [&](const std::vector<double>& coef) {  // p = -V coef
    p.assign(n, 0.0);
    for (int i = 0; i < n; i++) {
      double acc = 0;
      for (int k = 0; k < n; k++) acc += V[(size_t)i * n + k] * coef[k];
      p[i] = -acc;
    }
  };
  auto phi_and_derivative = [&](double a, double& phi, double& phi_prime) {
    double pn2 = 0, sum3 = 0;
    for (int k = 0; k < n; k++) {
      const double denom = s[k] * s[k] + a;
      const double q = suf[k] / denom;
      pn2 += q * q;
      sum3 += suf[k] * suf[k] / (denom * denom * denom);
    }
    const double p_norm = std::sqrt(pn2);
    phi = p_norm - Delta;
    phi_prime = -sum3 / p_norm;
  };
  bool full_rank = false;
  if (m >= n) {
    const double threshold = kEps * (double)m * s[0];
    full_rank = s[n - 1] > threshold;
  }
  std::vector<double> coef(n);
  if (full_rank) {
    for (int k = 0; k < n; k++) coef[k] = suf[k] / (s[k] * s[k]);  // uf / s
    apply(coef);
    if (norm2(p) <= Delta) {
      alpha = 0.0;
      return;
    }
  }
  double alpha_upper = norm2(suf) / Delta;
  double alpha_lower = 0.0;
  if (full_rank) {
    double phi, phi_prime;
    phi_and_derivative(0.0, phi, phi_prime);
    alpha_lower = -phi / phi_prime;
  }
  if (!full_rank && alpha == 0.0) alpha = std::max(0.001 * alpha_upper, std::sqrt(alpha_lower * alpha_upper));
  for (int it = 0; it < 10; it++) {
    if (alpha < alpha_lower || alpha > alpha_upper)
      alpha = std::max(0.001 * alpha_upper, std::sqrt(alpha_lower * alpha_upper));
    double phi, phi_prime;
    phi_and_derivative(alpha, phi, phi_prime);
    if (phi < 0) alpha_upper = alpha;
    const double ratio = phi / phi_prime;
    alpha_lower = std::max(alpha_lower, alpha - ratio);
    alpha -= (phi + Delta) * ratio / Delta;
    if (std::fabs(phi) < 0.01 * Delta) break;
  }
  for (int k = 0; k < n; k++) coef[k] = suf[k] / (s[k] * s[k] + alpha);
  apply(coef);
  const double pn = norm2(p);
  if (pn > 0)
    for (double& x : p) x *= Delta / pn;
}

// Workspace carved out of ctx->scratch[1..3].
struct BaWork {
  int C = 0, n = 0, NP = 0, ksplit = 1, uniformK = 1;
  int64_t N = 0, m = 0, m_pad = 0;
  size_t stride_Pq = 0, stride_RT = 0;
  double *d_x = nullptr, *d_params = nullptr, *d_hvec = nullptr, *d_Pq = nullptr, *d_RT = nullptr,
         *d_obs = nullptr, *d_r = nullptr, *d_Jaug = nullptr, *d_partial = nullptr, *d_G = nullptr,
         *d_cost = nullptr, *d_rho = nullptr;
  int32_t* d_valid = nullptr;
  std::vector<int32_t> valid;
  double *h_x = nullptr, *h_G = nullptr;  // pinned: parameter upload, [G | cost, finite] download
  // one-launch linearisation (ba_fused_kernel)
  bool fused = false;
  int chunks = 0, groups = 0;
  double *d_fpartial = nullptr, *d_fcost = nullptr, *d_fJaug = nullptr, *h_fout = nullptr, *h_mail = nullptr,
         *d_fmail = nullptr;
  bool prearm = false, armed = false;  // a linearisation kernel launched ahead of its base point (ba_fused_kernel)
  double armed_stamp = 0.0;
  // MOCAP_BA_PROFILE: where a slow linearisation spent its time.  A large gap between two consecutive clock reads of
  // the spinning host thread means the THREAD was not running (pre-empted / CFS-throttled), not that the GPU was slow.
  bool prof = false;
  int handovers = 0;         // points handed to a launched-ahead kernel in this solve
  int prof_relaunches = 0;   // linearisations repeated because a launched-ahead kernel had abandoned itself
  double prof_wait_max_ms = 0.0, prof_gap_max_ms = 0.0, prof_launch_max_ms = 0.0, prof_setup_ms = 0.0;
  int32_t* d_fcounters = nullptr;
};

int ba_setup(mocap_ctx* ctx, BaWork& w, int64_t N, const double* obs, int Pmax, bool want_jaug = false) {
  if (!ctx->C) return ctx->fail(MOCAP_E_NOCAMS, "mocap_set_cameras has not been called");
  if (N < 1 || !obs) return ctx->fail(MOCAP_E_ARG, "bundle adjustment: bad argument");
  const int C = ctx->C;
  w.C = C;
  w.n = 1 + 7 * (C - 1);
  w.NP = (w.n + 1 + 15) / 16 * 16;
  w.N = N;
  w.uniformK = ctx->cv.uniformK;
  w.stride_Pq = w.uniformK ? (size_t)12 * C : (size_t)12 * C * C;
  w.stride_RT = (size_t)12 * C;
  // a point enters the residual vector when >= 2 cameras see it (helpers.py:222-223, :207-208)
  w.valid.clear();
  for (int64_t i = 0; i < N; i++) {
    int v = 0;
    for (int c = 0; c < C; c++) {
      const double x = obs[(i * C + c) * 2], y = obs[(i * C + c) * 2 + 1];
      v += (std::isnan(x) || std::isnan(y)) ? 0 : 1;
    }
    if (v >= 2) w.valid.push_back((int32_t)i);
  }
  w.m = (int64_t)w.valid.size();
  w.m_pad = (w.m + 3) / 4 * 4;
  w.ksplit = ba_gram_ksplit(w.m_pad, w.NP);
  auto al = [](size_t x) { return (x + 31) / 32 * 32; };
  const size_t nd_x = al(w.n), nd_params = al((size_t)Pmax * w.n), nd_h = al(w.n),
               nd_Pq = al((size_t)Pmax * w.stride_Pq), nd_RT = al((size_t)Pmax * w.stride_RT),
               nd_obs = al((size_t)N * C * 2), nd_r = al((size_t)Pmax * N),
               nd_J = al((size_t)std::max<int64_t>(w.m_pad, 4) * w.NP),
               nd_part = al((size_t)w.ksplit * w.NP * w.NP), nd_G = al((size_t)w.NP * w.NP), nd_cost = 32,
               nd_rho = al((size_t)std::max<int64_t>(w.m, 1));
  const size_t total = nd_x + nd_params + nd_h + nd_Pq + nd_RT + nd_obs + nd_r + nd_J + nd_part + nd_G + nd_cost + nd_rho;
  void* zero_ptr = nullptr;
  size_t zero_words = 0;
  const auto ts0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!w.prof) return;
    static thread_local std::chrono::steady_clock::time_point last;
    const auto tn = std::chrono::steady_clock::now();
    const double d = std::chrono::duration<double, std::milli>(tn - (what ? last : ts0)).count();
    if (what && d > 0.3) fprintf(stderr, "[ba_setup] %s: %.3f ms\n", what, d);
    last = tn;
  };
  lap(nullptr);
  if (ctx->scratch[1].reserve(total * sizeof(double))) return ctx->fail(MOCAP_E_HIP, "hipMalloc(BA workspace) failed");
  if (ctx->scratch[2].reserve(sizeof(int32_t) * (size_t)std::max<int64_t>(w.m, 1)))
    return ctx->fail(MOCAP_E_HIP, "hipMalloc(BA valid list) failed");
  double* p = (double*)ctx->scratch[1].ptr;
  w.d_x = p;        p += nd_x;
  w.d_params = p;   p += nd_params;
  w.d_hvec = p;     p += nd_h;
  w.d_Pq = p;       p += nd_Pq;
  w.d_RT = p;       p += nd_RT;
  w.d_obs = p;      p += nd_obs;
  w.d_r = p;        p += nd_r;
  w.d_Jaug = p;     p += nd_J;
  w.d_partial = p;  p += nd_part;
  w.d_G = p;        p += nd_G;
  w.d_cost = p;     p += nd_cost;
  w.d_rho = p;
  w.d_valid = (int32_t*)ctx->scratch[2].ptr;
  lap("device workspace (hipMalloc)");
  // one launch per linearisation when the rig fits the fused kernel's LDS budget (MOCAP_BA_UNFUSED=1: the chain
  // of five launches it replaces, kept as the fallback for large rigs and for A/B measurements)
  // Any point count: the Gram partials are added by a 16-ary tree of last-arriver workgroups, so the serial tail
  // stays at two short sums (per iteration, fused vs chain: 41 vs 58 us at 1 000 points, 97-102 vs 141-147 us at
  // 16 000 points).  At 16 000 points BOTH schedules show a slow mode from one process to the next (fused 285-310 us in
  // 6 of 18 runs, chain 340 us in 1 of 4): a property of the box (host thread / PCIe placement), not of the schedule.
  w.fused = ba_fused_eligible(C, w.n, w.NP, w.uniformK != 0) && !getenv("MOCAP_BA_UNFUSED");
  size_t nd_fout = 0;
  if (w.fused) {
    w.chunks = (int)((N + 63) / 64);
    w.groups = ba_fused_groups(C);
    const size_t recs = ba_fused_records(w.chunks);
    const size_t nd_fp = al(recs * ((size_t)(w.n + 1) * (w.n + 2) / 2)), nd_fc = al(recs * 2),
                 nd_cnt = al(ba_fused_counters(w.chunks) / 2 + 1), nd_fJ = want_jaug ? al((size_t)N * w.NP) : 0, nd_mail = al(2 + 128);
    if (ctx->ba_fused.reserve((nd_fp + nd_fc + nd_cnt + nd_mail + nd_fJ) * sizeof(double)))
      return ctx->fail(MOCAP_E_HIP, "hipMalloc(BA fused workspace) failed");
    double* q = (double*)ctx->ba_fused.ptr;
    w.d_fpartial = q;  q += nd_fp;
    w.d_fcost = q;     q += nd_fc;
    w.d_fcounters = (int32_t*)q;  q += nd_cnt;
    w.d_fmail = q;     q += nd_mail;
    w.d_fJaug = want_jaug ? q : nullptr;
    zero_ptr = w.d_fcounters;  // (the kernel leaves its counters at zero; the mailbox mirror is re-tagged per launch)
    zero_words = (nd_cnt + nd_mail) * sizeof(double) / 4;
    nd_fout = al((size_t)w.NP * w.NP + 3) + al(8 * 19);  // + mailbox: 19 lines of {tag, 7 doubles} (n <= 127)
    // launch-ahead: the mailbox poll reads 16 lines of 7 parameters in one go; used where it was validated
    w.prearm = !getenv("MOCAP_BA_NO_PREARM") && w.n <= 112 && N <= 2048;
  }
  lap("fused workspace (hipMalloc)");
  const size_t pin_bytes = sizeof(double) * (nd_x + nd_G + nd_cost + nd_fout);
  if (pin_bytes > ctx->ba_pin_cap) {
    if (ctx->ba_pin) (void)hipHostFree(ctx->ba_pin);
    ctx->ba_pin = nullptr;
    ctx->ba_pin_cap = 0;
    // coherent (fine-grained): the fused kernel's completion stamp is polled by the host while the kernel runs
    HIP_TRY(ctx, hipHostMalloc(&ctx->ba_pin, pin_bytes, hipHostMallocCoherent));
    ctx->ba_pin_cap = pin_bytes;
  }
  if (!ctx->ba_event) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ba_event, hipEventDisableTiming));
  w.h_x = (double*)ctx->ba_pin;
  w.h_G = w.h_x + nd_x;
  w.h_fout = w.fused ? w.h_G + nd_G + nd_cost : nullptr;
  if (w.h_fout) {
    w.h_fout[(size_t)w.NP * w.NP + 2] = 0.0;  // neither a stamp (>= 1) nor an abandon mark (-stamp)
    w.h_mail = w.h_fout + al((size_t)w.NP * w.NP + 3);
    w.h_mail[0] = 0.0;
  }
  lap("pinned result buffer (hipHostMalloc)");
  // Stage-in.  Default: the inputs are copied into the context's own pinned staging buffer and pulled to the device by
  // ONE small kernel that also zeroes the arrival counters -- no copy engine, no pinning of the caller's pages, no
  // runtime blit kernel.  MOCAP_BA_STAGE=0 (hipMemcpyAsync from the caller's arrays + hipMemsetAsync, the round-2 path)
  // and =1 (pinned staging + hipMemcpyAsync) stay for A/B runs (scripts/diag_ba_stall.py).
  static const int stage_mode = getenv("MOCAP_BA_STAGE") ? atoi(getenv("MOCAP_BA_STAGE")) : 2;
  const size_t b_obs = sizeof(double) * (size_t)N * C * 2, b_valid = sizeof(int32_t) * (size_t)w.m;
  if (stage_mode == 0) {
    if (zero_words) HIP_TRY(ctx, hipMemsetAsync(zero_ptr, 0, zero_words * 4, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(w.d_obs, obs, b_obs, hipMemcpyHostToDevice, ctx->stream));
    if (w.m) HIP_TRY(ctx, hipMemcpyAsync(w.d_valid, w.valid.data(), b_valid, hipMemcpyHostToDevice, ctx->stream));
    lap("pageable copies + memset queued");
    return MOCAP_OK;
  }
  if (b_obs + b_valid > ctx->ba_stage_cap) {
    if (ctx->ba_stage) (void)hipHostFree(ctx->ba_stage);
    ctx->ba_stage = nullptr;
    ctx->ba_stage_cap = 0;
    const size_t want = std::max<size_t>((b_obs + b_valid) * 5 / 4, (size_t)1 << 20);
    HIP_TRY(ctx, hipHostMalloc(&ctx->ba_stage, want, hipHostMallocDefault));
    ctx->ba_stage_cap = want;
    lap("pinned staging buffer (hipHostMalloc)");
  }
  char* st = (char*)ctx->ba_stage;
  memcpy(st, obs, b_obs);
  if (w.m) memcpy(st + b_obs, w.valid.data(), b_valid);
  lap("host copy into the staging buffer");
  if (stage_mode == 1) {
    if (zero_words) HIP_TRY(ctx, hipMemsetAsync(zero_ptr, 0, zero_words * 4, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(w.d_obs, st, b_obs, hipMemcpyHostToDevice, ctx->stream));
    if (w.m) HIP_TRY(ctx, hipMemcpyAsync(w.d_valid, st + b_obs, b_valid, hipMemcpyHostToDevice, ctx->stream));
  } else {
    HIP_TRY(ctx, launch_ba_stage(st, w.d_obs, b_obs / 4, zero_ptr, zero_words, ctx->stream));
    if (w.m) HIP_TRY(ctx, launch_ba_stage(st + b_obs, w.d_valid, b_valid / 4, nullptr, 0, ctx->stream));
  }
  lap("stage-in queued");
  return MOCAP_OK;
}

// Wait for everything queued on the context's stream by polling an event: the LM loop waits twice per
// iteration on ~0.1 ms of GPU work, and a sleeping wait costs more than the work itself.
int ba_wait(mocap_ctx* ctx) {
  HIP_TRY(ctx, hipEventRecord(ctx->ba_event, ctx->stream));
  for (long spins = 0;; spins++) {
    const hipError_t e = hipEventQuery(ctx->ba_event);
    if (e == hipSuccess) return MOCAP_OK;
    if (e != hipErrorNotReady) return ctx->hip_fail(e, "hipEventQuery");
    if (spins > 2000000) {  // seconds of polling: something is badly stuck, fall back to a blocking wait
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      return MOCAP_OK;
    }
  }
}

// residuals for P parameter vectors already in w.d_params -> w.d_r [P][N]
int ba_eval_device(mocap_ctx* ctx, BaWork& w, int P, const double* params = nullptr, const double* fd_x = nullptr,
                   double rel_step = 0.0) {
  BaCamArgs ca;
  ca.C = w.C;
  ca.n = w.n;
  ca.P = P;
  ca.uniformK = w.uniformK;
  ca.params = fd_x ? nullptr : (params ? params : w.d_params);
  ca.x = fd_x;
  if (fd_x && w.n <= 64) {  // small rigs: the base point rides in the kernel arguments
    memcpy(ca.x_inline, fd_x, sizeof(double) * w.n);
    ca.x = nullptr;
  }
  ca.rel_step = rel_step;
  ca.hvec = w.d_hvec;
  ca.K = ctx->d_K9;
  ca.Pq = w.d_Pq;
  ca.RT = w.d_RT;
  ca.stride_Pq = w.stride_Pq;
  ca.stride_RT = w.stride_RT;
  HIP_TRY(ctx, launch_ba_build_cameras(ca, ctx->stream));
  TriArgs ta;
  ta.cv = ctx->cv;
  ta.cv.Pq = w.d_Pq;
  ta.cv.RT = w.d_RT;
  ta.N = w.N;
  ta.P = P;
  ta.stride_Pq = w.stride_Pq;
  ta.stride_RT = w.stride_RT;
  ta.obs = w.d_obs;
  ta.xyz = nullptr;
  ta.err = w.d_r;
  ta.xyz_in = nullptr;
  HIP_TRY(ctx, launch_triangulate(ta, ctx->stream));
  return MOCAP_OK;
}

// cost at x (host) -> cost, finite
int ba_cost_at(mocap_ctx* ctx, BaWork& w, const double* x, int f32, int cauchy, double& cost, bool& finite) {
  // zero-copy: the kernels read the parameter vector from, and write (cost, finite) to, pinned host memory
  // (device-visible, coherent) -- a copy-engine round trip costs more than the 2 us of work it would move
  memcpy(w.h_x, x, sizeof(double) * w.n);
  int rc = ba_eval_device(ctx, w, 1, w.h_x);
  if (rc) return rc;
  double* out = w.h_G + (w.d_cost - w.d_G);
  HIP_TRY(ctx, launch_ba_cost(w.d_r, w.d_valid, w.m, f32, cauchy, out, ctx->stream));
  rc = ba_wait(ctx);
  if (rc) return rc;
  cost = out[0];
  finite = out[1] != 0.0;
  return MOCAP_OK;
}

// linearise at x: G = [J|f]^T [J|f] (host copy, NP x NP), cost
int ba_fused_launch(mocap_ctx* ctx, BaWork& w, const double* x /* null: launched ahead, x comes by mailbox */, int f32,
                    int cauchy, double rel_step, double& stamp_out) {
  BaFusedArgs a;
  a.C = w.C;
  a.n = w.n;
  a.NP = w.NP;
  a.uniformK = w.uniformK;
  a.f32_rounding = ctx->cv.f32_rounding;
  a.f32_residuals = f32;
  a.use_cauchy = cauchy;
  a.chunks = w.chunks;
  a.groups = w.groups;
  static const int dbg = getenv("MOCAP_BA_DEBUG_STOP") ? atoi(getenv("MOCAP_BA_DEBUG_STOP")) : 0;
  a.debug_stop = dbg;
  a.N = w.N;
  a.rel_step = rel_step;
  ctx->ba_stamp += 1.0;
  a.stamp = stamp_out = ctx->ba_stamp;
  a.mailbox = x ? nullptr : w.h_mail;
  a.dev_mail = w.d_fmail;
  if (x) memcpy(a.x, x, sizeof(double) * w.n);
  a.K = ctx->d_K9;
  a.K4 = ctx->cv.K4;
  a.obs = w.d_obs;
  a.r = w.d_r;
  a.partial = w.d_fpartial;
  a.cost_part = w.d_fcost;
  a.counters = w.d_fcounters;
  a.Jaug_out = w.d_fJaug;
  a.out = w.h_fout;
  HIP_TRY(ctx, launch_ba_fused(a, ctx->stream));
  return MOCAP_OK;
}

// Tell a launched-ahead kernel that its base point will never come, and wait until the stream is empty.
int ba_disarm(mocap_ctx* ctx, BaWork& w) {
  if (!w.armed) return MOCAP_OK;
  w.armed = false;
  __atomic_thread_fence(__ATOMIC_RELEASE);
  *(volatile double*)w.h_mail = -1.0;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  *(volatile double*)w.h_mail = 0.0;  // the next launch-ahead must not read the quit tag before its point arrives
  return MOCAP_OK;
}
struct BaDisarmGuard {
  mocap_ctx* ctx;
  BaWork* w;
  ~BaDisarmGuard() { (void)ba_disarm(ctx, *w); }
};

// linearise at x: G = [J|f]^T [J|f] (host copy, NP x NP), cost
int ba_linearize_fused(mocap_ctx* ctx, BaWork& w, const double* x, int f32, int cauchy, double rel_step,
                       std::vector<double>& G, double& cost, bool* finite) {
  double stamp = 0.0;
  int rc;
  const auto t_launch0 = std::chrono::steady_clock::now();
  if (w.prearm) {
    // the kernel for THIS point was launched while the host was still computing the point (it is resident and
    // polls the mailbox): hand it x, then queue the next one behind it before waiting
    if (!w.armed) {
      rc = ba_fused_launch(ctx, w, nullptr, f32, cauchy, rel_step, w.armed_stamp);
      if (rc) return rc;
      w.armed = true;
    }
    stamp = w.armed_stamp;
    {  // test hook: MOCAP_BA_DEBUG_HOST_STALL="k:ms" holds the host up for ms before the k-th hand-over of a solve, long
       // enough (> 2 s) for the device watchdog to abandon the resident kernel (tests/test_gpu_ba.py)
      static const char* hook = getenv("MOCAP_BA_DEBUG_HOST_STALL");
      if (hook) {
        int k = 0, ms_ = 0;
        if (sscanf(hook, "%d:%d", &k, &ms_) == 2 && w.handovers == k) std::this_thread::sleep_for(std::chrono::milliseconds(ms_));
      }
      w.handovers++;
    }
    // mailbox = 64-byte lines {tag, 7 doubles of x}: data first, then the tag of the line
    for (int l = 0; l * 7 < w.n; l++) {
      volatile double* line = w.h_mail + 8 * l;
      for (int j = 0; j < 7 && l * 7 + j < w.n; j++) line[1 + j] = x[l * 7 + j];
      __atomic_thread_fence(__ATOMIC_RELEASE);
      line[0] = stamp;
    }
    w.armed = false;
    rc = ba_fused_launch(ctx, w, nullptr, f32, cauchy, rel_step, w.armed_stamp);
    if (rc) return rc;
    w.armed = true;
  } else {
    rc = ba_fused_launch(ctx, w, x, f32, cauchy, rel_step, stamp);
    if (rc) return rc;
  }
  if (w.prof)
    w.prof_launch_max_ms = std::max(w.prof_launch_max_ms, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_launch0).count());
  // the kernel's last workgroup stores G, the cost and then the stamp into this pinned buffer: spin on the stamp
  // (a few microseconds; an event query costs a driver call per poll)
  const size_t nG = (size_t)w.NP * w.NP;
  volatile double* done = w.h_fout + nG + 2;
  auto t0 = std::chrono::steady_clock::now();  // (restarted when the linearisation is launched again below)
  auto t_prev = t0;
  for (long spins = 0; *done != stamp; spins++) {
    if (*done == -stamp) {
      // the launched-ahead kernel gave up before the point reached it (device watchdog: the host was held up for
      // seconds between arming it and this call -- a progress callback, a descheduled thread): same linearisation
      // again, point passed by value, no launch-ahead until the next call re-arms
      rc = ba_disarm(ctx, w);  // the launch queued behind the dead one would wait for ITS point until its own watchdog fires
      if (rc) return rc;
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      rc = ba_fused_launch(ctx, w, x, f32, cauchy, rel_step, stamp);
      if (rc) return rc;
      w.prof_relaunches++;
      t0 = std::chrono::steady_clock::now();
      continue;
    }
    if (w.prof && (spins & 0xff) == 0xff) {
      const auto tn = std::chrono::steady_clock::now();
      w.prof_gap_max_ms = std::max(w.prof_gap_max_ms, std::chrono::duration<double, std::milli>(tn - t_prev).count());
      t_prev = tn;
    }
    if ((spins & 0xffff) == 0xffff) {
      const hipError_t e = hipStreamQuery(ctx->stream);  // a failed launch never writes the stamp
      if (e != hipSuccess && e != hipErrorNotReady) return ctx->hip_fail(e, "ba_fused_kernel");
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 20.0)
        return ctx->fail(MOCAP_E_HIP, "ba_fused_kernel: no completion stamp after 20 s");
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  if (w.prof)
    w.prof_wait_max_ms = std::max(w.prof_wait_max_ms, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  const int n = w.n, NP = w.NP;
  G.assign(nG, 0.0);
  for (int i = 0; i <= n; i++)
    for (int j = i; j <= n; j++) {
      const double v = w.h_fout[i * (n + 1) - i * (i - 1) / 2 + (j - i)];  // packed upper triangle, row-major
      G[(size_t)i * NP + j] = v;
      G[(size_t)j * NP + i] = v;
    }
  cost = w.h_fout[nG];
  if (finite) *finite = w.h_fout[nG + 1] != 0.0;
  return MOCAP_OK;
}

int ba_linearize(mocap_ctx* ctx, BaWork& w, const double* x, int f32, int cauchy, std::vector<double>& G,
                 double& cost, bool* finite = nullptr) {
  if (w.fused) {
    const double rs = f32 ? (double)std::sqrt(1.1920928955078125e-07f) : std::sqrt(kEps);
    return ba_linearize_fused(ctx, w, x, f32, cauchy, rs, G, cost, finite);
  }
  memcpy(w.h_x, x, sizeof(double) * w.n);  // pinned, read by the kernel directly (zero-copy)
  // scipy _numdiff: rel_step = sqrt(eps of the residual dtype) for the 2-point scheme
  // (_eps_for_method returns np.finfo(np.float32).eps ** 0.5, a np.float32 scalar under NumPy 2 promotion:
  // the square root is taken AND rounded in float32 -- 1.7e-8 off the float64 root, which is enough to move
  // x + h and flip float32 roundings in ~1 % of the differenced residuals)
  const double rel_step = f32 ? (double)std::sqrt(1.1920928955078125e-07f) : std::sqrt(kEps);
  int rc = ba_eval_device(ctx, w, w.n + 1, nullptr, w.h_x, rel_step);  // perturbation fused into the table build
  if (rc) return rc;
  BaJacArgs ja;
  ja.n = w.n;
  ja.NP = w.NP;
  ja.N = w.N;
  ja.m = w.m;
  ja.valid = w.d_valid;
  ja.r = w.d_r;
  ja.hvec = w.d_hvec;
  ja.f32_residuals = f32;
  ja.use_cauchy = cauchy;
  ja.Jaug = w.d_Jaug;
  ja.rho0 = w.d_rho;  // loss value per valid point: the cost is then a plain sum (fused into the Gram reduce)
  HIP_TRY(ctx, launch_ba_jacobian(ja, ctx->stream));
  // G and the (cost, finite) pair are written straight into pinned host memory by the reduce / cost kernels
  const size_t nG = (size_t)w.NP * w.NP, span = (size_t)(w.d_cost - w.d_G) + 2;
  HIP_TRY(ctx, launch_ba_gram_cost(w.d_Jaug, w.m_pad, w.NP, w.d_partial, w.ksplit, w.h_G, w.d_rho, nullptr, w.m, f32,
                                   -1 /* rho precomputed */, w.h_G + span - 2, ctx->stream));
  rc = ba_wait(ctx);
  if (rc) return rc;
  G.assign(w.h_G, w.h_G + nG);
  cost = w.h_G[span - 2];
  if (finite) *finite = w.h_G[span - 1] != 0.0;
  return MOCAP_OK;
}

}  // namespace

extern "C" int mocap_ba_residuals(mocap_ctx* ctx, int P, const double* params, int64_t N, const double* obs,
                                  double* r) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (P < 1 || !params || !r) return ctx->fail(MOCAP_E_ARG, "mocap_ba_residuals: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  BaWork w;
  int rc = ba_setup(ctx, w, N, obs, P);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(w.d_params, params, sizeof(double) * (size_t)P * w.n, hipMemcpyHostToDevice, ctx->stream));
  rc = ba_eval_device(ctx, w, P);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(r, w.d_r, sizeof(double) * (size_t)P * N, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MOCAP_OK;
}

extern "C" int mocap_ba_normal_eq(mocap_ctx* ctx, const double* x, int64_t N, const double* obs,
                                  int f32_residuals, int use_cauchy, double* JtJ, double* Jtr, double* cost,
                                  double* J_out, int64_t* m_out) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!x || !JtJ || !Jtr) return ctx->fail(MOCAP_E_ARG, "mocap_ba_normal_eq: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  BaWork w;
  int rc = ba_setup(ctx, w, N, obs, 1 + 7 * (ctx->C - 1) + 1, J_out != nullptr);
  if (rc) return rc;
  w.prearm = false;  // a single linearisation: nothing to launch ahead of
  BaDisarmGuard guard{ctx, &w};
  std::vector<double> G;
  double c = 0;
  rc = ba_linearize(ctx, w, x, f32_residuals, use_cauchy, G, c);
  if (rc) return rc;
  const int n = w.n, NP = w.NP;
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) JtJ[(size_t)i * n + j] = G[(size_t)i * NP + j];
    Jtr[i] = G[(size_t)i * NP + n];
  }
  if (cost) *cost = c;
  if (m_out) *m_out = w.m;
  if (J_out && w.m && w.fused) {  // rows by point index -> the valid rows in order
    std::vector<double> Jaug((size_t)N * NP);
    HIP_TRY(ctx, hipMemcpy(Jaug.data(), w.d_fJaug, sizeof(double) * Jaug.size(), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < w.m; i++)
      for (int j = 0; j < n; j++) J_out[(size_t)i * n + j] = Jaug[(size_t)w.valid[i] * NP + j];
  } else if (J_out && w.m) {
    std::vector<double> Jaug((size_t)w.m_pad * NP);
    HIP_TRY(ctx, hipMemcpy(Jaug.data(), w.d_Jaug, sizeof(double) * Jaug.size(), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < w.m; i++)
      for (int j = 0; j < n; j++) J_out[(size_t)i * n + j] = Jaug[(size_t)i * NP + j];
  }
  return MOCAP_OK;
}

// The subproblem solver as mocap_ba_solve drives it: dead parameters deflated exactly, then the Cholesky
// secular iteration when J has zero columns (scipy's rank-deficient branch) and the eigen path otherwise or when
// a pivot collapses.  Shared by the LM loop's test entry point below.
struct TrSubproblem {
  int n = 0;
  int64_t m = 0;
  std::vector<int> alive, order;
  std::vector<double> A, Va, lama, V, lam, s, suf, Vs, plive;
  CholSecular chol;  // csrc/tr_host.cpp
  bool use_chol = false, eig_ready = false;
  const double* JtJ = nullptr;
  const double* g = nullptr;
  // method: 0 = as mocap_ba_solve (Cholesky when rank-deficient), 1 = always eigen, 2 = Cholesky or fail
  void prepare(int n_, int64_t m_, const double* JtJ_, const double* g_, int method) {
    n = n_;
    m = m_;
    JtJ = JtJ_;
    g = g_;
    alive.clear();
    for (int i = 0; i < n; i++) {
      bool any = false;
      for (int j = 0; j < n && !any; j++) any = JtJ[(size_t)i * n + j] != 0.0;
      if (any) alive.push_back(i);
    }
    const int na = (int)alive.size();
    A.assign((size_t)na * na, 0.0);
    for (int a = 0; a < na; a++)
      for (int b = 0; b < na; b++) A[(size_t)a * na + b] = JtJ[(size_t)alive[a] * n + alive[b]];
    use_chol = method != 1 && na > 0 && (na < n || method == 2);
    eig_ready = false;
    if (use_chol) {
      chol.set(JtJ, g, n, alive.data(), na);
      plive.resize(na);
    }
  }
  void ensure_eigen() {
    if (eig_ready) return;
    eig_ready = true;
    const int na = (int)alive.size();
    sym_eig(na, A, Va, lama);
    V.assign((size_t)n * n, 0.0);
    lam.assign(n, 0.0);
    for (int k = 0; k < na; k++) {
      lam[k] = lama[k];
      for (int a = 0; a < na; a++) V[(size_t)alive[a] * n + k] = Va[(size_t)a * na + k];
    }
    {
      int k = na;
      for (int i = 0; i < n; i++)
        if (!std::binary_search(alive.begin(), alive.end(), i)) V[(size_t)i * n + k++] = 1.0;
    }
    order.resize(n);
    s.resize(n);
    suf.resize(n);
    Vs.resize((size_t)n * n);
    for (int i = 0; i < n; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return lam[a] > lam[b]; });
    for (int k = 0; k < n; k++) {
      const int src = order[k];
      s[k] = std::sqrt(std::max(lam[src], 0.0));
      double acc = 0;
      for (int i = 0; i < n; i++) {
        Vs[(size_t)i * n + k] = V[(size_t)i * n + src];
        acc += V[(size_t)i * n + src] * g[i];
      }
      suf[k] = acc;  // s * U^T f = V^T J^T f
    }
  }
  // returns the method that produced the step (1 eigen, 2 Cholesky)
  int solve(double Delta, double& alpha, std::vector<double>& step, bool chol_only = false) {
    if (use_chol && !chol.solve(Delta, alpha, plive.data())) {
      if (chol_only) return 0;
      use_chol = false;  // ill-conditioned live block
    }
    if (use_chol) {
      step.assign(n, 0.0);
      for (size_t a = 0; a < alive.size(); a++) step[alive[a]] = plive[a];
      return 2;
    }
    ensure_eigen();
    solve_tr(n, m, suf, s, Vs, Delta, alpha, step);
    return 1;
  }
};

extern "C" int mocap_ba_trust_region_step(mocap_ctx* ctx, int n, int64_t m, const double* JtJ, const double* Jtr,
                                          double Delta, double* alpha_io, int method, double* step, int32_t* info) {
  // host-only arithmetic on caller-owned buffers: no context state is touched (ctx may be NULL; it only
  // receives the error text)
  if (n < 1 || !JtJ || !Jtr || !alpha_io || !step || !(Delta > 0) || method < 0 || method > 2)
    return ctx ? ctx->fail(MOCAP_E_ARG, "mocap_ba_trust_region_step: bad argument") : MOCAP_E_ARG;
  TrSubproblem tr;
  tr.prepare(n, m, JtJ, Jtr, method);
  std::vector<double> p;
  double alpha = *alpha_io;
  const int used = tr.solve(Delta, alpha, p, method == 2);
  if (!used)
    return ctx ? ctx->fail(MOCAP_E_NOCONV, "mocap_ba_trust_region_step: Cholesky pivot collapsed (method 2 has no fallback)")
               : MOCAP_E_NOCONV;
  memcpy(step, p.data(), sizeof(double) * n);
  *alpha_io = alpha;
  if (info) {
    info[0] = used;
    info[1] = (int32_t)tr.alive.size();
  }
  return MOCAP_OK;
}

extern "C" int mocap_set_ba_progress(mocap_ctx* ctx, void (*cb)(const double* x, int n, void* user), void* user) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  ctx->ba_progress = cb;
  ctx->ba_progress_user = user;
  return MOCAP_OK;
}

extern "C" int mocap_ba_profile(mocap_ctx* ctx, const double* x, int64_t N, const double* obs, int f32_residuals,
                                int use_cauchy, int reps, double* out) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!x || !out || reps < 1) return ctx->fail(MOCAP_E_ARG, "mocap_ba_profile: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  BaWork w;
  int rc = ba_setup(ctx, w, N, obs, 1 + 7 * (ctx->C - 1) + 1);
  if (rc) return rc;
  BaDisarmGuard guard{ctx, &w};
  const bool prearm = w.prearm;
  w.prearm = false;  // kernel time first: each launch alone between two events
  std::vector<double> G;
  double cost = 0;
  for (int i = 0; i < 3; i++) {  // warm
    rc = ba_linearize(ctx, w, x, f32_residuals, use_cauchy, G, cost);
    if (rc) return rc;
  }
  hipEvent_t e0, e1;
  HIP_TRY(ctx, hipEventCreate(&e0));
  HIP_TRY(ctx, hipEventCreate(&e1));
  double gpu_ms = 0, wall_us = 0;
  for (int i = 0; i < reps; i++) {
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
    rc = ba_linearize(ctx, w, x, f32_residuals, use_cauchy, G, cost);
    if (rc) return rc;
    HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
    wall_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    HIP_TRY(ctx, hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
    gpu_ms += ms;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  // the host side of an iteration: the trust-region subproblem at the first radius scipy would use
  const int n = w.n, NP = w.NP;
  std::vector<double> JtJ((size_t)n * n), g(n), step;
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) JtJ[(size_t)i * n + j] = G[(size_t)i * NP + j];
    g[i] = G[(size_t)i * NP + n];
  }
  double Delta = 0;
  for (int i = 0; i < n; i++) Delta += x[i] * x[i];
  Delta = Delta > 0 ? std::sqrt(Delta) : 1.0;
  // an iteration as the LM loop runs it: host subproblem, then the linearisation at the new point (with launch-ahead
  // the kernel for it is already resident and polling while the host solves)
  w.prearm = prearm;
  double tr_us = 0, post_us = 0;
  for (int i = 0; i < reps; i++) {
    const auto ta = std::chrono::steady_clock::now();
    TrSubproblem tr;
    tr.prepare(n, w.m, JtJ.data(), g.data(), 0);
    double alpha = 0.0;
    tr.solve(Delta, alpha, step);
    const auto tb = std::chrono::steady_clock::now();
    rc = ba_linearize(ctx, w, x, f32_residuals, use_cauchy, G, cost);
    if (rc) return rc;
    const auto tc = std::chrono::steady_clock::now();
    tr_us += std::chrono::duration<double, std::micro>(tb - ta).count();
    post_us += std::chrono::duration<double, std::micro>(tc - tb).count();
  }
  tr_us /= reps;
  wall_us = post_us;
  out[0] = 1e3 * gpu_ms / reps;
  out[1] = wall_us / reps;
  out[2] = tr_us;
  out[3] = w.fused ? 1.0 : 5.0;
  out[4] = (double)w.m;
  out[5] = NP;
  out[6] = w.fused ? 1.0 : 0.0;
  out[7] = cost;
  return MOCAP_OK;
}

static int ba_solve_impl(mocap_ctx* ctx, double* x, int64_t N, const double* obs, double ftol, double xtol, double gtol,
                         int max_iter, int f32_residuals, int use_cauchy, double* info_out, int info_len);

// info [8]: the layout this symbol has had since it was first exported -- a caller's 8-double buffer stays valid
extern "C" int mocap_ba_solve(mocap_ctx* ctx, double* x, int64_t N, const double* obs, double ftol, double xtol,
                              double gtol, int max_iter, int f32_residuals, int use_cauchy, double* info) {
  return ba_solve_impl(ctx, x, N, obs, ftol, xtol, gtol, max_iter, f32_residuals, use_cauchy, info, 8);
}

// the same with the caller stating how many doubles `info` holds (at most MOCAP_BA_INFO_DOUBLES are written)
extern "C" int mocap_ba_solve_ex(mocap_ctx* ctx, double* x, int64_t N, const double* obs, double ftol, double xtol,
                                 double gtol, int max_iter, int f32_residuals, int use_cauchy, double* info, int info_len) {
  return ba_solve_impl(ctx, x, N, obs, ftol, xtol, gtol, max_iter, f32_residuals, use_cauchy, info, info_len);
}

static int ba_solve_impl(mocap_ctx* ctx, double* x, int64_t N, const double* obs, double ftol, double xtol, double gtol,
                         int max_iter, int f32_residuals, int use_cauchy, double* info_out, int info_len) {
  double info_buf[MOCAP_BA_INFO_DOUBLES] = {0};
  double* info = info_out ? info_buf : nullptr;
  struct InfoCopy {  // every return path hands over what was filled in
    double* dst; const double* src; int n;
    ~InfoCopy() { if (dst) memcpy(dst, src, sizeof(double) * (size_t)std::max(0, std::min(n, (int)MOCAP_BA_INFO_DOUBLES))); }
  } info_copy{info_out, info_buf, info_len};
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!x) return ctx->fail(MOCAP_E_ARG, "mocap_ba_solve: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const auto t_begin = std::chrono::steady_clock::now();
  const bool prof = getenv("MOCAP_BA_PROFILE") != nullptr;  // stderr breakdown of one solve
  double t_lin = 0, t_eig = 0, t_tr = 0, t_cost = 0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
  };
  BaWork w;
  w.prof = prof;
  int rc = ba_setup(ctx, w, N, obs, 1 + 7 * (ctx->C - 1) + 1);
  if (rc) return rc;
  w.prof_setup_ms = ms(t_begin, now());
  BaDisarmGuard guard{ctx, &w};  // every return path below abandons the kernel that was launched ahead
  const int n = w.n, NP = w.NP;
  if (w.m < 1) return ctx->fail(MOCAP_E_ARG, "mocap_ba_solve: no point is seen by two cameras");
  const int max_nfev = max_iter > 0 ? max_iter : 100 * n;  // scipy: max_nfev = 100 * n

  std::vector<double> xv(x, x + n), G, JtJ((size_t)n * n), g(n), step, x_new(n);
  double cost = 0;
  const auto t_first0 = now();
  rc = ba_linearize(ctx, w, xv.data(), f32_residuals, use_cauchy, G, cost);
  if (rc) return rc;
  const double t_first = ms(t_first0, now());
  const double cost0 = cost;
  int nfev = 1, njev = 1, iteration = 0, termination = 0;
  double Delta = norm2(xv);  // x_scale = 1 (scipy default)
  if (Delta == 0) Delta = 1.0;
  double alpha = 0.0, g_norm = 0.0;
  bool need_factor = true;
  std::vector<double> eig_ms, lin_ms;
  TrSubproblem tr;
  bool have_trial = false;
  const bool speculate = !getenv("MOCAP_BA_NO_SPECULATION");
  if (!speculate) w.prearm = false;  // cost-only evaluations queue plain kernels on the stream: nothing may be parked on it
  std::vector<double> G_trial;

  while (true) {
    if (need_factor) {
      for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) JtJ[(size_t)i * n + j] = G[(size_t)i * NP + j];
        g[i] = G[(size_t)i * NP + n];  // compute_grad: J^T f
      }
    }
    g_norm = 0;
    for (double v : g) g_norm = std::max(g_norm, std::fabs(v));
    if (g_norm < gtol) termination = 1;
    if (termination || nfev >= max_nfev) break;
    if (need_factor) {
      // Parameters without any effect (the dead focal entries, helpers.py:267-270) give exactly-zero
      // rows and columns of J^T J.  SciPy sees them as exactly-zero singular values of J; an
      // eigen-solver working on the squared matrix would return +-eps*lam_max instead, i.e. singular
      // values ~1e-8 that pass the full-rank test.  Deflate them exactly: solve the live block only
      // (TrSubproblem::prepare).  Exactly-zero columns of J <=> rank-deficient for scipy: the secular
      // iteration needs no spectrum (Cholesky path).
      const auto te0 = now();
      tr.prepare(n, w.m, JtJ.data(), g.data(), getenv("MOCAP_BA_EIGEN") ? 1 : 0);
      t_eig += ms(te0, now());
      need_factor = false;
    }
    double actual_reduction = -1, step_norm = 0, cost_new = cost;
    have_trial = false;
    while (actual_reduction <= 0 && nfev < max_nfev) {
      const auto tt0 = now();
      tr.solve(Delta, alpha, step);
      t_tr += ms(tt0, now());
      // predicted_reduction = -evaluate_quadratic(J, g, step) = -(0.5 |J step|^2 + step.g)
      double q = 0, l = 0;
      for (int i = 0; i < n; i++) {
        double acc = 0;
        for (int j = 0; j < n; j++) acc += JtJ[(size_t)i * n + j] * step[j];
        q += step[i] * acc;
        l += step[i] * g[i];
      }
      const double predicted_reduction = -(0.5 * q + l);
      for (int i = 0; i < n; i++) x_new[i] = xv[i] + step[i];
      bool finite = true;
      const auto tc0 = now();
      // Speculative linearisation: the trial point is evaluated together with its forward-difference batch
      // (one launch sequence, latency-bound either way).  Nine steps in ten are accepted, and then the
      // Jacobian scipy would compute next at x_new is already here; a rejected step discards it.
      if (speculate) {
        rc = ba_linearize(ctx, w, x_new.data(), f32_residuals, use_cauchy, G_trial, cost_new, &finite);
        have_trial = rc == MOCAP_OK;
      } else {
        rc = ba_cost_at(ctx, w, x_new.data(), f32_residuals, use_cauchy, cost_new, finite);
      }
      if (rc) return rc;
      t_cost += ms(tc0, now());
      nfev++;
      const double step_h_norm = norm2(step);
      if (!finite || !std::isfinite(cost_new)) {
        Delta = 0.25 * step_h_norm;
        continue;
      }
      actual_reduction = cost - cost_new;
      // update_tr_radius
      double ratio;
      if (predicted_reduction > 0)
        ratio = actual_reduction / predicted_reduction;
      else if (predicted_reduction == 0 && actual_reduction == 0)
        ratio = 1;
      else
        ratio = 0;
      double Delta_new = Delta;
      if (ratio < 0.25)
        Delta_new = 0.25 * step_h_norm;
      else if (ratio > 0.75 && step_h_norm > 0.95 * Delta)
        Delta_new = Delta * 2.0;
      step_norm = step_h_norm;
      // check_termination
      const bool ftol_ok = actual_reduction < ftol * cost && ratio > 0.25;
      const bool xtol_ok = step_norm < xtol * (xtol + norm2(xv));
      termination = (ftol_ok && xtol_ok) ? 4 : ftol_ok ? 2 : xtol_ok ? 3 : 0;
      if (termination) break;
      alpha *= Delta / Delta_new;
      Delta = Delta_new;
    }
    if (actual_reduction > 0) {
      xv = x_new;
      cost = cost_new;
      if (ctx->ba_progress) {
        // once per accepted step.  The callback is the caller's code (a socket emit under the GIL in the reference's
        // handler): no kernel may sit on the GPU polling for a point while it runs, however long it takes
        rc = ba_disarm(ctx, w);
        if (rc) return rc;
        ctx->ba_progress(xv.data(), n, ctx->ba_progress_user);
      }
      njev++;              // scipy re-linearises even when it is about to stop (trf.py:534); the result is unused,
      if (!termination) {  // so only the count is kept
        const auto tl0 = now();
        if (have_trial) {
          G.swap(G_trial);  // linearised at x_new already (the last trial of the loop above is the accepted one)
        } else {
          rc = ba_linearize(ctx, w, xv.data(), f32_residuals, use_cauchy, G, cost_new);
          if (rc) return rc;
        }
        t_lin += ms(tl0, now());
        if (prof) lin_ms.push_back(ms(tl0, now()));
        need_factor = true;
      }
    }
    iteration++;
    if (termination) break;
  }
  memcpy(x, xv.data(), sizeof(double) * n);
  if (prof && !lin_ms.empty()) {
    std::vector<double> t = lin_ms;
    std::sort(t.begin(), t.end());
    fprintf(stderr, "[mocap_ba_solve] linearize per call: min %.3f median %.3f p90 %.3f max %.3f ms; first slow calls at:", t.front(),
            t[t.size() / 2], t[t.size() * 9 / 10], t.back());
    int shown = 0;
    for (size_t i = 0; i < lin_ms.size() && shown < 12; i++)
      if (lin_ms[i] > 3 * t[t.size() / 2]) { fprintf(stderr, " #%zu=%.2f", i, lin_ms[i]); shown++; }
    fprintf(stderr, "\n");
  }
  if (prof && !eig_ms.empty()) {
    std::sort(eig_ms.begin(), eig_ms.end());
    fprintf(stderr, "[mocap_ba_solve] eigen per call: min %.3f median %.3f p90 %.3f max %.3f ms\n", eig_ms.front(),
            eig_ms[eig_ms.size() / 2], eig_ms[eig_ms.size() * 9 / 10], eig_ms.back());
  }
  if (prof)
    fprintf(stderr, "[mocap_ba_solve] iterations %d nfev %d njev %d: linearize %.2f ms, eigen %.2f ms, trust region %.2f ms, "
            "cost evals %.2f ms, total %.2f ms | setup %.3f ms, first linearise %.3f ms, longest launch call %.3f ms, longest "
            "wait for a stamp %.3f ms, longest gap between two clock reads of the spinning host thread %.3f ms\n",
            iteration, nfev, njev, t_lin, t_eig, t_tr, t_cost, ms(t_begin, now()), w.prof_setup_ms, t_first, w.prof_launch_max_ms,
            w.prof_wait_max_ms, w.prof_gap_max_ms);
  if (info) {
    const auto t_end = std::chrono::steady_clock::now();
    info[0] = iteration;
    info[1] = nfev;
    info[2] = termination;
    info[3] = cost0;
    info[4] = cost;
    info[5] = g_norm;
    info[6] = (double)w.m;
    info[7] = std::chrono::duration<double, std::milli>(t_end - t_begin).count();
    info[8] = njev;
    info[9] = w.prof_relaunches;
  }
  return termination ? MOCAP_OK : MOCAP_E_NOCONV;
}
