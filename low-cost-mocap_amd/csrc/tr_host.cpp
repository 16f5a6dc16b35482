// tr_host.cpp -- host-side dense kernels of the trust-region subproblem (plain C++, no HIP).
//
// Part of what replaces scipy.optimize._lsq.common.solve_lsq_trust_region (reference helpers.py:287 ->
// scipy _lsq/trf.py:495).  An LM iteration at the metric's size (8 cameras x 1 000 points) is ~30 us of GPU
// work and one host decision; at 15 us the scalar Cholesky-secular solve was a quarter of the iteration, so
// the 42 x 42 factorisation is written for the host's SIMD units: rows padded to a multiple of four doubles,
// dot products over contiguous row prefixes with two 4-wide accumulators (GCC/clang vector extensions; the
// avx2 clone is picked at load time, the baseline clone computes the same sums in the same order with SSE2
// pairs, so the step does not depend on which host ran it).
#include "tr_host.hpp"

#include <cmath>
#include <cstdint>
#include <cstring>

namespace mocap {
namespace {

typedef double v4d __attribute__((vector_size(32), aligned(8)));

#if defined(__x86_64__)
#define MOCAP_SIMD_CLONES __attribute__((target_clones("avx2", "default")))
#else
#define MOCAP_SIMD_CLONES
#endif

// sum_{k < n} a[k] * b[k], fixed association: two 4-wide partial sums over k = 0..8m-1, their lane-wise
// sum reduced as (l0 + l1) + (l2 + l3), then the tail left to right
static inline double dot_prefix(const double* a, const double* b, int n) {
  v4d s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
  int k = 0;
  for (; k + 8 <= n; k += 8) {
    s0 += *(const v4d*)(a + k) * *(const v4d*)(b + k);
    s1 += *(const v4d*)(a + k + 4) * *(const v4d*)(b + k + 4);
  }
  if (k + 4 <= n) {
    s0 += *(const v4d*)(a + k) * *(const v4d*)(b + k);
    k += 4;
  }
  const v4d s = s0 + s1;
  double r = (s[0] + s[1]) + (s[2] + s[3]);
  for (; k < n; k++) r += a[k] * b[k];
  return r;
}

// y[k] -= c * a[k] for k < n
static inline void axpy_neg(double* y, const double* a, double c, int n) {
  const v4d cv = {c, c, c, c};
  int k = 0;
  for (; k + 4 <= n; k += 4) *(v4d*)(y + k) -= cv * *(const v4d*)(a + k);
  for (; k < n; k++) y[k] -= c * a[k];
}

// L L^T = B + a I, column by column (Cholesky-Crout) on row-major storage: every entry of column j is an inner
// product of two contiguous row prefixes, and the entries of one column are independent of each other (the
// row-by-row order has a serial chain through each row: measured 2x slower on an out-of-order core; a right-looking
// version in blocks of four columns with a vectorised trailing update: 1.25x slower -- the strided panel solve eats
// what the update gains at n = 42)
MOCAP_SIMD_CLONES bool chol_factor(const double* B, double* L, double* invd, int na, int ld, double a) {
  for (int j = 0; j < na; j++) {
    double* lj = L + (size_t)j * ld;
    const double d0 = B[(size_t)j * ld + j] + a;
    const double d = d0 - dot_prefix(lj, lj, j);
    if (!(d > 1e-10 * d0)) return false;
    const double sd = std::sqrt(d);
    lj[j] = sd;
    const double id = 1.0 / sd;
    invd[j] = id;
    for (int i = j + 1; i < na; i++) {
      double* li = L + (size_t)i * ld;
      li[j] = (B[(size_t)i * ld + j] - dot_prefix(li, lj, j)) * id;
    }
  }
  return true;
}

// (L L^T) x = b in place: forward substitution by row dot products, backward substitution in axpy form (row i of L
// is column i of L^T), so both sweeps touch contiguous rows only
MOCAP_SIMD_CLONES void chol_solve(const double* L, const double* invd, int na, int ld, double* b) {
  for (int i = 0; i < na; i++) b[i] = (b[i] - dot_prefix(L + (size_t)i * ld, b, i)) * invd[i];
  for (int i = na - 1; i >= 0; i--) {
    const double xi = b[i] * invd[i];
    b[i] = xi;
    axpy_neg(b, L + (size_t)i * ld, xi, i);
  }
}

MOCAP_SIMD_CLONES void chol_forward(const double* L, const double* invd, int na, int ld, double* b) {
  for (int i = 0; i < na; i++) b[i] = (b[i] - dot_prefix(L + (size_t)i * ld, b, i)) * invd[i];
}

static inline double norm2n(const double* v, int n) { return std::sqrt(dot_prefix(v, v, n)); }

}  // namespace

void CholSecular::set(const double* B_full, const double* g_full, int n, const int* alive, int na) {
  na_ = na;
  ld_ = (na + 3) / 4 * 4;
  const size_t nB = (size_t)ld_ * ld_;
  store_.assign(2 * nB + 4 * (size_t)ld_ + 8, 0.0);
  double* p = store_.data();
  p += ((32 - (reinterpret_cast<uintptr_t>(p) & 31)) & 31) / sizeof(double);
  B_ = p;
  L_ = B_ + nB;
  invd_ = L_ + nB;
  g_ = invd_ + ld_;
  q_ = g_ + ld_;
  w_ = q_ + ld_;
  for (int a = 0; a < na; a++) {
    for (int b = 0; b < na; b++) B_[(size_t)a * ld_ + b] = B_full[(size_t)alive[a] * n + alive[b]];
    g_[a] = g_full[alive[a]];
  }
}

bool CholSecular::factor(double a) { return chol_factor(B_, L_, invd_, na_, ld_, a); }
void CholSecular::solve_inplace(double* b) const { chol_solve(L_, invd_, na_, ld_, b); }
void CholSecular::forward_inplace(double* b) const { chol_forward(L_, invd_, na_, ld_, b); }

bool CholSecular::solve(double Delta, double& alpha_io, double* p_live) {
  const int na = na_;
  double alpha = alpha_io;
  double alpha_upper = norm2n(g_, na) / Delta;  // |suf| = |V^T g| = |g|
  double alpha_lower = 0.0;
  if (alpha == 0.0) alpha = std::fmax(0.001 * alpha_upper, std::sqrt(alpha_lower * alpha_upper));
  for (int it = 0; it < 10; it++) {
    if (alpha < alpha_lower || alpha > alpha_upper)
      alpha = std::fmax(0.001 * alpha_upper, std::sqrt(alpha_lower * alpha_upper));
    if (!factor(alpha)) return false;
    std::memcpy(q_, g_, sizeof(double) * na);
    solve_inplace(q_);  // q = (B + a I)^{-1} g = -p
    const double p_norm = norm2n(q_, na);
    // phi' needs q^T (B + a I)^{-1} q = |L^{-1} q|^2: one forward substitution, not a second full solve
    std::memcpy(w_, q_, sizeof(double) * na);
    forward_inplace(w_);
    const double pw = dot_prefix(w_, w_, na);
    const double phi = p_norm - Delta, phi_prime = -pw / p_norm;
    if (phi < 0) alpha_upper = alpha;
    const double ratio = phi / phi_prime;
    alpha_lower = std::fmax(alpha_lower, alpha - ratio);
    alpha -= (phi + Delta) * ratio / Delta;
    if (std::fabs(phi) < 0.01 * Delta) break;
  }
  if (!factor(alpha)) return false;
  std::memcpy(q_, g_, sizeof(double) * na);
  solve_inplace(q_);
  const double pn = norm2n(q_, na);
  for (int i = 0; i < na; i++) p_live[i] = pn > 0 ? -q_[i] * (Delta / pn) : -q_[i];
  alpha_io = alpha;
  return true;
}

}  // namespace mocap
