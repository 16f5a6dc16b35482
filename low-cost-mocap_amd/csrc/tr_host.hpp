// tr_host.hpp -- host-side dense kernels of the trust-region subproblem (csrc/tr_host.cpp, plain C++).
#pragma once
#include <cstddef>
#include <vector>

namespace mocap {

// Secular iteration of scipy's solve_lsq_trust_region for a rank-deficient J (the case the reference is always
// in: exactly-zero Jacobian columns of the dead focal parameters), on the live block B = J^T J, g = J^T f:
//     phi(a) = |p(a)| - Delta,  p(a) = -(B + a I)^{-1} g,  phi'(a) = -(p^T (B + a I)^{-1} p) / |p|
// One Cholesky factorisation and two triangular solves per value of a.
class CholSecular {
 public:
  // B_full: n x n row-major; alive: indices of the live parameters (na of them); g_full: n
  void set(const double* B_full, const double* g_full, int n, const int* alive, int na);
  // scipy's iteration from alpha (0 = none).  p_live [na].  false: a pivot fell below 1e-10 of its diagonal
  // entry (the block is numerically singular at this shift) -- the caller falls back to the eigen path.
  bool solve(double Delta, double& alpha, double* p_live);
  int live() const { return na_; }

 private:
  bool factor(double a);
  void solve_inplace(double* b) const;
  void forward_inplace(double* b) const;
  int na_ = 0, ld_ = 0;
  std::vector<double> store_;  // aligned carve-out: B | L | invd | g | q | w
  double *B_ = nullptr, *L_ = nullptr, *invd_ = nullptr, *g_ = nullptr, *q_ = nullptr, *w_ = nullptr;
};

}  // namespace mocap
