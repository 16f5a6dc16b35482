// mocap_device.hpp -- per-lane FP64 geometry core shared by every kernel of the path.
//
// Replaces, per candidate correspondence group (one wave lane = one group):
//   * triangulate_point + DLT            (reference computer_code/api/helpers.py:293-327)
//   * calculate_reprojection_error       (helpers.py:214-241, incl. cv.projectPoints)
//
// Design (gfx950): everything for one candidate lives in VGPRs (10-entry packed symmetric
// B = A^T A, 16-entry eigenvector accumulator); camera tables are read with wave-uniform
// addresses so they come in over the scalar cache (s_load) when all intrinsics are equal.
// The 4x4 null vector comes from a shifted-Cholesky / Laguerre / inverse-iteration solve (a cyclic
// Jacobi eigen-solve with compile-time rotation indices is kept behind -DMOCAP_EIG_JACOBI); both
// run in registers only (no dynamic register indexing, no scratch).  FP64 throughout: B squares the condition
// number of A (helpers.py:319-321) and the contract is 1e-5 relative on the 3-D point.
//
// The file is compiled with -ffp-contract=off: expressions whose rounding the reference
// pins (OpenCV's scalar C loops, NumPy elementwise ops) are written unfused; fma() is
// explicit where the reference's own order is BLAS/LAPACK-internal and therefore unpinned.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mocap {

constexpr int kMaxCameras = 64;
constexpr int kMaxBlobs = 256;

// Camera tables are frame-invariant and never written by the kernels that read them, so they are
// addressed through the CONSTANT address space: wave-uniform reads then compile to scalar-cache
// loads (s_load_dwordx4/x8 into SGPRs) instead of 64-lane vector loads of one address.
typedef const double __attribute__((address_space(4))) * ctab_t;
__host__ __device__ inline ctab_t as_ctab(const double* p) { return (ctab_t)(uintptr_t)p; }

// Device view of the camera tables built by mocap_set_cameras (csrc/capi.hip).
struct CamView {
  int C;
  int uniformK;      // all intrinsic matrices identical -> Pq is [C][12]
  int f32_rounding;  // reproduce OpenCV's float32 roundings (MOCAP_OPT_F32_ROUNDING)
  int _pad;
  const double* Pq;  // uniformK: [cam][12], else [j][cam][12]: K[j] @ [R|t][cam]
                     // (intrinsics by compacted index j: helpers.py:296-298,305-307)
  const double* RT;  // [cam][12]: R row-major (9), t (3)
  const double* K4;  // [j][4]: fx, fy, cx, cy
  const double* F;   // [a][b][9]: fundamentalFromProjections(P_a, P_b) (helpers.py:362)
  // table accessors used by the geometry core below (offsets in doubles); wave-uniform -> scalar loads
  __device__ __forceinline__ ctab_t pq(size_t off) const { return as_ctab(Pq + off); }
  __device__ __forceinline__ ctab_t rt(size_t off) const { return as_ctab(RT + off); }
  __device__ __forceinline__ ctab_t k4(size_t off) const { return as_ctab(K4 + off); }
};

// The same view with the pose-dependent tables (Pq, RT) in LDS: bundle adjustment builds the cameras of a
// parameter vector inside the kernel that consumes them (csrc/ba_kernels.hip), where the scalar cache cannot be
// used (it is not coherent with stores of the same kernel).  Wave-uniform LDS addresses are broadcast reads.
typedef const double __attribute__((address_space(3))) * ltab_t;
struct LdsCamView {
  int C;
  int uniformK;
  int f32_rounding;
  int _pad;
  ltab_t Pq;         // layouts as CamView
  ltab_t RT;
  const double* K4;  // intrinsics do not depend on the parameter vector (dead focal entries, helpers.py:267-270)
  __device__ __forceinline__ ltab_t pq(size_t off) const { return Pq + off; }
  __device__ __forceinline__ ltab_t rt(size_t off) const { return RT + off; }
  __device__ __forceinline__ ctab_t k4(size_t off) const { return as_ctab(K4 + off); }
};

// CamView with the camera count as a compile-time constant (frame_bb.hip's 8-camera instantiation): the geometry
// core's loops over the cameras then unroll.
template <int CT>
struct CamViewFixed {
  const CamView& v;
  static constexpr int C = CT;
  __device__ __forceinline__ ctab_t pq(size_t off) const { return v.pq(off); }
  __device__ __forceinline__ ctab_t rt(size_t off) const { return v.rt(off); }
  __device__ __forceinline__ ctab_t k4(size_t off) const { return v.k4(off); }
};

// ---- packed symmetric 4x4: (0,0)=0 (0,1)=1 (0,2)=2 (0,3)=3 (1,1)=4 (1,2)=5 (1,3)=6 (2,2)=7 (2,3)=8 (3,3)=9
__host__ __device__ constexpr int sidx(int i, int j) {
  return i <= j ? (i == 0 ? j : i == 1 ? 3 + j : i == 2 ? 5 + j : 9)
                : (j == 0 ? i : j == 1 ? 3 + i : j == 2 ? 5 + i : 9);
}

// ---- FP64 building blocks written against the gfx950 instruction set.
// v_rsq_f64 / v_rcp_f64 deliver ~2^-26 relative accuracy; HIP's rsqrt()/sqrt()/operator/ wrap them
// in Newton steps PLUS range scaling and class fix-ups (9-20 instructions).  Where the operand range
// is known those wrappers are dead weight:

// 1/sqrt(d) for finite d > 0 away from the exponent limits (Cholesky pivots clamped from below):
// raw estimate + one third-order step -> < 1 ulp.  6 instructions instead of 9.
// The calling wave's global stores have been acknowledged (gfx9 counts stores on vmcnt).  Needed before a flag /
// counter that tells ANOTHER workgroup (or the host) "my data is there": a workgroup-scope release fence emits no
// such wait (visibility inside one CU needs none), and an agent-scope one writes the whole L2 back (16 us when every
// wave does it).  With sc1 (write-through) stores the acknowledgement means the data has left this XCD's L2.
__device__ __forceinline__ void wait_own_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// A workgroup barrier with its wait WRITTEN OUT.  __syncthreads() normally compiles to s_waitcnt lgkmcnt(0) + s_barrier; round 5
// found one barrier of frame_bb_kernel emitted without the wait (a wave read claim words while another wave's ds_or was
// still queued: one wrong frame in 1e5).  Used wherever tests/isa_barriers.py cannot prove the wait from the compiler's own
// output on every path (a barrier at a loop head that the previous iteration reaches through several exits); free when the
// compiler has already waited.  tests/test_code_objects_cpu.py checks every barrier of the big kernels in the shipped ISA.
__device__ __forceinline__ void block_sync_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
}

__device__ __forceinline__ double rsqrt_pos(double d) {
  const double y = __builtin_amdgcn_rsq(d);
  const double e = fma(-(d * y), y, 1.0);
  return fma(y * e, fma(e, 0.375, 0.5), y);
}

// r ~ 1/b refined to working precision (two Newton steps from v_rcp_f64), b finite, normal, non-zero
__device__ __forceinline__ double recip_refined(double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = fma(fma(-b, r, 1.0), r, r);
  r = fma(fma(-b, r, 1.0), r, r);
  return r;
}

// a / b with the final residual correction of the IEEE sequence the compiler emits for operator/
// (v_div_scale -> v_rcp -> 2 Newton -> q = a r -> q += (a - b q) r -> v_div_fmas -> v_div_fixup),
// minus the exponent scaling and the special-case fix-up: identical bits whenever neither operand
// nor the quotient comes near the exponent limits (|x| in 2^-500 .. 2^500, b != 0), i.e. for every
// camera depth and homogeneous coordinate this path sees.  r = recip_refined(b) is shared by
// quotients with the same denominator.
__device__ __forceinline__ double div_by(double a, double b, double r) {
  const double q = a * r;
  return fma(fma(-b, q, a), r, q);
}

// One Jacobi rotation in the (P,Q) plane; indices are compile-time so a[] / v[] stay in VGPRs.
template <int P, int Q>
__device__ __forceinline__ void jacobi_rot(double (&a)[10], double (&v)[16]) {
  const double apq = a[sidx(P, Q)];
  if (apq != 0.0) {
    const double app = a[sidx(P, P)], aqq = a[sidx(Q, Q)];
    // Rotation angle from the double-angle identities, division- and sqrt-free (two v_rsq_f64):
    //   h = aqq - app, w = 2 apq, r = hypot(h, w):  cos 2θ = |h| / r,  sin 2θ = sgn(h) w / r
    //   cos²θ = (1 + cos 2θ) / 2 = u,  c = sqrt(u) = u * rsqrt(u),  s = sin 2θ / (2c),  t = s / c
    const double h = aqq - app, w = apq + apq;
    const double ir = rsqrt(fma(h, h, w * w));
    const double c2 = fabs(h) * ir;
    const double s2 = (h < 0.0 ? -w : w) * ir;
    const double u = fma(0.5, c2, 0.5);
    const double ic = rsqrt(u);
    const double c = u * ic;
    const double s = 0.5 * s2 * ic;
    const double t = s * ic;
    a[sidx(P, P)] = fma(-t, apq, app);
    a[sidx(Q, Q)] = fma(t, apq, aqq);
    a[sidx(P, Q)] = 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (k != P && k != Q) {
        const double akp = a[sidx(k, P)], akq = a[sidx(k, Q)];
        a[sidx(k, P)] = fma(c, akp, -s * akq);
        a[sidx(k, Q)] = fma(s, akp, c * akq);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const double vkp = v[k * 4 + P], vkq = v[k * 4 + Q];
      v[k * 4 + P] = fma(c, vkp, -s * vkq);
      v[k * 4 + Q] = fma(s, vkp, c * vkq);
    }
  }
}

// Eigenvector of the smallest-magnitude eigenvalue of the symmetric 4x4 B (== the last
// right-singular vector scipy.linalg.svd(B) yields at helpers.py:320-321, up to sign).
__device__ __forceinline__ void smallest_eigvec4_jacobi(double (&a)[10], double (&out)[4]) {
  double v[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  for (int sweep = 0; sweep < 12; sweep++) {
    const double off2 = fma(a[1], a[1], fma(a[2], a[2], fma(a[3], a[3],
                        fma(a[5], a[5], fma(a[6], a[6], a[8] * a[8])))));
    const double dg2 = fma(a[0], a[0], fma(a[4], a[4], fma(a[7], a[7], a[9] * a[9])));
    // converged when the off-diagonal mass is ~1e-17 of the diagonal (below FP64 rounding)
    if (!(off2 > 1e-34 * dg2)) break;
    jacobi_rot<0, 1>(a, v);
    jacobi_rot<0, 2>(a, v);
    jacobi_rot<0, 3>(a, v);
    jacobi_rot<1, 2>(a, v);
    jacobi_rot<1, 3>(a, v);
    jacobi_rot<2, 3>(a, v);
  }
  const double d0 = fabs(a[0]), d1 = fabs(a[4]), d2 = fabs(a[7]), d3 = fabs(a[9]);
  int m = 0;
  double dm = d0;
  if (d1 < dm) { dm = d1; m = 1; }
  if (d2 < dm) { dm = d2; m = 2; }
  if (d3 < dm) { dm = d3; m = 3; }
#pragma unroll
  for (int k = 0; k < 4; k++)
    out[k] = m == 0 ? v[k * 4 + 0] : m == 1 ? v[k * 4 + 1] : m == 2 ? v[k * 4 + 2] : v[k * 4 + 3];
}

// Same vector, ~6x fewer instructions: shifted Cholesky + Laguerre + inverse iteration.
//   B is symmetric positive semi-definite, so p(x) = det(B - x I) has four real roots >= 0 and
//   Laguerre's iteration started at x = 0 climbs monotonically to the smallest one without ever
//   overshooting it (B - x I stays positive definite -> plain Cholesky is backward stable):
//       L L^T = B - x I,  M = L^{-1},  (B - x I)^{-1} = M^T M
//       s1 = tr (B - x I)^{-1} = |M|_F^2 = -p'/p,      s2 = tr (B - x I)^{-2} = |M^T M|_F^2
//       x += 4 / (s1 + sqrt(3 (4 s2 - s1^2)))                          (cubic convergence)
//   With rho = (lam1 - x) / (lam2 - x) the contraction of one inverse-iteration step,
//   1 - s2/s1^2 ~ 2 rho.  A factorisation costs ~160 instructions, an inverse-iteration step ~25,
//   so the shift only has to be good enough for a handful of steps: stop once 1 - s2/s1^2 < 1e-3
//   (rho < 5e-4) and run kInvIters = 5 steps from e4 (the first one is free: it is the last row
//   of M): rho^5 < 4e-17.  On 71 k candidate matrices of the 8 x 16 bench stream that is exactly
//   two factorisations for every lane of every wave (lam1/lam2 is ~1e-2 for the wrong groups that
//   dominate the candidate set, so shift 0 alone is never enough, and one Laguerre step always
//   is); max 2.3e-14 relative on X against LAPACK dgesdd.
constexpr int kInvIters = 5;
// lamcut / lam_lb: the exact cut-off of candidate selection (EigCut below).  trace((B - lam I)^-1) = s1 >= 1/(lam1 - lam),
// so every factorisation yields the rigorous lower bound lam1 >= lam + 1/s1.  After the FIRST factorisation (lam = 0)
// the candidate is dropped when s1 * lamcut < 1 (returns false, `out` untouched); otherwise lam_lb receives the bound
// of the last factorisation, shrunk by the rounding allowance (Cholesky backward error and the rounding of B itself
// are O(1e-15 tr); 2e-12 tr is charged).  lamcut = +inf switches the cut off.
__device__ __forceinline__ bool smallest_eigvec4_cholesky(const double (&a)[10], double (&out)[4], double lamcut,
                                                          double& lam_lb) {
  const double tr = (a[0] + a[4]) + (a[7] + a[9]);
  // pivots are clamped from below (fmax also swallows NaN): a shift that rounding pushed past lam1
  // yields one tiny pivot, i.e. a huge last row of M -- still the wanted vector -- and s2/s1^2 -> 1,
  // which ends the loop through the ordinary convergence test
  const double floor_piv = tr * 1e-30 + 1e-300;
  double lam = 0.0, piv3 = 1.0, s1_last = 1.0;
  double m[10];  // M = L^{-1}, lower triangular, packed like sidx with (row >= col) -> sidx(col,row)
  for (int it = 0; it < 8; it++) {
    // ---- Cholesky of B - lam I; r_i = 1 / l_ii
    const double r0 = rsqrt_pos(fmax(a[0] - lam, floor_piv));
    const double l10 = a[1] * r0, l20 = a[2] * r0, l30 = a[3] * r0;
    const double r1 = rsqrt_pos(fmax(fma(-l10, l10, a[4] - lam), floor_piv));
    const double l21 = fma(-l20, l10, a[5]) * r1, l31 = fma(-l30, l10, a[6]) * r1;
    const double r2 = rsqrt_pos(fmax(fma(-l21, l21, fma(-l20, l20, a[7] - lam)), floor_piv));
    const double l32 = fma(-l31, l21, fma(-l30, l20, a[8])) * r2;
    piv3 = fmax(fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, a[9] - lam))), floor_piv);
    const double r3 = rsqrt_pos(piv3);
    // ---- M = L^{-1}
    const double m00 = r0, m11 = r1, m22 = r2, m33 = r3;
    const double m10 = -(l10 * m00) * m11;
    const double m21 = -(l21 * m11) * m22;
    const double m32 = -(l32 * m22) * m33;
    const double m20 = -fma(l21, m10, l20 * m00) * m22;
    const double m31 = -fma(l32, m21, l31 * m11) * m33;
    const double m30 = -fma(l32, m20, fma(l31, m10, l30 * m00)) * m33;
    m[0] = m00; m[1] = m10; m[2] = m20; m[3] = m30; m[4] = m11; m[5] = m21; m[6] = m31; m[7] = m22;
    m[8] = m32; m[9] = m33;
    // ---- s1 = |M|_F^2 ; W = M^T M ; s2 = |W|_F^2
    const double w00 = fma(m00, m00, fma(m10, m10, fma(m20, m20, m30 * m30)));
    const double w11 = fma(m11, m11, fma(m21, m21, m31 * m31));
    const double w22 = fma(m22, m22, m32 * m32);
    const double w33 = m33 * m33;
    const double s1 = (w00 + w11) + (w22 + w33);
    if (it == 0 && s1 * fma(2e-12, tr, lamcut) < 1.0) return false;
    s1_last = s1;
    const double w01 = fma(m10, m11, fma(m20, m21, m30 * m31));
    const double w02 = fma(m20, m22, m30 * m32);
    const double w03 = m30 * m33;
    const double w12 = fma(m21, m22, m31 * m32);
    const double w13 = m31 * m33;
    const double w23 = m32 * m33;
    const double sd = fma(w00, w00, fma(w11, w11, fma(w22, w22, w33 * w33)));
    const double so = fma(w01, w01, fma(w02, w02, fma(w03, w03, fma(w12, w12, fma(w13, w13, w23 * w23)))));
    const double s2 = fma(2.0, so, sd);
    const double s1sq = s1 * s1;
    if (!(s1sq - s2 > 1e-3 * s1sq)) break;
    // Laguerre step 4 / (s1 + sqrt(3 (4 s2 - s1^2))).  Only the shift depends on it, so raw
    // v_rsq / v_rcp estimates (2^-26) are plenty; the step is shortened by 2^-20 so the estimate
    // errors can never carry the shift past lam1.
    const double disc = fmax(3.0 * fma(4.0, s2, -s1sq), 1e-300);
    const double den = fma(disc, __builtin_amdgcn_rsq(disc), s1);
    lam = fma(4.0 - 0x1p-18, __builtin_amdgcn_rcp(den), lam);
  }
  // ---- inverse iteration: x_1 = (B - lam I)^{-1} e4 ~ last row of M ; x_{k+1} = M^T (sc M x_k).
  // sc = last Cholesky pivot = (lam1 - lam) / v4^2 up to O(rho): |x| stays of order one however
  // tight the shift is (no overflow), at 4 multiplies per step.
  double x0 = m[3], x1 = m[6], x2 = m[8], x3 = m[9];
#pragma unroll
  for (int k = 1; k < kInvIters; k++) {
    const double y0 = (m[0] * x0) * piv3;
    const double y1 = fma(m[1], x0, m[4] * x1) * piv3;
    const double y2 = fma(m[2], x0, fma(m[5], x1, m[7] * x2)) * piv3;
    const double y3 = fma(m[3], x0, fma(m[6], x1, fma(m[8], x2, m[9] * x3))) * piv3;
    x0 = fma(m[0], y0, fma(m[1], y1, fma(m[2], y2, m[3] * y3)));
    x1 = fma(m[4], y1, fma(m[5], y2, m[6] * y3));
    x2 = fma(m[7], y2, m[8] * y3);
    x3 = m[9] * y3;
  }
  out[0] = x0;
  out[1] = x1;
  out[2] = x2;
  out[3] = x3;
  lam_lb = fma(1.0 - 1e-5, __builtin_amdgcn_rcp(s1_last), lam) - 2e-12 * tr;
  return true;
}

// s1 = trace(B^-1) from the Cholesky factor of B itself (EigCut's first test on its own: the frame kernel's
// branch-and-bound runs it on PARTIAL groups): lam1(B) >= 1 / s1.  tr = trace(B) for the rounding allowance.
__device__ __forceinline__ double eigcut_s1(const double (&a)[10], double& tr) {
  tr = (a[0] + a[4]) + (a[7] + a[9]);
  const double floor_piv = tr * 1e-30 + 1e-300;
  const double r0 = rsqrt_pos(fmax(a[0], floor_piv));
  const double l10 = a[1] * r0, l20 = a[2] * r0, l30 = a[3] * r0;
  const double r1 = rsqrt_pos(fmax(fma(-l10, l10, a[4]), floor_piv));
  const double l21 = fma(-l20, l10, a[5]) * r1, l31 = fma(-l30, l10, a[6]) * r1;
  const double r2 = rsqrt_pos(fmax(fma(-l21, l21, fma(-l20, l20, a[7])), floor_piv));
  const double l32 = fma(-l31, l21, fma(-l30, l20, a[8])) * r2;
  const double r3 = rsqrt_pos(fmax(fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, a[9]))), floor_piv));
  const double m10 = -(l10 * r0) * r1;
  const double m21 = -(l21 * r1) * r2;
  const double m32 = -(l32 * r2) * r3;
  const double m20 = -fma(l21, m10, l20 * r0) * r2;
  const double m31 = -fma(l32, m21, l31 * r1) * r3;
  const double m30 = -fma(l32, m20, fma(l31, m10, l30 * r0)) * r3;
  const double w00 = fma(r0, r0, fma(m10, m10, fma(m20, m20, m30 * m30)));
  const double w11 = fma(r1, r1, fma(m21, m21, m31 * m31));
  const double w22 = fma(r2, r2, m32 * m32);
  return (w00 + w11) + (w22 + r3 * r3);
}

// The same after moving the origin of the world to c (EigCut's bound holds in ANY homogeneous frame x = M x',
// M = [[I, c], [0, 1]]:  x^T B x = x'^T (M^T B M) x',  depths z_c = (P_c[2] M) . x'): with the origin inside the
// working volume |x'| stays near one and max |P_c[2] M| near the camera distance, which makes the X-free form of the
// bound 20 times tighter on a rig whose world frame sits in its first camera (median bound / error 0.03 -> 0.61).
// tr: allowance scale for the rounding of B and of the transform.
__device__ __forceinline__ double eigcut_s1_shifted(const double (&a)[10], const double (&c)[3], double& tr) {
  double b[10];
  b[0] = a[0]; b[1] = a[1]; b[2] = a[2]; b[4] = a[4]; b[5] = a[5]; b[7] = a[7];
  b[3] = fma(a[0], c[0], fma(a[1], c[1], fma(a[2], c[2], a[3])));
  b[6] = fma(a[1], c[0], fma(a[4], c[1], fma(a[5], c[2], a[6])));
  b[8] = fma(a[2], c[0], fma(a[5], c[1], fma(a[7], c[2], a[8])));
  b[9] = fma(c[0], b[3] + a[3], fma(c[1], b[6] + a[6], fma(c[2], b[8] + a[8], a[9])));
  double tr1;
  const double s1 = eigcut_s1(b, tr1);
  const double c2 = fma(c[0], c[0], fma(c[1], c[1], fma(c[2], c[2], 1.0)));
  tr = tr1 + 2.0 * c2 * ((a[0] + a[4]) + (a[7] + a[9]));
  return s1;
}

#ifdef MOCAP_EIG_JACOBI
__device__ __forceinline__ bool smallest_eigvec4(double (&a)[10], double (&out)[4], double, double& lam_lb) {
  smallest_eigvec4_jacobi(a, out);
  lam_lb = 0.0;  // no bound: nothing is ever cut
  return true;
}
#else
__device__ __forceinline__ bool smallest_eigvec4(double (&a)[10], double (&out)[4], double lamcut, double& lam_lb) {
  return smallest_eigvec4_cholesky(a, out, lamcut, lam_lb);
}
#endif

// Exact cut-off of candidate selection from the DLT matrix alone (frame path; everything else passes EigCut{}).
// For ANY homogeneous point x = (X, 1):  x^T B x = sum over the views of z_c^2 (du_c^2 + dv_c^2)  (z_c = P_c[2].x, the
// depth; rows ra.x = z (v - v^), rb.x = -z (u - u^) of helpers.py:315-316 when K = [[fx,0,cx],[0,fy,cy],[0,0,1]], which
// makes the projection of P = K[R|t] the one cv.projectPoints computes -- the host switches the cut off otherwise),
// and x^T B x >= lam1 |x|^2.  Hence  sum of squared residuals >= lam1 |x|^2 / max_c z_c^2  >= lam1 / max_c |P_c[2]|^2
// (Cauchy-Schwarz).  A candidate whose bound exceeds the best error found so far for its root can never be selected,
// so the rest of its evaluation is skipped: after the first factorisation with the X-free form (p3max2), after the
// null vector with the depths of the point that would be reprojected.  `limit` below is the bound on the SUM the
// reprojection cut-off uses; the allowances: float32 rounding of the projections (|fl(p) - p| <= 2^-24 |p|,
// |p| <= |o| + |d|  =>  sqrt(S32) >= (1 - 2^-23) sqrt(S) - sqrt(2v) 2^-23 omax, squared with (a+b)^2 <= (1+h) a^2 +
// (1+1/h) b^2, h = 2^-10), every other rounding inside the factor 1.002.
struct EigCut {
  double p3max2 = 0.0;   // (1 + 1e-5) * max over the cameras of |P[2]|^2; 0 = cut-off disabled
  double o2slack = 0.0;  // 1100 * 2^-46 * omax^2, omax = largest |coordinate| among the frame's blobs
};

// DLT contribution of one view: rows ra = y*P2 - P1 and rb = P0 - x*P2 (helpers.py:315-316),
// Bc = ra ra^T + rb rb^T (packed symmetric).  B = A^T A is the sum of the views' contributions in camera
// order (helpers.py:318 hands A^T A to LAPACK: its internal order is not pinned).  Keeping the contribution
// a function of (camera, observation) alone lets the frame kernel tabulate it once per blob instead of once
// per candidate group -- and every path (table or not, frame or explicit triangulation) rounds identically.
template <class Tab>
__device__ __forceinline__ void dlt_contribution(double (&Bc)[10], Tab P, double x, double y) {
  double ra[4], rb[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    ra[k] = y * P[8 + k] - P[4 + k];  // unfused on purpose: an fma would need two SGPR operands
    rb[k] = P[k] - x * P[8 + k];      // (constant-bus limit 1 on gfx9 -> extra v_mov pairs)
  }
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = i; j < 4; j++) Bc[sidx(i, j)] = fma(ra[i], ra[j], rb[i] * rb[j]);
}

template <class Tab>
__device__ __forceinline__ void dlt_accumulate(double (&B)[10], Tab P, double x, double y) {
  double Bc[10];
  dlt_contribution(Bc, P, x, y);
#pragma unroll
  for (int e = 0; e < 10; e++) B[e] = B[e] + Bc[e];
}

// cv.projectPoints restated (helpers.py:231-237; OpenCV cvProjectPoints2, 3x3 R, no distortion):
// squared pixel residuals of one view.  X already rounded to float32 when F32R.
template <bool F32R, class Tab>
__device__ __forceinline__ void reproject_sq(Tab RT, ctab_t K4, const double (&X)[3], double ox,
                                             double oy, double& du2, double& dv2) {
  double x = RT[0] * X[0] + RT[1] * X[1] + RT[2] * X[2] + RT[9];
  double y = RT[3] * X[0] + RT[4] * X[1] + RT[5] * X[2] + RT[10];
  double z = RT[6] * X[0] + RT[7] * X[1] + RT[8] * X[2] + RT[11];
  z = z != 0.0 ? div_by(1.0, z, recip_refined(z)) : 1.0;  // OpenCV: z = z ? 1./z : 1
  x *= z;
  y *= z;
  double pu = x * K4[0] + K4[2];
  double pv = y * K4[1] + K4[3];
  if (F32R) {
    pu = (double)(float)pu;
    pv = (double)(float)pv;
  }
  const double du = ox - pu, dv = oy - pv;
  du2 = du * du;
  dv2 = dv * dv;
}

// Triangulate one correspondence group and score it.
//   obs1(c, x, y) / obs2(c, x, y) -> true when camera c sees the point; obs1 feeds the DLT pass,
//   obs2 the reprojection pass (each called once per camera, ascending, c wave-uniform).
//   PAIRWISE: sum the squared residuals in NumPy's pairwise order when every camera is seen
//   (float64 array path of errors.mean(), helpers.py:241); otherwise left to right.
//   F32R: reproduce OpenCV's float32 roundings (MOCAP_OPT_F32_ROUNDING).
// Returns the number of views; X / err are valid when it is >= 2.
// Second half of triangulate_and_score: null vector of B (v views accumulated), point, reprojection error.
template <bool UNIFORM_K, bool PAIRWISE, bool F32R, int BATCH = 1, class View, class Obs2>
__device__ __forceinline__ void score_point(const View& cv, int v, Obs2&& obs2, const double (&X)[3], double& err,
                                            double limit);

// `limit` (sum of squared residuals, +inf = off): candidate selection only needs to know whether this group beats
// the best one so far, and the residuals are a sum of non-negative terms -- once the running left-to-right sum
// exceeds `limit` the group cannot win, and the remaining views are skipped (err = +inf).  The callers put a safety
// factor on the limit, so a group within rounding distance of the best is never cut short.
// DEPTH_CUT: after the null vector, the bound is taken again with the depths of the point that would be reprojected
// (EigCut's second form).  It pays in the exhaustive walk, where every candidate comes this far; behind the block search
// of frame_bb.hip, whose survivors are mostly near-winners, it costs more than it cuts (5.77 vs 6.00 ms per 100 k
// frames, round 3) and is compiled out there.
// BATCH > 1: obs2 is a functor with raw(c) / decode(raw, x, y) next to operator() -- see triangulate_and_score.
template <bool UNIFORM_K, bool PAIRWISE, bool F32R, bool DEPTH_CUT = true, int BATCH = 1, class View, class Obs2>
__device__ __forceinline__ void solve_and_score(const View& cv, double (&B)[10], int v, Obs2&& obs2,
                                                double (&X)[3], double& err,
                                                double limit = __builtin_huge_val(), const EigCut& ec = EigCut{}) {
  const double inf = __builtin_huge_val();
  const bool cut = ec.p3max2 > 0.0;
  // limit = +inf (no finite error for the root yet) keeps every comparison below false
  const double limit_adj = fma(1.002, limit, (double)(2 * v) * ec.o2slack);
  double vec[4], lam_lb;
  if (!smallest_eigvec4(B, vec, cut ? ec.p3max2 * limit_adj : inf, lam_lb)) {
    X[0] = X[1] = X[2] = 0.0;
    err = inf;  // like a reprojection that was cut short
    return;
  }
  const double rw = recip_refined(vec[3]);
  X[0] = div_by(vec[0], vec[3], rw);  // helpers.py:321
  X[1] = div_by(vec[1], vec[3], rw);
  X[2] = div_by(vec[2], vec[3], rw);
  if (cut && DEPTH_CUT) {
    double Xp[3] = {X[0], X[1], X[2]};
    if (F32R) {
      Xp[0] = (double)(float)X[0];
      Xp[1] = (double)(float)X[1];
      Xp[2] = (double)(float)X[2];
    }
    const double n2 = fma(Xp[0], Xp[0], fma(Xp[1], Xp[1], fma(Xp[2], Xp[2], 1.0)));
    double zmax2 = 1e-14 * (n2 * ec.p3max2);  // rounding of the depths below
    if constexpr (BATCH > 1) {
      const int C = cv.C;
      for (int c0 = 0; c0 < C; c0 += BATCH) {
        unsigned long long rw[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; u++) rw[u] = obs2.raw(c0 + u < C ? c0 + u : C - 1);
#pragma unroll
        for (int u = 0; u < BATCH; u++) {
          double ox, oy;
          if (c0 + u < C && obs2.decode(rw[u], ox, oy)) {
            const auto RT = cv.rt(12 * (c0 + u));
            const double z = fma(RT[6], Xp[0], fma(RT[7], Xp[1], fma(RT[8], Xp[2], RT[11])));
            zmax2 = fmax(zmax2, z * z);
          }
        }
      }
    } else {
      for (int c = 0; c < cv.C; c++) {
        double ox, oy;
        if (obs2(c, ox, oy)) {
          const auto RT = cv.rt(12 * c);
          const double z = fma(RT[6], Xp[0], fma(RT[7], Xp[1], fma(RT[8], Xp[2], RT[11])));
          zmax2 = fmax(zmax2, z * z);
        }
      }
    }
    if (lam_lb * n2 > zmax2 * limit_adj) {
      err = inf;
      return;
    }
  }
  score_point<UNIFORM_K, PAIRWISE, F32R, BATCH>(cv, v, obs2, X, err, limit);
}

// The point alone (helpers.py:318-321) of a group whose DLT matrix is B: solve_and_score's arithmetic up to the point,
// nothing cut (a caller that knows the group's error already and did not keep its point).
__device__ __forceinline__ void solve_point(double (&B)[10], double (&X)[3]) {
  double vec[4], lam_lb;
  smallest_eigvec4(B, vec, __builtin_huge_val(), lam_lb);
  const double rw = recip_refined(vec[3]);
  X[0] = div_by(vec[0], vec[3], rw);
  X[1] = div_by(vec[1], vec[3], rw);
  X[2] = div_by(vec[2], vec[3], rw);
}

// calculate_reprojection_error (helpers.py:214-241) of a GIVEN point X seen by v cameras.
template <bool UNIFORM_K, bool PAIRWISE, bool F32R, int BATCH, class View, class Obs2>
__device__ __forceinline__ void score_point(const View& cv, int v, Obs2&& obs2, const double (&X)[3], double& err,
                                            double limit) {
  const int C = cv.C;
  double Xp[3] = {X[0], X[1], X[2]};
  if (F32R) {
    Xp[0] = (double)(float)X[0];  // helpers.py:232 `.astype(np.float32)`
    Xp[1] = (double)(float)X[1];
    Xp[2] = (double)(float)X[2];
  }
  // NumPy's pairwise sum of the n = 2C residual components (n < 8: plain loop; else eight partial
  // sums over the leading n - n % 8 values, a fixed tree, then the rest one by one).  Cameras come
  // in fours (8 components); `seq` is the left-to-right sum used when a view is missing.
  const bool pw = PAIRWISE && v == C && 2 * C >= 8;
  const int C4 = C & ~3;
  double seq = 0.0, r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int j = 0;
  for (int c0 = 0; c0 < C4; c0 += 4) {
    if (!(seq <= limit)) continue;  // cut short: whole waves skip the block once every lane is out
    unsigned long long rw[4] = {0, 0, 0, 0};
    if constexpr (BATCH > 1) {  // the four observations are fetched before any of them is used
#pragma unroll
      for (int u = 0; u < 4; u++) rw[u] = obs2.raw(c0 + u);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      double x, y;
      bool seen;
      if constexpr (BATCH > 1)
        seen = obs2.decode(rw[u], x, y);
      else
        seen = obs2(c0 + u, x, y);
      if (seen) {
        double du2, dv2;
        reproject_sq<F32R>(cv.rt(12 * (c0 + u)), cv.k4(4 * (UNIFORM_K ? 0 : j)), Xp, x, y, du2, dv2);
        seq = seq + du2;
        seq = seq + dv2;
        if (PAIRWISE) {
          r[2 * u] = r[2 * u] + du2;
          r[2 * u + 1] = r[2 * u + 1] + dv2;
        }
        j++;
      }
    }
  }
  double spw = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (int c = C4; c < C; c++) {
    double x, y;
    if (obs2(c, x, y)) {
      double du2, dv2;
      reproject_sq<F32R>(cv.rt(12 * c), cv.k4(4 * (UNIFORM_K ? 0 : j)), Xp, x, y, du2, dv2);
      seq = seq + du2;
      seq = seq + dv2;
      if (PAIRWISE) {
        spw = spw + du2;
        spw = spw + dv2;
      }
      j++;
    }
  }
  err = (seq <= limit) ? (pw ? spw : seq) / (double)(2 * v) : __builtin_huge_val();
}

// BATCH > 1: the observations of BATCH cameras are fetched before any of them is used -- for callers whose observations
// are reads of device memory (wide frames keep a lane's group in an HBM-resident column: one dependent round trip per
// camera and pass otherwise, three passes per candidate).  The observation source is then a functor with
//   raw(c) -> the camera's 8 bytes, loaded unconditionally;  decode(raw, x, y) -> seen?;  operator()(c, x, y)
// (a lambda that tests x before it reads y compiles to two dependent loads per view, each followed by a wait -- measured).
// The views are still accumulated in camera order, so the result is the same to the bit.
template <bool UNIFORM_K, bool PAIRWISE, bool F32R, int BATCH = 1, bool DEPTH_CUT = true, class View, class Obs1, class Obs2>
__device__ __forceinline__ int triangulate_and_score(const View& cv, Obs1&& obs1, Obs2&& obs2,
                                                     double (&X)[3], double& err,
                                                     double limit_e = __builtin_huge_val(), const EigCut& ec = EigCut{}) {
  const int C = cv.C;
  double B[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int v = 0;
  if constexpr (BATCH > 1) {
    for (int c0 = 0; c0 < C; c0 += BATCH) {
      unsigned long long rw[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; u++) rw[u] = obs1.raw(c0 + u < C ? c0 + u : C - 1);
#pragma unroll
      for (int u = 0; u < BATCH; u++) {
        double x, y;
        if (c0 + u < C && obs1.decode(rw[u], x, y)) {
          dlt_accumulate(B, cv.pq(UNIFORM_K ? (size_t)12 * (c0 + u) : 12 * ((size_t)v * C + (c0 + u))), x, y);
          v++;
        }
      }
    }
  } else {
    for (int c = 0; c < C; c++) {
      double x, y;
      if (obs1(c, x, y)) {
        dlt_accumulate(B, cv.pq(UNIFORM_K ? (size_t)12 * c : 12 * ((size_t)v * C + c)), x, y);
        v++;
      }
    }
  }
  if (v <= 1) return v;  // helpers.py:300
  solve_and_score<UNIFORM_K, PAIRWISE, F32R, DEPTH_CUT, BATCH>(cv, B, v, obs2, X, err, limit_e * (double)(2 * v) * (1.0 + 0x1p-40), ec);
  return v;
}

// The same with the views' DLT contributions already tabulated (frame kernel, identical intrinsics):
// contrib(c, B) adds camera c's contribution for the group's blob to B and returns true, or returns false
// when the camera is not in the group.
template <bool PAIRWISE, bool F32R, class Contrib, class Obs2>
__device__ __forceinline__ int triangulate_and_score_tab(const CamView& cv, Contrib&& contrib, Obs2&& obs2,
                                                         double (&X)[3], double& err,
                                                         double limit_e = __builtin_huge_val(), const EigCut& ec = EigCut{}) {
  const int C = cv.C;
  double B[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int v = 0;
  for (int c = 0; c < C; c++) v += contrib(c, B) ? 1 : 0;
  if (v <= 1) return v;
  // limit_e is a bound on the ERROR (mean of 2 v squares) -> bound on their sum, with room for the rounding of
  // the two summation orders (~2 v ulp) and of the division
  solve_and_score<true, PAIRWISE, F32R>(cv, B, v, obs2, X, err, limit_e * (double)(2 * v) * (1.0 + 0x1p-40), ec);
  return v;
}

}  // namespace mocap
