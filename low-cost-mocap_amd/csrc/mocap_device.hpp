// mocap_device.hpp -- per-lane FP64 geometry core shared by every kernel of the path.
//
// Replaces, per candidate correspondence group (one wave lane = one group):
//   * triangulate_point + DLT            (reference computer_code/api/helpers.py:293-327)
//   * calculate_reprojection_error       (helpers.py:214-241, incl. cv.projectPoints)
//
// Design (gfx950): everything for one candidate lives in VGPRs (10-entry packed symmetric
// B = A^T A, 16-entry eigenvector accumulator); camera tables are read with wave-uniform
// addresses so they come in over the scalar cache (s_load) when all intrinsics are equal.
// The 4x4 null vector is a cyclic Jacobi eigen-solve with compile-time rotation indices
// (no dynamic register indexing, no scratch).  FP64 throughout: B squares the condition
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
};

// ---- packed symmetric 4x4: (0,0)=0 (0,1)=1 (0,2)=2 (0,3)=3 (1,1)=4 (1,2)=5 (1,3)=6 (2,2)=7 (2,3)=8 (3,3)=9
__host__ __device__ constexpr int sidx(int i, int j) {
  return i <= j ? (i == 0 ? j : i == 1 ? 3 + j : i == 2 ? 5 + j : 9)
                : (j == 0 ? i : j == 1 ? 3 + i : j == 2 ? 5 + i : 9);
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
__device__ __forceinline__ void smallest_eigvec4(double (&a)[10], double (&out)[4]) {
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

// DLT accumulation of one view: rows y*P2 - P1 and P0 - x*P2 (helpers.py:315-316) into B.
__device__ __forceinline__ void dlt_accumulate(double (&B)[10], ctab_t P, double x,
                                               double y) {
  double ra[4], rb[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    ra[k] = y * P[8 + k] - P[4 + k];
    rb[k] = P[k] - x * P[8 + k];
  }
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = i; j < 4; j++) B[sidx(i, j)] = fma(ra[i], ra[j], fma(rb[i], rb[j], B[sidx(i, j)]));
}

// cv.projectPoints restated (helpers.py:231-237; OpenCV cvProjectPoints2, 3x3 R, no distortion):
// squared pixel residuals of one view.  X already rounded to float32 when f32_rounding.
__device__ __forceinline__ void reproject_sq(ctab_t RT, ctab_t K4,
                                             const double (&X)[3], double ox, double oy, bool f32r,
                                             double& du2, double& dv2) {
  double x = RT[0] * X[0] + RT[1] * X[1] + RT[2] * X[2] + RT[9];
  double y = RT[3] * X[0] + RT[4] * X[1] + RT[5] * X[2] + RT[10];
  double z = RT[6] * X[0] + RT[7] * X[1] + RT[8] * X[2] + RT[11];
  z = z != 0.0 ? 1.0 / z : 1.0;
  x *= z;
  y *= z;
  double pu = x * K4[0] + K4[2];
  double pv = y * K4[1] + K4[3];
  if (f32r) {
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
// Returns the number of views; X / err are valid when it is >= 2.
template <bool UNIFORM_K, bool PAIRWISE, class Obs1, class Obs2>
__device__ __forceinline__ int triangulate_and_score(const CamView& cv, Obs1&& obs1, Obs2&& obs2,
                                                     double (&X)[3], double& err) {
  const int C = cv.C;
  double B[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int v = 0;
  for (int c = 0; c < C; c++) {
    double x, y;
    if (obs1(c, x, y)) {
      ctab_t P = as_ctab(UNIFORM_K ? cv.Pq + 12 * c : cv.Pq + 12 * ((size_t)v * C + c));
      dlt_accumulate(B, P, x, y);
      v++;
    }
  }
  if (v <= 1) return v;  // helpers.py:300
  double vec[4];
  smallest_eigvec4(B, vec);
  X[0] = vec[0] / vec[3];  // helpers.py:321
  X[1] = vec[1] / vec[3];
  X[2] = vec[2] / vec[3];

  const bool f32r = cv.f32_rounding != 0;
  double Xp[3] = {X[0], X[1], X[2]};
  if (f32r) {
    Xp[0] = (double)(float)X[0];  // helpers.py:232 `.astype(np.float32)`
    Xp[1] = (double)(float)X[1];
    Xp[2] = (double)(float)X[2];
  }
  const bool pw = PAIRWISE && v == C && 2 * C >= 8;
  double seq = 0.0, r[8] = {0, 0, 0, 0, 0, 0, 0, 0}, spw = 0.0;
  int j = 0;
  for (int c0 = 0; c0 < C; c0 += 4) {
    const bool full_chunk = c0 + 4 <= C;
    if (!full_chunk) spw = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int c = c0 + u;
      if (c < C) {
        double x, y;
        if (obs2(c, x, y)) {
          double du2, dv2;
          reproject_sq(as_ctab(cv.RT + 12 * c), as_ctab(cv.K4 + 4 * (UNIFORM_K ? 0 : j)), Xp, x, y, f32r, du2, dv2);
          seq = seq + du2;
          seq = seq + dv2;
          if (full_chunk) {
            r[2 * u] = r[2 * u] + du2;
            r[2 * u + 1] = r[2 * u + 1] + dv2;
          } else {
            spw = spw + du2;
            spw = spw + dv2;
          }
          j++;
        }
      }
    }
  }
  if ((C & 3) == 0) spw = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  err = (pw ? spw : seq) / (double)(2 * v);
  return v;
}

}  // namespace mocap
