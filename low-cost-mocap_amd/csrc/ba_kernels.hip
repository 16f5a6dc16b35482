// ba_kernels.hip -- bundle-adjustment building blocks around the batched residual kernel
// (csrc/tri_kernel.hip).  Replaces the inner work of
//   bundle_adjustment / residual_function          (reference computer_code/api/helpers.py:244-290)
//   scipy.optimize.least_squares(method="trf", loss="cauchy") linearisation: 2-point finite
//   differences (scipy/optimize/_numdiff.py), Cauchy scaling (scipy/optimize/_lsq/least_squares.py,
//   _lsq/common.py scale_for_robust_loss_function) and the dense J^T J / J^T f contraction.
//
// The contraction runs on the FP64 matrix cores (v_mfma_f64_16x16x4_f64): the Jacobian is
// stored augmented, Jaug = [ J | f | 0-pad ] with a row length NP that is a multiple of 16, so
// one Gram product G = Jaug^T Jaug yields J^T J, J^T f and f^T f together.  Reduction over the
// row (K) dimension is split across waves and finished in a fixed order -> bit-reproducible.
#include "kernels.hpp"

namespace mocap {

// ---------------------------------------------------------------- parameter perturbation
// scipy _numdiff._compute_absolute_step / _dense_difference (2-point):
//   h_j = rel_step * sign(x_j) * max(1, |x_j|),   dx_j = (x_j + h_j) - x_j
__global__ void ba_perturb_kernel(const double* __restrict__ x, int n, double rel_step,
                                  double* __restrict__ params, double* __restrict__ hvec) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;  // 0..n : row of params
  if (j > n) return;
  for (int k = 0; k < n; k++) {
    double v = x[k];
    if (j >= 1 && k == j - 1) {
      const double sgn = v >= 0.0 ? 1.0 : -1.0;
      const double h = rel_step * sgn * fmax(1.0, fabs(v));
      const double x1 = v + h;
      hvec[k] = x1 - v;
      v = x1;
    }
    params[(size_t)j * n + k] = v;
  }
}

hipError_t launch_ba_perturb(const double* x, int n, double rel_step, double* params, double* hvec,
                             hipStream_t stream) {
  hipLaunchKernelGGL(ba_perturb_kernel, dim3((n + 1 + 63) / 64), dim3(64), 0, stream, x, n, rel_step,
                     params, hvec);
  return hipGetLastError();
}

// ---------------------------------------------------------------- parameters -> camera tables
// params_to_camera_poses (helpers.py:247-262): camera 0 = (I, 0); camera i = (from_rotvec, t).
// Rotation.from_rotvec(...).as_matrix() restated: rotvec -> quaternion (series below 1e-3 rad)
// -> matrix.
__device__ inline void rotvec_to_matrix(const double* rv, double* R) {
  const double angle = sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
  double scale;
  if (angle <= 1e-3) {
    const double a2 = angle * angle;
    scale = 0.5 - a2 / 48 + a2 * a2 / 3840;
  } else {
    scale = sin(angle / 2) / angle;
  }
  const double x = scale * rv[0], y = scale * rv[1], z = scale * rv[2], w = cos(angle / 2);
  const double x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w;
  const double xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
  R[0] = x2 - y2 - z2 + w2;
  R[1] = 2 * (xy - zw);
  R[2] = 2 * (xz + yw);
  R[3] = 2 * (xy + zw);
  R[4] = -x2 + y2 - z2 + w2;
  R[5] = 2 * (yz - xw);
  R[6] = 2 * (xz - yw);
  R[7] = 2 * (yz + xw);
  R[8] = -x2 - y2 + z2 + w2;
}

// Pose parameters of camera `cam` in parameter set p of a forward-difference batch around xb: set 0 = xb,
// set 1 + j = xb + h_j e_j with scipy's step rule (see ba_perturb_kernel) -> rt = [R row-major | t].
__device__ inline void ba_fd_camera_pose(const double* xb, double rel_step, int p, int cam, double (&rt)[12]) {
  if (cam == 0) {
    for (int k = 0; k < 12; k++) rt[k] = 0.0;
    rt[0] = rt[4] = rt[8] = 1.0;
    return;
  }
  const int j = p - 1;  // perturbed parameter of this set (-1: none)
  double q[6];
  for (int k = 0; k < 6; k++) {
    const int idxp = (cam - 1) * 7 + 2 + k;
    double v = xb[idxp];
    if (idxp == j) v = v + rel_step * (v >= 0.0 ? 1.0 : -1.0) * fmax(1.0, fabs(v));
    q[k] = v;
  }
  rotvec_to_matrix(q, rt);
  rt[9] = q[3];
  rt[10] = q[4];
  rt[11] = q[5];
}

// one row-block of P = K @ [R | t] (helpers.py:306-307)
__device__ inline void ba_projection(const double* K, const double (&rt)[12], double* P) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) {
      const double r0 = c < 3 ? rt[c] : rt[9], r1 = c < 3 ? rt[3 + c] : rt[10], r2 = c < 3 ? rt[6 + c] : rt[11];
      P[r * 4 + c] = K[r * 3 + 0] * r0 + K[r * 3 + 1] * r1 + K[r * 3 + 2] * r2;
    }
}

__global__ void ba_build_cameras_kernel(BaCamArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.P * a.C) return;
  const int p = idx / a.C, cam = idx - p * a.C;
  double rt[12];
  // the camera's six pose parameters: from the batch, or base point + this set's forward-difference step
  // (ba_perturb_kernel's rule, fused: one launch less per linearisation)
  if (a.params) {
    const double* x = a.params + (size_t)p * a.n;
    double q[6];
    for (int k = 0; k < 6; k++) q[k] = cam ? x[(cam - 1) * 7 + 2 + k] : 0.0;
    if (cam == 0) {
      for (int k = 0; k < 12; k++) rt[k] = 0.0;
      rt[0] = rt[4] = rt[8] = 1.0;
    } else {
      rotvec_to_matrix(q, rt);
      rt[9] = q[3];
      rt[10] = q[4];
      rt[11] = q[5];
    }
  } else {
    const int j = p - 1;
    const double* xb = a.x ? a.x : a.x_inline;
    ba_fd_camera_pose(xb, a.rel_step, p, cam, rt);
    if (cam == 0 && j >= 0) {  // dx_j = (x_j + h_j) - x_j, also for the dead focal entries
      const double v = xb[j];
      const double x1 = v + a.rel_step * (v >= 0.0 ? 1.0 : -1.0) * fmax(1.0, fabs(v));
      a.hvec[j] = x1 - v;
    }
  }
  double* RT = a.RT + (size_t)p * a.stride_RT + 12 * cam;
  for (int k = 0; k < 12; k++) RT[k] = rt[k];
  // P = K[j] @ [R | t] (helpers.py:306-307); j = compacted view index <= cam
  const int jn = a.uniformK ? 1 : cam + 1;
  for (int j = 0; j < jn; j++)
    ba_projection(a.K + 9 * j, rt, a.Pq + (size_t)p * a.stride_Pq + 12 * (a.uniformK ? (size_t)cam : (size_t)j * a.C + cam));
}

hipError_t launch_ba_build_cameras(const BaCamArgs& a, hipStream_t stream) {
  const int total = a.P * a.C;
  hipLaunchKernelGGL(ba_build_cameras_kernel, dim3((total + 63) / 64), dim3(64), 0, stream, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------- Jacobian + robust scaling
// J[i][j] = (r_j[i] - r_0[i]) / dx_j.  Cauchy loss (f_scale = 1): z = f^2, rho0 = log1p(z), rho1 = 1/(1+z),
// rho2 = -1/(1+z)^2; J_scale = sqrt(max(rho1 + 2 rho2 f^2, EPS)); f <- f rho1 / J_scale ; J <- J_scale J.
//
// f32_residuals: the reference hands scipy a float32 residual vector (helpers.py:273), and NumPy's type rules
// then decide the precision of every intermediate.  Restated operation by operation (NumPy 2 promotion):
//   _numdiff._dense_difference:  df = fun(x1) - f0          float32 - float32 -> float32
//                                J^T[i] = df / dx            float32 array / np.float64 scalar -> float64
//   least_squares.loss_function: z = (f / f_scale) ** 2      float32
//   least_squares.cauchy:        rho[0] = log1p(z), t = 1 + z, rho[1] = 1 / t, rho[2] = -1 / t**2
//                                                            all float32, stored into the float64 `rho` array
//   common.scale_for_robust_loss_function:
//                                J_scale = rho[1] + 2 * rho[2] * f**2    float64 (f**2 is the float32 z)
//                                f *= rho[1] / J_scale       float64 product, rounded into the float32 `f`
// float32 log1p is taken correctly rounded (libm / SVML versions differ in the last bit; unpinned).
struct CauchyTerms {
  double f;       // residual as the optimizer sees it (float32-rounded when emulated)
  double rho0;    // loss value
  double jscale;  // row scale of J
  double fs;      // scaled residual
};
__device__ inline CauchyTerms cauchy_terms(double r, int f32_residuals, int use_cauchy) {
  CauchyTerms c;
  if (f32_residuals) {
    const float f = (float)r;
    const float z = f * f;
    c.f = (double)f;
    if (!use_cauchy) {  // loss="linear": cost = 0.5 * np.dot(f, f) on the float32 vector
      c.rho0 = (double)z;
      c.jscale = 1.0;
      c.fs = (double)f;
      return c;
    }
    const float t = 1.0f + z;
    const float rho1 = 1.0f / t;
    const float rho2 = -1.0f / (t * t);
    c.rho0 = (double)(float)log1p((double)z);
    double js = (double)rho1 + (2.0 * (double)rho2) * (double)z;
    if (js < 2.220446049250313e-16) js = 2.220446049250313e-16;
    js = sqrt(js);
    c.jscale = js;
    c.fs = (double)(float)((double)f * ((double)rho1 / js));
    return c;
  }
  c.f = r;
  const double z = r * r;
  if (!use_cauchy) {
    c.rho0 = z;
    c.jscale = 1.0;
    c.fs = r;
    return c;
  }
  const double t = 1.0 + z;
  const double rho1 = 1.0 / t, rho2 = -1.0 / (t * t);
  c.rho0 = log1p(z);
  double js = rho1 + 2.0 * rho2 * z;
  if (js < 2.220446049250313e-16) js = 2.220446049250313e-16;
  js = sqrt(js);
  c.jscale = js;
  c.fs = r * (rho1 / js);
  return c;
}

__global__ __launch_bounds__(256) void ba_jacobian_kernel(BaJacArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // valid-row index (fastest)
  const int j = blockIdx.y;                                          // column of Jaug
  const int64_t m_pad = (a.m + 3) / 4 * 4;
  if (i >= m_pad) return;
  double out = 0.0;
  if (i < a.m && j <= a.n) {
    const int64_t idx = a.valid[i];
    const double r0 = a.r[idx];
    const CauchyTerms c = cauchy_terms(r0, a.f32_residuals, a.use_cauchy);
    if (j == a.n) {
      out = c.fs;
      if (a.rho0) a.rho0[i] = c.rho0;
    } else {
      const double fj = a.r[(size_t)(1 + j) * a.N + idx];
      double df;
      if (a.f32_residuals)
        df = (double)((float)fj - (float)r0);
      else
        df = fj - r0;
      out = (df / a.hvec[j]) * c.jscale;
    }
  }
  if (j < a.NP) a.Jaug[(size_t)i * a.NP + j] = out;
}

hipError_t launch_ba_jacobian(const BaJacArgs& a, hipStream_t stream) {
  const int64_t m_pad = (a.m + 3) / 4 * 4;
  if (m_pad == 0) return hipSuccess;
  dim3 grid((unsigned)((m_pad + 255) / 256), (unsigned)a.NP);
  hipLaunchKernelGGL(ba_jacobian_kernel, grid, dim3(256), 0, stream, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------- Gram matrix on the matrix cores
// One wave per (16x16 output tile, K split).  v_mfma_f64_16x16x4_f64 operands (one f64 per lane):
//   A[i][k]: i = lane & 15, k = lane >> 4      -> Jaug[k0 + (lane>>4)][ti*16 + (lane&15)]
//   B[k][j]: k = lane >> 4, j = lane & 15      -> Jaug[k0 + (lane>>4)][tj*16 + (lane&15)]
//   D: 4 f64 per lane, col = lane & 15, row = (lane >> 4) + 4 * reg
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void ba_gram_kernel(const double* __restrict__ Jaug, int64_t m_pad,
                                                     int NP, int ksplit, double* __restrict__ partial) {
  const int nt = NP / 16;
  const int tile = blockIdx.x;
  const int ti = tile / nt, tj = tile - ti * nt;
  const int ks = blockIdx.y;
  const int lane = threadIdx.x;
  const int64_t steps = m_pad / 4;
  const int64_t s0 = steps * ks / ksplit, s1 = steps * (ks + 1) / ksplit;
  const double* pa = Jaug + (size_t)(lane >> 4) * NP + ti * 16 + (lane & 15);
  const double* pb = Jaug + (size_t)(lane >> 4) * NP + tj * 16 + (lane & 15);
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  for (int64_t s = s0; s < s1; s++) {
    const double av = pa[(size_t)s * 4 * NP];
    const double bv = pb[(size_t)s * 4 * NP];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
  }
  double* out = partial + (size_t)ks * NP * NP;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = ti * 16 + (lane >> 4) + 4 * r;
    const int col = tj * 16 + (lane & 15);
    out[(size_t)row * NP + col] = acc[r];
  }
}

__device__ void ba_cost_block(const double* __restrict__ r, const int32_t* __restrict__ valid, int64_t m,
                              int f32_residuals, int use_cauchy, double* __restrict__ out);

__global__ __launch_bounds__(256) void ba_gram_reduce_kernel(const double* __restrict__ partial, int NP, int ksplit,
                                                             double* __restrict__ G, const double* __restrict__ r,
                                                             const int32_t* __restrict__ valid, int64_t m,
                                                             int f32_residuals, int use_cauchy,
                                                             double* __restrict__ cost_out) {
  if (cost_out && blockIdx.x == gridDim.x - 1) {  // the extra workgroup: cost of residual row r
    if (use_cauchy < 0) {
      // r already holds the loss values rho(f_i^2) of the valid points (written by the Jacobian kernel, one
      // lane per point): same addends, same order as ba_cost_block, without 16 k serial log1p in one workgroup
      __shared__ double sh[256];
      __shared__ int bad;
      if (threadIdx.x == 0) bad = 0;
      __syncthreads();
      double s = 0.0;
      for (int64_t i = threadIdx.x; i < m; i += 256) {
        const double v = r[i];
        if (!isfinite(v)) bad = 1;
        s += v;
      }
      sh[threadIdx.x] = s;
      __syncthreads();
      for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
      }
      if (threadIdx.x == 0) {
        cost_out[0] = 0.5 * sh[0];
        cost_out[1] = bad ? 0.0 : 1.0;
      }
      return;
    }
    ba_cost_block(r, valid, m, f32_residuals, use_cauchy, cost_out);
    return;
  }
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= NP * NP) return;
  double s = 0.0;
  int k = 0;
  for (; k + 8 <= ksplit; k += 8) {  // eight loads in flight, added in slice order (fixed order)
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = partial[(size_t)(k + u) * NP * NP + idx];
#pragma unroll
    for (int u = 0; u < 8; u++) s += v[u];
  }
  for (; k < ksplit; k++) s += partial[(size_t)k * NP * NP + idx];
  G[idx] = s;
}

int ba_gram_ksplit(int64_t m_pad, int NP) {
  (void)NP;
  // one wave per 16 x 16 tile and K slice: a wave's loop is a chain of dependent load -> MFMA steps (~0.3 us
  // each, nothing to overlap with), so short slices finish sooner: 8 steps (32 rows) per wave, up to 256 slices
  int64_t ks = m_pad / 32;
  if (ks < 1) ks = 1;
  if (ks > 256) ks = 256;
  return (int)ks;
}

hipError_t launch_ba_gram(const double* Jaug, int64_t m_pad, int NP, double* partial, int ksplit,
                          double* G, hipStream_t stream) {
  const int nt = NP / 16;
  hipLaunchKernelGGL(ba_gram_kernel, dim3(nt * nt, ksplit), dim3(64), 0, stream, Jaug, m_pad, NP, ksplit,
                     partial);
  hipLaunchKernelGGL(ba_gram_reduce_kernel, dim3((NP * NP + 255) / 256), dim3(256), 0, stream, partial, NP,
                     ksplit, G, (const double*)nullptr, (const int32_t*)nullptr, (int64_t)0, 0, 0, (double*)nullptr);
  return hipGetLastError();
}

hipError_t launch_ba_gram_cost(const double* Jaug, int64_t m_pad, int NP, double* partial, int ksplit, double* G,
                               const double* r, const int32_t* valid, int64_t m, int f32_residuals, int use_cauchy,
                               double* cost_out, hipStream_t stream) {
  const int nt = NP / 16;
  hipLaunchKernelGGL(ba_gram_kernel, dim3(nt * nt, ksplit), dim3(64), 0, stream, Jaug, m_pad, NP, ksplit,
                     partial);
  hipLaunchKernelGGL(ba_gram_reduce_kernel, dim3((NP * NP + 255) / 256 + 1), dim3(256), 0, stream, partial, NP,
                     ksplit, G, r, valid, m, f32_residuals, use_cauchy, cost_out);
  return hipGetLastError();
}

// ---------------------------------------------------------------- one linearisation in ONE launch
// An LM iteration is a chain of small dependent steps, so its cost is launch and round-trip latency, not
// arithmetic.  This kernel does everything between "the host knows x" and "the host has
// G = [J|f]^T [J|f] and the cost": camera tables of the forward-difference parameter sets, all residual
// evaluations, float32 differencing + Cauchy scaling, the Gram matrix on the FP64 matrix cores, the
// cross-workgroup reduction, and the hand-over to the host through pinned memory.
//
//   workgroup (chunk k, group q), kFusedWaves waves: wave w evaluates ONE parameter set on the chunk's 64 points.  The
//     set's cameras are built by the workgroup itself into LDS (Rodrigues + K[R|t]) and read back as
//     broadcasts; residuals go to r[p][point] in HBM.  Only LIVE sets are evaluated: the focal entries of the
//     parameter vector have no effect on the residuals (helpers.py:267-270 writes them into a temporary), so
//     their forward differences are exact zeros in the reference as well -- 1 + 6 (C-1) sets instead of
//     2 + 7 (C-1).
//   the LAST workgroup to finish a chunk (device-scope counter) owns the chunk's 64 rows of Jaug = [J | f | 0]:
//     builds them in LDS, v_mfma_f64_16x16x4_f64 over the upper-triangle 16 x 16 tiles (tiles round-robin over
//     its waves, 16 K-steps each: a 16 x 16 x 4 FP64 MFMA occupies its SIMD for 64 clocks, so the Gram matrix has
//     to be spread over the chunks' workgroups -- one workgroup doing all of it would need 17 us), writes the
//     chunk's partial tiles.
//   the LAST chunk owner adds the partials in chunk order (fixed order -> bit-reproducible whichever
//     workgroup happens to be last) and stores G, cost and the completion stamp into pinned host memory.
// What the structure is shaped by (measured, gfx950): an agent-scope release (write-back of an XCD's L2) costs
// 16 us when every wave of 208 workgroups issues one, 3.5 us with one per workgroup and 64 workgroups; a round of
// dependent loads of lines another XCD just wrote is ~3 us; a spilled register reloaded inside a loop ~1 us --
// so: few, fat workgroups, one fence each, as few dependent rounds as possible.  Counters return to zero inside
// the launch (no memset between launches).
constexpr int kFusedFan = 16;  // records per node of the reduction tree
constexpr int kFusedWaves = 8, kFusedThreads = 64 * kFusedWaves;  // one parameter set per wave (A/B at 8 x 1 000: 4 waves 29.9 us, 8 waves 26.2, 16 waves 29.4 per launch)

// Cross-workgroup hand-off inside the launch (MI355X_MICROARCH.md, rows handoff-flag / publish-large): payload
// with agent-scope (sc1, write-through) stores, every wave drains its stores (vmcnt(0)), then ONE relaxed
// agent-scope atomic on the arrival counter; the consumer reads the payload with sc1 loads.  No L2 write-back
// fence: an agent-scope release costs 3.5 us per round here even when only one thread per workgroup issues it.
__device__ __forceinline__ void st_agent(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void drain_stores() { wait_own_stores(); }

// slot of a live parameter set (0 = base point, 1 + k = k-th live parameter) -> parameter set index p
__device__ __forceinline__ int ba_live_set(int slot) {
  if (slot == 0) return 0;
  const int k = slot - 1;
  return 1 + 7 * (k / 6) + 2 + k % 6;  // x = [f0, (f_i, rotvec 3, t 3) ...]: entries 0 and 1 + 7 i are the dead focals
}
__device__ __forceinline__ bool ba_live_param(int j, int n) { return j < n && j != 0 && (j - 1) % 7 != 0; }

template <bool UNIFORM_K, bool F32R>
__global__ __launch_bounds__(kFusedThreads) void ba_fused_kernel(BaFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ int sh_last;
  __shared__ double sh_red[8];
  __shared__ double xs[128];  // base point (dynamic per-thread indexing of the by-value argument would go through scratch)
  const int chunk = blockIdx.x, grp = blockIdx.y;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int C = a.C, n = a.n, NP = a.NP;
  const int64_t N = a.N;
  const int nsets = 1 + 6 * (C - 1);  // live parameter sets
  const double qnan = __longlong_as_double(0x7ff8000000000000ll);
  const size_t nPq = UNIFORM_K ? (size_t)12 * C : (size_t)12 * C * C;
  const size_t set_stride = (size_t)12 * C + nPq;  // [RT C*12 | Pq]
  double* tabs = lds;                               // [kFusedWaves][set_stride]
  double* obs_s = lds + kFusedWaves * set_stride;   // [C][2][64] the chunk's observations, shared by the waves
  if (a.mailbox) {
    // Launched ahead: the previous linearisation's result is still being turned into the next trial point by the
    // host (trust-region subproblem, ~9 us).  Being resident and polling hides the launch latency and the launch
    // floor behind that host work.  ONE workgroup polls the host's mailbox over PCIe (96 pollers serialise on the
    // link: measured +30 us per iteration) and republishes {x, stamp} in device memory, where the others poll.
    // A watchdog (2 s of the 100 MHz wall clock) ends an orphaned launch.
    double* dm = a.dev_mail;  // [0] stamp, [2..] x
    const bool lead = chunk == 0 && grp == 0;
    if (lead) {
      // the mailbox is a run of 64-byte lines {tag, 7 doubles of x}; the host writes a line's data, then its tag
      // (x86 store order), and a line is read whole: a line whose tag equals the stamp carries its new data, so the
      // poll that sees the command has also read x -- no second trip over PCIe
      const int lines = (n + 6) / 7;
      if (wave == 0) {
        const unsigned long long t0 = wall_clock64();
        int go = 0;
        while (true) {
          double v = 0.0;
          if (lane < 8 * lines) v = __hip_atomic_load(&a.mailbox[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          double v2 = 0.0;  // lines 8.. (n > 56)
          if (lane + 64 < 8 * lines) v2 = __hip_atomic_load(&a.mailbox[lane + 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          const bool is_tag = (lane & 7) == 0;
          const bool bad1 = is_tag && lane < 8 * lines && v != a.stamp, bad2 = is_tag && lane + 64 < 8 * lines && v2 != a.stamp;
          const bool quit = is_tag && lane == 0 && v < 0.0;
          if (!__ballot(bad1 || bad2)) {
            if (!is_tag) {
              const int k1 = (lane >> 3) * 7 + (lane & 7) - 1, k2 = ((lane + 64) >> 3) * 7 + (lane & 7) - 1;
              if (lane < 8 * lines && k1 < n) xs[k1] = v;
              if (lane + 64 < 8 * lines && k2 < n) xs[k2] = v2;
            }
            go = 1;
            break;
          }
          if (__ballot(quit) || wall_clock64() - t0 > 200000000ull) break;
          __builtin_amdgcn_s_sleep(2);
        }
        if (lane == 0) sh_last = go;
      }
      __syncthreads();
      if (sh_last)
        for (int k = tid; k < n; k += kFusedThreads) st_agent(&dm[2 + k], xs[k]);
      drain_stores();
      __syncthreads();
      if (tid == 0) st_agent(&dm[0], sh_last ? a.stamp : -a.stamp);  // -stamp: this launch is abandoned
      if (!sh_last) {
        // ... and the host is told so (it may be on its way to hand this launch a point: a watchdog exit, not a quit tag):
        // -stamp in the completion word makes it relaunch with the point passed by value instead of waiting for a stamp
        // that will never come
        if (tid == 0) {
          __threadfence_system();
          *(volatile double*)(a.out + (size_t)NP * NP + 2) = -a.stamp;
        }
        return;
      }
    } else {
      if (tid == 0) {
        const unsigned long long t0 = wall_clock64();
        int go = 0;
        while (true) {
          const double sq = ld_agent(&dm[0]);
          if (sq == a.stamp) {
            go = 1;
            break;
          }
          // ONLY the lead workgroup decides (its watchdog: 2 s after ITS start): the others wait for its verdict, +stamp or
          // -stamp.  Their own clock is a last resort far beyond that (30 s: a workgroup that became resident long before
          // the lead must not give up while the lead still accepts the point -- the completion counter would never fill).
          if (sq == -a.stamp || wall_clock64() - t0 > 3000000000ull) break;
          __builtin_amdgcn_s_sleep(2);
        }
        sh_last = go;
      }
      __syncthreads();
      if (!sh_last) return;
      for (int k = tid; k < n; k += kFusedThreads) xs[k] = ld_agent(&dm[2 + k]);
    }
  } else {
    for (int k = tid; k < n; k += kFusedThreads) xs[k] = a.x[k];
  }
  __syncthreads();
  if (a.debug_stop && chunk == 0 && grp == 0 && tid == 0) *(volatile double*)(a.out + (size_t)NP * NP + 2) = a.stamp;
  if (a.debug_stop == 1) return;
  // ---- phase 0: the chunk's observations -> LDS, transposed (every wave of the workgroup triangulates the same 64
  // points; read per lane from HBM they are 16 dependent, uncoalesced loads per point: ~3 us of the residual phase)
  for (int e = tid; e < 64 * C * 2; e += kFusedThreads) {
    const int pt = e / (2 * C), rem = e - pt * 2 * C;
    const int64_t idx = (int64_t)chunk * 64 + pt;
    obs_s[rem * 64 + pt] = idx < N ? a.obs[(size_t)idx * C * 2 + rem] : qnan;
  }
  // cameras of this group's parameter sets
  for (int t = tid; t < kFusedWaves * C; t += kFusedThreads) {
    const int w = t / C, cam = t - w * C, slot = kFusedWaves * grp + w;
    if (slot >= nsets) continue;
    double rt[12];
    ba_fd_camera_pose(xs, a.rel_step, ba_live_set(slot), cam, rt);
    double* T = tabs + (size_t)w * set_stride;
    for (int k = 0; k < 12; k++) T[12 * cam + k] = rt[k];
    const int jn = UNIFORM_K ? 1 : cam + 1;
    for (int j = 0; j < jn; j++)
      ba_projection(a.K + 9 * j, rt, T + 12 * C + 12 * (UNIFORM_K ? (size_t)cam : (size_t)j * C + cam));
  }
  __syncthreads();
  if (a.debug_stop == 2) return;
  // ---- phase 1: residual of (set p, point idx) = re-triangulation + mean squared reprojection error
  {
    const int slot = kFusedWaves * grp + wave;
    const int p = ba_live_set(slot);
    const int64_t idx = (int64_t)chunk * 64 + lane;
    if (slot < nsets && idx < N) {
      LdsCamView cv;
      cv.C = C;
      cv.uniformK = UNIFORM_K;
      cv.f32_rounding = F32R;
      cv.RT = (ltab_t)(tabs + (size_t)wave * set_stride);
      cv.Pq = (ltab_t)(tabs + (size_t)wave * set_stride + 12 * C);
      cv.K4 = a.K4;
      const double* o = obs_s + lane;
      auto obs = [&](int c, double& x, double& y) -> bool {
        x = o[(2 * c) * 64];
        y = o[(2 * c + 1) * 64];
        return !(isnan(x) || isnan(y));
      };
      double X[3], e = qnan;
      const int v = triangulate_and_score<UNIFORM_K, false, F32R>(cv, obs, obs, X, e);
      // NaN = fewer than two views (no residual entry, helpers.py:207-208); a seen point whose error is not
      // a number is handed on as +inf so that the all-finite check of the trust-region loop still sees it
      st_agent(&a.r[(size_t)p * N + idx], v < 2 ? qnan : (isnan(e) ? __longlong_as_double(0x7ff0000000000000ll) : e));
    }
  }
  if (a.debug_stop == 3) return;
  // ---- who finishes the chunk?
  drain_stores();
  __syncthreads();
  if (tid == 0)
    sh_last = __hip_atomic_fetch_add(&a.counters[chunk], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.groups - 1;
  __syncthreads();
  if (!sh_last) return;
  if (tid == 0) __hip_atomic_store(&a.counters[chunk], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (a.debug_stop == 4) return;
  // ---- phase 2: the chunk's rows of Jaug (LDS, aliases the camera tables), cost terms
  const int LD = NP + 1;  // row stride of Jaug in LDS: lanes = rows store without a 64-way bank conflict
  double* Jaug = lds;     // [64][LD]
  {
    const int row = lane, cg = wave;
    const int64_t idx = (int64_t)chunk * 64 + row;
    // all global loads first (the rows were written by other workgroups, possibly on another XCD: ~2 us each
    // if they were issued one after the other)
    constexpr int kMaxCols = 128 / kFusedWaves;  // NP <= 128
    double fj[kMaxCols];
    double r0 = __longlong_as_double(0x7ff8000000000000ll);
    if (idx < N) {
      r0 = ld_agent(&a.r[idx]);
#pragma unroll
      for (int q = 0; q < kMaxCols; q++) {
        const int j = cg + kFusedWaves * q;
        fj[q] = ba_live_param(j, n) ? ld_agent(&a.r[(size_t)(1 + j) * N + idx]) : 0.0;
      }
    }
    const bool valid = !isnan(r0);
    CauchyTerms ct = {0.0, 0.0, 0.0, 0.0};
    if (valid) ct = cauchy_terms(r0, a.f32_residuals, a.use_cauchy);
#pragma unroll
    for (int q = 0; q < kMaxCols; q++) {
      const int j = cg + kFusedWaves * q;
      if (j >= NP) break;
      double out = 0.0;
      if (valid) {
        if (ba_live_param(j, n)) {
          const double df = a.f32_residuals ? (double)((float)fj[q] - (float)r0) : fj[q] - r0;
          // dx_j = (x_j + h_j) - x_j (scipy _numdiff._dense_difference)
          const double xv = xs[j];
          const double dx = (xv + a.rel_step * (xv >= 0.0 ? 1.0 : -1.0) * fmax(1.0, fabs(xv))) - xv;
          out = (df / dx) * ct.jscale;
        } else if (j == n) {
          out = ct.fs;
        }
      }
      Jaug[row * LD + j] = out;
      if (a.Jaug_out && idx < N) a.Jaug_out[(size_t)idx * NP + j] = out;
    }
    if (cg == 0) {  // cost of the chunk: sum of the loss values in row order (fixed tree), all-finite flag
      double v = valid ? ct.rho0 : 0.0;
      const unsigned long long badm = __ballot(valid && !isfinite(ct.f));
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
      if (lane == 0) {
        st_agent(&a.cost_part[2 * chunk], v);
        st_agent(&a.cost_part[2 * chunk + 1], badm ? 0.0 : 1.0);
      }
    }
  }
  __syncthreads();
  if (a.debug_stop == 5) return;
  // ---- phase 3: partial Gram of the 64 rows, upper-triangle tiles round-robin over the waves
  const int nt = NP / 16;
  const int tri_n = (n + 1) * (n + 2) / 2;  // packed upper triangle of G's leading block
  {
    int u = 0;
    for (int ti = 0; ti < nt; ti++)
      for (int tj = ti; tj < nt; tj++, u++) {
        if (u % kFusedWaves != wave) continue;
        const double* pa = Jaug + (size_t)(lane >> 4) * LD + ti * 16 + (lane & 15);
        const double* pb = Jaug + (size_t)(lane >> 4) * LD + tj * 16 + (lane & 15);
        double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
        for (int s4 = 0; s4 < 16; s4++)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[(size_t)s4 * 4 * LD], pb[(size_t)s4 * 4 * LD], acc, 0, 0, 0);
        // only the upper triangle of the leading (n+1) x (n+1) block travels, packed row-major: a workgroup reads
        // fresh cross-XCD lines at ~65 GB/s (guide: handoff-payload), so the last owner's sum is bandwidth-bound
        double* out = a.partial + (size_t)chunk * tri_n;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = ti * 16 + (lane >> 4) + 4 * r, col = tj * 16 + (lane & 15);
          if (row <= col && col <= n) st_agent(&out[row * (n + 1) - row * (row - 1) / 2 + (col - row)], acc[r]);
        }
      }
  }
  if (a.debug_stop == 6) return;
  // ---- phase 4: tree of "last arriver" reductions, 16 records per node, summed in record order (fixed order ->
  // bit-reproducible whichever workgroup is last).  1 000 points = 16 chunks = one node: its last arriver writes G
  // (packed upper triangle of the leading (n+1) x (n+1) block; the host unpacks and mirrors it), the cost and the stamp
  // into pinned host memory.  16 000 points = 250 chunks = 16 nodes + a root: a single workgroup adding 250 records
  // would be bandwidth-bound for ~40 us (a workgroup reads fresh cross-XCD lines at ~65 GB/s).
  {
    int units = a.chunks, my = chunk;
    size_t rec_base = 0, cnt_base = (size_t)a.chunks;  // records / counters of the current level
    while (true) {
      const int nodes = (units + kFusedFan - 1) / kFusedFan, node = my / kFusedFan;
      const int first = node * kFusedFan, cnt = min(kFusedFan, units - first);
      drain_stores();
      block_sync_lds();  // (loop head; the wait written out: tests/isa_barriers.py)
      if (tid == 0) {
        sh_last = __hip_atomic_fetch_add(&a.counters[cnt_base + node], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == cnt - 1;
        if (sh_last) __hip_atomic_store(&a.counters[cnt_base + node], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      if (!sh_last) return;
      if (a.debug_stop == 7) return;
      const bool top = nodes == 1;
      const double* __restrict__ part = a.partial + (rec_base + first) * tri_n;
      double* __restrict__ dst = top ? a.out : a.partial + (rec_base + units + node) * tri_n;
      constexpr int EPT = 4;  // packed elements per thread and round: 4 x 16 records = 64 loads in flight
      for (int e0 = tid; e0 < tri_n; e0 += EPT * kFusedThreads) {
        double sum[EPT];
        double v[EPT][kFusedFan];
#pragma unroll
        for (int t = 0; t < EPT; t++)
#pragma unroll
          for (int q = 0; q < kFusedFan; q++)
            v[t][q] = (e0 + t * kFusedThreads < tri_n && q < cnt) ? ld_agent(&part[(size_t)q * tri_n + e0 + t * kFusedThreads]) : 0.0;
#pragma unroll
        for (int t = 0; t < EPT; t++) {
          sum[t] = 0.0;
#pragma unroll
          for (int q = 0; q < kFusedFan; q++)
            if (q < cnt) sum[t] += v[t][q];  // record order
        }
#pragma unroll
        for (int t = 0; t < EPT; t++)
          if (e0 + t * kFusedThreads < tri_n) {
            if (top)
              dst[e0 + t * kFusedThreads] = sum[t];  // pinned host memory
            else
              st_agent(&dst[e0 + t * kFusedThreads], sum[t]);
          }
      }
      // cost of the node: its records' sums in order, all-finite flag
      if (wave == 0) {
        const double* cp = a.cost_part + 2 * (rec_base + first);
        double c = lane < cnt ? ld_agent(&cp[2 * lane]) : 0.0;
        double fin = lane < cnt ? ld_agent(&cp[2 * lane + 1]) : 1.0;
        double acc = 0.0;
        for (int q = 0; q < cnt; q++) acc += __shfl(c, q);  // in record order
        for (int o = 8; o > 0; o >>= 1) fin = fmin(fin, __shfl_down(fin, o));
        if (lane == 0) {
          if (top) {
            sh_red[0] = 0.5 * acc;
            sh_red[1] = fin;
          } else {
            st_agent(&a.cost_part[2 * (rec_base + units + node)], acc);
            st_agent(&a.cost_part[2 * (rec_base + units + node) + 1], fin);
          }
        }
      }
      if (top) break;
      rec_base += units;
      cnt_base += nodes;
      units = nodes;
      my = node;
    }
  }
  // every wave waits for its own stores (workgroup-scope release), one thread publishes to the host
  wait_own_stores();
  __syncthreads();
  if (tid == 0) {
    double* tail = a.out + (size_t)NP * NP;
    tail[0] = sh_red[0];
    tail[1] = sh_red[1];
    __threadfence_system();
    *(volatile double*)(tail + 2) = a.stamp;  // the host spins on this word
  }
}

// records (Gram partials, cost pairs) and counters of the reduction tree over `chunks` leaves
size_t ba_fused_records(int chunks) {
  size_t total = 0;
  for (int u = chunks; ; u = (u + kFusedFan - 1) / kFusedFan) {
    total += (size_t)u;
    if (u <= kFusedFan) break;
  }
  return total;
}
size_t ba_fused_counters(int chunks) { return ba_fused_records(chunks) + 1; }

size_t ba_fused_lds_bytes(int C, int NP, bool uniformK) {
  const size_t set_stride = (size_t)12 * C + (uniformK ? (size_t)12 * C : (size_t)12 * C * C);
  const size_t tabs = (kFusedWaves * set_stride + (size_t)64 * C * 2) * sizeof(double),
               jaug = (size_t)64 * (NP + 1) * sizeof(double);
  return tabs > jaug ? tabs : jaug;
}

int ba_fused_groups(int C) { return (1 + 6 * (C - 1) + kFusedWaves - 1) / kFusedWaves; }

bool ba_fused_eligible(int C, int n, int NP, bool uniformK) {
  return n <= 127 && ba_fused_lds_bytes(C, NP, uniformK) <= 150 * 1024;
}

hipError_t launch_ba_fused(const BaFusedArgs& a, hipStream_t stream) {
  void (*k)(BaFusedArgs);
  if (a.f32_rounding)
    k = a.uniformK ? ba_fused_kernel<true, true> : ba_fused_kernel<false, true>;
  else
    k = a.uniformK ? ba_fused_kernel<true, false> : ba_fused_kernel<false, false>;
  const size_t lds = ba_fused_lds_bytes(a.C, a.NP, a.uniformK);
  if (lds > 64 * 1024) {  // above the default dynamic-LDS limit (non-identical intrinsics: tables per compacted index)
    const hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)a.chunks, (unsigned)a.groups), dim3(kFusedThreads), lds, stream, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------- cost of one residual vector
// cost = 0.5 * sum rho(f^2) over the valid points (scipy loss_function(..., cost_only=True));
// out[1] = 1 when every residual is finite.  Single workgroup, fixed-order tree -> reproducible.
__device__ void ba_cost_block(const double* __restrict__ r, const int32_t* __restrict__ valid, int64_t m,
                              int f32_residuals, int use_cauchy, double* __restrict__ out) {
  __shared__ double sh[256];
  __shared__ int bad;
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < m; i += 256) {
    const CauchyTerms c = cauchy_terms(r[valid[i]], f32_residuals, use_cauchy);
    if (!isfinite(c.f)) bad = 1;
    s += c.rho0;
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = 0.5 * sh[0];
    out[1] = bad ? 0.0 : 1.0;
  }
}

__global__ __launch_bounds__(256) void ba_cost_kernel(const double* __restrict__ r,
                                                      const int32_t* __restrict__ valid, int64_t m,
                                                      int f32_residuals, int use_cauchy,
                                                      double* __restrict__ out) {
  ba_cost_block(r, valid, m, f32_residuals, use_cauchy, out);
}

hipError_t launch_ba_cost(const double* r, const int32_t* valid, int64_t m, int f32_residuals,
                          int use_cauchy, double* out, hipStream_t stream) {
  hipLaunchKernelGGL(ba_cost_kernel, dim3(1), dim3(256), 0, stream, r, valid, m, f32_residuals, use_cauchy,
                     out);
  return hipGetLastError();
}

// Stage-in without a copy engine: 4-byte words from device-visible (pinned) host memory to device memory, and the
// arrival counters of the one-launch linearisation back to zero.  mocap_ba_solve's set-up used hipMemcpyAsync from the
// caller's pageable arrays and hipMemsetAsync; both go through runtime-internal machinery (user-pointer pinning, SDMA
// queues, blit kernels created on first use) that a solve of a few milliseconds should not depend on.
__global__ void ba_stage_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t n_words,
                                uint32_t* __restrict__ zero, size_t n_zero) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride) dst[i] = src[i];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_zero; i += stride) zero[i] = 0u;
}

hipError_t launch_ba_stage(const void* src, void* dst, size_t n_words, void* zero, size_t n_zero_words, hipStream_t stream) {
  const size_t work = n_words > n_zero_words ? n_words : n_zero_words;
  if (!work) return hipSuccess;
  size_t grid = (work + 255) / 256;
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(ba_stage_kernel, dim3((unsigned)grid), dim3(256), 0, stream, (const uint32_t*)src, (uint32_t*)dst, n_words,
                     (uint32_t*)zero, n_zero_words);
  return hipGetLastError();
}

}  // namespace mocap
