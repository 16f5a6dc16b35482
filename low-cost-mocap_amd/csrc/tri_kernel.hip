// tri_kernel.hip -- batched DLT triangulation + reprojection error for explicit correspondences.
// Replaces triangulate_points (reference computer_code/api/helpers.py:330-336) and
// calculate_reprojection_errors (helpers.py:203-211); with P > 1 camera sets it is also the
// inner evaluation of the bundle-adjustment residual_function (helpers.py:264-273), one grid
// row per parameter vector.  One lane per point, FP64, camera tables over the scalar cache.
#include "kernels.hpp"
#include "mocap_device.hpp"

namespace mocap {

template <bool UNIFORM_K, bool F32R>
__global__ __launch_bounds__(256) void tri_kernel(TriArgs a) {
  CamView cv = a.cv;
  const int p = blockIdx.y;
  cv.Pq += (size_t)p * a.stride_Pq;
  cv.RT += (size_t)p * a.stride_RT;
  const int C = cv.C;
  const double qnan = __longlong_as_double(0x7ff8000000000000ll);
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < a.N;
       n += (int64_t)gridDim.x * blockDim.x) {
    const double* o = a.obs + (size_t)n * C * 2;
    auto obs = [&](int c, double& x, double& y) -> bool {
      x = o[2 * c];
      y = o[2 * c + 1];
      return !(isnan(x) || isnan(y));
    };
    double X[3] = {qnan, qnan, qnan}, e = qnan;
    // explicit captures arrive as object ndarrays in the reference (index.py:232): errors.mean()
    // sums left to right there, so PAIRWISE = false
    int v;
    if (a.xyz_in) {
      // calculate_reprojection_error of a given point (helpers.py:214-241): the views decide whether there is an
      // entry at all (helpers.py:222-223); a NaN coordinate (the reference's None) yields NaN
      v = 0;
      for (int c = 0; c < C; c++) {
        double x, y;
        v += obs(c, x, y) ? 1 : 0;
      }
      X[0] = a.xyz_in[(size_t)n * 3 + 0];
      X[1] = a.xyz_in[(size_t)n * 3 + 1];
      X[2] = a.xyz_in[(size_t)n * 3 + 2];
      if (v >= 2) score_point<UNIFORM_K, false, F32R>(cv, v, obs, X, e, __builtin_huge_val());
    } else {
      v = triangulate_and_score<UNIFORM_K, false, F32R>(cv, obs, obs, X, e);
    }
    if (v < 2) {
      X[0] = X[1] = X[2] = qnan;  // the reference yields [None, None, None] (helpers.py:300-301)
      e = qnan;                   // and skips the error entry (helpers.py:207-208)
    }
    if (a.xyz) {
      double* d = a.xyz + ((size_t)p * a.N + n) * 3;
      d[0] = X[0];
      d[1] = X[1];
      d[2] = X[2];
    }
    if (a.err) a.err[(size_t)p * a.N + n] = e;
  }
}

hipError_t launch_triangulate(const TriArgs& a, hipStream_t stream) {
  if (a.N <= 0 || a.P <= 0) return hipSuccess;
  int64_t blocks = (a.N + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  dim3 grid((unsigned)blocks, (unsigned)a.P);
  void (*k)(TriArgs);
  if (a.cv.f32_rounding)
    k = a.cv.uniformK ? tri_kernel<true, true> : tri_kernel<false, true>;
  else
    k = a.cv.uniformK ? tri_kernel<true, false> : tri_kernel<false, false>;
  hipLaunchKernelGGL(k, grid, dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace mocap
