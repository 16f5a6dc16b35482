// post_kernels.hip -- the step right after the hot path in the reference's frame loop:
//   locate_objects (reference computer_code/api/helpers.py:424-480): find the 3-LED drone patterns
//   (two points 0.095 m from a lead point and 0.15 m from each other, +-0.025) among a frame's
//   triangulated points, heading from the pair, drone index from the side the lead point is on.
// The world-coordinate epilogue (helpers.py:96-103) is fused into the frame kernel's store
// (frame_kernel.hip, write_point).
//
// One lane per frame: the reference's scan is sequential (`already_matched_points`, first valid pair
// wins) and K is a few dozen points, so the parallelism is across the frame batch.  Distances are
// recomputed where the reference reads its K x K matrix (same expression, same bits).
#include "kernels.hpp"

namespace mocap {

__device__ __forceinline__ double dist3(const double* a, const double* b) {
  const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return sqrt((dx * dx + dy * dy) + dz * dz);  // np.sqrt(np.sum(d**2)), helpers.py:434
}

__global__ __launch_bounds__(64) void locate_objects_kernel(LocateArgs a) {
  const double dist1 = 0.095, dist2 = 0.15, tol = 0.025;  // helpers.py:425-426,441,448
  const double pi = 3.141592653589793, half_pi = 1.5707963267948966;
  for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < a.n_frames;
       f += (int64_t)gridDim.x * blockDim.x) {
    const double* P = a.xyz + (size_t)f * a.K_max * 3;
    const double* E = a.err + (size_t)f * a.K_max;
    int K = a.n_pts[f];
    K = K < 0 ? 0 : (K > a.K_max ? a.K_max : K);
    unsigned long long matched[4] = {0, 0, 0, 0};  // already_matched_points, K <= 256
    int no = 0;
    for (int i = 0; i < K; i++) {
      if (matched[i >> 6] >> (i & 63) & 1ull) continue;  // helpers.py:437-438
      int nm = 0;
      for (int j = 0; j < K; j++) nm += fabs(dist3(P + 3 * i, P + 3 * j) - dist1) < tol ? 1 : 0;
      if (nm < 2) continue;  // helpers.py:443
      bool done = false;
      // cartesian_product(matches, matches), first index slow (helpers.py:444,532-533)
      for (int p1 = 0; p1 < K && !done; p1++) {
        if (!(fabs(dist3(P + 3 * i, P + 3 * p1) - dist1) < tol)) continue;
        for (int p2 = 0; p2 < K; p2++) {
          if (!(fabs(dist3(P + 3 * i, P + 3 * p2) - dist1) < tol)) continue;
          const double pd = dist3(P + 3 * p1, P + 3 * p2);
          if (fabs(pd - dist2) > tol) continue;  // helpers.py:448-449
          matched[i >> 6] |= 1ull << (i & 63);
          matched[p1 >> 6] |= 1ull << (p1 & 63);
          matched[p2 >> 6] |= 1ull << (p2 & 63);
          const double* A = P + 3 * p1;
          const double* B = P + 3 * p2;
          const double loc[3] = {(A[0] + B[0]) / 2, (A[1] + B[1]) / 2, (A[2] + B[2]) / 2};
          const double error = ((E[i] + E[p1]) + E[p2]) / 3.0;  // np.mean of three (helpers.py:459)
          double hx = A[0] - B[0], hy = A[1] - B[1], hz = A[2] - B[2];
          const double nrm = sqrt((hx * hx + hy * hy) + hz * hz);  // linalg.norm (helpers.py:462)
          hx /= nrm;
          hy /= nrm;
          double heading = atan2(hy, hx);
          heading = heading > half_pi ? heading - pi : heading;   // helpers.py:465
          heading = heading < -half_pi ? heading + pi : heading;  // helpers.py:466
          const int drone = (P[3 * i + 1] - loc[1]) > 0 ? 0 : 1;   // helpers.py:469
          if (no < a.O_max) {
            const size_t o = (size_t)f * a.O_max + no;
            a.obj_pos[3 * o + 0] = loc[0];
            a.obj_pos[3 * o + 1] = loc[1];
            a.obj_pos[3 * o + 2] = loc[2];
            a.obj_heading[o] = -heading;
            a.obj_err[o] = error;
            a.obj_drone[o] = drone;
            if (a.obj_lead) a.obj_lead[o] = i;
          }
          no++;
          done = true;  // `break` leaves the pair loop (helpers.py:478)
          break;
        }
      }
    }
    a.n_obj[f] = no;  // may exceed O_max: the caller sees that objects were dropped
  }
}

hipError_t launch_locate_objects(const LocateArgs& a, hipStream_t stream) {
  if (a.n_frames <= 0) return hipSuccess;
  int64_t blocks = (a.n_frames + 63) / 64;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(locate_objects_kernel, dim3((unsigned)blocks), dim3(64), 0, stream, a);
  return hipGetLastError();
}

}  // namespace mocap
