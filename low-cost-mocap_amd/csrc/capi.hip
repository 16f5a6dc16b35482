// capi.hip -- the C ABI of include/mocap_core.h: context, camera tables, host/device entry
// points.  Host runtime only; all arithmetic of the hot path happens in the HIP kernels
// (frame_kernel.hip, tri_kernel.hip, ba_kernels.hip).  The only host-side numerics are the
// frame-invariant camera tables (P = K[R|t], the C x C fundamental table) and the n x n
// trust-region algebra of the LM loop (ba_solve.hip).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mocap_core.h"
#include "ctx.hpp"

using namespace mocap;

// ------------------------------------------------------------------ helpers
int mocap_ctx::fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  {
    std::lock_guard<std::mutex> lk(err_mu);
    err = buf;
  }
  return code;
}

int mocap_ctx::hip_fail(hipError_t e, const char* what) {
  frame_q_dirty();  // a launch that failed or aborted may have left the self-cleaning queue counters dirty
  return fail(MOCAP_E_HIP, "%s: %s", what, hipGetErrorString(e));
}

// after a "_dev" entry point enqueued work that is still running when it returns
int mocap_ctx::mark_enqueued() {
  if (!handover_event) {
    hipError_t e = hipEventCreateWithFlags(&handover_event, hipEventDisableTiming);
    if (e != hipSuccess) return hip_fail(e, "hipEventCreateWithFlags(handover)");
  }
  hipError_t e = hipEventRecord(handover_event, stream);
  if (e != hipSuccess) return hip_fail(e, "hipEventRecord(handover)");
  dev_outstanding = true;
  return MOCAP_OK;
}

#define HIP_TRY(ctx, expr)                                  \
  do {                                                      \
    hipError_t e__ = (expr);                                \
    if (e__ != hipSuccess) return (ctx)->hip_fail(e__, #expr); \
  } while (0)

int DevBuf::reserve(size_t bytes) {
  if (bytes <= cap) return 0;
  if (ptr) (void)hipFree(ptr);
  ptr = nullptr;
  cap = 0;
  size_t want = bytes + bytes / 4 + 256;
  hipError_t e = hipMalloc(&ptr, want);
  if (e != hipSuccess) {
    ptr = nullptr;
    return -1;
  }
  cap = want;
  return 0;
}
void DevBuf::release() {
  if (ptr) (void)hipFree(ptr);
  ptr = nullptr;
  cap = 0;
}

// ------------------------------------------------------------------ lifetime
extern "C" int mocap_create(int device_id, mocap_ctx** out) {
  if (!out) return MOCAP_E_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return MOCAP_E_HIP;
  mocap_ctx* c = new mocap_ctx();
  c->device = device_id;
  if (hipSetDevice(device_id) != hipSuccess || hipStreamCreate(&c->own_stream) != hipSuccess) {
    delete c;
    return MOCAP_E_HIP;
  }
  c->stream = c->own_stream;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->num_cus = prop.multiProcessorCount;
  const char* t = getenv("MOCAP_FRAME_THREADS");
  if (t) {
    int v = atoi(t);
    if (v == 64 || v == 128 || v == 256) c->frame_threads = v;
  }
  if ((t = getenv("MOCAP_HEAVY_THRESHOLD"))) c->heavy_threshold = atoi(t);  // 0 disables splitting
  if ((t = getenv("MOCAP_SLICE_SIZE"))) c->slice_size = atoi(t);
  if ((t = getenv("MOCAP_FORCE_WIDE"))) c->force_wide = atoi(t) ? 1 : 0;
  if ((t = getenv("MOCAP_PRUNE"))) c->prune = atoi(t) ? 1 : 0;
  if ((t = getenv("MOCAP_EIGCUT"))) c->eigcut = atoi(t) ? 1 : 0;
  if ((t = getenv("MOCAP_EVAL_BB"))) c->eval_bb = atoi(t) ? 1 : 0;
  if ((t = getenv("MOCAP_BB_PL")) && atoi(t) >= 1 && atoi(t) <= 64) c->bb_pl = atoi(t);
  if ((t = getenv("MOCAP_BB_FLUSH")) && atoi(t) >= 1) c->bb_flush = atoi(t);
  if ((t = getenv("MOCAP_BB_MIN_G")) && atoi(t) >= 0) c->bb_min_g = atoi(t);
  if ((t = getenv("MOCAP_FRAME_LAUNCHES"))) c->frame_launches = atoi(t) == 3 ? 3 : 1;  // 3: main / slice / merge launches (A/B)  // 0: every group is reprojected in full (A/B)
  *out = c;
  return MOCAP_OK;
}

extern "C" void mocap_destroy(mocap_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  ctx->tables.release();
  for (auto& b : ctx->scratch) b.release();
  ctx->ba_fused.release();
  ctx->compact_ws.release();
  ctx->frame_ws.release();
  ctx->resub.release();
  ctx->live_stage.release();
  ctx->resub_ctr.release();
  ctx->resub_q.release();
  ctx->heavy_recs.release();
  ctx->heavy_ws.release();
  ctx->heavy_enum.release();
  if (ctx->live_pin) (void)hipHostFree(ctx->live_pin);
  if (ctx->live_event) (void)hipEventDestroy(ctx->live_event);
  ctx->img_map.release();
  ctx->img_rot.release();
  ctx->img_mask.release();
  ctx->img_stage.release();
  ctx->img_tiles.release();
  ctx->img_lens.release();
  ctx->img_fix.release();
  ctx->img_fixidx.release();
  ctx->img_act.release();
  ctx->img_box.release();
  ctx->img_zero.release();
  ctx->world.release();
  if (ctx->ba_pin) (void)hipHostFree(ctx->ba_pin);
  if (ctx->ba_stage) (void)hipHostFree(ctx->ba_stage);
  if (ctx->ba_event) (void)hipEventDestroy(ctx->ba_event);
  if (ctx->handover_event) (void)hipEventDestroy(ctx->handover_event);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

// The message is copied, under the error string's own lock, into a buffer owned by the CALLING thread: another thread's
// failing call may rewrite ctx->err at any time (Flask-SocketIO handlers vs the MJPEG generator share one context).
extern "C" const char* mocap_last_error(const mocap_ctx* ctx) {
  if (!ctx) return "null context";
  static thread_local std::string mine;
  {
    std::lock_guard<std::mutex> lk(const_cast<mocap_ctx*>(ctx)->err_mu);
    mine = ctx->err;
  }
  return mine.c_str();
}
extern "C" const char* mocap_version(void) { return "mocap_core 0.1 (gfx950)"; }

extern "C" int mocap_set_stream(mocap_ctx* ctx, void* hip_stream) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  hipStream_t ns = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  // The per-context workspaces (work queues that clean themselves at the END of a launch, scratch) are shared by whatever
  // stream is current: work a "_dev" entry point left running on the previous stream must finish before the first launch
  // on the new one.  Ordered on the DEVICE (the new stream waits for the event recorded behind that work): the previous
  // stream belongs to the caller and may be destroyed by now, the host does not block under the context lock, and the
  // calling thread's current device is put back.
  if (ns != ctx->stream && ctx->dev_outstanding) {
    int prev_dev = -1;
    (void)hipGetDevice(&prev_dev);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const hipError_t e = hipStreamWaitEvent(ns, ctx->handover_event, 0);
    if (prev_dev >= 0 && prev_dev != ctx->device) (void)hipSetDevice(prev_dev);
    if (e != hipSuccess) return ctx->hip_fail(e, "hipStreamWaitEvent(handover)");
    ctx->frame_q_dirty();
  }
  ctx->stream = ns;
  return MOCAP_OK;
}

extern "C" const char* mocap_last_frame_kernel(mocap_ctx* ctx) {
  if (!ctx) return "none";
  std::lock_guard<std::mutex> lk(ctx->mu);
  return ctx->last_frame_kernel;  // string literals: valid for the life of the library
}

extern "C" int mocap_synchronize(mocap_ctx* ctx) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->dev_outstanding = false;
  return MOCAP_OK;
}

extern "C" int mocap_set_options(mocap_ctx* ctx, uint32_t flags) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  ctx->flags = flags;
  ctx->cv.f32_rounding = (flags & MOCAP_OPT_F32_ROUNDING) ? 1 : 0;
  ctx->exhaustive = (flags & MOCAP_OPT_EXHAUSTIVE_WALK) ? 1 : 0;
  return MOCAP_OK;
}

extern "C" int mocap_set_tuning(mocap_ctx* ctx, int frame_threads, int heavy_threshold, int slice_size) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (frame_threads != 0 && frame_threads != 64 && frame_threads != 128 && frame_threads != 256)
    return ctx->fail(MOCAP_E_ARG, "mocap_set_tuning: frame_threads must be 0, 64, 128 or 256");
  ctx->frame_threads = frame_threads;  // 0 = automatic
  ctx->heavy_threshold = heavy_threshold < 0 ? -1 : heavy_threshold;
  ctx->slice_size = slice_size < 0 ? 0 : slice_size;
  return MOCAP_OK;
}

extern "C" int mocap_set_frame_limits(mocap_ctx* ctx, int hit_cap, int force_wide) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (hit_cap < 0 || hit_cap > kMaxBlobs) return ctx->fail(MOCAP_E_ARG, "mocap_set_frame_limits: hit_cap must be 0..%d", kMaxBlobs);
  if (hit_cap) ctx->hit_cap = hit_cap;
  ctx->force_wide = force_wide ? 1 : 0;
  return MOCAP_OK;
}

extern "C" void mocap_limits(int* max_cameras, int* max_blobs) {
  if (max_cameras) *max_cameras = kMaxCameras;
  if (max_blobs) *max_blobs = kMaxBlobs;
}

// ------------------------------------------------------------------ cameras
namespace {

// cv::determinant for a 4x4 CV_64F (OpenCV core/src/lapack.cpp -> hal::LU64f, partial pivoting,
// eps = 100*DBL_EPSILON, p * prod(diag)); used by cv.sfm.fundamentalFromProjections (helpers.py:362).
// This TU is compiled with -ffp-contract=off so the loop rounds like the scalar x86-64 build.
double det4_lu(const double* M) {
  double A[16];
  memcpy(A, M, sizeof A);
  double p = 1.0;
  const double eps = 2.220446049250313e-16 * 100;
  for (int i = 0; i < 4; i++) {
    int k = i;
    for (int j = i + 1; j < 4; j++)
      if (std::fabs(A[j * 4 + i]) > std::fabs(A[k * 4 + i])) k = j;
    if (std::fabs(A[k * 4 + i]) < eps) return 0.0;
    if (k != i) {
      for (int j = i; j < 4; j++) std::swap(A[i * 4 + j], A[k * 4 + j]);
      p = -p;
    }
    const double d = -1.0 / A[i * 4 + i];
    for (int j = i + 1; j < 4; j++) {
      const double alpha = A[j * 4 + i] * d;
      for (int kk = i + 1; kk < 4; kk++) A[j * 4 + kk] = A[j * 4 + kk] + alpha * A[i * 4 + kk];
    }
  }
  double r = p;
  for (int i = 0; i < 4; i++) r = r * A[i * 4 + i];
  return r;
}

void projection(const double* K, const double* R, const double* t, double* P) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) {
      const double r0 = c < 3 ? R[c] : t[0], r1 = c < 3 ? R[3 + c] : t[1], r2 = c < 3 ? R[6 + c] : t[2];
      P[r * 4 + c] = K[r * 3 + 0] * r0 + K[r * 3 + 1] * r1 + K[r * 3 + 2] * r2;
    }
}

void fundamental(const double* P1, const double* P2, double* F) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double XY[16];
      memcpy(XY + 0, P1 + 4 * ((j + 1) % 3), 32);
      memcpy(XY + 4, P1 + 4 * ((j + 2) % 3), 32);
      memcpy(XY + 8, P2 + 4 * ((i + 1) % 3), 32);
      memcpy(XY + 12, P2 + 4 * ((i + 2) % 3), 32);
      F[i * 3 + j] = det4_lu(XY);
    }
}

}  // namespace

extern "C" int mocap_set_cameras(mocap_ctx* ctx, int C, const double* K, const double* R, const double* t) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!K || !R || !t || C < 1) return ctx->fail(MOCAP_E_ARG, "mocap_set_cameras: bad argument");
  if (C > kMaxCameras) return ctx->fail(MOCAP_E_LIMIT, "mocap_set_cameras: C=%d exceeds %d", C, kMaxCameras);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  ctx->hK.assign(K, K + 9 * C);
  ctx->hR.assign(R, R + 9 * C);
  ctx->ht.assign(t, t + 3 * C);
  bool uniform = true;
  for (int i = 1; i < C && uniform; i++) uniform = memcmp(K, K + 9 * i, 72) == 0;
  // layout of the device table block (doubles): Pq | RT | K4 | F | K9
  const size_t nPq = uniform ? (size_t)12 * C : (size_t)12 * C * C;
  const size_t nRT = (size_t)12 * C, nK4 = (size_t)4 * C, nF = (size_t)9 * C * C, nK9 = (size_t)9 * C;
  std::vector<double> h(nPq + nRT + nK4 + nF + nK9, 0.0);
  double* Pq = h.data();
  double* RT = Pq + nPq;
  double* K4 = RT + nRT;
  double* F = K4 + nK4;
  double* K9 = F + nF;
  std::vector<double> Ptrue((size_t)12 * C);
  for (int c = 0; c < C; c++) {
    projection(K + 9 * c, R + 9 * c, t + 3 * c, Ptrue.data() + 12 * c);
    memcpy(RT + 12 * c, R + 9 * c, 72);
    memcpy(RT + 12 * c + 9, t + 3 * c, 24);
    K4[4 * c + 0] = K[9 * c + 0];
    K4[4 * c + 1] = K[9 * c + 4];
    K4[4 * c + 2] = K[9 * c + 2];
    K4[4 * c + 3] = K[9 * c + 5];
  }
  if (uniform) {
    memcpy(Pq, Ptrue.data(), sizeof(double) * 12 * C);
  } else {
    // intrinsics by compacted view index j, pose by camera (helpers.py:296-298,305-307)
    for (int j = 0; j < C; j++)
      for (int c = j; c < C; c++) projection(K + 9 * j, R + 9 * c, t + 3 * c, Pq + 12 * ((size_t)j * C + c));
  }
  for (int a = 0; a < C; a++)
    for (int b = 0; b < C; b++)
      if (a != b) fundamental(Ptrue.data() + 12 * a, Ptrue.data() + 12 * b, F + 9 * ((size_t)a * C + b));
  memcpy(K9, K, sizeof(double) * 9 * C);
  ctx->hF.assign(F, F + nF);

  // a frame call may be in flight on another stream of the same context: the mutex serialises
  // host calls; wait for queued device work before replacing the tables it reads.
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->tables.reserve(h.size() * sizeof(double))) return ctx->fail(MOCAP_E_HIP, "hipMalloc(camera tables) failed");
  HIP_TRY(ctx, hipMemcpyAsync(ctx->tables.ptr, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  const double* d = (const double*)ctx->tables.ptr;
  ctx->C = C;
  ctx->cv.C = C;
  ctx->cv.uniformK = uniform ? 1 : 0;
  ctx->cv.f32_rounding = (ctx->flags & MOCAP_OPT_F32_ROUNDING) ? 1 : 0;
  ctx->cv.Pq = d;
  ctx->cv.RT = d + nPq;
  ctx->cv.K4 = ctx->cv.RT + nRT;
  ctx->cv.F = ctx->cv.K4 + nK4;
  ctx->d_K9 = ctx->cv.F + nF;
  // EigCut (mocap_device.hpp): the bound equates the DLT rows' residual with cv.projectPoints', which holds when every
  // K is [[fx,0,cx],[0,fy,cy],[0,0,1]] (projectPoints reads fx, fy, cx, cy only); then P[2] = (R[2], t[2])
  bool plainK = true;
  double p3 = 0.0;
  for (int c = 0; c < C; c++) {
    const double* k = K + 9 * c;
    plainK = plainK && k[1] == 0.0 && k[3] == 0.0 && k[6] == 0.0 && k[7] == 0.0 && k[8] == 1.0;
    const double* r = R + 9 * c;
    p3 = std::fmax(p3, r[6] * r[6] + r[7] * r[7] + r[8] * r[8] + t[3 * c + 2] * t[3 * c + 2]);
  }
  ctx->p3max2 = plainK && std::isfinite(p3) ? p3 * (1.0 + 1e-5) : 0.0;
  // the branch and bound takes its bounds in a frame whose origin is the point closest (least squares) to all optical
  // axes -- any point gives a valid bound, one inside the working volume gives a tight one (eigcut_s1_shifted)
  double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, bb[3] = {0, 0, 0};
  for (int c = 0; c < C; c++) {
    const double* r = R + 9 * c;
    const double* tt = t + 3 * c;
    const double d[3] = {r[6], r[7], r[8]};  // optical axis (R is orthonormal up to the user's rounding)
    const double o[3] = {-(r[0] * tt[0] + r[3] * tt[1] + r[6] * tt[2]), -(r[1] * tt[0] + r[4] * tt[1] + r[7] * tt[2]),
                         -(r[2] * tt[0] + r[5] * tt[1] + r[8] * tt[2])};  // camera centre -R^T t
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        const double pij = (i == j ? 1.0 : 0.0) - d[i] * d[j];
        A[3 * i + j] += pij;
        bb[i] += pij * o[j];
      }
  }
  double c0[3] = {0, 0, 0};
  {
    const double det = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
    if (std::isfinite(det) && std::fabs(det) > 1e-6 * C * C * C) {  // (parallel axes: keep the world origin)
      c0[0] = (bb[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (bb[1] * A[8] - A[5] * bb[2]) + A[2] * (bb[1] * A[7] - A[4] * bb[2])) / det;
      c0[1] = (A[0] * (bb[1] * A[8] - A[5] * bb[2]) - bb[0] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * bb[2] - bb[1] * A[6])) / det;
      c0[2] = (A[0] * (A[4] * bb[2] - bb[1] * A[7]) - A[1] * (A[3] * bb[2] - bb[1] * A[6]) + bb[0] * (A[3] * A[7] - A[4] * A[6])) / det;
      if (!(std::isfinite(c0[0]) && std::isfinite(c0[1]) && std::isfinite(c0[2]))) c0[0] = c0[1] = c0[2] = 0.0;
    }
  }
  double p3c = 0.0;
  for (int c = 0; c < C; c++) {
    const double* r = R + 9 * c;
    const double w = r[6] * c0[0] + r[7] * c0[1] + r[8] * c0[2] + t[3 * c + 2];
    p3c = std::fmax(p3c, r[6] * r[6] + r[7] * r[7] + r[8] * r[8] + w * w);
  }
  for (int i = 0; i < 3; i++) ctx->eig_c0[i] = c0[i];
  ctx->p3max2c = plainK && std::isfinite(p3c) ? p3c * (1.0 + 1e-5) : 0.0;
  return MOCAP_OK;
}

extern "C" int mocap_set_world_transform(mocap_ctx* ctx, const double* to_world) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!to_world) {
    ctx->world_on = false;
    return MOCAP_OK;
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // a queued frame batch may still read the old matrix
  if (ctx->world.reserve(16 * sizeof(double))) return ctx->fail(MOCAP_E_HIP, "hipMalloc(world matrix) failed");
  HIP_TRY(ctx, hipMemcpyAsync(ctx->world.ptr, to_world, 16 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->world_on = true;
  return MOCAP_OK;
}

extern "C" int mocap_get_fundamental(mocap_ctx* ctx, double* F) {
  if (!ctx || !F) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!ctx->C) return ctx->fail(MOCAP_E_NOCAMS, "mocap_set_cameras has not been called");
  memcpy(F, ctx->hF.data(), ctx->hF.size() * sizeof(double));
  return MOCAP_OK;
}

// ------------------------------------------------------------------ triangulation
static int triangulate_dev_locked(mocap_ctx* ctx, int64_t N, const double* d_obs, double* d_xyz, double* d_err) {
  if (!ctx->C) return ctx->fail(MOCAP_E_NOCAMS, "mocap_set_cameras has not been called");
  if (N < 0 || (N > 0 && !d_obs)) return ctx->fail(MOCAP_E_ARG, "mocap_triangulate: bad argument");
  TriArgs a;
  a.cv = ctx->cv;
  a.N = N;
  a.P = 1;
  a.stride_Pq = a.stride_RT = 0;
  a.obs = d_obs;
  a.xyz = d_xyz;
  a.err = d_err;
  a.xyz_in = nullptr;
  HIP_TRY(ctx, launch_triangulate(a, ctx->stream));
  return MOCAP_OK;
}

extern "C" int mocap_triangulate_dev(mocap_ctx* ctx, int64_t N, const double* d_obs, double* d_xyz, double* d_err) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int rc = triangulate_dev_locked(ctx, N, d_obs, d_xyz, d_err);
  return rc ? rc : ctx->mark_enqueued();
}

extern "C" int mocap_triangulate(mocap_ctx* ctx, int64_t N, const double* obs, double* xyz, double* err) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!ctx->C) return ctx->fail(MOCAP_E_NOCAMS, "mocap_set_cameras has not been called");
  if (N < 0 || (N > 0 && (!obs || !xyz))) return ctx->fail(MOCAP_E_ARG, "mocap_triangulate: bad argument");
  if (N == 0) return MOCAP_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int C = ctx->C;
  const size_t b_obs = sizeof(double) * (size_t)N * C * 2, b_xyz = sizeof(double) * (size_t)N * 3,
               b_err = sizeof(double) * (size_t)N;
  DevBuf& s = ctx->scratch[0];
  if (s.reserve(b_obs + b_xyz + b_err)) return ctx->fail(MOCAP_E_HIP, "hipMalloc(%zu) failed", b_obs + b_xyz + b_err);
  double* d_obs = (double*)s.ptr;
  double* d_xyz = d_obs + (size_t)N * C * 2;
  double* d_err = d_xyz + (size_t)N * 3;
  HIP_TRY(ctx, hipMemcpyAsync(d_obs, obs, b_obs, hipMemcpyHostToDevice, ctx->stream));
  int rc = triangulate_dev_locked(ctx, N, d_obs, d_xyz, d_err);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(xyz, d_xyz, b_xyz, hipMemcpyDeviceToHost, ctx->stream));
  if (err) HIP_TRY(ctx, hipMemcpyAsync(err, d_err, b_err, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MOCAP_OK;
}

extern "C" int mocap_reproject(mocap_ctx* ctx, int64_t N, const double* obs, const double* xyz, double* err) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!ctx->C) return ctx->fail(MOCAP_E_NOCAMS, "mocap_set_cameras has not been called");
  if (N < 0 || (N > 0 && (!obs || !xyz || !err))) return ctx->fail(MOCAP_E_ARG, "mocap_reproject: bad argument");
  if (N == 0) return MOCAP_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int C = ctx->C;
  const size_t b_obs = sizeof(double) * (size_t)N * C * 2, b_xyz = sizeof(double) * (size_t)N * 3,
               b_err = sizeof(double) * (size_t)N;
  DevBuf& s = ctx->scratch[0];
  if (s.reserve(b_obs + b_xyz + b_err)) return ctx->fail(MOCAP_E_HIP, "hipMalloc(%zu) failed", b_obs + b_xyz + b_err);
  double* d_obs = (double*)s.ptr;
  double* d_xyz = d_obs + (size_t)N * C * 2;
  double* d_err = d_xyz + (size_t)N * 3;
  HIP_TRY(ctx, hipMemcpyAsync(d_obs, obs, b_obs, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_xyz, xyz, b_xyz, hipMemcpyHostToDevice, ctx->stream));
  TriArgs a;
  a.cv = ctx->cv;
  a.N = N;
  a.P = 1;
  a.stride_Pq = a.stride_RT = 0;
  a.obs = d_obs;
  a.xyz = nullptr;
  a.err = d_err;
  a.xyz_in = d_xyz;
  HIP_TRY(ctx, launch_triangulate(a, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(err, d_err, b_err, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MOCAP_OK;
}

// ------------------------------------------------------------------ frame path
namespace {
// which kernel a frame batch of this shape takes (the decision is per launch and never changes a result)
struct FramePlan {
  int T = 256, hit_cap = 1;
  bool wide = false, use_bb = false;
  size_t lds = 0;
};
FramePlan plan_frame(const mocap_ctx* ctx, int M_max, int K_max, int hit_cap_override) {
  FramePlan pl;
  // automatic workgroup size: tiny frames (4 x 4: a handful of candidates) are latency-bound, one wave
  // per frame keeps 4x more frames in flight per CU; everything else wants 256 lanes per frame
  pl.T = ctx->frame_threads ? ctx->frame_threads : (ctx->C * M_max <= 32 ? 64 : 256);
  const int cap = hit_cap_override > 0 ? hit_cap_override : ctx->hit_cap;
  pl.hit_cap = cap < 1 ? 1 : (cap > M_max ? M_max : cap);
  // (a narrow frame with identical intrinsics keeps blob indices in one byte with 0xFF = none: 256 slots go wide)
  pl.wide = ctx->force_wide != 0 || (M_max > 255 && ctx->cv.uniformK);
  // The realistic rigs go to their own kernel (csrc/frame_bb.hip: exact branch and bound): identical plain intrinsics
  // (the eigenvalue bounds need K = [[fx,0,cx],[0,fy,cy],[0,0,1]]), <= 16 cameras, <= 64 blobs per camera, <= 255 roots,
  // frames big enough for a 256-lane workgroup (or 256 lanes asked for: MOCAP_FRAME_THREADS / mocap_set_tuning).  Everything
  // else -- and MOCAP_EVAL_BB=0 -- takes the exhaustive walk.  (Decided before narrow / wide: its layout has no odometer columns and fits where the general narrow one does not.)
  pl.use_bb = ctx->eval_bb && !ctx->exhaustive && !pl.wide && ctx->cv.uniformK && ctx->prune && ctx->eigcut && ctx->p3max2 > 0.0 && ctx->p3max2c > 0.0 &&
              (ctx->frame_threads == 256 || (ctx->frame_threads == 0 && ctx->C * M_max > 32)) && ctx->frame_launches != 3 &&
              frame_bb_fits(ctx->C, M_max, K_max);
  if (pl.use_bb) {
    pl.T = 256;
    pl.lds = frame_bb_lds_bytes(ctx->C, M_max, K_max);
  } else if (!pl.wide) {
    pl.lds = frame_lds_bytes(ctx->C, M_max, K_max, pl.T, pl.hit_cap, false, ctx->cv.uniformK != 0);
    while (pl.lds > 160 * 1024 && pl.T > 64) {
      pl.T /= 2;
      pl.lds = frame_lds_bytes(ctx->C, M_max, K_max, pl.T, pl.hit_cap, false, ctx->cv.uniformK != 0);
    }
    pl.wide = pl.lds > 160 * 1024;  // the frame state does not fit LDS: big tables go to an HBM workspace
  }
  if (pl.wide) {
    pl.T = kWideThreads;
    pl.lds = frame_lds_bytes(ctx->C, M_max, K_max, pl.T, pl.hit_cap, true, false);
    // Round 6: 512 lanes per frame and TWO frames per CU wherever the LDS holds two frame states (64 cameras x 256 blobs:
    // up to ~400 roots).  The same 16 waves per CU and 128 VGPRs, no arithmetic changed -- but two independent frames in
    // different phases (the matching's scalar-heavy pre-test loop, the geometry's FP64) share the SIMDs' issue slots, and
    // every barrier waits for 8 waves instead of 16: 33.3 -> 28.9 ms per 12 500 stress frames.  MOCAP_WIDE_THREADS=1024 = the old plan.
    const char* wt = getenv("MOCAP_WIDE_THREADS");
    const size_t l2 = frame_lds_bytes(ctx->C, M_max, K_max, 512, pl.hit_cap, true, false);
    if (!(wt && atoi(wt) == 1024) && 2 * l2 <= (size_t)160 * 1024 && ctx->frame_launches != 3) {
      pl.T = 512;
      pl.lds = l2;
    }
  }
  return pl;
}
}  // namespace

// hit_cap_override > 0: the hit-list cap of THIS launch (the re-submit pass keeps every gated hit) -- an argument, never a
// change of the context's state.  n_frames_dev != null: the batch is min(*n_frames_dev, n_frames) frames long.
// heavy: null, or the export buffer of the heavy-root search (re-submit pass, wide variant only): roots over G_cap are
// exported instead of flagging their frames (FrameArgs::heavy_bb)
struct HeavyHook {
  int32_t* count;
  unsigned char* recs;
  int cap;
};
static int match_dev_locked(mocap_ctx* ctx, int64_t n_frames, int M_max, const float* d_blobs,
                            const int32_t* d_counts, double gate_px, int K_max, int64_t G_cap, double* d_xyz,
                            double* d_err, int16_t* d_corr, int32_t* d_n_out, int32_t* d_status,
                            int32_t* d_n_cand, int hit_cap_override = 0, const int32_t* n_frames_dev = nullptr,
                            const HeavyHook* heavy = nullptr) {
  if (!ctx->C) return ctx->fail(MOCAP_E_NOCAMS, "mocap_set_cameras has not been called");
  if (n_frames < 0 || M_max < 1 || K_max < 1 || G_cap < 1)
    return ctx->fail(MOCAP_E_ARG, "mocap_match_triangulate: bad size argument");
  if (n_frames == 0) return MOCAP_OK;
  if (!d_blobs || !d_counts || !d_xyz || !d_err || !d_corr || !d_n_out || !d_status)
    return ctx->fail(MOCAP_E_ARG, "mocap_match_triangulate: null buffer");
  if (M_max > kMaxBlobs) return ctx->fail(MOCAP_E_LIMIT, "M_max=%d exceeds %d", M_max, kMaxBlobs);
  if (G_cap > (1ll << 24)) G_cap = 1ll << 24;  // 32-bit candidate offsets per frame
  FrameArgs a;
  a.cv = ctx->cv;
  a.n_frames = n_frames;
  a.n_frames_dev = n_frames_dev;
  a.M = M_max;
  a.K_max = K_max;
  a.gate_px = gate_px;
  a.G_cap = G_cap;
  a.blobs = d_blobs;
  a.counts = d_counts;
  a.xyz = d_xyz;
  a.err = d_err;
  a.corr = d_corr;
  a.n_out = d_n_out;
  a.status = d_status;
  a.n_cand = d_n_cand;
  a.world = ctx->world_on ? (const double*)ctx->world.ptr : nullptr;
  const FramePlan pl = plan_frame(ctx, M_max, K_max, hit_cap_override);
  int T = pl.T;
  const int hit_cap = pl.hit_cap;
  const bool wide = pl.wide, use_bb = pl.use_bb;
  size_t lds = pl.lds;
  if (wide && lds > 160 * 1024)
    return ctx->fail(MOCAP_E_LIMIT, "frame state needs %zu B of LDS (C=%d, M_max=%d, K_max=%d): lower K_max",
                     lds, ctx->C, M_max, K_max);
  a.H = hit_cap;
  a.wide = wide ? 1 : 0;
  if (wide) {  // A/B and tests: MOCAP_WIDE_SPEC=0 = the chain over the cameras strictly camera by camera (frame_kernel.hip spec_begin)
    const char* sp = getenv("MOCAP_WIDE_SPEC");
    if (sp && atoi(sp) == 0) a.wide = 2;
  }
  a.prune = ctx->prune && !ctx->exhaustive;
  a.p3max2 = a.prune && ctx->eigcut ? ctx->p3max2 : 0.0;
  if (!wide && ctx->frame_threads == 0 && T == 64) a.p3max2 = 0.0;  // tiny frames (a handful of candidates): the cut-offs cost more than they save
  a.eval_bb = use_bb ? 1 : 0;
  a.bb_pl = ctx->bb_pl;
  for (int i = 0; i < 3; i++) a.bb_c0[i] = ctx->eig_c0[i];
  a.p3max2c = ctx->p3max2c;
  a.bb_flush = ctx->bb_flush > 0 ? ctx->bb_flush : 256;
  a.bb_min_g = ctx->bb_min_g;
  while (a.bb_pl > 1 && (size_t)a.bb_pl * M_max * 2 * 256 >= ((size_t)1 << 22)) a.bb_pl /= 2;  // expanded-list counter: 22 bits
  a.ws = nullptr;
  a.ws_stride = 0;
  a.heavy_bb = 0;
  a.heavy_cap = 0;
  a.heavy_count = nullptr;
  a.heavy_recs = nullptr;
  a.heavy_stride = 0;
  if (heavy && wide) {
    a.heavy_bb = 1;
    a.heavy_cap = heavy->cap;
    a.heavy_count = heavy->count;
    a.heavy_recs = heavy->recs;
    a.heavy_stride = heavy_rec_bytes(ctx->C, hit_cap);
  }
  // persistent grid: enough workgroups to fill every CU at the LDS-limited occupancy
  int per_cu = (int)((160 * 1024) / lds);
  const int wave_cap = use_bb ? frame_bb_wg_per_cu_cap(ctx->C, M_max, K_max) : (16 / (T / 64) > 0 ? 16 / (T / 64) : 1);  // 128 VGPRs -> 16 waves per CU (the headline kernel: its instantiation's own budget)
  if (per_cu > wave_cap) per_cu = wave_cap;
  if (per_cu < 1) per_cu = 1;
  const int64_t full_grid = (int64_t)ctx->num_cus * per_cu;
  int64_t grid = full_grid < n_frames ? full_grid : n_frames;
  if (use_bb && frame_bb_ws_bytes(ctx->C)) {
    a.ws_stride = frame_bb_ws_bytes(ctx->C);
    if (ctx->frame_ws.reserve((size_t)full_grid * a.ws_stride))
      return ctx->fail(MOCAP_E_HIP, "hipMalloc(search workspace, %zu B) failed", (size_t)full_grid * a.ws_stride);
    a.ws = (unsigned char*)ctx->frame_ws.ptr;
  }
  if (wide) {
    a.ws_stride = frame_ws_bytes(ctx->C, M_max, K_max, T, hit_cap, true, false);
    if (ctx->frame_ws.reserve((size_t)full_grid * a.ws_stride))
      return ctx->fail(MOCAP_E_HIP, "hipMalloc(wide-frame workspace, %zu B) failed", (size_t)full_grid * a.ws_stride);
    a.ws = (unsigned char*)ctx->frame_ws.ptr;
  }

  // work queues: heavy-frame list + slice partials (scheduling note in frame_kernel.hip)
  const bool batch = n_frames >= 2 * full_grid;
  FrameQueues& q = a.q;
  q.heavy_threshold = ctx->heavy_threshold >= 0 ? (uint32_t)ctx->heavy_threshold : (batch ? 32768u : 16u * T);  // batch: swept under the single-launch schedule (16 k: 13.45, 24-32 k: 13.32, 48 k: 13.50, 64 k: 13.72 ms per 100 k frames); live calls: swept, p50 0.129 -> 0.117 ms vs 2T, same p99
  // (wide frames: every slice re-does the frame's matching, ~2/3 of an average frame's time, so the slices are three times as long --
  // swept at the stress shape on two streams, scripts/gpu_r06_t.sh: 8 192 -> 24 576 candidates: 26.97 -> 25.65 and 24.82 -> 24.56 ms per
  // 12 500 frames; 16 384 / 20 480 / 28 672 / 32 768 in between or worse, 65 536: 30.5)
  q.slice_size = ctx->slice_size > 0 ? (uint32_t)ctx->slice_size : (batch ? (wide ? 24576u : 8192u) : 4u * T);
  if (a.heavy_bb) q.heavy_threshold = 0;  // (a sliced frame would be matched, and its heavy roots exported, once per slice)
  // small frames: amortise the queue atomic over a chunk (keeps >= 64 chunks per workgroup for balance);
  // frames with real work keep the finest granularity, their candidate counts are heavy-tailed
  q.frame_chunk = 1;
  if (ctx->C * M_max <= 32) {
    int64_t ch = n_frames / (full_grid * 64);
    q.frame_chunk = (int)(ch < 1 ? 1 : (ch > 16 ? 16 : ch));
  }
  int64_t H = n_frames / 8;
  if (H < 64) H = 64;
  if (H > n_frames) H = n_frames;
  q.H_cap = (int)H;
  q.W_cap = (int)(H * 8 < 64 ? 64 : H * 8);
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  const size_t b_cnt = al(sizeof(int32_t) * QC_COUNT), b_heavy = al(sizeof(int32_t) * 4 * (size_t)q.H_cap),
               b_slice = al(sizeof(int32_t) * (size_t)q.W_cap), b_gen = b_slice, b_pe = al(sizeof(double) * (size_t)q.W_cap * K_max),
               b_pg = al(sizeof(uint32_t) * (size_t)q.W_cap * K_max), b_px = al(sizeof(double) * 3 * (size_t)q.W_cap * K_max);
  const int qs = n_frames_dev ? 1 : 0;  // the re-submit's second pass keeps queues of its own
  DevBuf& wq = qs ? ctx->resub_q : ctx->scratch[3];
  const void* wq_before = wq.ptr;
  if (wq.reserve(b_cnt + b_heavy + b_slice + b_gen + b_pe + b_pg + b_px))
    return ctx->fail(MOCAP_E_HIP, "hipMalloc(frame work queues) failed");
  // Big batches of tiny frames (one-wave workgroups: 4 x 4) are bound by how many frames are in flight, and the lean
  // kernel of the three-launch schedule keeps twice the waves of the all-in-one kernel resident (measured on 1 M frames
  // of 4 x 4: 4.8 vs 8.2 ms); everything else takes the one persistent launch
  const bool tiny_batch = !wide && ctx->frame_threads == 0 && T == 64 && n_frames >= 4096;
  const bool one_launch = ctx->frame_launches != 3 && !(tiny_batch && !getenv("MOCAP_FRAME_LAUNCHES"));
  // one-launch schedule: the kernel leaves the counters at zero and slices carry a launch generation, so the queue
  // needs clearing only when the buffer is new, the layout moved, or the other schedule used it last
  const bool fresh = wq.ptr != wq_before || ctx->frame_q_cap[qs] != q.W_cap || !ctx->frame_q_clean[qs];
  char* w = (char*)wq.ptr;
  q.counters = (int32_t*)w;     w += b_cnt;
  q.slice_heavy = (int32_t*)w;  w += b_slice;
  q.slice_gen = (int32_t*)w;    w += b_gen;
  q.gen = ++ctx->frame_gen;
  if (ctx->frame_gen == 0x7fffffff) {  // generation wrap: start over from a cleared queue
    ctx->frame_gen = 0;
    ctx->frame_q_dirty();
  }
  q.heavy = (int32_t*)w;        w += b_heavy;
  q.part_e = (double*)w;        w += b_pe;
  q.part_g = (uint32_t*)w;      w += b_pg;
  q.part_x = (double*)w;
  if ((!one_launch && !use_bb) || fresh) {
    HIP_TRY(ctx, hipMemsetAsync(q.counters, 0, b_cnt, ctx->stream));
    if (q.heavy_threshold) HIP_TRY(ctx, hipMemsetAsync(q.slice_heavy, 0xFF, b_slice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(q.slice_gen, 0, b_gen, ctx->stream));
  }
  ctx->frame_q_cap[qs] = q.W_cap;
  ctx->frame_q_clean[qs] = false;  // until the launch below is known to be queued
  if (use_bb) {
    // frames only: a frame's cost follows its surviving blocks, not its candidate count -- no heavy list, no slices
    ctx->last_frame_kernel = ctx->C <= 8 ? "frame_bb_kernel<CW=1>" : "frame_bb_kernel<CW=2>";
    HIP_TRY(ctx, launch_frame_bb(a, (int)grid, ctx->stream));
    ctx->frame_q_clean[qs] = true;
    return MOCAP_OK;
  }
  ctx->last_frame_kernel = wide ? (T == 512 ? "frame_kernel<512, wide>" : "frame_kernel<1024, wide>") : (T == 64 ? "frame_kernel<64>" : (T == 128 ? "frame_kernel<128>" : "frame_kernel<256>"));
  if (one_launch) {
    // one launch: frames, then slices of the heavy frames, merged by the workgroup that finishes a frame's last slice.
    // Few frames (live calls): still enough workgroups for a heavy frame's slices to run side by side.
    int64_t g1 = n_frames + (q.heavy_threshold ? 64 : 0);
    if (g1 > full_grid) g1 = full_grid;
    HIP_TRY(ctx, launch_frame_kernel(a, MODE_ALL, T, (int)g1, ctx->stream));
    ctx->frame_q_clean[qs] = true;
    return MOCAP_OK;
  }
  HIP_TRY(ctx, launch_frame_kernel(a, MODE_MAIN, T, (int)grid, ctx->stream));
  if (q.heavy_threshold) {
    HIP_TRY(ctx, launch_frame_kernel(a, MODE_SLICE, T, (int)(full_grid < q.W_cap ? full_grid : q.W_cap), ctx->stream));
    HIP_TRY(ctx, launch_frame_kernel(a, MODE_MERGE, T, (int)(full_grid < q.H_cap ? full_grid : q.H_cap), ctx->stream));
  }
  return MOCAP_OK;
}

// does a frame batch with these sizes fit one of the frame kernels?  (the size logic of plan_frame, hit lists uncapped)
static bool frame_shape_fits(const mocap_ctx* ctx, int M_max, int K) {
  if (frame_bb_fits(ctx->C, M_max, K) && ctx->cv.uniformK && !ctx->force_wide && M_max <= 255) return true;
  const bool must_wide = ctx->force_wide != 0 || (M_max > 255 && ctx->cv.uniformK);
  if (!must_wide && frame_lds_bytes(ctx->C, M_max, K, 64, M_max, false, ctx->cv.uniformK != 0) <= (size_t)160 * 1024) return true;
  return frame_lds_bytes(ctx->C, M_max, K, kWideThreads, M_max, true, false) <= (size_t)160 * 1024;
}

// ------------------------------------------------------------------ re-submit on the device (uncapped enumeration)
// The reference enumerates the full Cartesian product whatever its size (helpers.py:394-400); the frame path works under
// caps (K_max roots, G_cap groups per root, hit_cap hits per pair of the wide variant) and reports per frame when one was
// hit.  Behind a first pass that is already queued: the flagged frames are gathered, on the device, into a scratch batch
// whose length stays on the device; the frame kernel runs on it with the largest caps the core has (root capacity C *
// M_max as far as a kernel's LDS holds it, G_cap = 2^24 groups per root, every gated hit of a (root, camera) pair kept);
// results that fit the caller's K_max slots are scattered back, the others report ROOT_OVERFLOW and the slots they need.
// Nothing here waits for the GPU: three enqueues behind the first pass (an empty list costs ~15 us of GPU time).
// d_info: null, or [2] device-accessible: {frames flagged, frames re-run}.
static int resubmit_dev_locked(mocap_ctx* ctx, int64_t n_frames, int M_max, const float* d_blobs, const int32_t* d_counts,
                               double gate_px, int K_max, double* d_xyz, double* d_err, int16_t* d_corr, int32_t* d_n_out,
                               int32_t* d_status, int32_t* d_n_cand, int32_t* d_info) {
  if (n_frames <= 0) return MOCAP_OK;
  const int C = ctx->C;
  // worst-case root capacity: every blob its own root (never less than the caller asked for) ...
  int K_big = C * M_max < 1024 ? C * M_max : 1024;
  if (K_big < K_max) K_big = K_max;
  if (!frame_shape_fits(ctx, M_max, K_big)) {
    // ... as far as the frame state fits a kernel (64 cameras x 256 blobs: the per-root tables of the wide variant end at
    // a few hundred roots); a frame with more roots than that keeps its root-overflow status
    int lo = K_max, hi = K_big;  // largest K in [K_max, K_big) that fits (K_max itself ran above)
    while (lo < hi) {
      const int mid = (lo + hi + 1) / 2;
      if (frame_shape_fits(ctx, M_max, mid)) lo = mid; else hi = mid - 1;
    }
    K_big = lo;
  }
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  const size_t per_frame = sizeof(float) * C * M_max * 2 + sizeof(int32_t) * C + (size_t)K_big * (32 + 2 * C) + 16;
  // the scratch batch holds every frame of the caller's batch unless that takes more than MOCAP_RESUBMIT_SCRATCH_MB
  // (default 8192); beyond it, flagged frames keep their status (d_info[0] > d_info[1] says so: call again)
  size_t budget = (size_t)8192 << 20;
  if (const char* e = getenv("MOCAP_RESUBMIT_SCRATCH_MB")) budget = (size_t)(atol(e) > 0 ? atol(e) : 1) << 20;
  // The scratch batch is sized for the flagged share one expects, not for the whole batch (round-5 advice: a full copy of a
  // 100 k-frame batch was reserved up front although no frame might be flagged): the whole batch while that costs at most
  // 256 MB (a caller's tiny G_cap may flag every frame of a small batch), else one frame in eight, at least 1 024 and at least
  // what 256 MB hold.  More
  // flagged frames than that keep their status without MOCAP_ST_FINAL and d_info says so; mocap_resubmit_dev (and the
  // host-buffer entry points, in a loop) continue with them.  An allocation that fails is retried at half the size down to
  // one frame: the first pass has succeeded by now, a missing scratch must not fail the call.
  int64_t cap = n_frames / 8 < 1024 ? 1024 : n_frames / 8;
  if (cap < (int64_t)(((size_t)256 << 20) / per_frame)) cap = (int64_t)(((size_t)256 << 20) / per_frame);
  if (const char* e = getenv("MOCAP_RESUBMIT_SCRATCH_FRAMES")) cap = atol(e) > 0 ? atol(e) : 1;  // (tests: a scratch smaller than the flagged set)
  if (cap > n_frames) cap = n_frames;
  if ((size_t)cap * per_frame > budget) cap = (int64_t)(budget / per_frame);
  if (cap < 1) cap = 1;
  size_t F2, b_list, b_b2, b_c2, b_x2, b_e2, b_r2, b_i;
  for (;;) {
    F2 = (size_t)cap;
    b_list = al(4 * F2), b_b2 = al(sizeof(float) * F2 * C * M_max * 2), b_c2 = al(4 * F2 * C);
    b_x2 = al(8 * F2 * K_big * 3), b_e2 = al(8 * F2 * K_big), b_r2 = al(2 * F2 * K_big * C), b_i = al(4 * (F2 + 2));
    if (!ctx->resub.reserve(b_list + b_b2 + b_c2 + b_x2 + b_e2 + b_r2 + 3 * b_i)) break;
    (void)hipGetLastError();  // (the failed hipMalloc's sticky error)
    if (cap == 1) return ctx->fail(MOCAP_E_HIP, "hipMalloc(re-submit scratch, %zu B for ONE frame) failed", b_list + b_b2 + b_c2 + b_x2 + b_e2 + b_r2 + 3 * b_i);
    cap = (cap + 1) / 2;
  }
  if (!ctx->resub_ctr.ptr) {
    if (ctx->resub_ctr.reserve(256)) return ctx->fail(MOCAP_E_HIP, "hipMalloc(re-submit counters) failed");
    HIP_TRY(ctx, hipMemsetAsync(ctx->resub_ctr.ptr, 0, 256, ctx->stream));
  }
  char* w = (char*)ctx->resub.ptr;
  ResubmitArgs ra;
  ra.n_frames = n_frames;
  ra.cap = cap;
  ra.C = C;
  ra.M = M_max;
  ra.K_max = K_max;
  ra.K_big = K_big;
  ra.status = d_status;
  ra.blobs = d_blobs;
  ra.counts = d_counts;
  // two counters, 128 bytes apart, alternate between calls: this call's is zero (the previous call's gather cleared it)
  // (the parity advances only once the gather -- which zeroes the OTHER counter for the next call -- is known to be queued:
  // a failure before that leaves this call's counter untouched and still zero)
  int32_t* ctr = (int32_t*)ctx->resub_ctr.ptr;
  const uint32_t par = ctx->resub_calls & 1u;
  ra.count = ctr + 32 * par;
  ra.count_next = ctr + 32 * (par ^ 1u);
  ra.list = (int32_t*)w;   w += b_list;
  ra.b2 = (float*)w;       w += b_b2;
  ra.c2 = (int32_t*)w;     w += b_c2;
  double* x2 = (double*)w;   w += b_x2;
  double* e2 = (double*)w;   w += b_e2;
  int16_t* r2 = (int16_t*)w; w += b_r2;
  int32_t* n2 = (int32_t*)w; w += b_i;
  int32_t* s2 = (int32_t*)w; w += b_i;   // (+ 2 slots: the self-check builds count in status[n_frames .. n_frames + 1])
  int32_t* g2 = (int32_t*)w;
  ra.x2 = x2;  ra.e2 = e2;  ra.r2 = r2;  ra.n2 = n2;  ra.s2 = s2;  ra.g2 = g2;
  ra.xyz = d_xyz;
  ra.err = d_err;
  ra.corr = d_corr;
  ra.n_out = d_n_out;
  ra.status_out = d_status;
  ra.n_cand = d_n_cand;
  ra.info = d_info;
  // Roots whose product no enumeration reaches (two markers behind each other from the root's camera: 2^60 groups at 64
  // cameras) go to the heavy-root search (csrc/heavy_bb.hip) where the second pass runs the wide variant on cameras of the
  // form EigCut needs: the pass enumerates up to MOCAP_RESUBMIT_G_CAP groups per root (default 4096) and exports the
  // roots above it; elsewhere it enumerates up to 2^24 per root and flags what is larger, as before.
  const FramePlan pl2 = plan_frame(ctx, M_max, K_big, M_max);
  const bool heavy_ok = pl2.wide && ctx->cv.uniformK && ctx->prune && ctx->eigcut && ctx->p3max2 > 0.0 && !ctx->exhaustive && !getenv("MOCAP_NO_HEAVY_BB");
  int64_t G2 = (int64_t)1 << 24;
  HeavyHook hk{nullptr, nullptr, 0};
  int ncap = 4096;  // (swept on the stress stream: 16 384 and 65 536 solve 1-3 more of ~30 hard roots per 12 500 frames and double the step)
  const int hv_grid = 64;
  const int enum_grid = ctx->num_cus * 3;  // heavy_enum_kernel: 256-lane workgroups (168 VGPRs: three waves per SIMD), the whole GPU on one root at a time
  if (const char* e = getenv("MOCAP_HEAVY_NCAP")) ncap = atoi(e) >= 1 ? atoi(e) : 1;  // (tests: 1 = the search gives up at the first level that keeps two nodes)
  if (heavy_ok) {
    G2 = 4096;
    if (const char* e = getenv("MOCAP_RESUBMIT_G_CAP")) G2 = atol(e) > 0 ? atol(e) : 1;
    hk.cap = 2048;
    if (ctx->heavy_recs.reserve((size_t)hk.cap * heavy_rec_bytes(C, M_max)) || ctx->heavy_ws.reserve((size_t)hv_grid * heavy_bb_ws_bytes(ncap)) ||
        ctx->heavy_enum.reserve(heavy_enum_ws_bytes(kHeavyEnumMax, enum_grid)))
      return ctx->fail(MOCAP_E_HIP, "hipMalloc(heavy-root search buffers) failed");
    hk.recs = (unsigned char*)ctx->heavy_recs.ptr;
    hk.count = ctr + 16;  // (its own word of the counter block; the gather kernel zeroes it)
    ra.heavy_count = hk.count;
    ra.enum_count = ctr + 17;
  } else {
    ra.heavy_count = nullptr;
    ra.enum_count = nullptr;
  }
  HIP_TRY(ctx, launch_resubmit_gather(ra, ctx->stream));
  ctx->resub_calls++;
  const char* batch_kernel = ctx->last_frame_kernel;  // mocap_last_frame_kernel() keeps naming the pass that did the batch, not the repair of its flagged frames
  const int rc = match_dev_locked(ctx, cap, M_max, ra.b2, ra.c2, gate_px, K_big, G2, x2, e2, r2, n2, s2, g2,
                                  /*hit_cap_override=*/M_max, /*n_frames_dev=*/ra.count, heavy_ok ? &hk : nullptr);
  if (std::strcmp(batch_kernel, "none") != 0) ctx->last_frame_kernel = batch_kernel;
  if (rc) return rc;
  if (heavy_ok) {
    HeavyArgs ha;
    ha.cv = ctx->cv;
    ha.M = M_max;
    ha.K_big = K_big;
    for (int i = 0; i < 3; i++) ha.bb_c0[i] = ctx->eig_c0[i];
    ha.p3max2c = ctx->p3max2c;
    ha.p3max2 = ctx->p3max2;
    ha.blobs = ra.b2;
    ha.heavy_count = hk.count;
    ha.recs = hk.recs;
    ha.cap = hk.cap;
    ha.stride = heavy_rec_bytes(C, M_max);
    ha.xyz = x2;
    ha.err = e2;
    ha.corr = r2;
    ha.n_out = n2;
    ha.status = s2;
    ha.world = ctx->world_on ? (const double*)ctx->world.ptr : nullptr;
    ha.ws = (unsigned char*)ctx->heavy_ws.ptr;
    ha.ws_stride = heavy_bb_ws_bytes(ncap);
    ha.ncap = ncap;
    ha.enum_cap = (int64_t)1 << 16;  // (2^20 in place costs tens of ms on one CU: the first pass, which slices such roots over 64 workgroups, is the place for them)
    if (const char* e = getenv("MOCAP_HEAVY_ENUM_CAP")) ha.enum_cap = atol(e) >= 0 ? atol(e) : 0;
    ha.debug = getenv("MOCAP_HEAVY_DEBUG") ? 1 : 0;
    {
      // roots the search gives up on with at most 2^24 groups are enumerated by the whole GPU behind it (heavy_enum_kernel):
      // the pass stays exact up to 2^24 groups per root, like the enumeration it replaces (round-5 advice)
      char* e = (char*)ctx->heavy_enum.ptr;
      ha.enum_max = (getenv("MOCAP_NO_HEAVY_ENUM") || (ctx->flags & MOCAP_OPT_BOUNDED_RESUBMIT)) ? 0 : kHeavyEnumMax;
      ha.enum_grid = enum_grid;
      ha.enum_count = ctr + 17;
      ha.enum_list = (int32_t*)e;    e += 4 * (size_t)kHeavyEnumMax;
      ha.enum_slice = (int32_t*)e;   e += 4 * (size_t)kHeavyEnumMax;
      ha.enum_done = (int32_t*)e;    e += 4 * (size_t)kHeavyEnumMax;
      e = (char*)(((uintptr_t)e + 63) / 64 * 64);
      ha.enum_bound = (unsigned long long*)e;  e += 8 * (size_t)kHeavyEnumMax;
      ha.enum_part = (unsigned char*)e;
    }
    HIP_TRY(ctx, launch_heavy_bb(ha, hv_grid, ctx->stream));
    HIP_TRY(ctx, launch_heavy_enum(ha, ctx->stream));
  }
  HIP_TRY(ctx, launch_resubmit_scatter(ra, ctx->stream));
  return MOCAP_OK;
}

extern "C" int mocap_match_triangulate_dev(mocap_ctx* ctx, int64_t n_frames, int M_max, const float* d_blobs,
                                           const int32_t* d_counts, double gate_px, int K_max, int64_t G_cap,
                                           double* d_xyz, double* d_err, int16_t* d_corr, int32_t* d_n_out,
                                           int32_t* d_status, int32_t* d_n_cand) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int rc = match_dev_locked(ctx, n_frames, M_max, d_blobs, d_counts, gate_px, K_max, G_cap, d_xyz, d_err, d_corr,
                                  d_n_out, d_status, d_n_cand);
  return rc ? rc : ctx->mark_enqueued();
}

extern "C" int mocap_resubmit_dev(mocap_ctx* ctx, int64_t n_frames, int M_max, const float* d_blobs, const int32_t* d_counts,
                                  double gate_px, int K_max, double* d_xyz, double* d_err, int16_t* d_corr, int32_t* d_n_out,
                                  int32_t* d_status, int32_t* d_n_cand, int32_t* d_resubmitted) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!ctx->C) return ctx->fail(MOCAP_E_NOCAMS, "mocap_set_cameras has not been called");
  if (n_frames < 0 || M_max < 1 || K_max < 1) return ctx->fail(MOCAP_E_ARG, "mocap_resubmit_dev: bad size argument");
  if (n_frames > 0 && (!d_blobs || !d_counts || !d_xyz || !d_err || !d_corr || !d_n_out || !d_status))
    return ctx->fail(MOCAP_E_ARG, "mocap_resubmit_dev: null buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int rc = resubmit_dev_locked(ctx, n_frames, M_max, d_blobs, d_counts, gate_px, K_max, d_xyz, d_err, d_corr, d_n_out, d_status,
                                     d_n_cand, d_resubmitted);
  return rc ? rc : ctx->mark_enqueued();
}

extern "C" int mocap_match_triangulate_dev_auto(mocap_ctx* ctx, int64_t n_frames, int M_max, const float* d_blobs,
                                                const int32_t* d_counts, double gate_px, int K_max, int64_t G_cap,
                                                double* d_xyz, double* d_err, int16_t* d_corr, int32_t* d_n_out,
                                                int32_t* d_status, int32_t* d_n_cand, int32_t* d_resubmitted) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = match_dev_locked(ctx, n_frames, M_max, d_blobs, d_counts, gate_px, K_max, G_cap, d_xyz, d_err, d_corr,
                            d_n_out, d_status, d_n_cand);
  if (rc) return rc;
  rc = resubmit_dev_locked(ctx, n_frames, M_max, d_blobs, d_counts, gate_px, K_max, d_xyz, d_err, d_corr, d_n_out, d_status,
                           d_n_cand, d_resubmitted);
  return rc ? rc : ctx->mark_enqueued();
}

// host buffers in, host buffers out; resubmit: frames that hit a cap take the device-side second pass before the results
// travel back (one lock, one synchronisation)
static int match_host_locked(mocap_ctx* ctx, int64_t n_frames, int M_max, const float* blobs, const int32_t* counts,
                             double gate_px, int K_max, int64_t G_cap, double* xyz, double* err, int16_t* corr,
                             int32_t* n_out, int32_t* status, int32_t* n_cand, bool resubmit, int32_t* n_resubmitted) {
  if (n_resubmitted) *n_resubmitted = 0;
  if (!ctx->C) return ctx->fail(MOCAP_E_NOCAMS, "mocap_set_cameras has not been called");
  if (n_frames < 0 || M_max < 1 || K_max < 1) return ctx->fail(MOCAP_E_ARG, "mocap_match_triangulate: bad size argument");
  if (n_frames == 0) return MOCAP_OK;
  if (!blobs || !counts || !xyz || !err || !corr || !n_out || !status)
    return ctx->fail(MOCAP_E_ARG, "mocap_match_triangulate: null buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int C = ctx->C;
  const size_t F = (size_t)n_frames;
  const size_t b_blobs = sizeof(float) * F * C * M_max * 2, b_counts = sizeof(int32_t) * F * C,
               b_xyz = sizeof(double) * F * K_max * 3, b_err = sizeof(double) * F * K_max,
               b_corr = sizeof(int16_t) * F * K_max * C, b_i = sizeof(int32_t) * F;
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  const size_t total = al(b_xyz) + al(b_err) + al(b_blobs) + al(b_counts) + al(b_corr) + 3 * al(b_i) + 256;
  // Live tracking (one or a few frames per call, helpers.py:94): zero-copy through pinned host memory.
  // The kernels read the blobs from, and write the points to, device-visible host memory; eight small
  // copy-engine transfers and a sleeping stream synchronise cost several times the kernels themselves.
  // Not for frames that go to the wide variant: it reads the blobs IN PLACE for every root batch and candidate view,
  // which over PCIe from uncached host memory costs far more than one staged copy.
  if (total <= (size_t)256 * 1024 && !plan_frame(ctx, M_max, K_max, 0).wide) {
    if (total > ctx->live_pin_cap) {
      if (ctx->live_pin) (void)hipHostFree(ctx->live_pin);
      ctx->live_pin = nullptr;
      ctx->live_pin_cap = 0;
      HIP_TRY(ctx, hipHostMalloc(&ctx->live_pin, (size_t)256 * 1024, hipHostMallocDefault));
      ctx->live_pin_cap = (size_t)256 * 1024;
    }
    if (!ctx->live_event) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->live_event, hipEventDisableTiming));
    char* p = (char*)ctx->live_pin;
    double* h_xyz = (double*)p;       p += al(b_xyz);
    double* h_err = (double*)p;       p += al(b_err);
    float* h_blobs = (float*)p;       p += al(b_blobs);
    int32_t* h_counts = (int32_t*)p;  p += al(b_counts);
    int16_t* h_corr = (int16_t*)p;    p += al(b_corr);
    int32_t* h_n_out = (int32_t*)p;   p += al(b_i);
    int32_t* h_status = (int32_t*)p;  p += al(b_i);
    int32_t* h_n_cand = (int32_t*)p;  p += al(b_i);
    int32_t* h_info = (int32_t*)p;
    h_info[0] = h_info[1] = 0;
    memcpy(h_blobs, blobs, b_blobs);
    memcpy(h_counts, counts, b_counts);
    int rc = match_dev_locked(ctx, n_frames, M_max, h_blobs, h_counts, gate_px, K_max, G_cap, h_xyz, h_err, h_corr,
                              h_n_out, h_status, h_n_cand);
    if (rc) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->live_event, ctx->stream));
    for (long spins = 0;; spins++) {
      const hipError_t e = hipEventQuery(ctx->live_event);
      if (e == hipSuccess) break;
      if (e != hipErrorNotReady) return ctx->hip_fail(e, "hipEventQuery");
      if (spins > 2000000) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        break;
      }
    }
    bool flagged = false;
    for (size_t f = 0; f < F && resubmit; f++) flagged |= h_status[f] != 0;
    if (flagged) {  // rare: the second pass is queued only when the first one, already back, asks for it
      int total = 0;
      for (;;) {  // (more than one round only when the scratch batch was smaller than the flagged set)
        rc = resubmit_dev_locked(ctx, n_frames, M_max, h_blobs, h_counts, gate_px, K_max, h_xyz, h_err, h_corr, h_n_out, h_status,
                                 h_n_cand, h_info);
        if (rc) return rc;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        total += h_info[1];
        if (h_info[0] <= h_info[1] || h_info[1] <= 0) break;
      }
      if (n_resubmitted) *n_resubmitted = total;
    }
    memcpy(n_out, h_n_out, b_i);
    memcpy(status, h_status, b_i);
    if (n_cand) memcpy(n_cand, h_n_cand, b_i);
    // only the slots the kernel wrote (n_out per frame) carry data; the caller's buffers keep their fill beyond
    for (size_t f = 0; f < F; f++) {
      const size_t k = (size_t)((h_n_out[f] < 0 || h_n_out[f] > K_max) ? 0 : h_n_out[f]);  // > K_max: needs more slots, nothing written
      memcpy(xyz + f * K_max * 3, h_xyz + f * K_max * 3, sizeof(double) * 3 * k);
      memcpy(err + f * K_max, h_err + f * K_max, sizeof(double) * k);
      memcpy(corr + f * K_max * C, h_corr + f * K_max * C, sizeof(int16_t) * C * k);
    }
    return MOCAP_OK;
  }
  DevBuf& s = ctx->scratch[0];
  if (s.reserve(total)) return ctx->fail(MOCAP_E_HIP, "hipMalloc(%zu) failed", total);
  char* p = (char*)s.ptr;
  double* d_xyz = (double*)p;       p += al(b_xyz);
  double* d_err = (double*)p;       p += al(b_err);
  float* d_blobs = (float*)p;       p += al(b_blobs);
  int32_t* d_counts = (int32_t*)p;  p += al(b_counts);
  int16_t* d_corr = (int16_t*)p;    p += al(b_corr);
  int32_t* d_n_out = (int32_t*)p;   p += al(b_i);
  int32_t* d_status = (int32_t*)p;  p += al(b_i);
  int32_t* d_n_cand = (int32_t*)p;  p += al(b_i);
  int32_t* d_info = (int32_t*)p;
  HIP_TRY(ctx, hipMemcpyAsync(d_blobs, blobs, b_blobs, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_counts, counts, b_counts, hipMemcpyHostToDevice, ctx->stream));
  int rc = match_dev_locked(ctx, n_frames, M_max, d_blobs, d_counts, gate_px, K_max, G_cap, d_xyz, d_err, d_corr,
                            d_n_out, d_status, d_n_cand);
  if (rc) return rc;
  int32_t h_info[2] = {0, 0};
  if (resubmit) {
    int total = 0;
    for (;;) {  // (more than one round only when the scratch batch was smaller than the flagged set)
      rc = resubmit_dev_locked(ctx, n_frames, M_max, d_blobs, d_counts, gate_px, K_max, d_xyz, d_err, d_corr, d_n_out, d_status,
                               d_n_cand, d_info);
      if (rc) return rc;
      HIP_TRY(ctx, hipMemcpyAsync(h_info, d_info, sizeof h_info, hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      total += h_info[1];
      if (h_info[0] <= h_info[1] || h_info[1] <= 0) break;
    }
    h_info[1] = total;
  }
  HIP_TRY(ctx, hipMemcpyAsync(xyz, d_xyz, b_xyz, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(err, d_err, b_err, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(corr, d_corr, b_corr, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(n_out, d_n_out, b_i, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(status, d_status, b_i, hipMemcpyDeviceToHost, ctx->stream));
  if (n_cand) HIP_TRY(ctx, hipMemcpyAsync(n_cand, d_n_cand, b_i, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (n_resubmitted) *n_resubmitted = h_info[1];
  return MOCAP_OK;
}

extern "C" int mocap_match_triangulate(mocap_ctx* ctx, int64_t n_frames, int M_max, const float* blobs,
                                       const int32_t* counts, double gate_px, int K_max, int64_t G_cap,
                                       double* xyz, double* err, int16_t* corr, int32_t* n_out,
                                       int32_t* status, int32_t* n_cand) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return match_host_locked(ctx, n_frames, M_max, blobs, counts, gate_px, K_max, G_cap, xyz, err, corr, n_out, status, n_cand,
                           false, nullptr);
}

// ------------------------------------------------------------------ object locator
static int locate_dev_locked(mocap_ctx* ctx, int64_t n_frames, int K_max, const double* d_xyz, const double* d_err,
                             const int32_t* d_n_pts, int O_max, double* d_pos, double* d_heading, double* d_oerr,
                             int32_t* d_drone, int32_t* d_lead, int32_t* d_n_obj) {
  if (n_frames < 0 || K_max < 1 || O_max < 1) return ctx->fail(MOCAP_E_ARG, "mocap_locate_objects: bad size argument");
  if (K_max > 256) return ctx->fail(MOCAP_E_LIMIT, "mocap_locate_objects: K_max=%d exceeds 256", K_max);
  if (n_frames == 0) return MOCAP_OK;
  if (!d_xyz || !d_err || !d_n_pts || !d_pos || !d_heading || !d_oerr || !d_drone || !d_n_obj)
    return ctx->fail(MOCAP_E_ARG, "mocap_locate_objects: null buffer");
  LocateArgs a;
  a.n_frames = n_frames;
  a.K_max = K_max;
  a.O_max = O_max;
  a.xyz = d_xyz;
  a.err = d_err;
  a.n_pts = d_n_pts;
  a.obj_pos = d_pos;
  a.obj_heading = d_heading;
  a.obj_err = d_oerr;
  a.obj_drone = d_drone;
  a.obj_lead = d_lead;
  a.n_obj = d_n_obj;
  // small batches: one wave per frame (latency); big ones: one lane per frame (throughput).  Same results (tested);
  // MOCAP_LOCATE_KERNEL=lane|wave forces one.
  bool wave = n_frames < 16384;
  if (const char* k = getenv("MOCAP_LOCATE_KERNEL")) wave = k[0] == 'w';
  if (wave) {
    TrackExportArgs e;
    memset(&e, 0, sizeof e);
    HIP_TRY(ctx, launch_track_export(a, e, ctx->stream));
  } else {
    HIP_TRY(ctx, launch_locate_objects(a, ctx->stream));
  }
  return MOCAP_OK;
}

extern "C" int mocap_track_record_bytes(int C) { return C < 1 ? 0 : (32 + 2 * C + 7) / 8 * 8; }

extern "C" int mocap_compact_tracks_dev(mocap_ctx* ctx, int64_t n_frames, int K_max, const int32_t* d_n_out,
                                        const double* d_xyz, const double* d_err, const int16_t* d_corr,
                                        int64_t* d_offsets, void* d_records, int64_t capacity, int64_t* d_total) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!ctx->C) return ctx->fail(MOCAP_E_NOCAMS, "mocap_set_cameras has not been called");
  if (n_frames < 0 || K_max < 1 || capacity < 0) return ctx->fail(MOCAP_E_ARG, "mocap_compact_tracks: bad size argument");
  if (n_frames == 0) return MOCAP_OK;
  if (!d_n_out || !d_xyz || !d_err || !d_corr || !d_offsets || (!d_records && capacity > 0))
    return ctx->fail(MOCAP_E_ARG, "mocap_compact_tracks: null buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t n_blocks = (size_t)((n_frames + 1023) / 1024);
  if (ctx->compact_ws.reserve(sizeof(int64_t) * n_blocks)) return ctx->fail(MOCAP_E_HIP, "hipMalloc(scan workspace) failed");
  CompactArgs a;
  a.n_frames = n_frames;
  a.K_max = K_max;
  a.C = ctx->C;
  a.stride = mocap_track_record_bytes(ctx->C);
  a.n_out = d_n_out;
  a.xyz = d_xyz;
  a.err = d_err;
  a.corr = d_corr;
  a.offsets = d_offsets;
  a.block_sums = (int64_t*)ctx->compact_ws.ptr;
  a.records = (unsigned char*)d_records;
  a.capacity = capacity;
  a.total = d_total;
  HIP_TRY(ctx, launch_compact_tracks(a, ctx->stream));
  return ctx->mark_enqueued();
}

extern "C" int mocap_locate_objects_dev(mocap_ctx* ctx, int64_t n_frames, int K_max, const double* d_xyz,
                                        const double* d_err, const int32_t* d_n_pts, int O_max, double* d_pos,
                                        double* d_heading, double* d_oerr, int32_t* d_drone, int32_t* d_lead,
                                        int32_t* d_n_obj) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int rc = locate_dev_locked(ctx, n_frames, K_max, d_xyz, d_err, d_n_pts, O_max, d_pos, d_heading, d_oerr, d_drone,
                                   d_lead, d_n_obj);
  return rc ? rc : ctx->mark_enqueued();
}

extern "C" int mocap_locate_objects(mocap_ctx* ctx, int64_t n_frames, int K_max, const double* xyz, const double* err,
                                    const int32_t* n_pts, int O_max, double* pos, double* heading, double* oerr,
                                    int32_t* drone, int32_t* lead, int32_t* n_obj) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (n_frames < 0 || K_max < 1 || O_max < 1) return ctx->fail(MOCAP_E_ARG, "mocap_locate_objects: bad size argument");
  if (n_frames == 0) return MOCAP_OK;
  if (!xyz || !err || !n_pts || !pos || !heading || !oerr || !drone || !n_obj)
    return ctx->fail(MOCAP_E_ARG, "mocap_locate_objects: null buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t F = (size_t)n_frames;
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  const size_t b_xyz = al(sizeof(double) * F * K_max * 3), b_err = al(sizeof(double) * F * K_max),
               b_n = al(sizeof(int32_t) * F), b_pos = al(sizeof(double) * F * O_max * 3),
               b_o = al(sizeof(double) * F * O_max), b_i = al(sizeof(int32_t) * F * O_max);
  DevBuf& s = ctx->scratch[0];
  if (s.reserve(b_xyz + b_err + 2 * b_n + b_pos + 2 * b_o + 2 * b_i)) return ctx->fail(MOCAP_E_HIP, "hipMalloc failed");
  char* p = (char*)s.ptr;
  double* d_xyz = (double*)p;        p += b_xyz;
  double* d_err = (double*)p;        p += b_err;
  double* d_pos = (double*)p;        p += b_pos;
  double* d_head = (double*)p;       p += b_o;
  double* d_oerr = (double*)p;       p += b_o;
  int32_t* d_n = (int32_t*)p;        p += b_n;
  int32_t* d_nobj = (int32_t*)p;     p += b_n;
  int32_t* d_drone = (int32_t*)p;    p += b_i;
  int32_t* d_lead = (int32_t*)p;
  HIP_TRY(ctx, hipMemcpyAsync(d_xyz, xyz, sizeof(double) * F * K_max * 3, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_err, err, sizeof(double) * F * K_max, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_n, n_pts, sizeof(int32_t) * F, hipMemcpyHostToDevice, ctx->stream));
  int rc = locate_dev_locked(ctx, n_frames, K_max, d_xyz, d_err, d_n, O_max, d_pos, d_head, d_oerr, d_drone, d_lead, d_nobj);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(pos, d_pos, sizeof(double) * F * O_max * 3, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(heading, d_head, sizeof(double) * F * O_max, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(oerr, d_oerr, sizeof(double) * F * O_max, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(drone, d_drone, sizeof(int32_t) * F * O_max, hipMemcpyDeviceToHost, ctx->stream));
  if (lead) HIP_TRY(ctx, hipMemcpyAsync(lead, d_lead, sizeof(int32_t) * F * O_max, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(n_obj, d_nobj, sizeof(int32_t) * F, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MOCAP_OK;
}

// ------------------------------------------------------------------ C-level re-submit (uncapped enumeration)
// mocap_match_triangulate_auto gives every caller of the C ABI what mocap_core/capi.py used to do in Python: frames whose
// status is non-zero are re-submitted, on the GPU, with the largest caps the core has -- under ONE acquisition of the
// context lock, with the hit-list cap of the second pass an argument of its launch (round 4 flipped ctx->hit_cap between
// two locked calls: a concurrent mocap_set_frame_limits was overwritten, a concurrent frame call ran under the foreign cap).
extern "C" int mocap_match_triangulate_auto(mocap_ctx* ctx, int64_t n_frames, int M_max, const float* blobs,
                                            const int32_t* counts, double gate_px, int K_max, int64_t G_cap,
                                            double* xyz, double* err, int16_t* corr, int32_t* n_out, int32_t* status,
                                            int32_t* n_cand, int32_t* n_resubmitted) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return match_host_locked(ctx, n_frames, M_max, blobs, counts, gate_px, K_max, G_cap, xyz, err, corr, n_out, status, n_cand,
                           true, n_resubmitted);
}

// ------------------------------------------------------------------ double-precision centroids at the boundary
// The reference measures on whatever its image_points lists hold (helpers.py:367-373): int64 for _find_dot's int()
// centroids, float64 for anything else.  The kernels carry blob coordinates as float32 (half the LDS / HBM bytes of the
// one array every phase reads).  This entry takes doubles: coordinates float32 can represent -- every integer pixel below
// 2^24, every float32-valued sub-pixel centroid -- go through unchanged, i.e. EXACTLY as the reference would see them;
// anything else is rounded to the nearest float32 (|dx| <= 2^-24 |x|: 2e-5 px at 320 px, far inside north_star's 1e-5
// relative on the points) and the frame is FLAGGED (MOCAP_ST_ROUNDED, informational: the outputs are valid) instead of
// being refused or silently altered.  NaN / inf coordinates are an argument error.
extern "C" int mocap_match_triangulate_f64(mocap_ctx* ctx, int64_t n_frames, int M_max, const double* blobs,
                                           const int32_t* counts, double gate_px, int K_max, int64_t G_cap, double* xyz,
                                           double* err, int16_t* corr, int32_t* n_out, int32_t* status, int32_t* n_cand,
                                           int32_t* n_resubmitted) {
  if (!ctx) return MOCAP_E_ARG;
  int C;
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!ctx->C) return ctx->fail(MOCAP_E_NOCAMS, "mocap_set_cameras has not been called");
    if (n_frames < 0 || M_max < 1 || K_max < 1) return ctx->fail(MOCAP_E_ARG, "mocap_match_triangulate_f64: bad size argument");
    if (n_frames > 0 && (!blobs || !counts || !status)) return ctx->fail(MOCAP_E_ARG, "mocap_match_triangulate_f64: null buffer");
    C = ctx->C;
  }
  const size_t per = (size_t)C * M_max * 2;
  std::vector<float> b32;
  std::vector<uint8_t> rounded;
  try {
    b32.resize((size_t)n_frames * per);
    rounded.assign((size_t)n_frames, 0);
  } catch (const std::exception& ex) {
    return ctx->fail(MOCAP_E_HIP, "mocap_match_triangulate_f64: %s", ex.what());
  }
  for (int64_t f = 0; f < n_frames; f++)
    for (int c = 0; c < C; c++) {
      int n = counts[(size_t)f * C + c];
      n = n < 0 ? 0 : (n > M_max ? M_max : n);
      for (int k = 0; k < 2 * n; k++) {
        const size_t o = (size_t)f * per + (size_t)c * M_max * 2 + k;
        const double v = blobs[o];
        if (!(v - v == 0.0)) return ctx->fail(MOCAP_E_ARG, "mocap_match_triangulate_f64: frame %lld camera %d: coordinate is NaN or infinite", (long long)f, c);
        const float r = (float)v;
        b32[o] = r;
        if ((double)r != v) rounded[f] = 1;
      }
    }
  const int rc = mocap_match_triangulate_auto(ctx, n_frames, M_max, b32.data(), counts, gate_px, K_max, G_cap, xyz, err, corr, n_out,
                                              status, n_cand, n_resubmitted);
  if (rc) return rc;
  for (int64_t f = 0; f < n_frames; f++)
    if (rounded[f]) status[f] |= MOCAP_ST_ROUNDED;
  return MOCAP_OK;
}

// ------------------------------------------------------------------ the live loop in one call
// helpers.py:94-133 per frame: find_point_correspondance_and_object_points -> world coordinates -> locate_objects ->
// the `object-points` payload.  One enqueue: [blob stage ->] frame kernel (world epilogue fused in its store) -> one wave
// per frame that runs locate_objects and exports everything into pinned host memory; the host waits for ONE event.
namespace {

struct TrackOut {
  double* xyz; double* err; int16_t* corr; int32_t* n_pts; int32_t* status;
  int O_max; double* pos; double* heading; double* oerr; int32_t* drone; int32_t* n_obj;
  float* blobs; int32_t* counts; int32_t* blob_status;
};

int wait_live_event(mocap_ctx* ctx) {
  HIP_TRY(ctx, hipEventRecord(ctx->live_event, ctx->stream));
  for (long spins = 0;; spins++) {
    const hipError_t e = hipEventQuery(ctx->live_event);
    if (e == hipSuccess) break;
    if (e != hipErrorNotReady) return ctx->hip_fail(e, "hipEventQuery");
    if (spins > 2000000) {
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      break;
    }
  }
  return MOCAP_OK;
}

// images != null: raw frames [F][C][rows][cols][3] (host); else blobs / counts (host) are the input
int track_locked(mocap_ctx* ctx, int64_t n_frames, const uint8_t* images, int M_max, const float* blobs,
                 const int32_t* counts, double gate_px, int K_max, int64_t G_cap, const TrackOut& o) {
  if (!ctx->C) return ctx->fail(MOCAP_E_NOCAMS, "mocap_set_cameras has not been called");
  if (n_frames < 0 || M_max < 1 || K_max < 1 || G_cap < 1 || o.O_max < 0)
    return ctx->fail(MOCAP_E_ARG, "mocap_track_frame: bad size argument");
  if (n_frames == 0) return MOCAP_OK;
  if ((!images && (!blobs || !counts)) || !o.xyz || !o.err || !o.n_pts || !o.status)
    return ctx->fail(MOCAP_E_ARG, "mocap_track_frame: null buffer");
  if (o.O_max > 0 && (!o.pos || !o.heading || !o.oerr || !o.drone || !o.n_obj))
    return ctx->fail(MOCAP_E_ARG, "mocap_track_frame: null object buffer");
  if (o.O_max > 0 && K_max > 256) return ctx->fail(MOCAP_E_LIMIT, "mocap_track_frame: K_max=%d exceeds 256 with the object search on", K_max);
  if (images && (!ctx->img_C || ctx->img_C != ctx->C))
    return ctx->fail(MOCAP_E_NOCAMS, "mocap_set_image_params has not been called for this camera set");
  if (images && (!o.blobs || !o.counts || !o.blob_status)) return ctx->fail(MOCAP_E_ARG, "mocap_track_frame_images: null blob buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int C = ctx->C, O = o.O_max > 0 ? o.O_max : 1;
  const size_t F = (size_t)n_frames;
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  const size_t b_blobs = al(sizeof(float) * F * C * M_max * 2), b_counts = al(sizeof(int32_t) * F * C),
               b_xyz = al(sizeof(double) * F * K_max * 3), b_err = al(sizeof(double) * F * K_max),
               b_corr = al(sizeof(int16_t) * F * K_max * C), b_i = al(sizeof(int32_t) * F),
               b_pos = al(sizeof(double) * F * O * 3), b_o = al(sizeof(double) * F * O), b_oi = al(sizeof(int32_t) * F * O);
  // caller-visible side (pinned host memory): inputs | frame outputs | objects
  const size_t host_total = b_blobs + 2 * b_counts + b_xyz + b_err + b_corr + 3 * b_i + b_pos + 2 * b_o + b_oi + b_i;
  if (host_total > ctx->live_pin_cap) {
    if (ctx->live_pin) (void)hipHostFree(ctx->live_pin);
    ctx->live_pin = nullptr;
    ctx->live_pin_cap = 0;
    const size_t want = host_total < (size_t)256 * 1024 ? (size_t)256 * 1024 : host_total + host_total / 4;
    HIP_TRY(ctx, hipHostMalloc(&ctx->live_pin, want, hipHostMallocDefault));
    ctx->live_pin_cap = want;
  }
  if (!ctx->live_event) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->live_event, hipEventDisableTiming));
  char* p = (char*)ctx->live_pin;
  float* h_blobs = (float*)p;          p += b_blobs;
  int32_t* h_counts = (int32_t*)p;     p += b_counts;
  int32_t* h_bstat = (int32_t*)p;      p += b_counts;
  double* h_xyz = (double*)p;          p += b_xyz;
  double* h_err = (double*)p;          p += b_err;
  int16_t* h_corr = (int16_t*)p;       p += b_corr;
  int32_t* h_n = (int32_t*)p;          p += b_i;
  int32_t* h_status = (int32_t*)p;     p += b_i;
  int32_t* h_ncand = (int32_t*)p;      p += b_i;
  double* h_pos = (double*)p;          p += b_pos;
  double* h_head = (double*)p;         p += b_o;
  double* h_oerr = (double*)p;         p += b_o;
  int32_t* h_drone = (int32_t*)p;      p += b_oi;
  int32_t* h_nobj = (int32_t*)p;
  // device side: [raw images | blobs | counts | blob status |] frame outputs
  const size_t b_raw = images ? al(F * C * (size_t)ctx->img_rows * ctx->img_cols * 3) : 0;
  const size_t dev_total = b_raw + (images ? b_blobs + 2 * b_counts : 0) + b_xyz + b_err + b_corr + 3 * b_i;
  DevBuf& s = ctx->scratch[0];
  if (s.reserve(dev_total)) return ctx->fail(MOCAP_E_HIP, "hipMalloc(%zu) failed", dev_total);
  char* d = (char*)s.ptr;
  uint8_t* d_raw = nullptr;
  float* d_blobs = nullptr;
  int32_t *d_counts = nullptr, *d_bstat = nullptr;
  if (images) {
    d_raw = (uint8_t*)d;               d += b_raw;
    d_blobs = (float*)d;               d += b_blobs;
    d_counts = (int32_t*)d;            d += b_counts;
    d_bstat = (int32_t*)d;             d += b_counts;
  }
  double* d_xyz = (double*)d;          d += b_xyz;
  double* d_err = (double*)d;          d += b_err;
  int16_t* d_corr = (int16_t*)d;       d += b_corr;
  int32_t* d_n = (int32_t*)d;          d += b_i;
  int32_t* d_status = (int32_t*)d;     d += b_i;
  int32_t* d_ncand = (int32_t*)d;

  if (images) {
    HIP_TRY(ctx, hipMemcpyAsync(d_raw, images, F * C * (size_t)ctx->img_rows * ctx->img_cols * 3, hipMemcpyHostToDevice, ctx->stream));
    const int rc = mocap_blob_stage_locked(ctx, n_frames, d_raw, M_max, d_blobs, d_counts, d_bstat);
    if (rc) return rc;
  } else {
    memcpy(h_blobs, blobs, sizeof(float) * F * C * M_max * 2);
    memcpy(h_counts, counts, sizeof(int32_t) * F * C);
  }
  LocateArgs la;
  la.n_frames = n_frames;
  la.K_max = K_max;
  la.O_max = O;
  la.xyz = d_xyz;
  la.err = d_err;
  la.n_pts = d_n;
  la.obj_pos = h_pos;
  la.obj_heading = h_head;
  la.obj_err = h_oerr;
  la.obj_drone = h_drone;
  la.obj_lead = nullptr;
  la.n_obj = o.O_max > 0 ? h_nobj : nullptr;
  TrackExportArgs ea;
  memset(&ea, 0, sizeof ea);
  ea.C = C;
  ea.M = M_max;
  ea.corr = d_corr;
  ea.status = d_status;
  ea.n_cand = d_ncand;
  ea.out_xyz = h_xyz;
  ea.out_err = h_err;
  ea.out_corr = o.corr ? h_corr : nullptr;
  ea.out_n_pts = h_n;
  ea.out_status = h_status;
  ea.out_n_cand = h_ncand;
  if (images) {
    ea.blobs = d_blobs;
    ea.counts = d_counts;
    ea.blob_status = d_bstat;
    ea.out_blobs = h_blobs;
    ea.out_counts = h_counts;
    ea.out_blob_status = h_bstat;
  }
  // the frame kernel reads pinned host memory in place (zero-copy) -- except when the shape goes to the wide variant, which
  // re-reads the blobs for every root batch and candidate view: those are staged into device memory once
  const float* in_blobs = images ? d_blobs : h_blobs;
  const int32_t* in_counts = images ? d_counts : h_counts;
  if (!images && plan_frame(ctx, M_max, K_max, 0).wide) {
    DevBuf& st = ctx->live_stage;
    if (st.reserve(b_blobs + b_counts)) return ctx->fail(MOCAP_E_HIP, "hipMalloc(%zu) failed", b_blobs + b_counts);
    HIP_TRY(ctx, hipMemcpyAsync(st.ptr, h_blobs, sizeof(float) * F * C * M_max * 2, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync((char*)st.ptr + b_blobs, h_counts, sizeof(int32_t) * F * C, hipMemcpyHostToDevice, ctx->stream));
    in_blobs = (const float*)st.ptr;
    in_counts = (const int32_t*)((char*)st.ptr + b_blobs);
  }
  {
    int rc = match_dev_locked(ctx, n_frames, M_max, in_blobs, in_counts, gate_px, K_max, G_cap, d_xyz, d_err, d_corr, d_n,
                              d_status, d_ncand);
    if (rc) return rc;
    // frames that hit a cap (candidates, hit lists, roots): re-run, those frames only, with the largest caps the core has --
    // the reference has none (helpers.py:394-400) -- before the export; queued behind the first pass, no host round trip
    rc = resubmit_dev_locked(ctx, n_frames, M_max, in_blobs, in_counts, gate_px, K_max, d_xyz, d_err, d_corr, d_n, d_status,
                             d_ncand, nullptr);
    if (rc) return rc;
    HIP_TRY(ctx, launch_track_export(la, ea, ctx->stream));
    rc = wait_live_event(ctx);
    if (rc) return rc;
  }
  memcpy(o.n_pts, h_n, sizeof(int32_t) * F);
  memcpy(o.status, h_status, sizeof(int32_t) * F);
  for (size_t f = 0; f < F; f++) {
    const size_t k = (size_t)((h_n[f] < 0 || h_n[f] > K_max) ? 0 : h_n[f]);  // > K_max: needs more slots, nothing written
    memcpy(o.xyz + f * K_max * 3, h_xyz + f * K_max * 3, sizeof(double) * 3 * k);
    memcpy(o.err + f * K_max, h_err + f * K_max, sizeof(double) * k);
    if (o.corr) memcpy(o.corr + f * K_max * C, h_corr + f * K_max * C, sizeof(int16_t) * C * k);
    if (o.O_max > 0) {
      o.n_obj[f] = h_nobj[f];
      const size_t no = (size_t)(h_nobj[f] < 0 ? 0 : (h_nobj[f] > o.O_max ? o.O_max : h_nobj[f]));
      memcpy(o.pos + f * O * 3, h_pos + f * O * 3, sizeof(double) * 3 * no);
      memcpy(o.heading + f * O, h_head + f * O, sizeof(double) * no);
      memcpy(o.oerr + f * O, h_oerr + f * O, sizeof(double) * no);
      memcpy(o.drone + f * O, h_drone + f * O, sizeof(int32_t) * no);
    }
  }
  if (images) {
    memcpy(o.counts, h_counts, sizeof(int32_t) * F * C);
    memcpy(o.blob_status, h_bstat, sizeof(int32_t) * F * C);
    for (size_t i = 0; i < F * C; i++) {
      const size_t k = (size_t)(h_counts[i] < 0 ? 0 : (h_counts[i] > M_max ? M_max : h_counts[i]));
      memcpy(o.blobs + i * M_max * 2, h_blobs + i * M_max * 2, sizeof(float) * 2 * k);
    }
  }
  return MOCAP_OK;
}

}  // namespace

extern "C" int mocap_track_frame(mocap_ctx* ctx, int64_t n_frames, int M_max, const float* blobs, const int32_t* counts,
                                 double gate_px, int K_max, int64_t G_cap, double* xyz, double* err, int16_t* corr,
                                 int32_t* n_pts, int32_t* status, int O_max, double* pos, double* heading, double* oerr,
                                 int32_t* drone, int32_t* n_obj) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const TrackOut o{xyz, err, corr, n_pts, status, O_max, pos, heading, oerr, drone, n_obj, nullptr, nullptr, nullptr};
  return track_locked(ctx, n_frames, nullptr, M_max, blobs, counts, gate_px, K_max, G_cap, o);
}

extern "C" int mocap_track_frame_images(mocap_ctx* ctx, int64_t n_frames, const uint8_t* images, int M_max, double gate_px,
                                        int K_max, int64_t G_cap, float* blobs, int32_t* counts, int32_t* blob_status,
                                        double* xyz, double* err, int16_t* corr, int32_t* n_pts, int32_t* status, int O_max,
                                        double* pos, double* heading, double* oerr, int32_t* drone, int32_t* n_obj) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!images) return ctx->fail(MOCAP_E_ARG, "mocap_track_frame_images: null image buffer");
  const TrackOut o{xyz, err, corr, n_pts, status, O_max, pos, heading, oerr, drone, n_obj, blobs, counts, blob_status};
  return track_locked(ctx, n_frames, images, M_max, nullptr, nullptr, gate_px, K_max, G_cap, o);
}

extern "C" int mocap_track_frame_dev(mocap_ctx* ctx, int64_t n_frames, int M_max, const float* d_blobs,
                                     const int32_t* d_counts, double gate_px, int K_max, int64_t G_cap, double* d_xyz,
                                     double* d_err, int16_t* d_corr, int32_t* d_n_pts, int32_t* d_status, int O_max,
                                     double* d_pos, double* d_heading, double* d_oerr, int32_t* d_drone, int32_t* d_n_obj) {
  if (!ctx) return MOCAP_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = match_dev_locked(ctx, n_frames, M_max, d_blobs, d_counts, gate_px, K_max, G_cap, d_xyz, d_err, d_corr, d_n_pts,
                            d_status, nullptr);
  if (rc) return rc;
  // frames that hit a cap are re-run on the device with the largest caps before the object search reads the points
  rc = resubmit_dev_locked(ctx, n_frames, M_max, d_blobs, d_counts, gate_px, K_max, d_xyz, d_err, d_corr, d_n_pts, d_status,
                           nullptr, nullptr);
  if (rc) return rc;
  if (O_max > 0) {
    rc = locate_dev_locked(ctx, n_frames, K_max, d_xyz, d_err, d_n_pts, O_max, d_pos, d_heading, d_oerr, d_drone, nullptr, d_n_obj);
    if (rc) return rc;
  }
  return ctx->mark_enqueued();
}
