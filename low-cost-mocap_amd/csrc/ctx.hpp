// ctx.hpp -- the opaque context behind `mocap_ctx*` (include/mocap_core.h).
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <vector>

#include "kernels.hpp"

struct DevBuf {
  void* ptr = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes);  // grow-only; 0 on success
  void release();
};

struct mocap_ctx {
  int device = 0;
  int num_cus = 256;
  int frame_threads = 0;    // workgroup size of the frame kernel, 0 = automatic (MOCAP_FRAME_THREADS=64|128|256)
  int heavy_threshold = -1; // -1 = automatic; 0 = never split heavy frames (MOCAP_HEAVY_THRESHOLD)
  int slice_size = 0;       // 0 = automatic (MOCAP_SLICE_SIZE)
  int hit_cap = 32;         // wide frames: hits kept per (root, camera) (mocap_set_frame_limits)
  int force_wide = 0;       // route every frame batch through the wide (HBM workspace) variant
  int32_t frame_gen = 0;    // generation of the last frame-path launch (tags the slices it publishes)
  int frame_q_cap[2] = {0, 0};  // W_cap the work-queue buffer was laid out for ([0] a batch's own pass, [1] the re-submit's second pass:
                            // two buffers, so that neither pass finds the other's layout and clears the queue again -- six fills per call)
  const char* last_frame_kernel = "none";  // which kernel the last frame batch went to (mocap_last_frame_kernel)
  bool frame_q_clean[2] = {false, false};  // the queue counters were left at zero by the one-launch schedule
  void frame_q_dirty() { frame_q_clean[0] = frame_q_clean[1] = false; }
  int frame_launches = 1;   // 1: one persistent launch per batch (MODE_ALL); 3: main / slice / merge launches
  int prune = 1;            // stop a candidate group's reprojection once it cannot beat its root's best (exact)
  int exhaustive = 0;       // MOCAP_OPT_EXHAUSTIVE_WALK: no branch and bound, no cut-offs (verification mode)
  int eval_bb = 1;          // branch-and-bound selection (csrc/frame_bb.hip) wherever it applies; 0: always the exhaustive walk
  int bb_pl = 16;           // ... candidates per block (at least)
  int bb_min_g = 0;         // ... frames with fewer candidates queue every block untested (swept: 0-512 equal, 2048 +13 %)
  int bb_flush = 0;         // ... queued candidates that trigger their evaluation (0 = one per lane)
  int eigcut = 1;           // ... and drop it before the null vector / the reprojection on an eigenvalue bound (EigCut)
  double eig_c0[3] = {0, 0, 0};  // ... origin for the branch-and-bound's bounds: the point closest to all optical axes
  double p3max2c = 0.0;     // ... and the constant in that frame
  double p3max2 = 0.0;      // EigCut constant of the current camera set (0: intrinsics not of the form the bound needs)
  hipStream_t own_stream = nullptr, stream = nullptr;
  // stream hand-over (mocap_set_stream): every "_dev" entry point records this event behind what it enqueued; a new
  // stream is ordered behind it with hipStreamWaitEvent -- the previous stream's handle (the caller's: it may be gone)
  // is never touched again and the host never blocks
  hipEvent_t handover_event = nullptr;
  bool dev_outstanding = false;
  int mark_enqueued();
  std::mutex mu;            // one context = one serialised caller (include/mocap_core.h)
  std::string err;          // guarded by err_mu (written by a failing call, copied out by mocap_last_error)
  std::mutex err_mu;
  uint32_t flags = 1u;      // MOCAP_OPT_F32_ROUNDING on by default
  int C = 0;
  std::vector<double> hK, hR, ht, hF;  // host copies (intrinsics are reused by bundle adjustment)
  DevBuf tables;            // Pq | RT | K4 | F | K9
  const double* d_K9 = nullptr;
  mocap::CamView cv{};
  // bundle adjustment: pinned host staging (async copies that really are async) and the event the
  // LM loop spin-waits on (hipStreamSynchronize may sleep on an interrupt: +50..400 us per wait)
  void* ba_pin = nullptr;
  size_t ba_pin_cap = 0;
  hipEvent_t ba_event = nullptr;
  void* ba_stage = nullptr;  // pinned staging of a solve's inputs (observations | valid list): no pageable copies
  size_t ba_stage_cap = 0;
  void (*ba_progress)(const double* x, int n, void* user) = nullptr;  // mocap_set_ba_progress
  void* ba_progress_user = nullptr;
  DevBuf ba_fused;          // one-launch linearisation: chunk partial tiles | chunk costs | counters | (Jaug dump)
  double ba_stamp = 0.0;    // completion stamp of the last fused launch (monotonic per context)
  void* live_pin = nullptr;  // zero-copy staging of the live (few frames per call) host entry point
  size_t live_pin_cap = 0;
  hipEvent_t live_event = nullptr;
  DevBuf world;             // 16 doubles: the to-world matrix of the fused epilogue
  bool world_on = false;
  // blob extraction (mocap_set_image_params): frame geometry, undistortion maps, mask workspace
  int img_C = 0, img_rows = 0, img_cols = 0, img_S = 0, img_ay = 0;
  DevBuf img_map, img_rot, img_mask, img_stage, img_tiles, img_lens, img_fix, img_fixidx, img_act, img_box, img_zero;
  int img_n_lt = 0;           // (lens, rotation) x tiles: entries of the fix-up index
  int blob_skip_dark = 1;     // exact early-out for tiles whose source bytes span a range <= 2 (mocap_set_blob_options)
  DevBuf compact_ws;        // block totals of the track-compaction scan
  DevBuf frame_ws;          // wide-frame workspace: [workgroup][hit lists | group columns | ...]
  DevBuf live_stage;        // mocap_track_frame: device copy of a wide frame's blobs (narrow frames are read from pinned host memory in place)
  DevBuf resub;             // device-side re-submit: counters | frame list | gathered inputs | second-pass outputs
  DevBuf resub_ctr;         // ... its two alternating counters (never re-allocated while a call is in flight)
  DevBuf resub_q;           // ... the second pass's work queues (frame_q_cap[1])
  DevBuf heavy_recs;        // ... heavy roots exported by the second pass (csrc/heavy_bb.hip)
  DevBuf heavy_ws;          // ... the search's frontier workspace
  DevBuf heavy_enum;        // ... queue + per-workgroup winners of the roots enumerated over the whole GPU (heavy_enum_kernel)
  uint32_t resub_calls = 0; // ... parity selects the counter of the current call
  DevBuf scratch[4];        // [0] host-API staging, [1..3] bundle adjustment workspace

  int fail(int code, const char* fmt, ...);
  int hip_fail(hipError_t e, const char* what);
};

// internal entry points shared between the translation units of the C ABI (context lock held by the caller)
int mocap_blob_stage_locked(mocap_ctx* ctx, int64_t n_frames, const uint8_t* d_images, int M_max, float* d_blobs,
                            int32_t* d_counts, int32_t* d_status);
