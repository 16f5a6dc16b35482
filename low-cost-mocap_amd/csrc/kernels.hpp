// kernels.hpp -- host-visible launch interface of the HIP kernels (internal to the .so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mocap_device.hpp"

namespace mocap {

// status bits, mirrored from include/mocap_core.h (MOCAP_ST_*)
constexpr int MOCAP_ST_ROOT_OVERFLOW_ = 1;
constexpr int MOCAP_ST_CAND_OVERFLOW_ = 2;
constexpr int MOCAP_ST_HIT_OVERFLOW_ = 4;
constexpr int MOCAP_ST_INTRACTABLE_ = 16;
constexpr int MOCAP_ST_FINAL_ = 32;
constexpr int MOCAP_ST_LOG2_GROUPS_SHIFT_ = 20;

// device-side work queues of the frame path (see the scheduling note in frame_kernel.hip)
constexpr int MODE_MAIN = 0, MODE_SLICE = 1, MODE_MERGE = 2, MODE_ALL = 3;
enum { QC_NEXT_FRAME = 0, QC_N_HEAVY = 1, QC_N_SLICES = 2, QC_NEXT_SLICE = 3, QC_NEXT_MERGE = 4, QC_FRAMES_DONE = 5, QC_EXITED = 6, QC_COUNT = 8 };
struct FrameQueues {
  int32_t* counters;     // [QC_COUNT], zeroed before every batch
  int32_t* heavy;        // [H_cap][4]: frame (or -1), first slice id, slice count, slices finished (MODE_ALL)
  int32_t* slice_heavy;  // [W_cap]: heavy-list index of each slice id (-1 = unused)
  int32_t* slice_gen;    // [W_cap] MODE_ALL: slice id s is published for this launch when slice_gen[s] == gen
  int32_t gen;           // launch generation (no per-launch memset of the queue: the counters clean themselves)
  double* part_e;        // [W_cap][K_max]     per-slice, per-root partial winners
  uint32_t* part_g;      // [W_cap][K_max]
  double* part_x;        // [W_cap][K_max][3]
  int H_cap, W_cap;
  uint32_t heavy_threshold;  // candidate count above which a frame is deferred (0 = never)
  uint32_t slice_size;       // target candidates per slice
  int frame_chunk;           // frames a workgroup takes per pull of the frame queue (>= 1)
};

struct FrameArgs {
  CamView cv;
  FrameQueues q;
  int64_t n_frames;
  int M;      // blob slots per camera
  int K_max;  // root / output capacity per frame
  double gate_px;
  int64_t G_cap;
  const float* blobs;     // [F][C][M][2]
  const int32_t* counts;  // [F][C]
  double* xyz;            // [F][K_max][3]
  double* err;            // [F][K_max]
  int16_t* corr;          // [F][K_max][C]
  int32_t* n_out;         // [F]
  int32_t* status;        // [F]
  int32_t* n_cand;        // [F] or null
  const double* world;    // null, or the 4x4 to-world matrix: xyz leaves the kernel in world coordinates
  // wide frames (state does not fit LDS): per-workgroup workspace in HBM, hit-list cap per (root, camera)
  unsigned char* ws;
  size_t ws_stride;
  int H;
  int wide;   // 1: wide variant; 2: ... with the chain over the cameras strictly sequential (A/B, tests: MOCAP_WIDE_SPEC=0)
  int prune;  // cut the reprojection of a group short once it cannot beat the best of its root (exact, see evaluate())
  int eval_bb;  // (host only) the batch goes to frame_bb.hip
  int bb_pl;    // ... candidates per block (at least)
  int bb_flush; // ... queued candidates that trigger their evaluation
  int bb_min_g; // ... frames with fewer candidates are walked exhaustively
  double bb_c0[3]; // branch and bound: origin of the frame its bounds are taken in (a point inside the working volume)
  double p3max2c;  // ... (1 + 1e-5) * max |P[2] M|^2 in that frame
  double p3max2;  // EigCut: (1 + 1e-5) * max |P[2]|^2 over the cameras, 0 = eigenvalue cut-off off (see mocap_device.hpp)
  const int32_t* n_frames_dev;  // null, or a device-side frame count: the batch is min(*n_frames_dev, n_frames) frames long
                                // (the re-submit pass of mocap_match_triangulate_dev_auto: its length is only known on the device)
  // heavy roots (wide variant, re-submit pass): a root whose Cartesian product exceeds G_cap does not flag its frame -- its
  // candidate 0 (the closest hit in every camera) is evaluated like any other group and the root's hit lists are exported
  // to the heavy-root search (csrc/heavy_bb.hip), which runs behind this launch and replaces the point if it finds a better group
  int heavy_bb;
  int heavy_cap;               // records the export buffer holds
  int32_t* heavy_count;        // [1] records exported so far (may exceed heavy_cap: the surplus frames are flagged)
  unsigned char* heavy_recs;   // [heavy_cap][heavy_stride]
  size_t heavy_stride;         // heavy_rec_bytes(C, H)
};

// record of one heavy root: header, hit counts [C] (uint16; 1 at the root's own camera, 0 before it), hit lists [C][Hs]
// (blob indices, ascending distance)
struct HeavyRecHdr {
  int32_t frame, root, outslot, rc, rb, omax_bits, views, Hs;
};
__host__ __device__ inline size_t heavy_rec_counts_off() { return sizeof(HeavyRecHdr); }
__host__ __device__ inline size_t heavy_rec_hits_off(int C) { return (sizeof(HeavyRecHdr) + 2 * (size_t)C + 15) / 16 * 16; }
__host__ __device__ inline size_t heavy_rec_bytes(int C, int Hs) { return (heavy_rec_hits_off(C) + (size_t)C * Hs + 15) / 16 * 16; }

// the heavy-root search (csrc/heavy_bb.hip): exact branch and bound over the digits of ONE root's candidate space, level by
// level, for roots whose product no enumeration reaches (two markers behind each other as seen from the root's camera:
// two hits in nearly every camera, 2^60 groups)
struct HeavyArgs {
  CamView cv;
  int M, K_big;
  double bb_c0[3], p3max2c, p3max2;
  const float* blobs;          // [frames][C][M][2] the batch the records' frame indices refer to
  const int32_t* heavy_count;
  const unsigned char* recs;
  int cap;
  size_t stride;
  double* xyz;                 // the batch's outputs ([frames][K_big] slots): a better group overwrites the root's slot
  double* err;
  int16_t* corr;
  int32_t* n_out;              // a root whose frontier outgrows the workspace flags its frame (status |= candidate overflow, n_out = 0)
  int32_t* status;
  const double* world;
  unsigned char* ws;           // [grid][ws_stride] frontier workspace
  size_t ws_stride;
  int ncap;                    // nodes per frontier buffer
  int64_t enum_cap;            // a root the search gives up on is enumerated in place when its product is at most this
  int debug;                   // MOCAP_HEAVY_DEBUG: one printf per root
  // roots the search gives up on whose product is above enum_cap and at most 2^24: queued for heavy_enum_kernel, which
  // enumerates them over the whole GPU (the contract of the re-submit pass: exact up to 2^24 groups per root)
  int32_t* enum_count;         // [1] queued roots (zeroed by the gather kernel)
  int32_t* enum_list;          // [enum_max] their record numbers
  int32_t* enum_slice;         // [enum_max] next slice of the root's group range
  int32_t* enum_done;          // [enum_max] workgroups that have reported
  unsigned long long* enum_bound;  // [enum_max] smallest error seen so far (bit pattern): every workgroup's cut-off
  unsigned char* enum_part;    // [enum_max][enum_grid][kHeavyEnumPartBytes] per-workgroup winners
  int enum_max, enum_grid;
};
constexpr int kHeavyEnumPartBytes = 40;  // {error bits u64, group u32, pad u32, X[3]}
constexpr int kHeavyEnumMax = 256;  // roots per re-submit call (beyond: the frame keeps its candidate-overflow flag)
size_t heavy_enum_ws_bytes(int enum_max, int grid);
hipError_t launch_heavy_enum(const HeavyArgs& a, hipStream_t stream);
size_t heavy_bb_ws_bytes(int ncap);
hipError_t launch_heavy_bb(const HeavyArgs& a, int grid, hipStream_t stream);

constexpr int kWideThreads = 1024;  // workgroup size of the wide-frame variant (one workgroup per CU)
// table = identical intrinsics (CamView::uniformK): per-blob DLT contributions tabulated in LDS (narrow frames only)
size_t frame_lds_bytes(int C, int M, int R, int T, int H, bool wide, bool table);
size_t frame_ws_bytes(int C, int M, int R, int T, int H, bool wide, bool table);
hipError_t launch_frame_kernel(const FrameArgs& a, int mode, int threads, int grid, hipStream_t stream);
// csrc/frame_bb.hip: identical plain intrinsics, C <= 16, M <= 64, K_max <= 255 -- its own kernel (256 lanes per frame)
// and LDS layout, exact branch-and-bound selection; uses q.counters / q.frame_chunk of FrameArgs::q only
bool frame_bb_fits(int C, int M, int R);
size_t frame_bb_lds_bytes(int C, int M, int R);
hipError_t launch_frame_bb(const FrameArgs& a, int grid, hipStream_t stream);
int frame_bb_wg_per_cu_cap(int C, int M, int R);  // workgroups per CU the kernel's register budget allows (its waves per SIMD)
size_t frame_bb_ws_bytes(int C);  // bytes of global workspace per workgroup (the probe scheme's parked candidates), 0 = none

// object (drone) locator over the frame path's output (reference helpers.py:424-480), csrc/post_kernels.hip
struct LocateArgs {
  int64_t n_frames;
  int K_max, O_max;
  const double* xyz;     // [F][K_max][3] (world coordinates)
  const double* err;     // [F][K_max]
  const int32_t* n_pts;  // [F]
  double* obj_pos;       // [F][O_max][3]
  double* obj_heading;   // [F][O_max]
  double* obj_err;       // [F][O_max]
  int32_t* obj_drone;    // [F][O_max]
  int32_t* obj_lead;     // [F][O_max] index of the point the pattern was found from, or null
  int32_t* n_obj;        // [F]
};
hipError_t launch_locate_objects(const LocateArgs& a, hipStream_t stream);
// live path (mocap_track_frame): one wave per frame runs the same scan (LocateArgs; n_obj == null = no object search) and
// copies the valid slots of the frame path's device-side outputs to caller-visible (e.g. pinned host) buffers
struct TrackExportArgs {
  int C;
  const int16_t* corr;      // [F][K_max][C] device
  const int32_t* status;    // [F]
  const int32_t* n_cand;    // [F] or null
  double* out_xyz;          // null = no export; else [F][K_max][3]
  double* out_err;          // [F][K_max]
  int16_t* out_corr;        // [F][K_max][C] or null
  int32_t* out_n_pts;       // [F]
  int32_t* out_status;      // [F]
  int32_t* out_n_cand;      // [F] or null
  // images -> payload (mocap_track_frame_images): the blob stage's device-side outputs travel with the rest
  int M;                    // blob slots per camera
  const float* blobs;       // null, or [F][C][M][2] device
  const int32_t* counts;    // [F][C]
  const int32_t* blob_status;  // [F][C]
  float* out_blobs;
  int32_t* out_counts;
  int32_t* out_blob_status;
};
hipError_t launch_track_export(const LocateArgs& a, const TrackExportArgs& e, hipStream_t stream);

// compaction of a frame batch's valid points into fixed-stride records (the payload of the multi-GPU exchange)
struct CompactArgs {
  int64_t n_frames;
  int K_max, C, stride;     // stride = mocap_track_record_bytes(C)
  const int32_t* n_out;     // [F]
  const double* xyz;        // [F][K_max][3]
  const double* err;        // [F][K_max]
  const int16_t* corr;      // [F][K_max][C]
  int64_t* offsets;         // [F + 1] out: first record of every frame, total at [F]
  int64_t* block_sums;      // [ceil(F / 1024)] scratch
  unsigned char* records;   // [capacity][stride] out
  int64_t capacity;
  int64_t* total;           // null, or [1] out (e.g. pinned host memory)
};
hipError_t launch_compact_tracks(const CompactArgs& a, hipStream_t stream);

// device-side re-submit of the frames that hit a cap (mocap_match_triangulate_dev_auto, csrc/post_kernels.hip): the flagged
// frames' inputs are gathered into a scratch batch whose length lives on the device, the frame kernel runs on it with the
// largest caps, and the results that fit the caller's K_max slots are scattered back
struct ResubmitArgs {
  int64_t n_frames;          // frames of the caller's batch
  int64_t cap;               // frames the scratch batch holds
  int C, M, K_max, K_big;    // K_big: root / output capacity of the second pass
  const int32_t* status;     // [F] first pass
  const float* blobs;        // [F][C][M][2]
  const int32_t* counts;     // [F][C]
  int32_t* list;             // [cap] flagged frames (unordered)
  int32_t* count;            // [1] number of flagged frames (may exceed cap); zero on entry
  int32_t* count_next;       // [1] the NEXT call's counter: zeroed by the gather kernel (no memset between calls)
  float* b2;                 // [cap][C][M][2]
  int32_t* c2;               // [cap][C]
  const double* x2;          // [cap][K_big][3] second pass
  const double* e2;          // [cap][K_big]
  const int16_t* r2;         // [cap][K_big][C]
  const int32_t* n2;         // [cap]
  const int32_t* s2;         // [cap]
  const int32_t* g2;         // [cap]
  double* xyz;               // the caller's outputs ([F][K_max] slots)
  double* err;
  int16_t* corr;
  int32_t* n_out;
  int32_t* status_out;
  int32_t* n_cand;           // or null
  int32_t* info;             // null, or [2] out: {frames flagged, frames re-run}
  int32_t* heavy_count;      // null, or the heavy-root export counter of the second pass: zeroed by the gather kernel
  int32_t* enum_count;       // null, or the counter of roots queued for heavy_enum_kernel: zeroed by the gather kernel
};
hipError_t launch_resubmit_gather(const ResubmitArgs& a, hipStream_t stream);
hipError_t launch_resubmit_scatter(const ResubmitArgs& a, hipStream_t stream);

// blob extraction, the step before the frame path (reference helpers.py:68-82, 143-163), csrc/blob_kernels.hip
constexpr int BLOB_ST_POINT_OVERFLOW_ = 1;  // more centroids than M_max: the first M_max are kept
constexpr int BLOB_ST_CAP_OVERFLOW_ = 2;    // more border pairs / contours than the workgroup's tables hold
constexpr int kBlobTile = 64;  // output tile edge of blob_mask_kernel
constexpr int kBlobHalo = 6;   // 4 (9x9 Gaussian) + 2 (5x5 filter)
constexpr int kBlobRegion = (kBlobTile + 2 * kBlobHalo) * (kBlobTile + 2 * kBlobHalo);  // 5776 region pixels per tile
constexpr int kBlobGather = ((kBlobRegion + 255) / 256) * 256;  // gather-table entries per tile (padded to 256)
constexpr int kSquareRows = 16;  // squared rows per workgroup of the activity pass = one band of the activity map
constexpr int kBlobMaxEdge = 832;  // widest frame: contour tables + padded mask must fit 160 KB of LDS
constexpr int kBlobActSlots = (kBlobMaxEdge * 3 / 16 + 63) / 64;  // 16-byte row segments per pre-pass lane (3)
struct BlobArgs {
  int64_t n_images;        // images of this launch; camera = (img_base + image) % C
  int64_t img_base;        // index of the launch's first image in the caller's batch
  int C, rows, cols;       // raw frame size
  int S, ay;               // squared frame edge (= cols), first frame row inside it
  int M_max;
  const uint8_t* raw;      // [n_images][rows][cols][3] RGB
  const int32_t* rot;      // [C] quarter turns (0 or 2)
  // per (distinct lens and rotation, tile): for each pixel of the tile's 76 x 76 region (reflect-101 already applied)
  // byte offset of the first tap pair in the RAW frame | fx << 22 | fy << 27 (fractions already swapped for
  // rotated cameras); offset 0x3fffff = zero pixel, or a pixel the fix-up list overwrites
  const uint32_t* gather;
  // fix-up list of (lens, tile) lt: fix_cnt[lt] records of 5 words from fix_rec[5 * fix_off[lt]]:
  // {region index | fx << 16 | fy << 24, tap TL, TR, BL, BR = raw byte offset | scale << 22 (0 = zero pixel)}
  const uint32_t* fix_rec;
  const int32_t* fix_off;
  const int32_t* fix_cnt;
  const int32_t* cam_lens;   // [C] index of the camera's lens table
  // dark-tile early-out (exact): the activity pass records min / max of the squared frame's bytes per (16-row band, 16-byte
  // segment); a tile whose source bounding box spans a value range <= 2 cannot produce a set mask bit
  uint8_t* activity;         // [n_images][bands][segs][2] (min, max); bands = ceil((rows + 16) / 16), segs = cols * 3 / 16
  const int16_t* tile_box;   // [lens][tiles^2][4]: first band, last band, first segment, last segment; band < 0 = never skip
  const uint8_t* tile_zero;  // [lens][tiles^2] 1 = the tile's taps reach zero rows / the zero frame (value 0 takes part)
  int skip_dark;
  unsigned long long* mask;  // [n_images][S][ceil(S / 64)] thresholded frame, 1 bit per pixel
  uint8_t* processed;      // [n_images][S][S][3] BGR frame as the reference streams it, or null
  float* blobs;            // [n_images][M_max][2]
  int32_t* counts;         // [n_images]
  int32_t* status;         // [n_images]
  int32_t* n_contours;     // [n_images] or null
};
hipError_t launch_blob_activity(const BlobArgs& a, hipStream_t stream);
hipError_t launch_blob_mask(const BlobArgs& a, hipStream_t stream);
hipError_t launch_blob_contours(const BlobArgs& a, int P_cap, int N_cap, int only_overflowed, hipStream_t stream);
size_t blob_contour_lds_bytes(int S, int P_cap, int N_cap);

// explicit-correspondence triangulation, optionally batched over P camera sets (bundle adjustment)
struct TriArgs {
  CamView cv;             // tables of camera set 0; set p is offset by the strides below
  int64_t N;              // points
  int P;                  // camera sets (1 for plain triangulation)
  size_t stride_Pq, stride_RT;  // doubles between consecutive camera sets
  const double* obs;      // [N][C][2], NaN = unseen
  double* xyz;            // [P][N][3] or null
  double* err;            // [P][N] or null
  const double* xyz_in;   // null, or [N][3]: score these points instead of triangulating (helpers.py:214-241)
};
hipError_t launch_triangulate(const TriArgs& a, hipStream_t stream);

// ---- bundle adjustment building blocks (csrc/ba_kernels.hip)
struct BaCamArgs {
  int C, n, P, uniformK;
  const double* params;  // [P][n], or null: the finite-difference batch of `x` is formed on the fly
  const double* x;       // [n] base point (params == null): set 0 = x, set 1 + j = x + h_j e_j
  double x_inline[64];   // the same by value when n <= 64 (x == null): kernel arguments, no PCIe read
  double rel_step;       // h_j = rel_step * sign(x_j) * max(1, |x_j|) (scipy _numdiff), written to hvec
  double* hvec;          // [n]
  const double* K;       // [C][9]
  double* Pq;            // [P][...]
  double* RT;            // [P][C][12]
  size_t stride_Pq, stride_RT;
};
hipError_t launch_ba_build_cameras(const BaCamArgs& a, hipStream_t stream);

// x (n) -> params [n+1][n]: row 0 = x, row 1+j = x + h_j e_j ; h written to hvec [n]
hipError_t launch_ba_perturb(const double* x, int n, double rel_step, double* params, double* hvec,
                             hipStream_t stream);

struct BaJacArgs {
  int n, NP;             // parameters, padded row length of Jaug (multiple of 16, >= n + 1)
  int64_t N, m;          // points, valid points
  const int32_t* valid;  // [m] indices of valid points
  const double* r;       // [n+1][N] residuals (row 0 at x)
  const double* hvec;    // [n]
  int f32_residuals, use_cauchy;
  double* Jaug;          // [m_pad][NP]: scaled J | scaled f | 0 padding; m_pad = ceil(m/4)*4
  double* rho0;          // [m] loss values (cost = 0.5 * sum)
};
hipError_t launch_ba_jacobian(const BaJacArgs& a, hipStream_t stream);

// G = Jaug^T Jaug on the matrix cores (v_mfma_f64_16x16x4_f64); G [NP][NP]
hipError_t launch_ba_gram(const double* Jaug, int64_t m_pad, int NP, double* partial, int ksplit,
                          double* G, hipStream_t stream);
// the same, with the cost of residual row r (launch_ba_cost) evaluated by one extra workgroup of the reduce launch
hipError_t launch_ba_gram_cost(const double* Jaug, int64_t m_pad, int NP, double* partial, int ksplit, double* G,
                               const double* r, const int32_t* valid, int64_t m, int f32_residuals, int use_cauchy,
                               double* cost_out, hipStream_t stream);
int ba_gram_ksplit(int64_t m_pad, int NP);

// one linearisation in one launch (see ba_fused_kernel)
struct BaFusedArgs {
  int C, n, NP, uniformK, f32_rounding, f32_residuals, use_cauchy;
  int chunks;            // ceil(N / 64)
  int groups;            // ba_fused_groups(n) workgroups per chunk
  int debug_stop;        // 0; timing experiments: leave the kernel after phase (debug_stop - 1), results invalid
  int64_t N;
  double rel_step;
  double stamp;          // written to out[NP*NP + 2] when G and the cost are in place
  // null: the base point is `x` below.  Else pinned host memory, 64-byte lines {tag, 7 doubles of x}: the kernel was
  // launched AHEAD of the host's decision and waits until every line's tag == stamp (tag of line 0 negative = abandon)
  const double* mailbox;
  double* dev_mail;      // device memory [2 + 128]: the lead workgroup republishes the mailbox here for the others
  double x[128];         // base point by value (n <= 127): kernel arguments, no PCIe read
  const double* K;       // [C][9]
  const double* K4;      // [j][4]
  const double* obs;     // [N][C][2]
  double* r;             // [n+1][N] residuals
  double* partial;       // [ba_fused_records][(n+1)(n+2)/2] packed upper triangle of each chunk's / tree node's Gram matrix
  double* cost_part;     // [ba_fused_records][2] (sum of loss values, all-finite flag)
  int32_t* counters;     // [ba_fused_counters], zero on entry and on exit
  double* Jaug_out;      // null, or [N][NP]: the rows of Jaug by point index (tests)
  double* out;           // pinned host memory [NP*NP + 3]: packed upper triangle of G ... | cost | finite | stamp (last 3)
};
size_t ba_fused_lds_bytes(int C, int NP, bool uniformK);
bool ba_fused_eligible(int C, int n, int NP, bool uniformK);
int ba_fused_groups(int C);    // workgroups per chunk: ceil(live parameter sets / sets per workgroup)
size_t ba_fused_records(int chunks);   // records of the reduction tree (one per chunk + the inner nodes)
size_t ba_fused_counters(int chunks);  // arrival counters: one per chunk, one per tree node
hipError_t launch_ba_fused(const BaFusedArgs& a, hipStream_t stream);

// words (4 bytes) from pinned host memory -> device memory, plus `n_zero_words` words at `zero` set to 0, in one launch
hipError_t launch_ba_stage(const void* src, void* dst, size_t n_words, void* zero, size_t n_zero_words, hipStream_t stream);

// cost-only evaluation: sum of rho over valid points of residual row r [N]
hipError_t launch_ba_cost(const double* r, const int32_t* valid, int64_t m, int f32_residuals,
                          int use_cauchy, double* out /*[2]: cost, finite flag*/, hipStream_t stream);

}  // namespace mocap
