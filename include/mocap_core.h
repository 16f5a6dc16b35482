/* mocap_core.h -- C ABI of the MI355X-native multi-view marker-tracking core.
 *
 * Drop-in boundary for the hot path of jyjblrd/Low-Cost-Mocap
 * (computer_code/api/helpers.py:203-421).  The reference has no FFI/plugin layer:
 * the seam is four module-level Python functions imported by name at
 * computer_code/api/index.py:1.  A Python module with the same four names
 * (low-cost-mocap_amd/mocap_core/helpers.py) binds these entry points with ctypes;
 * INTEGRATION.md shows the stub a maintainer adds to the reference.
 *
 * Conventions
 *   - plain pointers and sizes only; all matrices row-major; doubles unless stated.
 *   - "host" entry points take caller-owned host buffers, copy in/out and return
 *     after the work finished.  "_dev" entry points take DEVICE pointers, enqueue on
 *     the context's stream and return immediately (mocap_synchronize to wait).
 *   - the library never retains a caller pointer past return.
 *   - return value: 0 = ok, negative = error (MOCAP_E_*); text in mocap_last_error.
 *     Per-frame conditions (candidate/root cap overflow) are reported in `status`,
 *     never by aborting.
 *   - a context is internally locked: calls on one context from several threads
 *     serialise (Flask-SocketIO handlers and the MJPEG generator run on different
 *     threads with no locking, reference helpers.py:84-94 vs :165-186).
 *   - unseen observations are NaN (the reference uses None / [None, None]).
 */
#ifndef MOCAP_CORE_H
#define MOCAP_CORE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mocap_ctx mocap_ctx;

enum {
  MOCAP_OK = 0,
  MOCAP_E_ARG = -1,      /* bad argument (null pointer, size out of range)          */
  MOCAP_E_HIP = -2,      /* a HIP runtime call failed                                */
  MOCAP_E_NOCAMS = -3,   /* mocap_set_cameras has not been called                    */
  MOCAP_E_LIMIT = -4,    /* configuration exceeds a compiled limit (see mocap_limits) */
  MOCAP_E_NOCONV = -5    /* solver hit its iteration cap (result still written)      */
};

/* per-frame status bits written by mocap_match_triangulate* */
enum {
  MOCAP_ST_ROOT_OVERFLOW = 1, /* more roots than K_max: frame output invalid          */
  MOCAP_ST_CAND_OVERFLOW = 2, /* a root has more than G_cap candidate groups: invalid */
  MOCAP_ST_HIT_OVERFLOW = 4,  /* wide frames only: a (root, camera) pair has more gated hits
                                 than hit_cap (mocap_set_frame_limits): invalid          */
  MOCAP_ST_ROUNDED = 8,       /* mocap_match_triangulate_f64 only, INFORMATIONAL (outputs valid): a coordinate of the
                                 frame was not representable in float32 and was rounded to the nearest float32 */
  MOCAP_ST_INTRACTABLE = 16,  /* set WITH MOCAP_ST_CAND_OVERFLOW by the re-submit pass: a root of the frame has more than 2^24
                                 candidate groups and the exact search over its multi-hit cameras (csrc/heavy_bb.hip) could not
                                 bound it -- no enumeration reaches it, the reference's own (helpers.py:394-400) included;
                                 bits 20..28 of the status word then hold ceil(log2(groups)) of the largest such root */
  MOCAP_ST_FINAL = 32         /* set by the re-submit pass on a frame it leaves flagged: larger caps do not exist, a repeated
                                 re-submit skips the frame (frames WITHOUT this bit that are still flagged were not reached:
                                 scratch batch full -- mocap_resubmit_dev continues with them) */
};
#define MOCAP_ST_LOG2_GROUPS_SHIFT 20
#define MOCAP_ST_LOG2_GROUPS_MASK 0x1FF

/* flags for mocap_set_options */
enum {
  MOCAP_OPT_F32_ROUNDING = 1 /* default ON: reproduce OpenCV's float32 roundings of the
                                epipolar line (helpers.py:363) and of the projected point /
                                pixel (helpers.py:231-237).  OFF = all-double arithmetic. */
  ,
  MOCAP_OPT_EXHAUSTIVE_WALK = 2 /* default OFF.  ON: every candidate group of the Cartesian product is triangulated and
                                   reprojected (helpers.py:408-421 as written) -- no eigenvalue bound drops or cuts anything
                                   (csrc/frame_kernel.hip without its cut-offs instead of csrc/frame_bb.hip).  Same results
                                   bit for bit, ~3 x the time: the verification mode bench.py's full-batch parity field and
                                   tests/test_gpu_bench_scale.py compare the shipped selection against. */
  ,
  MOCAP_OPT_BOUNDED_RESUBMIT = 4 /* default OFF.  ON: the re-submit pass of the wide variant does not ENUMERATE the roots its
                                   exact search gives up on (2^16 .. 2^24 groups: ~3 ms of the whole GPU each, csrc/heavy_bb.hip
                                   heavy_enum_kernel); their frames keep MOCAP_ST_CAND_OVERFLOW | MOCAP_ST_FINAL instead.  For
                                   callers that prefer a bounded step time to the last 0.1 % of the frames of a stress batch;
                                   every frame that is returned is still exact. */
};

/* ---------------------------------------------------------------- lifetime */
int mocap_create(int device_id, mocap_ctx** out);
void mocap_destroy(mocap_ctx* ctx);
const char* mocap_last_error(const mocap_ctx* ctx);
const char* mocap_version(void);
/* mocap_last_frame_kernel: which kernel the last frame batch of this context went to ("frame_bb_kernel<CW=1>" = the
 * exact branch-and-bound kernel of csrc/frame_bb.hip, "frame_kernel<256>" = the exhaustive walk, ...).  Diagnostic:
 * results never depend on it.  The string is a literal owned by the library. */
const char* mocap_last_frame_kernel(mocap_ctx* ctx);

/* enqueue on an existing hipStream_t (e.g. the host framework's current stream);
 * NULL restores the context's own stream. */
int mocap_set_stream(mocap_ctx* ctx, void* hip_stream);
int mocap_synchronize(mocap_ctx* ctx);
int mocap_set_options(mocap_ctx* ctx, uint32_t flags);
/* scheduling knobs of the frame path; results are bit-identical for every setting (tested).
 *   frame_threads    workgroup size 64 | 128 | 256 (0 = automatic: 64 for tiny frames, else 256)
 *   heavy_threshold  candidate count above which a frame is cut into slices evaluated by several
 *                    workgroups (-1 = automatic, 0 = never split)
 *   slice_size       target candidates per slice (0 = automatic) */
int mocap_set_tuning(mocap_ctx* ctx, int frame_threads, int heavy_threshold, int slice_size);
/* Frames whose state does not fit the 160 KB of LDS (e.g. 64 cameras x 256 blobs) run through a
 * "wide" variant that keeps hit lists and per-lane group columns in an HBM workspace and caps the
 * gated hits kept per (root, camera) at hit_cap (default 32, 0 = keep; the reference has no cap:
 * overflow sets MOCAP_ST_HIT_OVERFLOW and the frame is re-submitted with a larger cap).
 * force_wide != 0 routes every batch through that variant (tests).  Results are identical. */
int mocap_set_frame_limits(mocap_ctx* ctx, int hit_cap, int force_wide);
/* compiled limits: max cameras, max blobs per camera */
void mocap_limits(int* max_cameras, int* max_blobs);

/* ---------------------------------------------------------------- cameras
 * Replaces the per-call rebuild of P = K [R|t] (helpers.py:305-308, :351-355) and the
 * per-(root, camera) cv.sfm.fundamentalFromProjections (helpers.py:362): builds the
 * projection table (with the reference's "intrinsics by compacted index" quirk,
 * helpers.py:296-298,305-307) and the C x C fundamental table once.
 * K [C][9], R [C][9], t [C][3]: host pointers, copied.  Natural call site:
 * Cameras.start_trangulating_points (helpers.py:171-175). */
int mocap_set_cameras(mocap_ctx* ctx, int C, const double* K, const double* R, const double* t);
/* read back the host-side fundamental table F [C][C][9] (tests / debugging) */
int mocap_get_fundamental(mocap_ctx* ctx, double* F);

/* ---------------------------------------------------------------- triangulation
 * Replaces triangulate_points (helpers.py:330-336) + calculate_reprojection_errors
 * (helpers.py:203-211) for explicit correspondences.
 *   obs [N][C][2]  pixel observations, NaN = unseen
 *   xyz [N][3]     DLT point (NaN row when < 2 views; the reference yields [None]*3)
 *   err [N]        mean squared reprojection error in px^2 (NaN when < 2 views; the
 *                  reference skips the entry).  err may be NULL. */
int mocap_triangulate(mocap_ctx* ctx, int64_t N, const double* obs, double* xyz, double* err);
int mocap_triangulate_dev(mocap_ctx* ctx, int64_t N, const double* d_obs, double* d_xyz, double* d_err);

/* mocap_reproject: replaces calculate_reprojection_errors (helpers.py:203-241) for GIVEN object points:
 * err[i] = mean over the seen cameras' 2 v pixel components of (obs - cv.projectPoints(xyz[i]))^2, NaN when
 * fewer than two cameras see point i (the reference skips the entry).  The three call sites of the reference
 * pass the triangulation of the same observations (helpers.py:272,416; index.py:275), for which
 * mocap_triangulate's `err` output is the same number; this entry honours the function's own contract.
 *   obs [N][C][2] (NaN unseen), xyz [N][3], err [N] */
int mocap_reproject(mocap_ctx* ctx, int64_t N, const double* obs, const double* xyz, double* err);

/* ---------------------------------------------------------------- frame path
 * Replaces find_point_correspondance_and_object_points (helpers.py:339-421) for a
 * batch of independent frames.
 *   blobs  [F][C][M_max][2] float32 pixel centroids (slots >= counts are ignored)
 *   counts [F][C]           int32 blobs per camera (0 = the reference's [[None, None]])
 *   gate_px                 epipolar gate, the reference hard-codes 5 (helpers.py:375,383)
 *   K_max                   capacity for roots / output points per frame
 *   G_cap                   cap on candidate groups per root (the reference enumerates the
 *                           full Cartesian product, helpers.py:394-400)
 * outputs, root order as the reference (camera-0 blobs first, then leftovers of camera 1, ...):
 *   xyz    [F][K_max][3]    winning DLT point per kept root
 *   err    [F][K_max]       its mean squared reprojection error (px^2)
 *   corr   [F][K_max][C]    int16 blob index per camera of the winning group, -1 = none
 *   n_out  [F]              number of points written for the frame (slots beyond are untouched)
 *   status [F]              MOCAP_ST_* bits; non-zero = that frame's outputs are invalid,
 *                           re-submit it with larger caps
 *   n_cand [F]              (may be NULL) candidate groups evaluated for the frame */
int mocap_match_triangulate(mocap_ctx* ctx, int64_t n_frames, int M_max, const float* blobs,
                            const int32_t* counts, double gate_px, int K_max, int64_t G_cap,
                            double* xyz, double* err, int16_t* corr, int32_t* n_out,
                            int32_t* status, int32_t* n_cand);
int mocap_match_triangulate_dev(mocap_ctx* ctx, int64_t n_frames, int M_max, const float* d_blobs,
                                const int32_t* d_counts, double gate_px, int K_max, int64_t G_cap,
                                double* d_xyz, double* d_err, int16_t* d_corr, int32_t* d_n_out,
                                int32_t* d_status, int32_t* d_n_cand);

/* mocap_resubmit_dev: the re-submit stage of mocap_match_triangulate_dev_auto on its own -- for a batch whose first pass has
 * run (same arguments, same buffers): every frame that is flagged and not MOCAP_ST_FINAL is re-run with the largest caps.
 * A caller that reads d_resubmitted[0] > d_resubmitted[1] after its own synchronisation calls this until the two agree;
 * every call repairs or finalises at least one frame. */
int mocap_resubmit_dev(mocap_ctx* ctx, int64_t n_frames, int M_max, const float* d_blobs, const int32_t* d_counts,
                       double gate_px, int K_max, double* d_xyz, double* d_err, int16_t* d_corr, int32_t* d_n_out,
                       int32_t* d_status, int32_t* d_n_cand, int32_t* d_resubmitted);

/* mocap_match_triangulate_dev_auto: mocap_match_triangulate_dev, then -- queued behind it on the context's stream, without
 * the host ever waiting -- every frame whose status is non-zero is re-run ON THE DEVICE with the largest caps the core has
 * (root capacity C * M_max as far as a kernel's LDS holds it, G_cap = 2^24 groups per root, every gated hit of a (root,
 * camera) pair kept): the reference enumerates the full product whatever its size (helpers.py:394-400).  The flagged
 * frames are gathered into a scratch batch whose length stays on the device, the frame kernel runs on it, results that
 * fit the caller's K_max slots are scattered back in place.  Status after the call has run: as
 * mocap_match_triangulate_auto (MOCAP_ST_ROOT_OVERFLOW = the frame needs n_out[f] > K_max slots and nothing was written
 * for it; consumers treat n_out > K_max as "no valid slot").  d_resubmitted (may be NULL): [2] device-accessible int32,
 * {frames flagged by the first pass, frames re-run}; the two differ only when the scratch batch could not hold every flagged
 * frame (the whole batch while that takes at most 256 MB, else one frame in eight, within MOCAP_RESUBMIT_SCRATCH_MB, default 8192; halved
 * until the allocation succeeds) -- those keep their status WITHOUT MOCAP_ST_FINAL: mocap_resubmit_dev continues with them
 * (the host-buffer entry points loop until none is left). */
int mocap_match_triangulate_dev_auto(mocap_ctx* ctx, int64_t n_frames, int M_max, const float* d_blobs,
                                     const int32_t* d_counts, double gate_px, int K_max, int64_t G_cap,
                                     double* d_xyz, double* d_err, int16_t* d_corr, int32_t* d_n_out,
                                     int32_t* d_status, int32_t* d_n_cand, int32_t* d_resubmitted);

/* mocap_match_triangulate_auto: mocap_match_triangulate, then every frame whose status is non-zero is re-submitted on the
 * GPU with the largest caps the core has (root capacity C * M_max, G_cap = 2^24 groups per root, every gated hit of a
 * (root, camera) pair kept) -- the reference enumerates the full product whatever its size (helpers.py:394-400).  After
 * return a non-zero status means: MOCAP_ST_ROOT_OVERFLOW = the frame is fine but needs n_out[f] > K_max output slots
 * (call again with a larger K_max); other bits = the frame exceeds even the largest caps.  *n_resubmitted (may be NULL)
 * = frames that took the second pass. */
int mocap_match_triangulate_auto(mocap_ctx* ctx, int64_t n_frames, int M_max, const float* blobs,
                                 const int32_t* counts, double gate_px, int K_max, int64_t G_cap, double* xyz,
                                 double* err, int16_t* corr, int32_t* n_out, int32_t* status, int32_t* n_cand,
                                 int32_t* n_resubmitted);

/* mocap_match_triangulate_f64: the same for DOUBLE centroids (the reference measures on whatever its image_points hold,
 * helpers.py:367-373: int64 for _find_dot's int() centroids, float64 for anything else).  blobs [F][C][M_max][2] float64.
 * Coordinates float32 can represent (every integer pixel, every float32-valued centroid) are used exactly; any other is
 * rounded to the nearest float32 and the frame's status gets MOCAP_ST_ROUNDED (informational); NaN / infinite
 * coordinates are MOCAP_E_ARG.  Everything else as mocap_match_triangulate_auto. */
int mocap_match_triangulate_f64(mocap_ctx* ctx, int64_t n_frames, int M_max, const double* blobs, const int32_t* counts,
                                double gate_px, int K_max, int64_t G_cap, double* xyz, double* err, int16_t* corr,
                                int32_t* n_out, int32_t* status, int32_t* n_cand, int32_t* n_resubmitted);

/* ---------------------------------------------------------------- before the path
 * Blob extraction: replaces the per-camera preprocessing of Cameras._camera_read (helpers.py:68-82:
 * np.rot90, make_square helpers.py:507-523, cv.undistort, cv.GaussianBlur (9,9), cv.filter2D with the 5x5
 * sharpening kernel, cv.cvtColor RGB2BGR) and Cameras._find_dot (helpers.py:143-163: grey, threshold
 * 255*0.2, cv.findContours RETR_TREE / CHAIN_APPROX_SIMPLE, cv.moments, int() centroid of every contour
 * with m00 != 0, in findContours' order) for a batch of frame sets.
 *
 * mocap_set_image_params: frame geometry and lens model, once per camera set (natural call site:
 * Cameras.__init__ / set_camera_params, helpers.py:20-22,195-201).  Builds the fixed-point undistortion
 * map cv.undistort would rebuild per frame.
 *   rows, cols   raw frame size (pseyepy RES_SMALL: 240 x 320); must be landscape with >= 8 rows of square
 *                padding above and below -- outside that domain the reference's make_square raises
 *   K [C][9]     intrinsic_matrix, dist [C][5] distortion_coef (k1 k2 p1 p2 k3)   (camera-params.json)
 *   rotation [C] quarter turns of np.rot90 (may be NULL = 0); odd values are rejected like the reference */
int mocap_set_image_params(mocap_ctx* ctx, int C, int rows, int cols, const double* K, const double* dist,
                           const int32_t* rotation);
/* read back camera `camera`'s packed map [S][S] (S = cols): fx | fy << 5 | (sx + 1) << 10 | (sy + 1) << 21,
 * sx field 2047 = every tap outside the frame (tests / debugging) */
int mocap_get_undistort_map(mocap_ctx* ctx, int camera, uint32_t* map);

/* scheduling knob of the blob stage; results are bit-identical for either setting (tested).
 *   skip_dark_tiles  default 1: a 64 x 64 tile whose source bytes span a value range <= 2 (activity map of
 *                    the pre-pass) provably yields no mask bit and is not filtered (exact early-out; IR
 *                    frames are black but for the dots).  0 = filter every tile.  2 (round 6) = the same early-out decided
 *                    INSIDE the mask pass on the range of the tile's undistorted region (no activity pass, no second read
 *                    of the image; a dark tile then pays its gather first). */
int mocap_set_blob_options(mocap_ctx* ctx, int skip_dark_tiles);

/* per-image status bits written by mocap_find_blobs* */
enum {
  MOCAP_BLOB_ST_POINT_OVERFLOW = 1, /* more centroids than M_max: the first M_max were kept            */
  MOCAP_BLOB_ST_CAP_OVERFLOW = 2    /* more than 8192 border pairs / 1024 contours in one image (noise,
                                       not blobs): count = 0                                            */
};
/* mocap_find_blobs
 *   images    [F][C][rows][cols][3] uint8 RGB, what pseyepy's Camera.read() returns per camera
 *   M_max     blob slots per camera
 *   blobs     [F][C][M_max][2] float32 centroids (x, y): exactly the frame path's input layout, so the
 *             _dev variant chains into mocap_match_triangulate_dev without leaving HBM
 *   counts    [F][C] centroids per camera (0 = the reference's [[None, None]], helpers.py:158-159)
 *   status    [F][C] MOCAP_BLOB_ST_* bits
 *   processed [F][C][cols][cols][3] (may be NULL) the BGR frame the reference streams to the UI
 *             (helpers.py:82,141; without the debug drawings of helpers.py:148,155-156)
 *   n_contours [F][C] (may be NULL) contours found, including the zero-area ones the reference skips */
int mocap_find_blobs(mocap_ctx* ctx, int64_t n_frames, const uint8_t* images, int M_max, float* blobs,
                     int32_t* counts, int32_t* status, uint8_t* processed, int32_t* n_contours);
int mocap_find_blobs_dev(mocap_ctx* ctx, int64_t n_frames, const uint8_t* d_images, int M_max, float* d_blobs,
                         int32_t* d_counts, int32_t* d_status, uint8_t* d_processed);

/* ---------------------------------------------------------------- after the path
 * World-coordinate epilogue of the frame loop (helpers.py:96-103), fused into the frame path's store:
 * with a matrix set, `xyz` of mocap_match_triangulate* leaves the kernel as
 *   p' = diag(-1,-1,1) p ;  h = to_world [p'; 1] ;  q = h[:3] / h[3] ;  (q.x, q.z, q.y)
 * to_world: 16 doubles row-major (Cameras.to_world_coords_matrix), copied; NULL switches it off. */
int mocap_set_world_transform(mocap_ctx* ctx, const double* to_world);

/* locate_objects (helpers.py:424-480): the 3-LED drone patterns among each frame's points.
 *   xyz [F][K_max][3], err [F][K_max], n_pts [F]   the frame path's outputs (world coordinates)
 *   pos [F][O_max][3]   midpoint of the 0.15 m pair            ("pos")
 *   heading [F][O_max]  -atan2 of the pair direction, folded into [-pi/2, pi/2]   ("heading")
 *   oerr [F][O_max]     mean error of the three points          ("error")
 *   drone [F][O_max]    0 / 1 by the side the lead point is on  ("droneIndex")
 *   lead [F][O_max]     (may be NULL) index of the lead point
 *   n_obj [F]           objects found (entries beyond O_max are dropped, the count is not) */
int mocap_locate_objects(mocap_ctx* ctx, int64_t n_frames, int K_max, const double* xyz, const double* err,
                         const int32_t* n_pts, int O_max, double* pos, double* heading, double* oerr,
                         int32_t* drone, int32_t* lead, int32_t* n_obj);
int mocap_locate_objects_dev(mocap_ctx* ctx, int64_t n_frames, int K_max, const double* d_xyz,
                             const double* d_err, const int32_t* d_n_pts, int O_max, double* d_pos,
                             double* d_heading, double* d_oerr, int32_t* d_drone, int32_t* d_lead,
                             int32_t* d_n_obj);

/* mocap_track_frame: the body of the reference's live loop in ONE call (Cameras._camera_read, helpers.py:94-133):
 *   find_point_correspondance_and_object_points (helpers.py:94) -> world coordinates (helpers.py:96-103, needs
 *   mocap_set_world_transform; without it the points stay in camera-0 coordinates) -> locate_objects (helpers.py:107-108,
 *   when O_max > 0 = Cameras.is_locating_objects) -> everything the `object-points` event carries (helpers.py:128-133).
 * One enqueue (frame kernel, then one wave per frame that runs the object search and exports the valid slots into pinned
 * host memory), one event wait; frames that hit a candidate / hit-list cap are re-submitted with the core's largest caps
 * before the call returns -- those frames only, root-capacity overflow included.  Meant for one or a few frames per call
 * (host buffers; for batches use the _dev form).
 *   blobs, counts, gate_px, K_max, G_cap, xyz, err, corr (may be NULL), status   as mocap_match_triangulate
 *   n_pts [F]      points of the frame (mocap_match_triangulate's n_out)
 *   O_max          object slots per frame; 0 = no object search (pos .. n_obj may then be NULL)
 *   pos [F][O_max][3], heading [F][O_max], oerr [F][O_max], drone [F][O_max], n_obj [F]   as mocap_locate_objects
 * status & MOCAP_ST_ROOT_OVERFLOW after return: the frame has n_pts[f] > K_max points (call again with that K_max). */
int mocap_track_frame(mocap_ctx* ctx, int64_t n_frames, int M_max, const float* blobs, const int32_t* counts,
                      double gate_px, int K_max, int64_t G_cap, double* xyz, double* err, int16_t* corr,
                      int32_t* n_pts, int32_t* status, int O_max, double* pos, double* heading, double* oerr,
                      int32_t* drone, int32_t* n_obj);
/* the same from RAW camera frames (helpers.py:68-133: _camera_read's preprocessing + _find_dot in front, needs
 * mocap_set_image_params): images [F][C][rows][cols][3] uint8 RGB (host); additionally returns the image points
 * (blobs [F][C][M_max][2], counts [F][C], blob_status [F][C] as mocap_find_blobs; the `image-points` event of
 * helpers.py:92 reads them). */
int mocap_track_frame_images(mocap_ctx* ctx, int64_t n_frames, const uint8_t* images, int M_max, double gate_px,
                             int K_max, int64_t G_cap, float* blobs, int32_t* counts, int32_t* blob_status,
                             double* xyz, double* err, int16_t* corr, int32_t* n_pts, int32_t* status, int O_max,
                             double* pos, double* heading, double* oerr, int32_t* drone, int32_t* n_obj);
/* device-pointer form for batches: frame kernel, device-side re-submission of the frames that hit a cap (as
 * mocap_match_triangulate_dev_auto) and object search, all enqueued on the context's stream. */
int mocap_track_frame_dev(mocap_ctx* ctx, int64_t n_frames, int M_max, const float* d_blobs, const int32_t* d_counts,
                          double gate_px, int K_max, int64_t G_cap, double* d_xyz, double* d_err, int16_t* d_corr,
                          int32_t* d_n_pts, int32_t* d_status, int O_max, double* d_pos, double* d_heading,
                          double* d_oerr, int32_t* d_drone, int32_t* d_n_obj);

/* ---------------------------------------------------------------- initial poses
 * The pose-chaining loop of the `calculate-camera-pose` handler (index.py:229-270), i.e. the caller of
 * bundle_adjustment: per neighbouring camera pair cv.findFundamentalMat(FM_RANSAC, threshold, confidence)
 * on the points both cameras saw (index.py:241-246), cv.sfm.essentialFromFundamental with the intrinsics of
 * cameras 0 and 1 (index.py:247), cv.sfm.motionFromEssential (index.py:248), the cheirality vote over the four
 * candidates through triangulate_points (index.py:250-262), R = R_c R_prev, t = t_prev + R_prev t_c (:264-265).
 *   obs [N][C][2]   calibration points, NaN = unseen (the handler's `cameraPoints` payload, index.py:232)
 *   K [C][9]        intrinsics; like the reference only those of cameras 0 and 1 are read
 *   threshold, confidence, max_iters   cv.findFundamentalMat arguments (reference: 1, 0.99999, default 1000)
 *   R [C][9], t [C][3]   poses, camera 0 = (I, 0); t has the unit scale of motionFromEssential per pair
 *   info [C-1][4]   (may be NULL) per pair: correspondences, RANSAC inliers, RANSAC iterations, candidate index
 * RANSAC draws its subsets from cv::RNG((uint64)-1) exactly like cv::RANSACPointSetRegistrator and replays its
 * bookkeeping over per-model inlier counts computed on the GPU.  Fewer than 15 common points: MOCAP_E_ARG
 * (OpenCV would silently switch to LMedS). */
int mocap_initial_poses(mocap_ctx* ctx, int C, int64_t N, const double* obs, const double* K, double threshold,
                        double confidence, int max_iters, double* R, double* t, int32_t* info);
/* cv.findFundamentalMat(p1, p2, FM_RANSAC, threshold, confidence, max_iters): p1, p2 [n][2] float32 (host),
 * F [9] row-major with F[8] = 1, mask [n] (may be NULL) 1 = inlier, info [3] (may be NULL) = inliers,
 * iterations run, iteration that produced F.  The result is the best minimal 7-point model (OpenCV does not
 * refit on the inliers). */
int mocap_find_fundamental(mocap_ctx* ctx, int64_t n, const float* p1, const float* p2, double threshold,
                           double confidence, int max_iters, double* F, uint8_t* mask, int32_t* info);

/* ---------------------------------------------------------------- exchange payload (multi-GPU gather)
 * The frame path's outputs are fixed-capacity ([F][K_max] slots).  For the one exchange of a frame-sharded run
 * (final tracks -> the gathering rank over RCCL/xGMI) only the valid slots need to travel:
 * mocap_compact_tracks_dev packs them, in frame order, into records of mocap_track_record_bytes(C) bytes
 *     { xyz f64[3] | err f64 | corr i16[C] | zero pad to a multiple of 8 }
 * and writes the exclusive prefix sum of n_out: offsets[f] = first record of frame f, offsets[F] = number of
 * records (also to *d_total when given -- e.g. pinned host memory, so the host learns the payload size without
 * a separate copy).  Records beyond `capacity` are dropped (offsets still count them).  All pointers are DEVICE
 * pointers (d_total: device-accessible); enqueued on the context's stream. */
int mocap_track_record_bytes(int C);
int mocap_compact_tracks_dev(mocap_ctx* ctx, int64_t n_frames, int K_max, const int32_t* d_n_out,
                             const double* d_xyz, const double* d_err, const int16_t* d_corr,
                             int64_t* d_offsets, void* d_records, int64_t capacity, int64_t* d_total);

/* ---------------------------------------------------------------- bundle adjustment
 * Parameter vector as the reference (helpers.py:278-285):
 *   x = [f0, (f_i, rotvec_i[3], t_i[3]) for i = 1..C-1],  n = 1 + 7 (C-1); camera 0 = (I, 0).
 * The focal entries are carried but have no effect, exactly as in the reference
 * (helpers.py:267-270 writes them into a temporary copy).  Intrinsics come from
 * mocap_set_cameras (R, t given there are ignored by the BA entry points).
 *
 * mocap_ba_residuals: replaces residual_function (helpers.py:264-276) for a batch of P
 * parameter vectors: r[p][i] = mean squared reprojection error of point i re-triangulated
 * with the poses of params[p] (float64; the reference then casts to float32);
 * NaN where the point has < 2 views (the reference drops the entry).
 *   params [P][n], obs [N][C][2] (NaN unseen), r [P][N] */
int mocap_ba_residuals(mocap_ctx* ctx, int P, const double* params, int64_t N, const double* obs,
                       double* r);

/* mocap_ba_normal_eq: one linearisation at x: forward-difference Jacobian (step rule of
 * scipy.optimize._numdiff for the given residual precision: rel_step = sqrt(eps),
 * h = rel_step * sign(x) * max(1, |x|)), Cauchy loss scaling (scipy _lsq/least_squares.py
 * `cauchy`, `scale_for_robust_loss_function`), then the dense contractions on the matrix
 * cores:  JtJ [n][n] = J^T J,  Jtr [n] = J^T f,  cost = 0.5 * sum(rho(f^2)).
 *   f32_residuals != 0 reproduces the reference's float32 cast of the residuals
 *   (helpers.py:273) and the resulting float32 step size.
 *   J_out (may be NULL): [m][n] the scaled Jacobian, m = number of valid points (returned in *m_out) */
int mocap_ba_normal_eq(mocap_ctx* ctx, const double* x, int64_t N, const double* obs,
                       int f32_residuals, int use_cauchy, double* JtJ, double* Jtr, double* cost,
                       double* J_out, int64_t* m_out);

/* mocap_ba_trust_region_step: the subproblem of one iteration exactly as mocap_ba_solve solves it -- replaces
 * scipy.optimize._lsq.common.solve_lsq_trust_region (scipy _lsq/common.py:57-, called from _lsq/trf.py:495) --
 *     min 0.5 p^T JtJ p + Jtr^T p   subject to |p| <= Delta
 * stated on the normal equations (JtJ [n][n], Jtr [n] as mocap_ba_normal_eq returns them; m = rows of J, used
 * only by scipy's full-rank threshold eps*m*s_max).  *alpha_io: Levenberg-Marquardt parameter in (scipy's
 * initial_alpha; 0 = none) and out.  method 0 = what mocap_ba_solve does (exactly-zero rows/columns deflated;
 * Cholesky secular iteration when any were found, i.e. scipy's rank-deficient branch; symmetric
 * eigen-decomposition otherwise or when a pivot collapses), 1 = eigen-decomposition always, 2 = Cholesky
 * (MOCAP_E_NOCONV when a pivot collapses).  step [n] out; info [2] (may be NULL) = {method used (1|2), live
 * parameters}.  Host-only arithmetic: exported so that parity tests can compare the step itself. */
int mocap_ba_trust_region_step(mocap_ctx* ctx, int n, int64_t m, const double* JtJ, const double* Jtr,
                               double Delta, double* alpha_io, int method, double* step, int32_t* info);

/* mocap_set_ba_progress: the reference's residual_function emits the current poses to the UI on every
 * evaluation (socketio.emit("camera-pose"), helpers.py:274; App.tsx animates them).  mocap_ba_solve has no
 * per-evaluation host round trip; it calls `cb(x, n, user)` once per ACCEPTED step with the parameter vector
 * (on the calling thread of mocap_ba_solve, context lock held: do not call back into the same context).
 * NULL switches it off. */
int mocap_set_ba_progress(mocap_ctx* ctx, void (*cb)(const double* x, int n, void* user), void* user);

/* mocap_ba_profile: measurement aid (bench.py's `ba.roofline`).  `reps` linearisations at x exactly as the LM loop
 * issues them, timed with HIP events on the context's stream, then `reps` trust-region subproblems on the
 * resulting normal equations.  out [8] = {GPU microseconds per linearisation (event to event: every launch of
 * one linearisation), wall-clock microseconds per linearisation (launch + wait for the result on the host),
 * host microseconds per trust-region subproblem, kernel launches per linearisation, valid points m, padded
 * row length NP of [J | f], 1 if the one-launch kernel ran, cost at x}. */
int mocap_ba_profile(mocap_ctx* ctx, const double* x, int64_t N, const double* obs, int f32_residuals,
                     int use_cauchy, int reps, double* out);

/* mocap_ba_solve: resident Levenberg-Marquardt / trust-region loop (the algorithm of
 * scipy.optimize.least_squares(method="trf", loss="cauchy"), helpers.py:287-289, restated
 * on the normal equations).  x [n] in/out.
 *   ftol, xtol, gtol    termination tolerances (reference: ftol=1e-2, others 1e-8)
 *   max_iter            cap on LM iterations (0 = 100*n like scipy's max_nfev)
 *   f32_residuals       see above
 *   info [8]            (may be NULL) {iterations, nfev, status, cost0, cost, optimality, m, elapsed_ms}:
 *                       nfev / status / cost / optimality as scipy's OptimizeResult reports them; iterations also
 *                       counts passes whose every trial was rejected.  EXACTLY 8 doubles are written: this symbol
 *                       keeps the buffer size it was first exported with. */
int mocap_ba_solve(mocap_ctx* ctx, double* x, int64_t N, const double* obs, double ftol, double xtol,
                   double gtol, int max_iter, int f32_residuals, int use_cauchy, double* info);

/* mocap_ba_solve_ex: the same solve; `info_len` = doubles the caller's `info` holds, min(info_len,
 * MOCAP_BA_INFO_DOUBLES) are written: the 8 above, then [8] njev (= 1 + accepted steps, scipy's count), [9] number of
 * linearisations that had to be launched a second time because the kernel launched ahead of the host's decision had
 * abandoned itself (device watchdog, 2 s: the host was held up between two iterations; 0 in normal operation). */
#define MOCAP_BA_INFO_DOUBLES 10
int mocap_ba_solve_ex(mocap_ctx* ctx, double* x, int64_t N, const double* obs, double ftol, double xtol,
                      double gtol, int max_iter, int f32_residuals, int use_cauchy, double* info, int info_len);

#ifdef __cplusplus
}
#endif
#endif /* MOCAP_CORE_H */
