/*
 * boslam_hip.h — C ABI of libboslam_hip.so: the MI355X (gfx950) implementation of boslam's data-parallel hot path.
 *
 * The reference (BOpermanis/boslam) has no plugin/operator API: its hot path is a set of third-party Python call
 * sites (cv2 / g2o / pyDBoW3).  Every entry point below names the reference call site it replaces (file:line in
 * the reference tree).  INTEGRATION.md shows the ctypes stubs a maintainer adds at those sites.
 *
 * Conventions
 *   - plain pointers and sizes; the caller owns every buffer; the library never keeps a caller pointer past a call.
 *   - `mem` says where the bulk arrays of THAT call live: BH_MEM_HOST (the library stages them through its own
 *     device workspace and synchronises before returning) or BH_MEM_DEVICE (pointers are device pointers; the work
 *     is enqueued on the context's stream and the call returns without synchronising; small count arrays such as
 *     n_kp are device arrays too).
 *   - every function returns BH_OK (0) or a negative error; bh_last_error(ctx) gives the text.  No exceptions, no
 *     CPU fallback: without a usable HIP device bh_ctx_create fails.
 *   - one bh_ctx per calling thread (the reference runs the tracker on the main thread and local mapping on a
 *     daemon thread, slam/main.py:37-47): a context owns its stream and workspace, there is no global mutable state.
 */
#ifndef BOSLAM_HIP_H
#define BOSLAM_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BH_OK             0
#define BH_ERR_INVALID   -1   /* bad argument */
#define BH_ERR_HIP       -2   /* HIP runtime error (no device, launch failure, ...) */
#define BH_ERR_CAPACITY  -3   /* an output or internal list overflowed; results truncated in canonical order */
#define BH_ERR_NOMEM     -4
#define BH_ERR_NUMERIC   -5   /* linear solve failed / non-finite state (the shims map this to the reference's
                                 (False, None, None, None), slam/optimizer.py:140-142) */

#define BH_MEM_HOST       0
#define BH_MEM_DEVICE     1

typedef struct bh_ctx bh_ctx;

/* cv2.ORB_create(**orb_params) — camera.py:49-62.  WTA_K = 2 and scoreType = HARRIS_SCORE are the implicit
 * defaults the reference relies on; firstLevel must be 0 and patchSize 31 (the only values the reference uses). */
typedef struct {
    int   nfeatures;
    float scaleFactor;
    int   nlevels;
    int   edgeThreshold;
    int   firstLevel;
    int   patchSize;
    int   fastThreshold;
} bh_orb_params;

/* cv2.KeyPoint fields the reference reads (k.pt at camera.py:142) plus the rest of the detector output. */
typedef struct {
    float x, y;        /* pt, level-0 coordinates */
    float size;        /* patchSize * scale(level) */
    float angle;       /* degrees, [0,360) */
    float response;    /* Harris response */
    int   octave;      /* pyramid level */
} bh_kp;

/* ---- context ------------------------------------------------------------------------------------------------- */
int         bh_ctx_create(int device, bh_ctx** out);
void        bh_ctx_destroy(bh_ctx* ctx);
const char* bh_last_error(const bh_ctx* ctx);
/* Adopt an existing hipStream_t (e.g. the caller's torch stream); NULL restores the context's own stream. */
int         bh_ctx_set_stream(bh_ctx* ctx, void* hip_stream);
int         bh_ctx_synchronize(bh_ctx* ctx);
/* Blocking copy between host and device memory on the context's device (to_device != 0: host -> device).  For callers
 * without a HIP runtime binding of their own, e.g. an all-reduce hook that stages through host memory. */
int         bh_ctx_copy(bh_ctx* ctx, void* dst, const void* src, size_t bytes, int to_device);
/* Deferred status of BH_MEM_DEVICE work: capacity flags raised on the device are STICKY (every enqueued call ORs into one status
 * word; nothing clears it but this function), so one check after many enqueued steps reports a flag raised by any of them.
 * Synchronises the stream, returns the first error (or BH_OK) and clears the word. */
int         bh_ctx_status(bh_ctx* ctx);
const char* bh_version(void);
/* Optional per-kernel timing with hipEvents recorded on the context's stream around every kernel launch
 * (bench.py uses it for the roofline figure).  bh_ctx_timing(ctx, 1) resets and enables, (ctx, 0) disables;
 * bh_ctx_timing_read synchronises and returns, per kernel, the summed duration (ms) and the launch count. */
int         bh_ctx_timing(bh_ctx* ctx, int enable);
int         bh_ctx_timing_read(bh_ctx* ctx, int max_k, double* total_ms, long long* launches,
                               const char** names, int* n_k);

/* ---- ORB detect + describe ------------------------------------------------------------------------------------
 * Replaces feature_extractor.detectAndCompute(frame, mask=(depth_image > 0) * 255) — camera.py:141 — for a batch
 * of B frames.
 *   img      u8[B][H][W][channels], channels 3 (BGR, camera.py:70 bgr8) or 1 (gray)
 *   mask     u8[B][H][W] or NULL      (cv2 semantics: keypoints on mask==0 pixels are dropped)
 *   depth    u16[B][H][W] or NULL     (when given and mask is NULL the mask is (depth>0)*255, fused on device)
 *   kps      bh_kp[B][max_kp], desc u8[B][max_kp][32], n_kp int[B]
 * Keypoint order is canonical: level-major, raster (y, x) inside a level (DESIGN.md "canonical order").
 * max_kp must allow for ties at the retainBest threshold (nfeatures + slack); overflow raises BH_ERR_CAPACITY. */
int bh_orb_detect_describe(bh_ctx* ctx, const uint8_t* img, int channels, const uint8_t* mask,
                           const uint16_t* depth, int B, int H, int W, const bh_orb_params* params,
                           int max_kp, bh_kp* kps, uint8_t* desc, int* n_kp, int mem);

/* Debug/inspection: copy the per-frame intermediates of the LAST bh_orb_detect_describe call for frame `frame`
 * to host buffers (each may be NULL).  Layout: level l at off[l] with row pitch pitch[l] (bh_orb_geometry). */
int bh_orb_geometry(int H, int W, const bh_orb_params* params, int* w, int* h, int* pitch, int* off,
                    int* nfeat, int* total_bytes);
int bh_orb_debug_fetch(bh_ctx* ctx, int frame, uint8_t* pyr, uint8_t* mask_pyr, uint8_t* nms, uint8_t* blur);

/* kp_arr = [(int(x), int(y)) for k in kp] — camera.py:142 — and the keypoint gather of
 * convert_depth_frame_to_pointcloud — camera.py:90-118: xyz = ((u-ppx)/fx*z, (v-ppy)/fy*z, z), z = depth/1000.
 *   kp_xy i64[B][max_kp][2] (x,y) out ; cloud_kp f64[B][max_kp][3] out (may be NULL) */
int bh_kp_backproject(bh_ctx* ctx, const bh_kp* kps, const int* n_kp, const uint16_t* depth, int B, int H, int W,
                      int max_kp, double fx, double fy, double ppx, double ppy,
                      int64_t* kp_xy, double* cloud_kp, int mem);

/* ---- brute-force Hamming matcher with cross-check ------------------------------------------------------------------
 * Replaces cv2.BFMatcher_create(cv2.NORM_HAMMING, crossCheck=True).match(q, t) — slam/tracking.py:45,56,121,
 * slam/local_mapping.py:21, slam/covisibility_graph.py:34, experiments/pnp_two_way_tracking.py:11,42 — for B
 * independent (query, train) pairs.  Pair b: q + b*q_stride*32 with nq[b] rows, t + b*t_stride*32 with nt[b] rows.
 * Output pair b at out_stride*b: qidx/tidx/dist (ascending queryIdx, OpenCV's crosscheck rule), n_match[b]. */
int bh_match_hamming_crosscheck(bh_ctx* ctx, const uint8_t* q, const int* nq, int q_stride,
                                const uint8_t* t, const int* nt, int t_stride, int B,
                                int* qidx, int* tidx, int* dist, int* n_match, int out_stride, int mem);

/* Map-point medoid descriptor — replaces the double loop of MapPoint.add_observation,
 * slam/nodes.py:142-154: for M map points with k[m] <= 10 stored descriptors each (desc u8[M][10][32]) the index of the
 * descriptor whose column of the uint8 Hamming matrix has the smallest numpy-median (first minimum); -1 when k[m] <= 0. */
int bh_medoid_descriptor(bh_ctx* ctx, const uint8_t* desc, const int* k, int M, int* medoid_idx, int mem);
/* MapPoint.add_observation as a whole - slam/nodes.py:142-156, the per-feature body of CovisibilityGraph.add_kf
 * (slam/covisibility_graph.py:125-136) for features already associated with a map point - for M map points, in place:
 *   n = new_n / |new_n|;  n_obs += 1;  if fewer than 10 stored observations: the descriptor is appended and feat becomes the medoid
 *   of the stored descriptors (as bh_medoid_descriptor);  normal = (n + (n_obs - 1) normal) / n_obs.
 * obs_desc u8[M][10][32], obs_cnt int[M] (0..10), n_obs int[M], normal f64[M][3], feat u8[M][32] are updated; new_desc u8[M][32],
 * new_n f64[M][3] (not normalised, non-zero: the caller's `norm > 0` test, covisibility_graph.py:128). */
int bh_mp_add_observation(bh_ctx* ctx, int M, uint8_t* obs_desc, int* obs_cnt, int* n_obs, double* normal, uint8_t* feat,
                          const uint8_t* new_desc, const double* new_n, int mem);

/* The per-feature loop of CovisibilityGraph.add_kf as a whole - slam/covisibility_graph.py:123-136 - for ONE keyframe against a map-point
 * table whose row index is the map point's id (MapPoint.cnt, slam/nodes.py:106-110), in place:
 *   for every feature i (des u8[N][32], cloud_kp f64[N][3], des2mp int[N]): n = frame_t - cloud_kp[i]; if |n| > 0:
 *     des2mp[i] == -1 or not a live row -> a new map point: id = next_id, next_id + 1, ... in FEATURE ORDER; row = MapPoint(feat, d3, n / |n|,
 *                                          kf_id) (nodes.py:105-123: direction normalised once more, n_obs = 1, obs = [(kf_id, feat)]); des2mp[i] = id
 *     else                              -> MapPoint.add_observation(feat, n, kf_id) on that row (as bh_mp_add_observation, plus the observing
 *                                          keyframe in obs_kf); several features tied to one map point are applied in feature order.
 * Table (capacity cap rows): alive u8[cap], pt3d / normal f64[cap][3], feat u8[cap][32], obs_desc u8[cap][10][32], obs_kf int[cap][10],
 * obs_cnt / n_obs / first_kf int[cap].  action int[N]: 0 = skipped (|n| == 0), 1 = created, 2 = observed - what add_edge_to_cg (:136)
 * and KeyFrame.make_mp2kp (nodes.py:36-39: {des2mp[i]: kp_arr[i]}) need, both of which stay with the host's graph.
 * *n_created: the new map points (the caller's next MapPoint.cnt = next_id + *n_created).  BH_ERR_CAPACITY when the table is full.
 * Precondition (both memory modes; device arrays are not validated): des2mp[i] is -1 or an id the map has handed out, i.e. < next_id.  An entry
 * >= next_id counts as "not in self.mps" and creates a map point, independent of the rows this call itself creates.
 * BH_MEM_HOST copies the whole table up and back (tests, small maps); the live use keeps the table in HBM (BH_MEM_DEVICE). */
int bh_kf_add_map_points(bh_ctx* ctx, int kf_id, const double frame_t[3], int N, const uint8_t* des, const double* cloud_kp, int* des2mp,
                         int cap, int next_id, uint8_t* alive, double* pt3d, double* normal, uint8_t* feat, uint8_t* obs_desc, int* obs_kf,
                         int* obs_cnt, int* n_obs, int* first_kf, int* action, int* n_created, int mem);

/* ---- pose-only bundle adjustment ----------------------------------------------------------------------------------
 * Replaces CamOnlyBA(pixels, points, cam, ...) — slam/optimizer.py:82-142, called at slam/tracking.py:73,128 —
 * for B independent problems (one free VertexCam initialised to identity, n fixed points, EdgeProjectP2MC,
 * information I2, Huber(delta), Levenberg with g2o's lambda schedule, `iters` outer iterations).
 *   pix f64[B][stride][2], pts f64[B][stride][3], n int[B]; K = {fx, fy, cx, cy, baseline}
 *   R f64[B][9] row-major, t f64[B][3]  (SBACam = camera-in-world, as optimizer.py:137-139 returns it)
 *   chi2 f64[B][stride] (per-edge e'e at the state g2o leaves them in), inlier u8[B][stride] (chi2 <= chi2_thr)
 *   status int[B]: BH_OK or BH_ERR_NUMERIC per problem */
int bh_pose_only_ba(bh_ctx* ctx, const double* pix, const double* pts, const int* n, int B, int stride,
                    const double K[5], int iters, double huber_delta, double chi2_thr,
                    double* R, double* t, double* chi2, uint8_t* inlier, int* status, int mem);

/* ---- one tracking step on the device ------------------------------------------------------------------------------
 * Tracker._track_wrt_refkf (slam/tracking.py:54-89) and the matching half of Tracker._track_local_map (slam/tracking.py:121-136),
 * batched over B (frame, reference) pairs with no host hop between the stages:
 *   matcher.match(frame.des, ref_des) -> keep distance < d_max (`inclusive` != 0: <= d_max, tracking.py:121) -> need >= min_matches
 *   -> CamOnlyBA(frame.kp_arr[inds_f], ref_points[inds_ref], cam) (identity start, optimizer.py:82-142) -> need >= min_inliers.
 * Inputs: q_desc u8[B][q_stride][32], nq[B], q_xy i64[B][q_stride][2] (frame.kp_arr); t_desc u8[B][t_stride][32], nt[B],
 *   t_pts f64[B][t_stride][3] (kf_ref.cloud_kp, or the map points that passed bh_visibility_gate); K = {fx, fy, cx, cy, baseline}.
 * Outputs, pair b at out_stride*b: qidx / tidx / dist of the matches that pass the distance test (ascending queryIdx), n_match[b];
 *   R f64[B][9], t f64[B][3] (camera-in-world as CamOnlyBA returns them), inlier u8 per kept match, n_inlier[B];
 *   status[b] = BH_TRACK_OK | BH_TRACK_FEW_MATCHES (CamOnlyBA not run, R / t untouched) | BH_TRACK_BA_FAILED | BH_TRACK_FEW_INLIERS. */
#define BH_TRACK_OK           0
#define BH_TRACK_FEW_MATCHES  1
#define BH_TRACK_BA_FAILED    2
#define BH_TRACK_FEW_INLIERS  3
#define BH_TRACK_NO_CANDIDATES 4   /* bh_track_local_map: no map point passed the gate (`if len(feats) == 0: return False`, tracking.py:116-117) */
#define BH_TRACK_SKIPPED       5   /* bh_track_local_map: the frame's first stage did not return BH_TRACK_OK (`if is_ok:`, tracking.py:205-212) */
int bh_track_refkf(bh_ctx* ctx, const uint8_t* q_desc, const int* nq, const int64_t* q_xy, int q_stride,
                   const uint8_t* t_desc, const int* nt, const double* t_pts, int t_stride, int B,
                   int d_max, int inclusive, int min_matches, int min_inliers,
                   const double K[5], int iters, double huber_delta, double chi2_thr,
                   int* qidx, int* tidx, int* dist, int* n_match, double* R, double* t, uint8_t* inlier, int* n_inlier,
                   int* status, int out_stride, int mem);

/* ---- the local-map half of Tracker.update on the device ------------------------------------------------------------------
 * Tracker._track_local_map (slam/tracking.py:91-162) for B frames with no host hop: the visibility gate over the local map's
 * candidates (:97-111) -> feats / pts3d of the survivors stacked in loop order (:119-120) -> matcher.match(frame.des, feats) with
 * `distance <= d_max` (:121) -> need >= min_matches (:123) -> CamOnlyBA(frame.kp_arr[inds_frame], pts3d[inds], cam) (:128) -> need
 * >= min_inliers (:136).
 * Inputs: q_desc u8[B][q_stride][32], nq[B], q_xy i64[B][q_stride][2] (frame.des, frame.kp_arr); R_in f64[B][9], t_in f64[B][3] (frame.R,
 *   frame.t: the pose the first half set); see f64[B][3] (frame.see_vector) or NULL = R (0,0,1) normalised, what Frame.setPose leaves
 *   on a fresh frame (camera.py:26-29); stage1_status int[B] or NULL: the verdicts of bh_track_refkf for the same frames - Tracker.update calls
 *   _track_local_map only `if is_ok` (tracking.py:205-212), so a frame whose first stage is not BH_TRACK_OK is not gated, matched or
 *   optimised (n_gate = n_match = n_inlier = 0, R / t untouched) and reports BH_TRACK_SKIPPED; NULL runs every frame (a caller that has
 *   already filtered); the candidates in the loop order of :97-99 - every (local keyframe, its map point) pair, so a map
 *   point seen from several keyframes appears several times, as in the reference: mp_pts / mp_normals f64[B][m_stride][3] (mp.pt3d,
 *   mp.nf()), mp_feat u8[B][m_stride][32] (mp.featf()), M[B]; K = {fx, fy, cx, cy, baseline} (the gate uses fx, cx, cy like cam_map).
 * Outputs: gate_idx int[B][m_stride], n_gate[B]: the candidates that pass the gate, in order (ids_matching_kfs / ids_matching_mps are the
 *   caller's id arrays at these indices; num_frames_visible_increment applies to them); qidx / tidx / dist / n_match as bh_track_refkf
 *   (tidx indexes the GATED list, like `inds`); cand int[B][out_stride] = gate_idx[tidx]: the candidate index of each match, so
 *   ids_matching_mps[inds[inliers]] = mp_ids[cand[inlier]]; R, t, inlier, n_inlier as bh_track_refkf;
 *   status[b] = BH_TRACK_OK | BH_TRACK_SKIPPED | BH_TRACK_NO_CANDIDATES | BH_TRACK_FEW_MATCHES | BH_TRACK_BA_FAILED | BH_TRACK_FEW_INLIERS. */
int bh_track_local_map(bh_ctx* ctx, const uint8_t* q_desc, const int* nq, const int64_t* q_xy, int q_stride,
                       const double* R_in, const double* t_in, const double* see, const int* stage1_status,
                       const double* mp_pts, const double* mp_normals, const uint8_t* mp_feat, const int* M, int m_stride, int B,
                       int width, int height, double cos_thr, int d_max, int min_matches, int min_inliers,
                       const double K[5], int iters, double huber_delta, double chi2_thr,
                       int* gate_idx, int* n_gate, int* qidx, int* tidx, int* cand, int* dist, int* n_match,
                       double* R, double* t, uint8_t* inlier, int* n_inlier, int* status, int out_stride, int mem);

/* ---- loop-closure verification: rigid alignment of matched 3-D points ---------------------------------------------------
 * The stage slam/loop_closing.py:17-19 declares - GlobalMapManager._similiarity_transformation(self) -> (R, t), "R, t that
 * transforms start of the loop with the end" - and leaves as `pass`.  No reference behaviour exists; the
 * algorithm is specified here: for B independent problems, pairs (p_i, q_i) - e.g. cloud_kp of the loop candidate and of the
 * current keyframe at the indices of their descriptor matches - find R, t with q ~ R p + t by RANSAC over 3-point minimal sets
 * (cv::RNG((uint64)-1) sampling stream, OpenCV's adaptive iteration count) + the Horn / Kabsch closed form, then two rounds of
 * closed form over the inliers (|q - (R p + t)| <= inlier_thr, metres).  Bit-identical to the CPU checker oracle/align_oracle.c.
 *   P, Q f64[B][stride][3], n int[B] pairs used per problem; R f64[B][9] row-major, t f64[B][3], inlier u8[B][stride], n_inlier[B];
 *   status[b]: BH_ALIGN_OK | BH_ALIGN_FEW_POINTS (n < 3: R = I, t = 0) | BH_ALIGN_NO_MODEL (no non-degenerate sample reached 3 inliers,
 *   or fewer than min_inliers inliers: R, t hold the best estimate, or identity). */
#define BH_ALIGN_OK          0
#define BH_ALIGN_FEW_POINTS  1
#define BH_ALIGN_NO_MODEL    2
int bh_align_se3_ransac(bh_ctx* ctx, const double* P, const double* Q, const int* n, int B, int stride, int iterations, double inlier_thr,
                        double confidence, int min_inliers, double* R, double* t, uint8_t* inlier, int* n_inlier, int* status, int mem);

/* ---- loop closure: duplicate map points of the loop's two ends ------------------------------------------------------------
 * GlobalMapManager._loop_fusion(R, t) - slam/loop_closing.py:21-26, "finds duplicated mappoints and fuses them together", body `pass`:
 * PARITY UNPINNED BY CONSTRUCTION (no reference behaviour; specified by this project, checked against oracle.loop_fuse_pairs).
 * Side A = the map points around the current keyframe (feat_a u8[Na][32], pt_a f64[Na][3], nobs_a int[Na] = MapPoint.n_obs), side B = those
 * around the loop candidate, (R, t) from bh_align_se3_ransac (a ~ R b + t).  (i, j) is a duplicate when the cross-check Hamming matcher
 * pairs feat_a[i] with feat_b[j] with distance < d_max and |pt_a[i] - (R pt_b[j] + t)| <= radius.  Out: pair_a / pair_b int[cap] in
 * ascending i, keep_a u8[cap] (1: the A point survives - it has at least as many observations), *n_pairs. */
int bh_loop_fuse_pairs(bh_ctx* ctx, const uint8_t* feat_a, const double* pt_a, const int* nobs_a, int Na,
                       const uint8_t* feat_b, const double* pt_b, const int* nobs_b, int Nb,
                       const double R[9], const double t[3], int d_max, double radius,
                       int* pair_a, int* pair_b, uint8_t* keep_a, int* n_pairs, int cap, int mem);

/* ---- loop closure: essential-graph optimisation -----------------------------------------------------------------------------
 * GlobalMapManager._essential_graph_optimization - slam/loop_closing.py:28-29, body `pass`: PARITY UNPINNED BY CONSTRUCTION (specified by
 * this project, checked against oracle.pose_graph_optimize).  K keyframe poses R f64[K][9], t f64[K][3] (camera in world, updated in place),
 * fixed u8[K] (at least one: the gauge); E edges (ei, ej) with the measured relative pose Zr f64[E][9], Zt f64[E][3] ~ T_i^-1 T_j and a
 * weight w f64[E] - the spanning tree and strong covisibility edges as they were before the loop closed, plus the loop edge from
 * bh_align_se3_ransac.  Minimises sum_e w_e |(t_E, log R_E)|^2, E = Z^-1 T_i^-1 T_j, by `iters` Gauss-Newton steps (each: conjugate
 * gradients with a 6 x 6 block-Jacobi preconditioner to 1e-10; a step is kept only if the cost falls).  cost_log f64[iters + 1]: the cost
 * at the start and after every kept step (0 beyond); *steps_kept, *cg_iters.  Host arrays only (one call per closed loop). */
int bh_pose_graph_optimize(bh_ctx* ctx, int K, double* R, double* t, const uint8_t* fixed, int E, const int* ei, const int* ej,
                           const double* Zr, const double* Zt, const double* w, int iters, double* cost_log, int* steps_kept, int* cg_iters);

/* ---- local bundle adjustment --------------------------------------------------------------------------------------
 * Replaces the g2o graph of BundleAdjustment — slam/optimizer.py:8-79 — as driven by
 * CovisibilityGraph.optimize_local — slam/covisibility_graph.py:186-235.
 *   poses: R f64[C][9] row-major + t f64[C][3] (camera-in-world, as add_pose passes them, optimizer.py:21-38),
 *          fixed u8[C]
 *   points f64[P][3] (all free + marginalised, optimizer.py:40-47)
 *   edges  pose index i32[E], point index i32[E], measurement f64[E][2] (optimizer.py:49-62; the 640x480
 *          admission gate of optimizer.py:52-54 is applied by the caller-side shim or by bh_ba_edge_gate)
 * bh_ba_optimize runs `iters` Levenberg iterations (Schur complement on the points, dense Cholesky of the reduced
 * camera system); `verbose_errors` != 0 reproduces SparseOptimizer::optimize's verbose branch, which recomputes
 * the active errors after every iteration (BundleAdjustment.optimize passes flag_verbose=True, optimizer.py:16-19). */
typedef struct bh_ba bh_ba;
int  bh_ba_create(bh_ctx* ctx, const double K[5], bh_ba** out);
void bh_ba_destroy(bh_ba* ba);
int  bh_ba_set_problem(bh_ba* ba, int C, const double* R, const double* t, const uint8_t* fixed,
                       int P, const double* points, int E, const int* edge_pose, const int* edge_point,
                       const double* meas, int mem);
int  bh_ba_optimize(bh_ba* ba, int iters, double huber_delta, int verbose_errors, int* iters_done);
int  bh_ba_get_poses(bh_ba* ba, double* R, double* t, int mem);
int  bh_ba_get_points(bh_ba* ba, double* points, int mem);
int  bh_ba_get_chi2(bh_ba* ba, double* chi2, int mem);
/* per-iteration log of the last optimize: f64[iters_done][4] = {robust chi2 after, lambda, trials, accepted} */
int  bh_ba_get_log(bh_ba* ba, double* log, int max_rows);
/* which reduced-camera-system solver bh_ba_set_problem selected for the current problem (so that a test which claims to
 * compare two of them can assert it): 0 = panel kernel, 1 = whole factor on chip (n <= 288), 2 = on-chip sliding window,
 * 3 = two-way window elimination, 4 = block cyclic reduction over the whole chip, 5 = the same cut into one segment per rank
 * (sharded mode: bh_ba_set_allreduce); < 0 = no problem set. */
int  bh_ba_solver(bh_ba* ba);
/* Test hook: one structural array of the current problem as bh_ba_set_problem built it (the three builders - one workgroup for local
 * windows, chip-wide kernels, host loops behind BH_BA_BUILD=general|host - must produce the same arrays).  which: 0 e_pt[E],
 * 1 p_start[Pa+1], 2 p_edges[E], 3 c_start[Cf+1], 4 c_edges[c_total], 5 pairs[npairs] (int2), 6 j_start[npairs+1], 7 join[M] (int2),
 * 8 cslot[C]; out receives at most cap_ints 32-bit values, *n_ints how many. */
int  bh_ba_debug_structure(bh_ba* ba, int which, int* out, long long cap_ints, long long* n_ints);

/* Batched local BA: W windows, each a bh_ba handle of `ctx` that holds a problem (bh_ba_set_problem; windows may differ in size),
 * advance together - every kernel of a Levenberg trial is launched once for all of them, gridded over (item, window), and each
 * window's dense reduced-camera factorisation takes its own workgroup, so W windows keep W CUs busy where bh_ba_optimize keeps one.
 * The reference optimises one window at a time on the mapping thread (slam/local_mapping.py:46-47 -> slam/covisibility_graph.py:186-235);
 * SURVEY 8(e) "local BA: replicas only (different windows)".  A window's result is bitwise the one bh_ba_optimize gives; afterwards
 * the usual getters (bh_ba_get_poses / _points / _chi2 / _log) read each window.  Windows must be local ones (<= 48 free keyframes,
 * not sharded); iters_done (may be NULL) receives int[W]. */
typedef struct bh_ba_batch bh_ba_batch;
/* One window's arguments of bh_ba_set_problem (host arrays). */
typedef struct bh_ba_problem {
    int C; const double* R; const double* t; const uint8_t* fixed;
    int P; const double* points;
    int E; const int* edge_pose; const int* edge_point; const double* meas;
} bh_ba_problem;
/* bh_ba_set_problem for W windows of one context at once: what W calls would leave behind, but the local windows among them
 * (<= 64 free keyframes, <= 12 288 observations, <= 4 096 landmarks) are built together - one copy up, one workgroup per window in ONE
 * launch (slots, edges by landmark / by keyframe, keyframe pairs, join lists, camera blocks), one copy back - instead of ~60 launches
 * and four host reads each.  Windows that are not local ones go through bh_ba_set_problem one by one. */
int  bh_ba_batch_set_problems(bh_ctx* ctx, int W, bh_ba* const* windows, const bh_ba_problem* problems);
int  bh_ba_batch_create(bh_ctx* ctx, int W, bh_ba* const* windows, bh_ba_batch** out);
void bh_ba_batch_destroy(bh_ba_batch* batch);
int  bh_ba_batch_optimize(bh_ba_batch* batch, int iters, double huber_delta, int verbose_errors, int* iters_done);

/* Test hook: one banded SPD solve S x = rhs with the reduced-camera-system kernels.  band f64[n][bw+1] holds the lower band,
 * element (r, c) at band[r*(bw+1) + c - r + bw]; mode 0 = panel kernel, 1 = on-chip sliding-window kernel (15 <= bw <= 256),
 * 3 = two workgroups eliminating from both ends (needs a band of exactly 16 tiles, 241 <= bw <= 256, and >= 40 block columns),
 * 4 = the whole factor on chip (n <= 288; the local-BA window and the separator system),
 * 5 = block cyclic reduction in 256 x 256 blocks over the whole chip (bw <= 256; the global-BA solver for n >= 2048);
 * clocks (may be NULL) receives i64[8] phase clocks of the window kernel (100 MHz ticks). */
int  bh_ba_debug_band_solve(bh_ctx* ctx, const double* band, const double* rhs, int n, int bw, int mode, double* x,
                            long long* clocks);

/* Global BA over several GPUs (BASELINE configs[4]; no counterpart in the reference, whose covisibility_graph.py:186-235
 * optimises on one CPU).  Landmarks (with all their observations) are dealt out to the ranks; every rank passes all C poses (same
 * order, same fixed flags) and its own subset of points with their edges to bh_ba_set_problem.  Linearisation and the Schur products
 * are local.  The reduced camera system S is cut BY KEYFRAME into one contiguous segment of 256-row blocks per rank:
 *   - the partial S of all ranks is reduce-scattered: a rank receives the sum of its own segment's band rows (S / world bytes);
 *   - each rank eliminates the interior of its segment (block cyclic reduction on its own chip), leaving its two end blocks;
 *   - the end blocks (4 x 512 KB per rank) are all-gathered and the chain of 2 x world blocks is solved on every rank - the only
 *     replicated arithmetic; each rank back-substitutes its segment; the keyframes' step (6Cf f64) is all-gathered;
 *   - Hpp / bp (1.7 MB at configs[4]), b~ and three scalars per trial are all-reduced as before.
 * Systems too short for that (fewer than two blocks per rank, or not banded) keep the round-2 scheme: S all-reduced, solve replicated.
 *   fn(user, dev_buf, count, op) works in place on `count` f64 at device address `dev_buf`:
 *     op 0 all-reduce sum, op 1 all-reduce max,
 *     op 2 reduce-scatter sum: the buffer is `world` equal chunks; afterwards chunk `rank` holds the sum over ranks of that chunk,
 *     op 3 all-gather: chunk `rank` holds this rank's data; afterwards every chunk holds its owner's.
 *   The library's stream is idle when fn is called; fn returns (0 = ok) when the result is visible to the device.
 * Call before bh_ba_set_problem.  fn == NULL restores the single-GPU behaviour. */
typedef int (*bh_allreduce_fn)(void* user, void* dev_buf, long long count, int op);
int  bh_ba_set_allreduce(bh_ba* ba, bh_allreduce_fn fn, void* user, int rank, int world);
/* Stream-ordered hooks (round 5).  By default the library synchronises its stream before it calls fn and expects the result to be visible
 * when fn returns - ten host synchronisations per Levenberg trial of the segment scheme.  A hook that ENQUEUES its collective on the context's
 * own stream (bh_ctx_set_stream: e.g. a torch stream RCCL's work is ordered against) needs neither: with ordered != 0 the library calls fn
 * without synchronising and goes on enqueueing behind it.  bh_ba_get_comm's milliseconds are then the enqueue time only. */
int  bh_ba_set_allreduce_ordered(bh_ba* ba, int ordered);
/* wall time spent inside fn and bytes handed to it since the last bh_ba_optimize began */
int  bh_ba_get_comm(bh_ba* ba, double* ms, long long* bytes);
/* Sharded mode in detail since the last bh_ba_optimize began, out f64[8]: bytes received through plain all-reduces, through the
 * reduce-scatter of S (this rank's band rows), through the all-gathers (end blocks of the other segments + their rows of the step);
 * milliseconds and count of the replicated part of the reduced-camera solves (the chain of 2 x world end blocks); 1 if the
 * segment-wise solve is active; blocks of 256 rows this rank holds; blocks per rank. */
int  bh_ba_get_comm_detail(bh_ba* ba, double* out);

/* ---- local-BA window and problem construction ----------------------------------------------
 * Replaces CovisibilityGraph.get_local_map(kf, flag_inner_outer=True, flag_with_input_kf=False) — slam/covisibility_graph.py:51-78
 * — and the graph walk of optimize_local — :188-218 — up to the add_pose / add_point / add_edge calls.
 *   covisibility graph, CSR over K keyframes: nbr_start i32[K+1], nbr_id i32[], nbr_common i32[] (kf2kf_num_common_mps of the
 *     pair; a pair that is not listed counts 0, as the reference's defaultdict does)
 *   observations, CSR over keyframes: obs_start i32[K+1], obs_mp i32[] (map point id < n_mp), obs_px i64[][2] (kf.mp2kp[mp])
 *   kf: the keyframe being optimised; min_common: 15 in the reference (:65,69)
 * Outputs (the reference walks Python sets; here every set is enumerated in ascending id):
 *   win_kf i32[cap_kf] = [kf] + inner + (outer minus inner), win_fixed u8[cap_kf] (outer empty: kf fixed, :195; else outer fixed,
 *     :197-199), pt_mp i32[cap_pt] = map points seen by more than one window keyframe in order of first appearance along that
 *     chain (:205-215), edges (index into win_kf, index into pt_mp, pixel) in chain order (:216)
 *   counts i32[5] = {n_window, n_inner, n_outer, n_points, n_edges}; n_window == 0 is the reference's "nothing to optimize".
 * Host arrays only (the graph lives in Python objects in the reference); BH_ERR_CAPACITY if an output is too small. */
int bh_ba_window_problem(bh_ctx* ctx, int K, const int* nbr_start, const int* nbr_id, const int* nbr_common,
                         const int* obs_start, const int* obs_mp, const long long* obs_px, int n_mp, int kf, int min_common,
                         int cap_kf, int* win_kf, uint8_t* win_fixed, int cap_pt, int* pt_mp, int cap_e, int* edge_pose,
                         int* edge_point, long long* edge_px, int* counts);

/* ---- PnP + RANSAC, Rodrigues -----------------------------------------------------------------------------------------
 * Replaces cv2.solvePnPRansac(obj, img, K, distCoeffs=0, flags=cv2.SOLVEPNP_ITERATIVE) and cv2.Rodrigues(v)[0] —
 * experiments/pnp_two_way_tracking.py:57-61,68-69 (slam/tracking.py:68-71 is dead code) — for B problems.
 *   obj f64[B][stride][3], img f64[B][stride][2], n int[B] (n >= 4.  n == 4 takes OpenCV's four-point branch: model_points = 4 with
 *   SOLVEPNP_P3P - ONE hypothesis from all four points, the first three correspondences solved for up to four poses and the fourth picking
 *   the one it reprojects best, accepted only when all four are inliers; n < 4 reports ok = 0)
 *   K = {fx, fy, cx, cy}; distortion is zero in the reference (camera.py:82, config.py:34)
 *   defaults of the reference call: iterations 100, reproj_error 8.0, confidence 0.99 (any iterationsCount: the sample stream of
 *   cv::RNG((uint64)-1) is drawn hypothesis by hypothesis)
 *   the final pose is the Levenberg-Marquardt minimiser of the reprojection error over the RANSAC inliers, started from the RANSAC
 *   model (OpenCV: solvePnP(ITERATIVE) from a DLT guess, CvLevMarq 20 iterations / FLT_EPSILON): same minimum, poses agree to ~1e-6
 *   rvec/tvec f64[B][3] (world -> camera), inliers int[B][stride] ascending, n_inliers int[B], ok int[B] (1/0)
 * bh_rodrigues is plain host math (3-vector <-> 3x3 row-major matrix; in_len 3 or 9). */
int bh_solve_pnp_ransac(bh_ctx* ctx, const double* obj, const double* img, const int* n, int B, int stride,
                        const double K[4], int iterations, double reproj_error, double confidence,
                        double* rvec, double* tvec, int* inliers, int* n_inliers, int* ok, int mem);
int bh_rodrigues(const double* in, int in_len, double* out);
/* Test hooks (host arrays).  bh_debug_pnp_trace = bh_solve_pnp_ransac that also returns the intermediates of the first 32
 * RANSAC hypotheses of every problem (trace f64[B][32][512]: sample, control points, barycentric coordinates, M'M, its
 * eigen-decomposition, L, rho, betas / R / t / error for N = 1..3, chosen R t, inlier count), for locating the first operation
 * that differs from the CPU oracle's trace.  bh_debug_f64_ops evaluates the f64 primitives the pose kernels rely on for n operand
 * pairs: out f64[8][n] = a/b, sqrt|a|, sin a, cos a, acos(clamp a), log|a|, pow(|a|, 5), rint a. */
int bh_debug_pnp_trace(bh_ctx* ctx, const double* obj, const double* img, const int* n, int B, int stride,
                       const double K[4], int iterations, double reproj_error, double confidence,
                       double* rvec, double* tvec, int* inliers, int* n_inliers, int* ok, double* trace);
int bh_debug_f64_ops(bh_ctx* ctx, const double* a, const double* b, int n, double* out);
/* the four-point model (n == 4 branch of bh_solve_pnp_ransac) alone: obj f64[B][4][3], img f64[B][4][2] -> pose f64[B][13] = R, t, found */
int bh_debug_p3p(bh_ctx* ctx, const double* obj, const double* img, int B, const double K[4], double* pose);

/* ---- local-map visibility gate ------------------------------------------------------------------------------------
 * Replaces the per-map-point loop of Tracker._track_local_map — slam/tracking.py:97-111 — for B frames:
 *   q = R p + t ; pixel = (q.x/q.z*f + cx, q.y/q.z*f + cy)   [cam.cam_map(g2o.SE3Quat(frame.R, frame.t) * mp.pt3d)]
 *   keep iff 0 <= u < width and 0 <= v < height and dot(see_vector, n) < cos_thr (cos60, tracking.py:19,104)
 *   pts / normals f64[B][stride][3], M int[B]; R f64[B][9] row-major, t f64[B][3], see f64[B][3]
 *   idx int[B][stride]: surviving indices in input order, n_out int[B] */
int bh_visibility_gate(bh_ctx* ctx, const double* pts, const double* normals, const int* M, int B, int stride,
                       const double* R, const double* t, const double* see, double f, double cx, double cy,
                       int width, int height, double cos_thr, int* idx, int* n_out, int mem);

/* ---- DBoW3 L1 database query -----------------------------------------------------------------------------------------
 * Replaces Database::query(features, [], max_results, -1) — slam/bow_db.py:37-42 -> docker/pydbow3/src/dbow3.cpp:147-152
 * — at the BoW-vector level (vocabulary-tree quantisation of descriptors is SURVEY §8(f) rank 1):
 *   database: CSR over entries (keyframes): words u32 ascending per entry, weights f64 (TF-IDF, L1-normalised)
 *   query   : strictly ascending word ids + weights
 *   result  : up to max_results (id, Score = -sum(|q-d|-|q|-|d|)/2) best first; entries sharing no word are absent;
 *             equal scores are ordered by ascending entry id.
 * Multi-GPU: entries shard across ranks; each rank queries its shard and the per-rank top-k lists are all-gathered
 * (boslam_amd/bow.py). */
typedef struct bh_bow bh_bow;
int  bh_bow_create(bh_ctx* ctx, bh_bow** out);
void bh_bow_destroy(bh_bow* bow);
int  bh_bow_set_db(bh_bow* bow, long long n_entries, const int64_t* start, const uint32_t* words, const double* vals, int mem);
int  bh_bow_query_l1(bh_bow* bow, const uint32_t* q_words, const double* q_vals, int nq, int max_results,
                     int* ids, double* scores, int* n_results, int mem);

/* Q queries in one call (host arrays): q_words u32[Q][q_stride], q_vals f64[Q][q_stride], nq int[Q]; ids int[Q][max_results],
 * scores f64[Q][max_results], n_results int[Q] - each query answered exactly as bh_bow_query_l1 answers it.  With the inverted
 * file (the default mode, max_results <= 16) up to 16 queries share every launch: a loop closer or relocaliser working through a
 * queue of keyframes (slam/loop_closing.py:29-30 takes them from bow_queue one by one) pays the launch latency once per 16. */
int  bh_bow_query_l1_batch(bh_bow* bow, const uint32_t* q_words, const double* q_vals, const int* nq, int Q, int q_stride, int max_results,
                           int* ids, double* scores, int* n_results);

/* The fork's Database::query(features, target_inds, max_results, max_id) with a non-empty target_inds — slam/bow_db.py:37-39 ->
 * dbow3.cpp:147-152: only the entries listed in `subset` can appear in the result (host arrays). */
int  bh_bow_query_l1_subset(bh_bow* bow, const uint32_t* q_words, const double* q_vals, int nq, const int* subset, int n_subset,
                            int max_results, int* ids, double* scores, int* n_results);
/* The fork's Database::compareBowsL1(i, j, cnt, score) behind Dbow.compareFrames — slam/bow_db.py:47-57 -> dbow3.cpp:154-162:
 * number of words the two stored entries share and their L1 score -sum(|a - b| - |a| - |b|) / 2 over those words. */
int  bh_bow_compare_l1(bh_bow* bow, int entry_i, int entry_j, int* n_common, double* score);
/* How host-side queries are answered: 0 (default) = through the inverted file (word -> (entry, weight) rows, DBoW3's own structure,
 * built on the device after the database changed; ~1 MB touched per query of configs[3]) with the streaming scan as the fallback
 * (word ids >= 2^20, more than 2^21 database entries, max_results > 16, or a query whose hits exceed the staging: 65 536 overflow
 * records, 4 096 records in one bucket of 64 entries, 1 024 accumulators at the selection threshold); 1 = always the scan (400 MB
 * streamed per query).  A single host query returns when its pinned result block carries the query's tag (two launches; no stream wake-up).
 * Both give bit-identical ids and scores.  BH_MEM_DEVICE queries (bh_bow_query_l1 with device pointers; words strictly ascending is the
 * caller's promise there) take the inverted file too: its kernels run on the caller's query arrays, the results go straight into the
 * caller's device arrays, and the scan that is enqueued behind them returns at once when a device-side flag says the query has been
 * answered - no host decision, no synchronisation. */
int  bh_bow_set_mode(bh_bow* bow, int mode);

/* Database::add at the BoW-vector level — slam/bow_db.py:26 -> dbow3.cpp:138-140: appends one entry (host arrays, words
 * strictly ascending), returns its entry id. */
int  bh_bow_add_entry(bh_bow* bow, const uint32_t* words, const double* vals, int n, int* entry_id);
/* Database::erase (fork method, slam/bow_db.py:33 -> dbow3.cpp:142-145): the entry keeps its id and loses its words. */
int  bh_bow_erase_entry(bh_bow* bow, int entry_id);

/* ---- vocabulary-tree quantisation ---------------------------------------------------------
 * Replaces DBoW3 Vocabulary::transform behind Database.add(des) / Database.query(des, ...) — slam/bow_db.py:26,39 ->
 * docker/pydbow3/src/dbow3.cpp:138-152: descriptors -> sparse BoW vector (TF-IDF weights, L1-normalised).
 *   tree: node 0 is the root; children of node i are child_idx[child_start[i] .. child_start[i+1]) in their stored order
 *   (first minimum wins); leaves carry (word id < 2^20, weight; weight <= 0 = stop word); node_desc u8[n_nodes][32].
 *   bh_voc_transform: B descriptor sets (desc u8[B][stride][32], n[b] <= min(stride, 4096)) ->
 *   words u32[B][stride] ascending, vals f64[B][stride], n_words[B]. */
typedef struct bh_voc bh_voc;
int  bh_voc_create(bh_ctx* ctx, int n_nodes, const int* child_start, const int* child_idx, const uint8_t* node_desc,
                   const double* node_weight, const int* node_word, bh_voc** out);
void bh_voc_destroy(bh_voc* voc);
int  bh_voc_transform(bh_voc* voc, const uint8_t* desc, const int* n, int B, int stride, uint32_t* words, double* vals,
                      int* n_words, int mem);

#ifdef __cplusplus
}
#endif
#endif /* BOSLAM_HIP_H */
