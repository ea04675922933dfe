/*
 * eacham_hip.h — C-ABI of the MI355X (gfx950) matching + bundle-adjustment hot path.
 *
 * This is the drop-in boundary for the two hot spots of fatlipp/eacham:
 *
 *   matching : FeatureMatcherFlann::Match        modules/base/features/FeatureMatcherFlann.cpp:14-30
 *              IFeatureMatcher<T>::Match         modules/base/features/IFeatureMatcher.h:8-20
 *              pair loop + mutual cross-check    apps/sfm/main.cpp:84-147
 *   BA       : RefineBA                          modules/sfm/reconstruction/BundleAdjuster.h:13-17
 *                                                modules/sfm/reconstruction/BundleAdjuster.cpp:40-250
 *              OptimizerConfig                   modules/sfm/config/SfmConfig.h:15-22
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types, no exceptions across the ABI.
 *   - every call returns EACHAM_OK (0) or a negative error code; eacham_last_error(ctx)
 *     returns the message of the last failure on that context.
 *   - one context = one HIP device + one HIP stream; calls on one context are serialised by an
 *     internal mutex, so the reference's pattern of calling Match() concurrently on one shared
 *     matcher instance (apps/sfm/main.cpp:98-109) stays legal. Distinct contexts are independent.
 *   - pointers named *_dev are device pointers valid on the context's device; everything else is
 *     host memory. *_dev entry points enqueue on the context stream and do not synchronise;
 *     use eacham_ctx_sync().
 *   - match lists are emitted SORTED BY QUERY INDEX: the reference returns an unordered_map whose
 *     iteration order is unspecified; sorted is the canonical form used for bit-exact parity.
 */
#ifndef EACHAM_HIP_H
#define EACHAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EACHAM_OK 0
#define EACHAM_ERR_INVALID (-1)      /* bad argument */
#define EACHAM_ERR_HIP (-2)          /* a HIP runtime call failed */
#define EACHAM_ERR_CAPACITY (-3)     /* caller-provided output buffer too small */
#define EACHAM_ERR_UNSUPPORTED (-4)  /* e.g. descriptor dimension not supported by the kernel */
#define EACHAM_ERR_NOT_INTEGER (-5)  /* descriptors are not integer-valued in [0,255] (exact i8 path) */
#define EACHAM_ERR_NO_DEVICE (-6)    /* no HIP device / extension unusable: the product path never falls back to CPU */

typedef struct eacham_ctx eacham_ctx;

/* ---- context ------------------------------------------------------------------------------ */

/* Creates a context on HIP device `device_id`. Fails (no CPU fallback) if no device exists. */
int eacham_ctx_create(int device_id, eacham_ctx** out_ctx);
void eacham_ctx_destroy(eacham_ctx* ctx);
const char* eacham_last_error(const eacham_ctx* ctx);
/* Blocks until everything enqueued on the context stream has finished. */
int eacham_ctx_sync(eacham_ctx* ctx);
/* Diagnostic: what the search for a second stream on a hardware queue of its own decided at eacham_ctx_create (the work behind a
 * batch's distance sweep runs on it beside the next sweep). *attempt = which candidate was kept (0..3 probed, 4 = the last one,
 * kept unprobed; -1 = no search: EACHAM_STREAM2_PRIORITY), *lead_ms = how long before the end of a 40 us spin on the first stream
 * the empty kernel on it finished (above 0.010: the two do not share a queue; -1 = not measured). */
int eacham_ctx_stream2_info(const eacham_ctx* ctx, int* attempt, float* lead_ms);
/* Returns the hipStream_t of the context (as void*), so callers can order their own work. */
void* eacham_ctx_stream(eacham_ctx* ctx);
/* Library / build identification ("eacham_hip <version> gfx950"). */
const char* eacham_version(void);

/* ---- descriptor store (input of Match: Node::GetDescriptors(), modules/sfm/data/Node.h:136-139) */

/* Uploads the N x dim row-major fp32 descriptor matrix of frame `frame_id` (the layout of the
 * cv::Mat returned by FeatureExtractorSift::Extract, modules/base/features/FeatureExtractorSift.cpp:14-26)
 * and keeps it resident on the device in the kernel's fragment-major int8 layout.
 * Values must be integers in [0,255] (OpenCV SIFT descriptors are) —
 * otherwise EACHAM_ERR_NOT_INTEGER. dim must be a multiple of 16 and <= 256. n may be 0.
 * Re-uploading a frame id replaces it. */
int eacham_upload_descriptors(eacham_ctx* ctx, int frame_id, const float* rowmajor, int n, int dim);
/* Same, source already on the device (enqueued on the context stream; the integrality check is
 * reported by the next synchronising call). */
int eacham_upload_descriptors_dev(eacham_ctx* ctx, int frame_id, const float* rowmajor_dev, int n, int dim);
/* Float descriptors (SuperPoint / LightGlue style, modules/onnx/lightglue/feature/Types.h:11-14: 256-D
 * fp32, any values): kept resident as fp32 MFMA fragments and matched with
 *   d2 = max(fma(-2, a.b, |a|^2 + |b|^2), 0)   (fp32, a.b and the norms as k-ordered fma chains)
 * on the f32 matrix cores. Same Match / mutual-check semantics, ties -> lower index. All resident
 * frames must be of one kind (int8 via eacham_upload_descriptors, or fp32 via this call). dim <= 256. */
int eacham_upload_descriptors_f32(eacham_ctx* ctx, int frame_id, const float* rowmajor, int n, int dim);
/* Number of rows of a resident frame, or a negative error code. */
int eacham_frame_rows(eacham_ctx* ctx, int frame_id);
/* Drops all resident frames. */
int eacham_clear_descriptors(eacham_ctx* ctx);

/* ---- directed match: FeatureMatcherFlann::Match (FeatureMatcherFlann.cpp:14-30) -------------
 * For each row q of frame f1: the two nearest rows of frame f2 under L2 (exact; ties -> lower
 * train index first); keep q -> t0 iff (float)(d0 / d1) < ratio with d = sqrtf(squared L2), the
 * quotient promoted to double as in the reference (ratio = 0.8 there, FeatureMatcherFlann.cpp:23).
 * Output sorted by q. If frame f2 has fewer than 2 rows the result is empty (the reference would
 * read m[1] out of bounds). */
int eacham_match_pair(eacham_ctx* ctx, int f1, int f2, double ratio,
                      uint32_t* out_q, uint32_t* out_t, int cap, int* out_count);

/* Batched form: npairs ORDERED pairs {f1, f2}, each one FeatureMatcherFlann::Match(d[f1], d[f2]) call, in one
 * launch sequence — what the reference's std::for_each(par_unseq) + std::async issues concurrently on its shared
 * matcher (apps/sfm/main.cpp:98-109); the C++ adapter funnels concurrent Match() callers into this entry point.
 * No thresholds, no mutual check. Result in CSR form over the pairs: counts[p], offsets (npairs + 1), and for
 * k in [offsets[p], offsets[p+1]) the match q[k] -> t[k] of pair p, sorted by q. cap = capacity of out_q / out_t
 * (sum of the query frames' rows always suffices); *out_total = matches found (EACHAM_ERR_CAPACITY if > cap). */
int eacham_match_pairs_directed(eacham_ctx* ctx, const int32_t* pairs, int npairs, double ratio,
                                int32_t* counts, int64_t* offsets, uint32_t* out_q, uint32_t* out_t, int64_t cap,
                                int64_t* out_total);

/* ---- all-pairs match + mutual check: apps/sfm/main.cpp:84-147 -------------------------------
 * pairs = npairs x {f1, f2} (unordered pairs; both directions are evaluated from one distance
 * tile). For each pair: directed matches m12, m21 as above; if |m12| < min_dir or |m21| < min_dir
 * the pair is dropped (main.cpp:111, literal 30); mutual = {(q,t) in m12 : m21[t] == q}
 * (main.cpp:133-140); the pair becomes an edge iff |mutual| > min_mutual (main.cpp:142, literal 30).
 *
 * Result (CSR over pairs): counts[p] = |mutual| for edges, 0 otherwise; offsets[p] = prefix sum
 * (npairs+1 entries); (q[k], t[k]) for k in [offsets[p], offsets[p+1]) sorted by q, q indexing
 * frame pairs[2p], t indexing frame pairs[2p+1]  (= Graph::Connect(n1,n2,best12); the reverse
 * edge Connect(n2,n1,best21) is its inverse, main.cpp:144-145).
 * stats (optional, may be NULL): npairs x {|m12|, |m21|, |mutual|, edge?1:0}.
 * ratio must be in (0, 1] here (the reference uses 0.8): the mutual check identifies m21[t] == q by
 * "column t passes the ratio test and its minimum is d(q,t)", which relies on a passing column having
 * a unique minimum; EACHAM_ERR_INVALID otherwise. eacham_match_pair accepts any ratio. */
int eacham_match_all_pairs(eacham_ctx* ctx, const int32_t* pairs, int npairs, double ratio,
                           int min_dir, int min_mutual,
                           int32_t* counts, int64_t* offsets,
                           uint32_t* out_q, uint32_t* out_t, int64_t cap, int64_t* out_total,
                           int32_t* stats);

/* Device-resident, asynchronous form used by the benchmark and the multi-GPU shard driver.
 * pairs_dev: npairs x 2 int32. counts_dev: npairs int32. offsets_dev: npairs+1 int64.
 * edges_dev: edge_cap x {uint32 q, uint32 t}; entries beyond edge_cap are dropped and
 * *total_dev (int64, device) still holds the uncapped total. stats_dev: npairs x 4 int32 or NULL.  * A pair naming a frame that is not resident yields no match (count 0) and raises a sticky error that
 * the next eacham_ctx_sync returns as EACHAM_ERR_INVALID; it never reaches the kernels. */
int eacham_match_all_pairs_dev(eacham_ctx* ctx, const int32_t* pairs_dev, int npairs, double ratio,
                               int min_dir, int min_mutual,
                               int32_t* counts_dev, int64_t* offsets_dev,
                               uint32_t* edges_dev, int64_t edge_cap, int64_t* total_dev,
                               int32_t* stats_dev);

/* Debug getter (tests, bench.py's parity gate): how a job of `npairs` pairs of the frames resident NOW would be cut into
 * launches by eacham_match_all_pairs[_dev] (with_stats = 1: the full-column form that the `stats` argument selects). starts[b] =
 * index of the first pair of launch b (the first `cap` of them are written), *n_batches = launches, *n_slots = workspace copies
 * they rotate through (2: the work behind launch b runs on a second stream beside launch b + 1). Lets a parity test place its
 * oracle samples on both sides of every launch boundary of the job of apps/sfm/main.cpp:84-147. */
int eacham_match_debug_batches(eacham_ctx* ctx, int npairs, int with_stats, int32_t* starts, int cap, int* n_batches, int* n_slots);

/* ---- bundle adjustment: RefineBA (modules/sfm/reconstruction/BundleAdjuster.cpp:40-250) --------
 *
 * The caller (the C++ adapter in include/eacham/BundleAdjusterHip.hpp) performs the reference's
 * graph walk (window selection :123-162, landmark filter `status && observers >= 2` :84) and
 * hands over plain arrays; the library restates everything from the factor graph on: factors and
 * noise models (:57-121, :171-178), Levenberg-Marquardt with GTSAM's Ceres defaults (:182-216),
 * error evaluation (:218-219). Write-back (:221-249) is the inverse mapping done by the adapter.
 *
 * Variable blocks: camera i -> Pose3 camera->world (the inverse of cam_T_wc, :65), tangent
 * [omega, v]; landmark j -> Point3; one shared Cal3_S2 (fx, fy, s = 0, u0, v0) (:47-49).
 * All arithmetic is fp64.
 */
typedef struct eacham_ba_problem {
    int32_t n_cams;
    int32_t n_points;
    int32_t n_obs;
    int32_t ordering;                /* EACHAM_BA_ORDER_* below: elimination order of the reduced camera system;  */
                                     /* 0 (a zero-initialised problem) = chosen by the library's cost model       */
    const double* cam_T_wc;          /* n_cams x 16, row-major world->camera (Node::GetTransform())   */
    const int32_t* cam_fixed;        /* n_cams, Graph::IsFixed(id) (:69)                              */
    const double* points;            /* n_points x 3 (Map::Get(id3d), :102)                          */
    const int32_t* point_observers;  /* n_points, Map::GetObservers(id3d).size(): the GLOBAL count,  */
                                     /* also inside a local window (:109) -> prior sigma 1/obs       */
    const uint32_t* obs_cam;         /* n_obs, index into cams                                       */
    const uint32_t* obs_point;       /* n_obs, index into points                                     */
    const double* obs_uv;            /* n_obs x 2, keypoint in pixels (cv::Point2f widened, :93)     */
    double K[4];                     /* fx, fy, cx, cy = K(0,0), K(1,1), K(0,2), K(1,2) (:47-49)     */
} eacham_ba_problem;

/* Elimination order of the reduced camera system (the counterpart of GTSAM's COLAMD ordering, which the reference
 * gets through LevenbergMarquardtParams::SetCeresDefaults, BundleAdjuster.cpp:182-190). Every choice solves the same
 * system; they differ in the height of the elimination tree (the number of dependent launches) and in fill.
 * AUTO evaluates the candidates with a cost model and keeps the cheapest (eacham_amd/csrc/ba_plan.hpp). */
#define EACHAM_BA_ORDER_AUTO 0
#define EACHAM_BA_ORDER_NATURAL 1  /* the caller's camera order: a band for a sequence                                */
#define EACHAM_BA_ORDER_RCM 2      /* reverse Cuthill-McKee: minimal band, the tree is a path                        */
#define EACHAM_BA_ORDER_ND 3       /* nested dissection: independent subtrees are factorised in the same launch      */

#define EACHAM_BA_LM 0
#define EACHAM_BA_DOGLEG 1

/* OptimizerConfig (modules/sfm/config/SfmConfig.h:15-22), fields verbatim + the reference's literal. */
typedef struct eacham_ba_options {
    int32_t method;             /* "LM" -> EACHAM_BA_LM, "DogLeg" -> EACHAM_BA_DOGLEG               */
    int32_t max_iter;           /* maxIter                                                           */
    float max_tolerance;        /* maxTolerance: absoluteErrorTol = relativeErrorTol (:187-188)      */
    float delta;                /* DogLeg deltaInitial (:211)                                        */
    int32_t use_preconditioner; /* usePreconditioner (LM only): solve every damped system with PCG +   */
                                /* block-Jacobi at 1e-10 / 1e-10 (:192-200) instead of the direct solve */
    int32_t min_landmarks;      /* literal 50 (:166): fewer landmarks -> silently do nothing         */
    int32_t lm_factor_policy;   /* EACHAM_BA_LM_FACTOR_* below; 0 (zero-initialised options) = RESET */
    int32_t reserved;
} eacham_ba_options;

/* What LevenbergMarquardtState::decreaseLambda does to the growth factor after an ACCEPTED step
 * (GTSAM 4.1.1 gtsam/nonlinear/internal/LevenbergMarquardtState.h, not in the reference tree; the
 * reference selects it through SetCeresDefaults, BundleAdjuster.cpp:184-190: lambdaFactor = 2,
 * useFixedLambdaFactor = false). Two readings of that line exist (SURVEY.md Appendix A.4 note):
 *   RESET  : currentFactor = 2 * params.lambdaFactor  (= 4: Ceres' decrease_factor reset; the default)
 *   DOUBLE : currentFactor = 2 * currentFactor        (the factor never shrinks)
 * They give the same lambda schedule until the first rejected step that follows two accepted ones. */
#define EACHAM_BA_LM_FACTOR_RESET 0
#define EACHAM_BA_LM_FACTOR_DOUBLE 1

#define EACHAM_BA_DONE 0     /* optimised                                                            */
#define EACHAM_BA_SKIPPED 1  /* fewer than min_landmarks landmarks: inputs copied through (:166-169) */
#define EACHAM_BA_INDETERMINATE 2 /* DogLeg only: the Gauss-Newton system is not positive definite. GTSAM's
                                   * DoglegOptimizer throws IndeterminantLinearSystemException there and
                                   * RefineBA never reaches its write-back; the values returned are those of
                                   * the last completed iteration (the inputs if it was the first). */

/* One row per tryLambda() call of the LM loop (for parity tests and reporting). */
typedef struct eacham_ba_trace_row {
    double lambda;          /* lambda used for this try                                              */
    double new_error;       /* nonlinear error at the tentative values (inf if not evaluated)        */
    double lin_change;      /* oldLinearizedError - newLinearizedError                               */
    int32_t accepted;       /* 1 = step taken                                                        */
    int32_t outer;          /* outer iteration index (iterate() call)                                */
} eacham_ba_trace_row;

typedef struct eacham_ba_result {
    double* cam_T_wc;        /* out: n_cams x 16 world->camera (Node::SetTransform, :247)            */
    double* points;          /* out: n_points x 3 (Map::UpdatePoint, :239)                           */
    double K[4];             /* out: fx, fy, cx, cy (:224-227)                                       */
    double initial_error;    /* graph.error(initial)   (:218)                                        */
    double final_error;      /* graph.error(result)    (:219)                                        */
    double final_lambda;
    int32_t status;          /* EACHAM_BA_DONE / EACHAM_BA_SKIPPED                                   */
    int32_t outer_iterations; /* successful LM iterations (NonlinearOptimizer::iterations())         */
    int32_t inner_iterations; /* tryLambda() calls = linear solves                                   */
    int32_t trace_cap;       /* in: capacity of `trace` (may be 0)                                   */
    int32_t trace_len;       /* out: rows written                                                    */
    int32_t reserved;        /* out: PCG iterations in total when use_preconditioner is set, else 0   */
    eacham_ba_trace_row* trace; /* optional                                                          */
} eacham_ba_result;

/* Runs RefineBA's optimisation on the device. Host pointers in, host pointers out; all state stays
 * resident on the device during the LM loop, only scalars (errors, lambda decisions) cross PCIe
 * per inner iteration. Returns EACHAM_OK also when the problem is skipped (see result->status).
 * Method DogLeg (BundleAdjuster.cpp:204-214) runs GTSAM's DoglegOptimizer control flow (mode
 * ONE_STEP_PER_ITERATION, deltaInitial = options->delta) on the same linearisation and Gauss-Newton solve;
 * its trace rows carry the trust-region radius in `lambda` and the model decrease in `lin_change`. */
int eacham_ba_solve(eacham_ctx* ctx, const eacham_ba_problem* problem, const eacham_ba_options* options,
                    eacham_ba_result* result);

/* Split form for callers that solve the same window repeatedly (the benchmark): prepare uploads the
 * problem and builds the device-side structure (observations grouped by landmark and by camera,
 * the camera-pair lists of the Schur complement); run restarts from the uploaded initial values,
 * so the timed region holds only device work and the per-try scalar read-backs. */
typedef struct eacham_ba_handle eacham_ba_handle;
int eacham_ba_prepare(eacham_ctx* ctx, const eacham_ba_problem* problem, eacham_ba_handle** out_handle);
int eacham_ba_run(eacham_ctx* ctx, eacham_ba_handle* handle, const eacham_ba_options* options,
                  eacham_ba_result* result);
void eacham_ba_release(eacham_ctx* ctx, eacham_ba_handle* handle);

/* What the analysis of the reduced camera system decided for a prepared problem (reporting: the benchmark's BA line
 * carries it): panels of 64 columns, tiles of the symbolic factor, height of the elimination tree = number of
 * dependent factorisation launches, the ordering used (EACHAM_BA_ORDER_NATURAL / _RCM / _ND), rank-64 tile updates of
 * one factorisation, the cost model's estimate for factorisation + back-substitution in microseconds, and where the
 * host time of eacham_ba_prepare went. */
typedef struct eacham_ba_plan_info {
    int32_t n_panels, n_tiles, n_levels, ordering, nd_leaf, reserved;
    int64_t tile_updates;
    double est_us;
    double prepare_us[3];  /* host time of eacham_ba_prepare: observation / pair structures | ordering + symbolic
                            * analysis | arena + upload + synchronisation */
} eacham_ba_plan_info;
int eacham_ba_get_plan_info(eacham_ctx* ctx, const eacham_ba_handle* handle, eacham_ba_plan_info* out);

/* Test/diagnostic entry point: one array of a prepared problem's device-side structure, as built by eacham_ba_prepare
 * (0 lm_ptr, 1 cam_ptr, 2 cam_obs, 3 obs_pos, 4 obs_cam, 5 obs_lm, 6 obs_uv, 7 cam_uv, 8 cam_lm, 9 pos_cam, 10 cam_chunks,
 * 11 cam_chunk_ptr, 12 blocks, 13 pair_chunks, 14 pair_entries, 15 pose0, 16 pt0, 17 lmprior, 18 K0, 19 fixed): the tests
 * hold the structure built by device sorts and scans against the one built by host loops, array by array. */
int eacham_ba_debug_structure(eacham_ctx* ctx, const eacham_ba_handle* handle, int which, void* out, int64_t cap_bytes,
                              int64_t* out_bytes);

/* Test/diagnostic entry point: linearises at the problem's initial values and returns the reduced
 * camera system of one damped Gauss-Newton step: S (n x n, row-major, n = 6*n_cams + 5, cameras
 * first, then K), its right-hand side g (n), the step delta for cameras+K (n) and for the points
 * (3*n_points), the nonlinear error at the linearisation point, and the linearised cost change. */
int eacham_ba_debug_step(eacham_ctx* ctx, const eacham_ba_problem* problem, double lambda,
                         double* S, double* g, double* delta_cams, double* delta_points,
                         double* error, double* lin_change);

/* ---- triangulation -------------------------------------------------
 * Batch form of TriangulateFrame's per-point call of TriangulatePointRansac
 * (/root/reference/modules/sfm/reconstruction/Triangulator.cpp:96-186 and :248-275; call sites
 * apps/sfm/main.cpp:203-210 with config.maxReprError / config.minTriAngle in radians).
 *   transforms   n_frames x 16 row-major world->camera matrices (Node::GetTransform)
 *   track_ptr    n_tracks+1 CSR offsets into the observation arrays (track = one candidate point, any number
 *                of observations: the reference has no limit and a long sequence can see a point a hundred times)
 *   obs_frame    row of `transforms` per observation;  obs_uv  pixel (x, y) per observation
 *   K            fx, fy, cx, cy
 * Outputs (host): points n_tracks x 3 = the `point3d` the reference leaves behind (the LAST pair's
 * triangulation); status[t] bit 0 = TriangulatePointRansac's return value, bit 1 = mask non-empty
 * and every observation an inlier — TriangulateFrame adds the point iff status[t] == 3 (:270-275);
 * masks = the reference's `inliers` vector per observation.
 * Tracks with fewer than 2 observations get status 0 and a zero point, as :104-107 does. */
int eacham_triangulate_tracks(eacham_ctx* ctx, const double* transforms, int n_frames, int n_tracks,
                              const int32_t* track_ptr, const uint32_t* obs_frame, const double* obs_uv,
                              const double* K, float max_repr_error, float min_tri_angle, double* points,
                              int32_t* status, uint8_t* masks);

/* Reprojection error of n (frame, map point, pixel) items: CalcReprojectionError(uv,
 * transformPoint3d(point, T[frame]), K) as a float (ProjectionHelper.cpp:32-38), the re-observation
 * gate of TriangulateFrame (Triangulator.cpp:222-236). */
int eacham_reprojection_errors(eacham_ctx* ctx, const double* transforms, int n_frames, int n,
                               const uint32_t* frame, const double* points, const double* uv, const double* K,
                               float* err);

/* Two-view structure for candidate relative poses (first part of SURVEY.md §8(f) rank 3): the per-match
 * loops of RecoverPoseTwoView (/root/reference/modules/sfm/reconstruction/ReconstructionManager.cpp:118-143
 * for the solutions of cv::decomposeHomographyMat, :162-186 for the pose of cv::recoverPose). Camera 1
 * is the identity; transforms[k] (row-major 4x4) maps camera-1 to camera-2 coordinates. For every
 * (transform k, match i): points[k][i] = TriangulatePoint(p1, p2, K, transform); keep[k][i] = 1 iff
 * z > 0 && reprojection error in camera 1 (rounded to float) < max_repr_error && the triangulation
 * angle passes: > min_tri_angle when angle_strict != 0 (the homography branch, :132), >= otherwise
 * (:170). counts[k] = kept matches (the reference picks the first k with the strictly largest count
 * and needs more than 20, :139-150). The robust E/H estimation itself stays with the caller. */
int eacham_two_view_points(eacham_ctx* ctx, int n_matches, const double* uv1, const double* uv2, const double* K,
                           int n_transforms, const double* transforms, float max_repr_error, float min_tri_angle,
                           int angle_strict, double* points, uint8_t* keep, int32_t* counts);

/* ---- hypothesis scoring for the robust estimators (rest of SURVEY.md §8(f) rank 3) -----------------------------
 * The part of cv::findEssentialMat / cv::findHomography (LMEDS) and cv::solvePnPRansac (EPNP, 10 000 iterations,
 * 4 px) — /root/reference/modules/sfm/reconstruction/ReconstructionManager.cpp:57-61, :75, :227-228 — that is
 * data-parallel over (hypothesis, correspondence): the error every candidate model assigns to every point, as
 * OpenCV 4.5.5's estimator callbacks define it (EMEstimatorCallback / HomographyEstimatorCallback /
 * PnPRansacCallback ::computeError), its inlier count under a
 * threshold (RANSAC: err <= threshold, the caller passes the squared pixel threshold as OpenCV does) and its median
 * (LMedS). One call scores all candidates; drawing the minimal samples and solving them stays with the caller.
 *   kind ESSENTIAL   a, b = n x 2 image points of view 1 / view 2; models = n_models x 9 row-major E. With K
 *                    (fx, fy, cx, cy) the points are normalised first, (u - cx) / fx, as findEssentialMat does;
 *                    K = NULL takes them as already normalised. err = Sampson distance (x2' E x1)^2 / (...), float.
 *   kind HOMOGRAPHY  a, b = n x 2; models = n_models x 9 row-major H (H[8] is taken as 1, as OpenCV's callback does);
 *                    err = |H a - b|^2 in float arithmetic. K unused.
 *   kind PNP         a = n x 3 object points, b = n x 2 image points; models = n_models x 12 = R (row-major) | t
 *                    (cv::Rodrigues of the candidate rvec is the caller's); K required; err = squared reprojection
 *                    error, projection in double, difference in float.
 * Outputs (each optional): errors n_models x n_points, inlier_counts n_models, medians n_models (sorted middle, or
 * the mean of the two middle values for even n; NaN for n = 0). */
#define EACHAM_SCORE_ESSENTIAL 0
#define EACHAM_SCORE_HOMOGRAPHY 1
#define EACHAM_SCORE_PNP 2
int eacham_score_hypotheses(eacham_ctx* ctx, int kind, int n_points, const double* a, const double* b, int n_models,
                            const double* models, const double* K, float threshold, float* errors,
                            int32_t* inlier_counts, float* medians);

/* ---- minimal solvers of the robust estimators --------------------------------------------
 * The model-generating half of cv::findHomography / cv::findEssentialMat
 * (/root/reference/modules/sfm/reconstruction/ReconstructionManager.cpp:75, :57-61): every minimal sample of the
 * caller's list -> its model(s), one launch for the whole list (the reference asks for 100 / 1000 iterations).
 * OpenCV draws the samples from its own RNG inside the estimator; here the sample INDICES are an argument, so what is
 * defined — and tested against the CPU restatement bit for bit — is "these correspondences -> these models".
 *   kind HOMOGRAPHY4  a, b = n_points x 2 (source, destination); 4 indices per sample; 1 model = 9 doubles row-major,
 *                     H[8] = 1 (HomographyEstimatorCallback::runKernel). K unused.
 *   kind ESSENTIAL5   a, b = n_points x 2 pixels of view 1 / view 2, K = fx fy cx cy (NULL: already normalised);
 *                     5 indices per sample; up to 10 models of 9 doubles (unit Frobenius norm, x2' E x1 = 0), the slots
 *                     beyond n_models[s] are zero (Nister's five-point algorithm, EMEstimatorCallback::runKernel).
 * models: n_samples x max_models x 9 with max_models = 1 / 10; n_models: n_samples (0 = degenerate sample).
 * Pair it with eacham_score_hypotheses (all models against all correspondences) for the RANSAC / LMedS choice. */
#define EACHAM_SOLVE_HOMOGRAPHY4 0
#define EACHAM_SOLVE_ESSENTIAL5 1
int eacham_solve_minimal(eacham_ctx* ctx, int kind, int n_points, const double* a, const double* b, const double* K,
                         int n_samples, const int32_t* sample_idx, double* models, int32_t* n_models);

/* EPnP, the solver inside cv::solvePnPRansac(..., 10000, 4.0f, 0.999f, inliers, cv::SOLVEPNP_EPNP)
 * (/root/reference/modules/sfm/reconstruction/ReconstructionManager.cpp:227-228): OpenCV runs it on 5-point samples inside
 * the RANSAC loop and once more on the inliers of the winning model. Here every row of sample_idx (sample_size >= 5 indices
 * into the n_points object / image points) is one EPnP problem, one launch for the whole list: 10 000 rows of 5 for the
 * loop, one row of all inliers for the refit. models: n_samples x 12 = R (row-major) | t with x_cam = R X + t — the layout
 * eacham_score_hypotheses(kind PNP) scores; n_models[s] = 1, or 0 (model zeroed) for a degenerate sample (collinear or
 * coincident points). A COPLANAR sample (smallest spread of its object points <= 1e-12 of the largest) gets a pose too, as
 * cv::solvePnPRansac returns one for a planar target: the three-control-point form of the EPnP paper (oracle/solve_oracle.c's
 * header states it). K = fx fy cx cy, no distortion (the reference passes zeros).
 * Bit-identical with oracle/solve_oracle.c's oracle_solve_pnp. */
int eacham_solve_pnp(eacham_ctx* ctx, int n_points, const double* object_points, const double* image_points, const double* K,
                     int sample_size, int n_samples, const int32_t* sample_idx, double* models, int32_t* n_models);

/* ---- view-graph query on the CSR match graph --------------------------
 * Graph::GetBestPairForValid (/root/reference/modules/sfm/data/Graph.h:59-106) evaluated directly on the
 * wire format of eacham_match_all_pairs: pair p with counts[p] > 0 is the factor f1 -> f2 with matches
 * q -> t and the factor f2 -> f1 with t -> q (Graph::Connect both ways, apps/sfm/main.cpp:144-145).
 *   valid[f]        Node::IsValid();  excluded[f] (may be NULL) = the `excluded` set
 *   kp_offsets      n_frames+1 offsets into kp_has3d;  kp_has3d[kp_offsets[f] + k] = 1 iff
 *                   node f HasPoint3d(k) && !IsPoint3dTwoView(k)
 * best = {id, id2, points3dCount} of the reference's tuple ({UINT_MAX, UINT_MAX, 0} when no valid node
 * has a not-yet-valid, not-excluded neighbour). A candidate replaces the best unless
 * `bestScore > count`, so among equal counts the last visited wins; nodes are visited in ascending id
 * as in the reference and neighbours in ascending id (the reference's unordered_map has no order).
 * edge_counts (optional, 2*npairs): points3dCount of f1 -> f2 and of f2 -> f1 for every pair. */
int eacham_graph_best_pair(eacham_ctx* ctx, int n_frames, const int32_t* pairs, int npairs, const int32_t* counts,
                           const int64_t* offsets, const uint32_t* q, const uint32_t* t, const uint8_t* valid,
                           const uint8_t* excluded, const int64_t* kp_offsets, const uint8_t* kp_has3d,
                           uint32_t* edge_counts, uint32_t* best);

/* The resident form of the same query, for the incremental loop of apps/sfm/main.cpp:188-214, which asks it after every frame
 * it adds: the CSR match graph is validated and uploaded once (eacham_graph_create: same arguments as above minus the per-frame
 * state; kp_offsets gives every frame's keypoint count), the per-frame state — Node::IsValid() and, per keypoint,
 * HasPoint3d(k) && !IsPoint3dTwoView(k) — is set frame by frame as the loop changes it (eacham_graph_set_frame: the frame it
 * just posed and triangulated and that frame's factor neighbours are the only ones that change between two queries), and
 * eacham_graph_query(excluded set) is two small kernels. Same result as eacham_graph_best_pair on the same state. The graph
 * belongs to its context (calls are serialised with the context's others); destroy it before the context. */
typedef struct eacham_graph eacham_graph;
int eacham_graph_create(eacham_ctx* ctx, int n_frames, const int32_t* pairs, int npairs, const int32_t* counts, const int64_t* offsets,
                        const uint32_t* q, const uint32_t* t, const int64_t* kp_offsets, eacham_graph** out_graph);
void eacham_graph_destroy(eacham_graph* graph);
int eacham_graph_set_frame(eacham_graph* graph, int frame, int valid, const uint8_t* has3d, int n_keypoints);
/* Several frames in one call — what the loop does after every frame it adds: the frame and its ~20 factor neighbours
 * (apps/sfm/main.cpp:203-209: TriangulateFrame's SetPoint3d reaches exactly those). frames[i], valid[i], and the frames' flag
 * arrays one behind the other in has3d (frame i's start at has3d_offsets[i], has3d_offsets[n] = total; every frame's full
 * keypoint count). One copy and one kernel instead of two small copies per frame. */
int eacham_graph_set_frames(eacham_graph* graph, int n, const int32_t* frames, const uint8_t* valid, const uint8_t* has3d,
                            const int64_t* has3d_offsets);
int eacham_graph_query(eacham_graph* graph, const int32_t* excluded_frames, int n_excluded, uint32_t* best);

/* ---- kernel timing (HIP events on the context stream; used for roofline reporting) ---------- */

#define EACHAM_KERNEL_MATCH_TILE 0     /* all-pairs int8 MFMA distance + fused row/col top-2      */
#define EACHAM_KERNEL_MATCH_FINALIZE 1 /* partial merge + ratio + mutual check + compaction       */
#define EACHAM_KERNEL_BA_LINEARIZE 2
#define EACHAM_KERNEL_BA_SCHUR 3
#define EACHAM_KERNEL_BA_SOLVE 4
#define EACHAM_KERNEL_BA_ERROR 5
#define EACHAM_KERNEL_TRIANGULATE 6     /* pair DLT + scoring and the per-track selection          */
#define EACHAM_KERNEL_SCORE 7           /* hypothesis scoring (errors, inlier counts, medians)      */
#define EACHAM_KERNEL_COUNT 8

/* Enables (1) / disables (0) per-launch HIP-event timing of the kernels above. */
int eacham_profile_enable(eacham_ctx* ctx, int on);
int eacham_profile_reset(eacham_ctx* ctx);
/* Synchronises, then returns launches and total milliseconds recorded for `kernel_id`. */
int eacham_profile_get(eacham_ctx* ctx, int kernel_id, int64_t* launches, double* total_ms);

/* ---- multi-GPU sharding of the pair loop (host-side helpers, no device needed) ---------------------
 * One process per GPU, one context each: every rank orders the pair list the same
 * way, takes its contiguous shard, runs eacham_match_all_pairs(_dev) on it and all-gathers counts + edges with
 * RCCL (bench.py / eacham_amd/shard.py show the torch.distributed form; the single-process form is
 * eacham_match_all_pairs_sharded below).
 * Replaces the std::for_each(par_unseq) over pairs of apps/sfm/main.cpp:98-109 across devices. */
/* Sorts [npairs][2] in place by train frame (second column), then query frame: consecutive workgroups stream
 * the same train frame, which keeps it in the XCD's L2. */
int eacham_order_pairs(int32_t* pairs, int npairs);
/* Contiguous shard of rank `rank` of `world`: [*begin, *end), sizes differ by at most one. */
int eacham_shard_bounds(int npairs, int world, int rank, int* begin, int* end);
/* Work-balanced contiguous cut of the ordered list: bounds[r] = first pair of shard r, bounds[world] = npairs. weights[k] =
 * cost of ordered pair k — the matcher's is the pair's distance matrix, rows(f1) * rows(f2), which is what makes shards of
 * ragged frames take equal time (equal pair COUNTS do not: apps/sfm/main.cpp:98-109 hands every pair to whichever thread is
 * free, a static cut has to weigh them). Shard r starts at the smallest k whose prefix weight P[k] satisfies
 * P[k] * world >= r * P[npairs]. weights == NULL (or all zero): the equal-count cut of eacham_shard_bounds. */
int eacham_shard_bounds_weighted(int npairs, int world, const int64_t* weights, int32_t* bounds);

/* ---- single-process multi-GPU form of the pair loop -----------------------
 * The reference app is ONE C++ process (apps/sfm/main.cpp:31) whose pair loop (:84-147) fans out over host threads
 * (:98-109). A communicator owns one context and one host thread per device and an RCCL communicator over them
 * (ncclCommInitAll; RCCL is loaded at run time, EACHAM_ERR_UNSUPPORTED if it cannot be): descriptors are replicated on
 * every device, the pair list is ordered by train frame and cut into contiguous shards (eacham_order_pairs /
 * eacham_shard_bounds), every device matches its shard, and the match graph is assembled on every device by
 * ncclAllGather of the per-pair counts and of the edge lists (padded to the largest shard) over xGMI, enqueued on each
 * context's stream behind its matching. devices = NULL means devices 0 .. ndev-1. */
typedef struct eacham_comm eacham_comm;
int eacham_comm_init(int ndev, const int* devices, eacham_comm** out_comm);
void eacham_comm_destroy(eacham_comm* comm);
const char* eacham_comm_last_error(const eacham_comm* comm);
int eacham_comm_size(const eacham_comm* comm);
/* The context of rank `rank` (owned by the communicator): for uploads of other kinds, profiling, BA replicas. */
eacham_ctx* eacham_comm_ctx(eacham_comm* comm, int rank);
/* eacham_upload_descriptors on every device of the communicator. */
int eacham_comm_upload_descriptors(eacham_comm* comm, int frame_id, const float* rowmajor, int n, int dim);
/* eacham_match_all_pairs over all devices: same arguments, same result (CSR over the caller's pair order, sorted by q). */
int eacham_match_all_pairs_sharded(eacham_comm* comm, const int32_t* pairs, int npairs, double ratio,
                                   int min_dir, int min_mutual, int32_t* counts, int64_t* offsets,
                                   uint32_t* out_q, uint32_t* out_t, int64_t cap, int64_t* out_total);
/* The two halves of eacham_match_all_pairs_sharded. run: order the pairs, cut them into shards (balance != 0: by work,
 * eacham_shard_bounds_weighted with rows(f1) * rows(f2); 0: by count), match every shard on its device and all-gather; the
 * gathered graph stays resident on EVERY device, *out_total = matches in it. fetch: read device 0's copy back and assemble
 * the CSR in the caller's pair order (any number of times until the next run). */
int eacham_comm_match_run(eacham_comm* comm, const int32_t* pairs, int npairs, double ratio, int min_dir, int min_mutual,
                          int balance, int64_t* out_total);
int eacham_comm_match_fetch(eacham_comm* comm, int32_t* counts, int64_t* offsets, uint32_t* out_q, uint32_t* out_t,
                            int64_t cap, int64_t* out_total);
/* Host logic of the gather's send buffers (no device needed; exported for the CPU tests): ncclAllGather sends the same
 * number of elements from every rank, so the edge region of EVERY rank's buffer holds the largest shard bound. */
int eacham_comm_edge_region(int world, const int64_t* shard_bound, int64_t* region);
/* Host-side assembly of gathered shards (what eacham_match_all_pairs_sharded does after its all-gather; also for callers
 * that run one process per GPU and gather with their own collective): g_counts = world x shard_cap per-pair counts,
 * g_edges = world x edge_cap x {q, t}; shard r holds pairs [eacham_shard_bounds(npairs, world, r)) of the ORDERED list,
 * sorted_index[k] = position of ordered pair k in the caller's list (NULL = identity). */
int eacham_assemble_match_graph(const int32_t* g_counts, const uint32_t* g_edges, int npairs, int world, int shard_cap,
                                int64_t edge_cap, const int32_t* sorted_index, int32_t* counts, int64_t* offsets,
                                uint32_t* out_q, uint32_t* out_t, int64_t cap, int64_t* out_total);

/* Same with explicit shard boundaries (bounds[world + 1] over the ordered list, as eacham_shard_bounds_weighted returns
 * them; NULL = the equal-count cut). */
int eacham_assemble_match_graph_bounds(const int32_t* g_counts, const uint32_t* g_edges, int npairs, int world, int shard_cap,
                                       int64_t edge_cap, const int32_t* bounds, const int32_t* sorted_index, int32_t* counts,
                                       int64_t* offsets, uint32_t* out_q, uint32_t* out_t, int64_t cap, int64_t* out_total);

#ifdef __cplusplus
}
#endif
#endif /* EACHAM_HIP_H */
