/*
 * rslam.h -- C ABI of the MI355X-native 1-point-RANSAC EKF update hot path.
 *
 * Drop-in boundary for plumewind/ransac_slam.  The reference exposes no plugin
 * or FFI interface for this path: it is reached by five direct C++ member
 * calls from System::TrackRunning (src/System.cpp:117,120,123,126,129) that
 * communicate only through ExtendKF's public members
 * (include/ransac_slam/ExtendKF.h:154-169).  The entry points below are what a
 * ~40-line adapter replacing the bodies of those five calls binds to (see
 * INTEGRATION.md).  All matrices are FP64, column-major (Eigen default).
 *
 * Conventions
 *   - every function returns RSLAM_OK (0) or a negative RSLAM_ERR_* code and
 *     never aborts the host process (the reference exit()s / Eigen-asserts);
 *   - pointers are borrowed for the duration of the call, caller keeps
 *     ownership; "host" / "device" says where the memory must live;
 *   - a context is not thread-safe; different contexts are independent;
 *   - there is NO CPU fallback: rslam_create fails with RSLAM_ERR_NO_DEVICE
 *     when no HIP device is usable.
 */
#ifndef RSLAM_H
#define RSLAM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSLAM_OK                  0
#define RSLAM_ERR_ARG            -1  /* null pointer / size out of range              */
#define RSLAM_ERR_NO_DEVICE      -2  /* no usable HIP device (no CPU fallback exists)  */
#define RSLAM_ERR_HIP            -3  /* a HIP runtime call failed                      */
#define RSLAM_ERR_STATE          -4  /* call order violated (e.g. update before predict) */
#define RSLAM_ERR_REF_ASSERT     -5  /* input on which the reference hits an Eigen assert
                                        (compat mode, Tracking.cpp:498, m_euc != m_id)  */
#define RSLAM_ERR_NOT_SPD        -6  /* innovation covariance not positive definite    */
#define RSLAM_ERR_IC_NOT_VISIBLE -7  /* ic[i] set for a feature predicted not visible  */
#define RSLAM_ERR_COMM           -8  /* RCCL not loadable or a collective failed        */

#define RSLAM_FEAT_INVERSE_DEPTH 0   /* "inversedepth": 6 state entries (ExtendKF.cpp:71) */
#define RSLAM_FEAT_CARTESIAN     1   /* "cartesian":    3 state entries (ExtendKF.cpp:80) */

/* CamParam, include/ransac_slam/System.h:69-82, filled at src/System.cpp:34-58.
 * f is the reference's cam.f (= YAML Camera.fps, focal length in the unit of dx). */
typedef struct rslam_camera {
    double k1, k2;      /* radial distortion                       */
    double Cx, Cy;      /* principal point, pixels                 */
    double f;           /* focal length                            */
    double dx, dy;      /* pixel pitch                             */
    int32_t nRows, nCols;
} rslam_camera;

#define RSLAM_PIN_HOST_COV 1
typedef struct rslam_config {
    rslam_camera cam;
    double  sigma_z;     /* RANSAC pixel threshold = std_z, Tracking.cpp:356            */
    double  p_success;   /* 0.99, Tracking.cpp:354                                      */
    int32_t n_hyp_init;  /* 1000, Tracking.cpp:357                                      */
    double  chi2_gate;   /* 5.9915, Tracking.cpp:576                                    */
    int32_t compat;      /* 1 = reproduce reference quirks Q1 (Tracking.cpp:448),
                            Q2 (:498), Q6 (ExtendKF.cpp:627), Q7 (Tracking.cpp:589);
                            0 = corrected arithmetic                                    */
    int32_t adaptive;    /* 1 = replay the adaptive termination of Tracking.cpp:531-537;
                            0 = evaluate exactly n_draws hypotheses (benchmark)         */
    int32_t dedup;       /* 1 = score each distinct hypothesised feature once and map
                            supports back (identical results, <= m hypotheses scored)   */
    int32_t reserved;    /* bit 0 (RSLAM_PIN_HOST_COV): page-lock the caller's covariance buffers.  The drop-in API
                            (rslam_predict: P_pred, rslam_ransac_update / rslam_fetch_cov: P_new) then registers them with
                            hipHostRegister on first use and keeps the registration while the same pointer comes back -- the
                            reference reuses ExtendKF's members p_k_km1 / p_k_k frame after frame (ExtendKF.h:154-169) -- so
                            the two 26 MB transfers of a frame run at the PCIe rate.  The caller promises that such a buffer
                            stays allocated until rslam_unpin_host_buffers has been called or the context is destroyed (a
                            buffer freed and re-allocated at the same address would still look registered); a change of the
                            state dimension drops every registration by itself (the reference re-allocates both matrices
                            when Map::map_management resizes the state); off by default for that reason */
} rslam_config;

/* State-vector layout: x = [r(3) q(4) v(3) w(3) | feature 0 | feature 1 | ...]
 * (ExtendKF.cpp:59-61,69-88).  offset[i] is the state index of feature i. */
typedef struct rslam_layout {
    int32_t n;               /* state dimension = 13 + 6*#id + 3*#cartesian */
    int32_t L;               /* number of features                          */
    const uint8_t* type;     /* L entries, RSLAM_FEAT_*                     */
    const int32_t* offset;   /* L entries                                   */
} rslam_layout;

/* per-stage device times of the last frame, microseconds (hipEvent) */
typedef struct rslam_stage_times {
    double predict_us;        /* K1  h, Jacobians, S_i                            */
    double pht_us;            /* K2+K3 P*H^T for matched features, innovations     */
    double score_us;          /* K4  hypothesis scoring                            */
    double select_us;         /* K5  consensus replay + winner's inlier set        */
    double update_li_us;      /* K6-K11 low-innovation update (all kernels)        */
    double rescue_us;         /* K12 rescue gate                                   */
    double update_hi_us;      /* high-innovation update (all kernels)              */
    double factor_li_us;      /* K8 blocked Cholesky sweep, LI pass                */
    double rank_update_li_us; /* K10 covariance rank-r update kernel, LI pass      */
    double factor_hi_us;      /* K8, HI pass                                       */
    double rank_update_hi_us; /* K10, HI pass                                      */
    double total_us;
} rslam_stage_times;

typedef struct rslam_ctx rslam_ctx;

int rslam_create (const rslam_config* cfg, int device, rslam_ctx** out);
int rslam_destroy(rslam_ctx* ctx);
const char* rslam_error_string(int code);
const char* rslam_version(void);

/* ------------------------------------------------------------------ *
 *  Drop-in (host-pointer) API.  P round-trips over PCIe.             *
 * ------------------------------------------------------------------ */

/* Segment 1: replaces Tracking::search_IC_matches lines 35-44
 * (predict_camera_measurements, calculate_derivatives, S_i = H_i P H_i^T + R_i).
 * x_pred = x_k_km1 (n), P_pred = p_k_km1 (n*n, ld = n), both host; both NULL = use the
 * prior that rslam_ekf_prediction left resident (layout may then be NULL too).
 * Outputs (host): h (L*2, row i = feature i), visible (L), S (L*4, 2x2 col-major).
 * For features that are not visible h and S entries are left untouched. */
int rslam_predict(rslam_ctx* ctx, const rslam_layout* layout,
                  const double* x_pred, const double* P_pred,
                  double* h, uint8_t* visible, double* S);

/* Segment 2: replaces System.cpp:120-129 (ransac_hypotheses,
 * ekf_update_li_inliers, rescue_hi_inliers, ekf_update_hi_inliers).
 * z (L*2) measured pixel of feature i (read only where ic[i]); ic (L) =
 * individually_compatible; draws[n_draws] in [0,1) replace the reference's
 * rand() stream (ExtendKF.cpp:230, one draw per hypothesis, Tracking.cpp:412).
 * Outputs (host): x_new = x_k_k (n); P_new = p_k_k (n*n) or NULL to keep it
 * device-resident (rslam_fetch_cov); li/hi (L) inlier flags; scalars. */
int rslam_ransac_update(rslam_ctx* ctx, const double* z, const uint8_t* ic,
                        const double* draws, int32_t n_draws,
                        double* x_new, double* P_new,
                        uint8_t* li, uint8_t* hi,
                        int32_t* best_hyp, int32_t* best_support,
                        int32_t* hyps_evaluated);

/* EKF prediction on the device (the step just before the hot path):
 * replaces ExtendKF::ekf_prediction (src/ExtendKF.cpp:333-388) for the "constant_velocity"
 * filter the reference instantiates (src/System.cpp:63).  It turns the resident posterior --
 * left by rslam_ransac_update, or uploaded with rslam_set_posterior after Map::map_management
 * edited x_k_k / p_k_k on the host -- into the resident prior x_k_km1 / p_k_km1; a following
 * rslam_predict(ctx, layout, NULL, NULL, ...) then runs without any upload.  delta_t = 1 and
 * std_a = Sigma.a, std_alpha = Sigma.alpha in the reference (ExtendKF.cpp:336,347-348). */
int rslam_set_posterior (rslam_ctx* ctx, const rslam_layout* layout, const double* x_kk, const double* P_kk);
int rslam_ekf_prediction(rslam_ctx* ctx, double delta_t, double std_a, double std_alpha);
int rslam_fetch_prior   (rslam_ctx* ctx, double* x_pred /* host n, may be NULL */, double* P_pred /* host n*n, may be NULL */);

/* Tracking::matching (Tracking.cpp:279-351): the NCC search between the two segments,
 * on the h / S that rslam_predict left on the device.  image: cam.nRows * cam.nCols uint8, row-major (the
 * cv::Mat of the frame); patches: NULL = the ones rslam_predict_patches left on the device, else L * 169
 * doubles, feature f's predicted 13 x 13 patch
 * (features_info[f].patch_when_matching, column-major, written by pred_patch_fc at Tracking.cpp:277;
 * ignored where the feature was not predicted).  Outputs (host): z (L*2) = (column, row) of the best
 * candidate where ic[f] = 1; ic (L) = individually_compatible; corr (L, may be NULL) = best normalised
 * correlation (-2 when the feature was not searched or had no candidate).  The correlation is
 * Converter::corrcoef_opencv's (Converter.cpp:188-209: patches through float32, accumulation in double),
 * threshold 0.80 and gate 5.9915 as at Tracking.cpp:281-283; candidates are visited column by column and
 * the first maximum wins, as Eigen's maxCoeff does (:342). */
int rslam_match(rslam_ctx* ctx, const uint8_t* image, const double* patches, double* z, uint8_t* ic, double* corr);

/* Tracking::pred_patch_fc (Tracking.cpp:164-278): the 13 x 13 patch every predicted
 * feature is expected to show, warped from its initialisation record.  The records are what
 * Map::initialize_a_features stores (Map.cpp:286-292) and live in a device-side feature store:
 *   rslam_set_feature_records    all L features at once: uv (L*2) = uv_when_initialized, R_wc (L*9, column-major)
 *                                = R_wc_when_initialized, r_wc (L*3) = r_wc_when_initialized, patches (L*1681) =
 *                                patch_when_initialized (41 x 41, column-major as Eigen stores it)
 *   rslam_append_feature_record  one more feature (after rslam_map_add_feature); rslam_map_delete_feature drops
 *                                the record of the feature it removes
 *   rslam_predict_patches        after rslam_predict: warps every feature that has a prediction, on x_k_km1 and h
 *                                resident on the device.  patches (host L*169, column-major 13 x 13 each, may be
 *                                NULL) = patch_when_matching; status (L, may be NULL): 1 warped, 0 zero patch
 *                                because h is within half a patch of the image border (Tracking.cpp:174-175,276),
 *                                2 no prediction, -1 the reference's meshgrid would not be 13 x 13.
 * The patches stay on the device: rslam_match(ctx, image, NULL, ...) searches with them.
 * compat = 1 keeps two quirks of the reference: the remap coordinates are one pixel off (MATLAB indices at
 * Tracking.cpp:263-264) and a Cartesian feature is warped about the world point of the last inverse-depth
 * feature before it (Tracking.cpp:52-61); compat = 0 corrects both.  cv::remap is restated as OpenCV
 * documents it for CV_32F sources: 1/32-pixel coordinates, float weights, BORDER_CONSTANT 0. */
int rslam_set_feature_records  (rslam_ctx* ctx, int32_t L, const double* uv, const double* R_wc, const double* r_wc,
                                const double* patches);
int rslam_append_feature_record(rslam_ctx* ctx, const double* uv, const double* R_wc, const double* r_wc, const double* patch);
int rslam_predict_patches      (rslam_ctx* ctx, double* patches, int32_t* status);

/* Map::map_management's edits of x_k_k / p_k_k on the resident posterior, so
 * that the covariance never leaves HBM between frames.  The host keeps features_info (patches,
 * counters) and mirrors each call there; the context tracks the layout (rslam_get_layout).
 *   rslam_map_delete_feature  Map::delete_a_feature (Map.cpp:69-104), feature index 0-based
 *   rslam_map_convert         Map::inversedepth_2_cartesian (Map.cpp:105-196): the first inverse-depth
 *                             feature whose linearity index is < threshold (0.1 in the reference) becomes
 *                             Cartesian; *converted = its index or -1; linearity (L, may be NULL) receives
 *                             every index (-1 for Cartesian features)
 *   rslam_map_add_feature     ExtendKF::hinv (ExtendKF.cpp:236-265) + Map::add_a_feature_covariance_inverse_depth
 *                             (Map.cpp:339-400) for the distorted pixel uvd[2]; initial_rho = std_rho = 1 in
 *                             the reference (Map.cpp:217-218,384)
 *   rslam_map_predict         ExtendKF::predict_camera_measurements(x_k_k) as Map::initialize_a_features uses it
 *                             for its occupancy test (Map.cpp:221-229,252-261); h (L*2), visible (L)
 * Each returns after the edit completed; follow with rslam_ekf_prediction and
 * rslam_predict(ctx, &new_layout, NULL, NULL, ...). */
int rslam_map_delete_feature(rslam_ctx* ctx, int32_t feature);
int rslam_map_convert       (rslam_ctx* ctx, double linearity_threshold, int32_t* converted, double* linearity);
int rslam_map_add_feature   (rslam_ctx* ctx, const double* uvd, double initial_rho, double std_rho);
int rslam_map_predict       (rslam_ctx* ctx, double* h, uint8_t* visible);
int rslam_get_layout        (rslam_ctx* ctx, int32_t* n, int32_t* L, uint8_t* type /* L, may be NULL */,
                             int32_t* offset /* L, may be NULL */);

int rslam_fetch_cov  (rslam_ctx* ctx, double* P /* host, n*n */);
/* Drop the page-lock registrations of RSLAM_PIN_HOST_COV (synchronises first).  To be called before the caller frees or
 * re-allocates a covariance buffer it has passed to the drop-in calls WITHOUT changing the state dimension -- the reference's
 * Map::map_management (Map.cpp:88-103) deletes and adds features, which can bring back the same n at the same address. */
int rslam_unpin_host_buffers(rslam_ctx* ctx);
int rslam_fetch_state(rslam_ctx* ctx, double* x /* host, n   */);
/* Per-stage hipEvent timing of eager (non-graph) frames: off by default. */
int rslam_enable_timing(rslam_ctx* ctx, int on);
int rslam_timings    (rslam_ctx* ctx, rslam_stage_times* out);

/* ------------------------------------------------------------------ *
 *  Resident API: inputs are loaded once, every step runs from HBM,   *
 *  nothing synchronises with the host until rslam_sync.  Used by the *
 *  benchmark and by the multi-GPU hypothesis sharding (one process   *
 *  per GPU; the exchange of supports happens between step_score and  *
 *  step_update on the caller's side, e.g. an RCCL all-gather).       *
 * ------------------------------------------------------------------ */

/* hipStream_t to enqueue on (NULL = the context's own stream). */
int rslam_set_stream(rslam_ctx* ctx, void* hip_stream);

/* Upload one frame's inputs (host pointers) into context-owned HBM buffers. */
int rslam_load_frame(rslam_ctx* ctx, const rslam_layout* layout,
                     const double* x_pred, const double* P_pred,
                     const double* z, const uint8_t* ic,
                     const double* draws, int32_t n_draws);

/* New measurements for the resident prior (a frame sequence whose covariance never leaves HBM):
 * z / ic / draws as for rslam_ransac_update, uploaded into the context's buffers; the gather tables
 * that replace Converter::find/select (Converter.cpp:210-287) are rebuilt.  Returns after the upload. */
int rslam_load_measurements(rslam_ctx* ctx, const double* z, const uint8_t* ic,
                            const double* draws, int32_t n_draws);

/* Segment 1 on the resident frame. */
int rslam_step_predict(rslam_ctx* ctx);

/* K2-K4 for hypotheses [hyp_begin, hyp_end) of the draw list: writes
 * d_supports[hyp_begin..hyp_end) (DEVICE pointer, int32, n_draws entries). */
int rslam_step_score(rslam_ctx* ctx, int32_t hyp_begin, int32_t hyp_end,
                     int32_t* d_supports);

/* K5-K12 + both updates, from the complete support list (DEVICE pointer). */
int rslam_step_update(rslam_ctx* ctx, const int32_t* d_supports);

/* Whole frame = step_predict + step_score(0, n_draws) + step_update with a
 * context-owned support buffer; replayed from a hipGraph when use_graph != 0.
 * Which to use: the replay is ONE host call per frame (7-9 us of host time on the measured host instead of ~22 for the frame's
 * seven launches), but the runtime leaves the device idle for ~8 us in front of every replay and not between stream-ordered
 * launches (ROCm 7.2, kernel trace: DESIGN.md section 4, "hipGraph"): frames enqueued back to back are ~4 % faster with
 * use_graph = 0 as long as the host stays ahead of the device; use_graph = 1 is for hosts that do not. */
int rslam_step_frame(rslam_ctx* ctx, int32_t use_graph);

/* The same frame in two hipGraph-replayed halves for the hypothesis-sharded (multi-GPU) case:
 * phase 0 = step_predict + step_score(hyp_begin, hyp_end), phase 1 = step_update; the caller
 * exchanges the supports (e.g. RCCL all-gather) between the two.  d_supports is a DEVICE pointer. */
int rslam_step_phase(rslam_ctx* ctx, int32_t phase, int32_t hyp_begin, int32_t hyp_end,
                     int32_t* d_supports, int32_t use_graph);

/* The hypothesis-sharded frame of one rank in ONE call, for the C++ System loop that owns an RCCL
 * communicator (one process per GPU; caller = the body of System::TrackRunning, System.cpp:117-129, on every rank):
 * the draw list 0 .. n_draws-1 is cut into `world` contiguous slices of chunk = ceil(n_draws / world); this rank scores
 * slice `rank` (phase 0 of rslam_step_phase), the int32 supports are exchanged with ONE ncclAllGather of chunk entries
 * per rank on the context's stream (4 KB per 1000 hypotheses: latency-bound, never chunked), and every rank replays the
 * consensus scan on the gathered list and applies both updates (phase 1): bit-identical posteriors on all ranks, no
 * covariance broadcast.  nccl_comm is the caller's ncclComm_t (rccl.h) of `world` ranks; NULL is allowed for
 * world == 1 (no collective).  RCCL is bound at the first call (the librccl.so.1 already in the process, else the
 * system one): librslam_hip.so itself does not depend on it; before the first collective of a communicator its size and
 * this rank's index in it are checked against `world` / `rank` (ncclCommCount, ncclCommUserRank: RSLAM_ERR_COMM on a
 * mismatch instead of a hung all-gather).  Every rank must have loaded the same frame
 * (rslam_load_frame / rslam_load_measurements).  Stream-ordered like rslam_step_frame: results via rslam_sync and
 * rslam_fetch_results. */
int rslam_shard_frame(rslam_ctx* ctx, void* nccl_comm, int32_t rank, int32_t world, int32_t use_graph);
/* The same frame with the exchange as ONE 8-byte all-reduce (north_star: "an RCCL all-reduce ... for the consensus inlier
 * count"; SURVEY 8e's alternative): each rank folds its slice into key = support << 32 | (0xFFFFFFFF - hypothesis index),
 * ncclAllReduce(ncclUint64, ncclMax) on the context's stream leaves the LARGEST support and, among equal supports, the
 * SMALLEST index on every rank -- the earliest strict maximum the loop of Tracking.cpp:507-537 keeps -- and every rank
 * recomputes the winner's inlier mask itself (one scoring workgroup) before phase 1.  Only for contexts created with
 * adaptive = 0 (every draw is evaluated; with the adaptive stop of :531-537 the evaluated count depends on the whole list and
 * the all-gather form is the one to use): RSLAM_ERR_ARG otherwise.  Same results as rslam_shard_frame, bit for bit; same
 * communicator checks; nccl_comm may be NULL for world == 1. */
int rslam_shard_frame_allreduce(rslam_ctx* ctx, void* nccl_comm, int32_t rank, int32_t world, int32_t use_graph);

/* Diagnostics of the resident pipeline since rslam_create (any pointer may be NULL): hipGraph captures
 * (a change of the launch sequence re-captures); update stages that had to be re-run -- because a hand-over of the
 * persistent sweep (or of its tile workers) ran into its bounded wait (its workgroups were not all resident: another user
 * of the GPU), in which case the stage is re-run with the launch-per-step sweep and the context keeps to that sweep for the
 * next 64 frames (doubling while the timeouts keep coming), or because the Jnorm hand-over of the stand-alone rank update
 * timed out (re-run with the x-update riders dispatched first), or -- compat = 1 contexts, at most once per context -- because
 * a frame had more than two low-innovation inliers: the launch sequence of that mode has no low-innovation sweep (the
 * consensus launch does the one- or two-inlier update of ExtendKF.cpp:559-634 itself); the stage is re-run with the sweep and
 * the context keeps it in the sequence (raw status -40; the device says so out of band, not as a status value: a frame of
 * that shape runs its rescue stage and high-innovation pass on a posterior that was never made, and whatever those report
 * is discarded with it). */
int rslam_get_counters(rslam_ctx* ctx, int32_t* graph_captures, int32_t* sweep_reruns);
/* How the update stage of the loaded frame shape runs: 0 = launch-per-step factor sweep + stand-alone rank update (systems too
 * large for one persistent launch, or the fallback after a timed-out hand-over), 1 = persistent sweep + stand-alone rank
 * update, 2 = persistent sweep with the x / covariance update inside its launch (ExtendKF.cpp:602-609 in one kernel),
 * 3 = as 0, and updates of 12 and more column blocks of this context have taken the staged form of that route: the factor
 * sweep of the innovation covariance alone on a few compute units (a stream with a CU mask) while the others solve
 * P H^T L^-T and apply P - Y Y^T group by group of column blocks (ExtendKF.cpp:603 beside :608). */
int rslam_update_mode(rslam_ctx* ctx);
/* The raw device-side code of the last bounded wait that ran out (0: none): which hand-over it was (rslam_sync folds all of
 * them into RSLAM_ERR_HIP or recovers by re-running the update stage, see rslam_get_counters). */
int rslam_last_raw_status(rslam_ctx* ctx);
/* Which bounded wait ran out FIRST (0: none since the context was created): code | workgroup << 8 | needed value << 20 with
 * code = -(raw status) of that wait.  One wait that runs out makes the waits behind it run out too and the status word keeps the
 * smallest code only; this word names the root.  Diagnosis only. */
int rslam_last_wait_detail(rslam_ctx* ctx);
/* How that wait ran out: (its own polls, saturating at 65535) << 16 | (wall clock it saw go by in microseconds, saturating at
 * 65535).  A wait on another workgroup of a persistent launch expires only when BOTH bounds are passed: 1 ms of the device's
 * 100 MHz wall clock AND 768 polls of the waiting wave itself (>= 0.5 us each) -- the wall clock keeps running while the
 * process's queues are off the device, and a wave that was not running has not waited.  Few polls in a long time therefore
 * means the waiter was descheduled (the wait does NOT expire: the frame finishes late, with no re-run); >= 768 polls and
 * >= 1000 us means the hand-over really was withheld.  Diagnosis only. */
int rslam_last_wait_polls(rslam_ctx* ctx);

/* Block until the stream is idle; returns the device-side status of the
 * frame (RSLAM_OK, RSLAM_ERR_NOT_SPD, RSLAM_ERR_IC_NOT_VISIBLE, ...).
 * Frames may be enqueued back to back without rslam_sync: the factor sweep of an update is one persistent launch
 * sized for the largest inlier count the frame can have, and it carries the x / covariance update of the same stage
 * (no dependence on the previous frame, nothing to re-capture).  The status of a frame nobody synchronised on is not
 * lost: the next rslam_sync / rslam_fetch_* reports the smallest status seen since the last report, and
 * rslam_ekf_prediction / rslam_map_* / rslam_load_measurements check the frame in flight first.  A bounded wait inside that
 * launch that runs out (its workgroups were not all resident: another user of the GPU) is such a status: the call that
 * sees it re-runs the update stage on the launch-per-step path (see rslam_get_counters).
 * Systems too large for that path (more than 16 column blocks of S or more 16-row strips than compute units, e.g. 1000
 * landmarks) run one launch sequence per block step whose length the HOST reads from the device (one integer after the
 * consensus, one after the rescue gate): rslam_step_frame / rslam_step_update block for those two round trips there and such
 * frames are never replayed from a hipGraph. */
int rslam_sync(rslam_ctx* ctx);

/* Results of the last frame (host pointers; any may be NULL). Synchronises. */
int rslam_fetch_prediction(rslam_ctx* ctx, double* h, uint8_t* visible, double* S);
int rslam_fetch_results(rslam_ctx* ctx, double* x_new, uint8_t* li, uint8_t* hi,
                        int32_t* best_hyp, int32_t* best_support,
                        int32_t* hyps_evaluated, int32_t* n_li, int32_t* n_hi);
/* Per-hypothesis supports and 64-bit inlier masks of the last scored slice
 * (host pointers; masks: n_draws * ceil(m/64) words, bit j = j-th matched
 * feature in feature order). */
int rslam_fetch_supports(rslam_ctx* ctx, int32_t* supports, uint64_t* masks,
                         int32_t* n_mask_words);

/* ------------------------------------------------------------------ *
 *  Kernel-level entry points (device pointers) used by the roofline  *
 *  measurements and kernel parity tests.                             *
 * ------------------------------------------------------------------ */

/* C(n x n, ldc) = sym(A)(n x n, lda) - Y(n x r, ldy) * Y^T  with
 * sym(A) = (A + A^T)/2: the covariance rank-r update of ExtendKF.cpp:608-609
 * in its factored form (K S K^T = Y Y^T, Y = P H^T L^-T).  n, r arbitrary;
 * buffers must be padded: lda, ldc, ldy >= round_up(n, 64), Y has
 * round_up(r, 32) columns, pad entries zero.  FP64 MFMA kernel (K10). */
int rslam_k_rank_update(rslam_ctx* ctx, int32_t n, int32_t r,
                        const double* dA, int32_t lda,
                        const double* dY, int32_t ldy,
                        double* dC, int32_t ldc);

/* The same kernel timed on its own (roofline measurement): `reps` back-to-back launches on context-owned buffers of
 * the shape (n, r), bracketed by hipEvents on the context's stream; *us_per_launch = mean duration of one launch.  In the
 * frame itself the rank update of systems up to ~3000 states runs inside the factor sweep's launch (see rslam_sync). */
int rslam_k_rank_update_time(rslam_ctx* ctx, int32_t n, int32_t r, int32_t reps, double* us_per_launch);

/* C(m x n) = alpha * A(m x k) * B(n x k)^T + beta * C, dense FP64 MFMA GEMM
 * (the dense form of P*H^T, Tracking.cpp:42,420-421).  All dimensions must be
 * padded to multiples of 64 (k: 32) with zero fill. */
int rslam_k_gemm_nt(rslam_ctx* ctx, int32_t m, int32_t n, int32_t k,
                    double alpha, const double* dA, int32_t lda,
                    const double* dB, int32_t ldb,
                    double beta, double* dC, int32_t ldc);

/* Peak-rate probe: issues back-to-back v_mfma_f64_16x16x4_f64 from every CU
 * and returns the measured TFLOP/s (prints nothing). */
int rslam_k_mfma_f64_peak(rslam_ctx* ctx, double* tflops);
/* Same probe with (waves_and_mode & 15) resident waves per SIMD and instruction mix
 * (waves_and_mode >> 4): 0 = 4 accumulators of 16x16x4, 1 = 8 accumulators, 2 = 4x4x4_4b;
 * also reports the shader cycles one SIMD spends per MFMA and the in-kernel clock
 * (either may be NULL). */
int rslam_k_mfma_f64_probe(rslam_ctx* ctx, int32_t waves_and_mode, double* tflops,
                           double* cycles_per_mfma, double* clock_mhz);
/* One v_mfma_f64_4x4x4_4b_f64 on caller-supplied per-lane operands (host pointers, 64 doubles
 * each): d = mfma(a, b, c) with the given CBSZ / ABID.  Unit test of the operand lane maps the
 * tile engine relies on. */
int rslam_k_mfma4_raw(rslam_ctx* ctx, int32_t cbsz, int32_t abid, const double* a, const double* b,
                      const double* c, double* d);
/* Streaming-copy probe: measured HBM GB/s for a bytes-sized device copy. */
int rslam_k_hbm_copy_peak(rslam_ctx* ctx, int64_t bytes, double* gbps);

#ifdef __cplusplus
}
#endif
#endif /* RSLAM_H */
