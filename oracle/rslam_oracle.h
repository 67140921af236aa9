/*
 * rslam_oracle.h -- CPU restatement of the reference's 1-point-RANSAC EKF
 * update path.  TEST INFRASTRUCTURE ONLY: nothing in the product path may
 * include, link or call this; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py do.
 *
 * PARITY UNPINNED: the reference (plumewind/ransac_slam) ships no tests, golden
 * vectors or fixtures for this path and cannot be compiled in the build image
 * (Eigen, OpenCV and ROS absent), so this restatement is pinned only by
 * known-answer tests derivable from the reference source (tests/test_oracle_kat.py)
 * and by an independent numpy/scipy cross-check of its linear algebra.
 *
 * Every function cites the reference file:line it follows
 * (paths relative to /root/reference).
 */
#ifndef RSLAM_ORACLE_H
#define RSLAM_ORACLE_H

#include <stdint.h>
#include "../include/rslam.h"   /* plain-data structs and error codes only */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

int orc_create (const rslam_config* cfg, orc_ctx** out);
int orc_destroy(orc_ctx* c);

/* structure = 0: reference structure -- dense H_i (2 x n), dense H_i*P*H_i^T
 *                and P*H_i^T recomputed per RANSAC iteration
 *                (Tracking.cpp:419-422), dense update in the reference's GEMM
 *                order (ExtendKF.cpp:602-609).  This is the CPU baseline.
 * structure = 1: same arithmetic with the structural zeros of H_i skipped and
 *                repeated hypotheses cached (used to check full-size cases in
 *                seconds; validated against structure 0 in tests). */
int orc_set_structure(orc_ctx* c, int structure);

/* Same contracts as rslam_predict / rslam_ransac_update (include/rslam.h). */
int orc_predict(orc_ctx* c, const rslam_layout* layout,
                const double* x_pred, const double* P_pred,
                double* h, uint8_t* visible, double* S);
int orc_ransac_update(orc_ctx* c, const double* z, const uint8_t* ic,
                      const double* draws, int32_t n_draws,
                      double* x_new, double* P_new,
                      uint8_t* li, uint8_t* hi,
                      int32_t* best_hyp, int32_t* best_support,
                      int32_t* hyps_evaluated);

/* RANSAC stage only (Tracking.cpp:352-539); max_iters > 0 bounds the number of
 * loop iterations (for the bounded CPU-baseline sample). */
int orc_ransac_only(orc_ctx* c, const double* z, const uint8_t* ic,
                    const double* draws, int32_t n_draws, int32_t max_iters,
                    uint8_t* li, int32_t* best_hyp, int32_t* best_support,
                    int32_t* hyps_evaluated);

/* Updates + rescue (System.cpp:123-129) from the flags orc_ransac_only left. */
int orc_finish_update(orc_ctx* c, double* x_new, double* P_new, uint8_t* li, uint8_t* hi);

/* Introspection for tests (valid after orc_ransac_update / orc_ransac_only). */
int orc_get_supports(orc_ctx* c, int32_t* supports /* hyps_evaluated */,
                     int32_t* positions /* hyps_evaluated: hypothesised feature */,
                     uint64_t* masks /* hyps_evaluated * words */, int32_t* words);
/* smallest |residual - sigma_z| over all scored pairs, smallest |d2 - chi2|
 * over all rescue candidates (the margin audit of DESIGN.md) */
int orc_get_margins(orc_ctx* c, double* score_margin, double* rescue_margin);
/* residual capture: pixels, m x m (row = rank of the hypothesised matched feature, column = rank of the scored one;
 * NaN where a position was never hypothesised); switch on before orc_ransac_update / orc_ransac_only */
int orc_enable_residuals(orc_ctx* c, int on);
int orc_get_residuals(orc_ctx* c, double* out /* m*m, nullable */, int32_t* m);
/* dense Jacobians of the last linearisation, L * (2 x n) col-major blocks */
int orc_get_H(orc_ctx* c, double* H);
/* intermediate filter state after the low-innovation update */
int orc_get_li_state(orc_ctx* c, double* x_li, double* P_li);

/* ---- unit functions (known-answer tests) ---- */
void orc_q2r(const double q[4], double R[9]);                          /* ExtendKF.cpp:91-102   */
void orc_hu(const rslam_camera* cam, const double y[3], double uv[2]); /* ExtendKF.cpp:153-174  */
void orc_distort_fm(const rslam_camera* cam, const double uv[2], double uvd[2]);   /* :175-204 */
void orc_undistort_fm(const rslam_camera* cam, const double uvd[2], double uvu[2]); /* :266-285 */
void orc_jacob_undistor_fm(const rslam_camera* cam, const double uvd[2], double J[4]); /* :312-332 */
void orc_dRq_times_a_by_dq(const double q[4], const double a[3], double out[12]);  /* :286-311 */
int  orc_hi_cartesian(const rslam_camera* cam, const double hrl[3], double uv[2]); /* :103-132 */
/* n_hyp after an improvement, Tracking.cpp:531-532 */
int  orc_adaptive_n_hyp(double p, int support, int num_ic);
/* ExtendKF::update, ExtendKF.cpp:597-639.  H is r x n col-major (ld = r). */
int  orc_update(int compat, int n, int r, const double* x, const double* P,
                const double* H, const double* z, const double* h,
                double* x_out, double* P_out);
/* ExtendKF::ekf_prediction (ExtendKF.cpp:333-388), "constant_velocity" filter (System.cpp:63) */
int  orc_ekf_prediction(int n, const double* x_kk, const double* P_kk, double delta_t, double std_a,
                        double std_alpha, double* x_pred, double* P_pred);
/* its 13-state motion model: fv (:389-400), dfv_by_dxv (:444-465), Q = G Pn G' (:347-376) */
void orc_motion_model(const double xv[13], double delta_t, double std_a, double std_alpha,
                      double xv_pred[13], double F[169], double Q[169]);
/* Map state surgery; outputs sized for the new state dimension */
int  orc_map_delete_feature(int n, int L, const uint8_t* type, const double* x, const double* P,
                            int feature, double* x_out, double* P_out);                    /* Map.cpp:69-104  */
int  orc_map_convert(int n, int L, const uint8_t* type, const double* x, const double* P, double threshold,
                     double* x_out, double* P_out, int* converted);                         /* Map.cpp:105-196 */
double orc_linearity_index(const double* x, const double* P, int n, int offset);            /* Map.cpp:124-149 */
int  orc_map_add_feature(const rslam_camera* cam, double std_z, int n, const double* x, const double* P,
                         const double uvd[2], double initial_rho, double std_rho,
                         double* x_out, double* P_out);                                     /* Map.cpp:281-292,339-400 */
void orc_hinv(const rslam_camera* cam, const double uvd[2], const double Xv[13], double initial_rho,
              double y[6]);                                                                 /* ExtendKF.cpp:236-265 */
void orc_add_feature_jacobians(const rslam_camera* cam, double std_z, double std_rho, const double uvd[2],
                               const double Xv[13], double D[78], double Rn[36]);           /* Map.cpp:339-388 */
/* NCC search: Tracking::matching, Tracking.cpp:279-351 + Converter::corrcoef_opencv */
int  orc_match_feature(const rslam_camera* cam, const uint8_t* image, const double* patch, int half,
                       const double h[2], const double S[4], double corr_threshold, double chi2,
                       double z[2], double* best_corr, int* n_cand, double margins[3]);
void orc_matching(const rslam_camera* cam, const uint8_t* image, int L, const double* patches, int half,
                  const double* h, const uint8_t* has_h, const double* S, double* z, uint8_t* ic, double* corr,
                  double margins[3]);
/* Patch prediction: Tracking::pred_patch_fc, Tracking.cpp:164-278 and its caller :46-65 */
int  orc_pred_patch(const rslam_camera* cam, int compat, const double xv[7], const double h[2], const double uv_f[2],
                    const double R_f[9], const double r_f[3], const double* patch_f, int half_f, const double XYZ_w[3],
                    int half, double* out, double* margin);
void orc_pred_patches(const rslam_camera* cam, int compat, int L, const uint8_t* type, const int32_t* offset,
                      const double* x, const double* h, const uint8_t* has_h, const double* uv_f, const double* R_f,
                      const double* r_f, const double* patch_f, int half_f, int half, double* out, int32_t* status,
                      double* margin);
/* dynamic-size inverse as Eigen does it (PartialPivLU), Tracking.cpp:421 */
int  orc_inverse_lu(int n, const double* A, double* Ainv);

#ifdef __cplusplus
}
#endif
#endif
