// ransac_slam_hip.hpp -- host-side mirror of the reference's interface for the hot path,
// written above the C ABI (include/rslam.h).  The reference is compiled C++ whose path
// is reached through ExtendKF / Tracking member calls that communicate via public data
// members (include/ransac_slam/ExtendKF.h:154-169, Tracking.h:19-48); this header keeps
// the same names, argument meaning and call order so that System::TrackRunning
// (src/System.cpp:117-129) can switch by changing two types.  Eigen is replaced by two
// minimal column-major containers (the image has no Eigen); with Eigen available the
// members map 1:1 onto Eigen::VectorXd / MatrixXd buffers.
//
// Error behaviour: the reference exit()s or Eigen-asserts; these methods throw
// ransac_slam_hip::Error carrying the RSLAM_ERR_* code instead of killing the ROS node.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <array>
#include <cstring>
#include <vector>

#include "../../include/rslam.h"

namespace ransac_slam_hip {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& where) : std::runtime_error(where + ": " + rslam_error_string(c)), code(c) {}
};

// column-major dense matrix / vector with the few members the adapter needs
struct MatrixXd {
    int r = 0, c = 0;
    std::vector<double> v;
    void resize(int rows, int cols) { r = rows; c = cols; v.assign((size_t)rows * cols, 0.0); }
    int rows() const { return r; }
    int cols() const { return c; }
    double& operator()(int i, int j) { return v[(size_t)j * r + i]; }
    double operator()(int i, int j) const { return v[(size_t)j * r + i]; }
    double* data() { return v.data(); }
    const double* data() const { return v.data(); }
};
struct VectorXd {
    std::vector<double> v;
    void resize(int n) { v.assign((size_t)n, 0.0); }
    int rows() const { return (int)v.size(); }
    double& operator()(int i) { return v[(size_t)i]; }
    double operator()(int i) const { return v[(size_t)i]; }
    double* data() { return v.data(); }
    const double* data() const { return v.data(); }
};

// CamParam, System.h:69-82 (hot fields)
struct CamParam { double k1, k2; int nRows, nCols; double Cx, Cy, f, dx, dy; };

// struct Feature, ExtendKF.h:14-42 (hot fields; the dense 2 x n H is never materialised)
struct Feature {
    std::string type;                    // "inversedepth" | "cartesian"
    bool individually_compatible = false;
    bool low_innovation_inlier = false;
    bool high_innovation_inlier = false;
    VectorXd z;                          // 2 entries when matched, empty otherwise
    VectorXd h;                          // 2 entries when predicted visible, empty otherwise
    MatrixXd S;                          // 2 x 2
    MatrixXd patch_when_matching;        // 13 x 13, written by pred_patch_fc (Tracking.cpp:277)
    // initialisation record (Map.cpp:286-292), mirrored into the device-side feature store
    MatrixXd patch_when_initialized;     // 41 x 41
    MatrixXd R_wc_when_initialized;      // 3 x 3
    VectorXd r_wc_when_initialized;      // 3
    VectorXd uv_when_initialized;        // 2
};

// ExtendKF members and methods on the hot path (ExtendKF.h:154-169)
class ExtendKF {
public:
    std::vector<Feature> features_info;
    CamParam* cam = nullptr;
    VectorXd x_k_k, x_k_km1;
    MatrixXd p_k_k, p_k_km1;
    double std_z = 1.0;
    double std_a = 0.007, std_alpha = 0.007;     // Sigma.a, Sigma.alpha (initialize_param.yaml:40-41)

    // n_draws: length of the draw list handed to the RANSAC loop per frame (>= the
    // reference's initial n_hyp = 1000, Tracking.cpp:357)
    // pin_covariance: page-lock p_k_km1 / p_k_k for the transfers of the drop-in calls (RSLAM_PIN_HOST_COV, include/rslam.h);
    // the two matrices are members of this object, so they outlive every call -- but a resize (Map::map_management adds or
    // deletes a feature) re-allocates them: the library drops its registrations whenever the state dimension changes, and
    // map_management code that re-allocates p_k_k / p_k_km1 at an UNCHANGED dimension (a delete followed by an add) calls
    // covariance_buffers_reallocated() first (rslam_unpin_host_buffers)
    // adaptive: the adaptive stop of the RANSAC loop (Tracking.cpp:531-537); false = every draw is evaluated (benchmarks, and the
    // one-key all-reduce form of the sharded frame, rslam_shard_frame_allreduce)
    ExtendKF(CamParam* param, int device = 0, int compat = 1, int n_draws = 1400, bool pin_covariance = false, bool adaptive = true)
        : cam(param), n_draws_(n_draws)
    {
        rslam_config cfg{};
        cfg.cam.k1 = param->k1; cfg.cam.k2 = param->k2; cfg.cam.Cx = param->Cx; cfg.cam.Cy = param->Cy;
        cfg.cam.f = param->f; cfg.cam.dx = param->dx; cfg.cam.dy = param->dy;
        cfg.cam.nRows = param->nRows; cfg.cam.nCols = param->nCols;
        cfg.sigma_z = std_z; cfg.p_success = 0.99; cfg.n_hyp_init = 1000; cfg.chi2_gate = 5.9915;
        cfg.compat = compat; cfg.adaptive = adaptive ? 1 : 0; cfg.dedup = 1;
        cfg.reserved = pin_covariance ? RSLAM_PIN_HOST_COV : 0;
        const int rc = rslam_create(&cfg, device, &ctx_);
        if (rc) throw Error(rc, "rslam_create");
    }
    ~ExtendKF() { if (ctx_) rslam_destroy(ctx_); }
    ExtendKF(const ExtendKF&) = delete;
    ExtendKF& operator=(const ExtendKF&) = delete;

    rslam_ctx* ctx() { return ctx_; }
    int n_draws() const { return n_draws_; }

    // predict_state_and_covariance (ExtendKF.cpp:333-388) on the device.  map_changed = true
    // when Map::map_management edited x_k_k / p_k_k on the host this frame (System.cpp:111):
    // they are uploaded first; otherwise the posterior of the previous frame is still resident
    // and nothing crosses PCIe.  The prior stays on the device for search_IC_matches_predict.
    void ekf_prediction(bool map_changed, const std::vector<uint8_t>& type, const std::vector<int32_t>& offset)
    {
        int rc;
        if (map_changed) {
            rslam_layout lay{x_k_k.rows(), (int32_t)type.size(), type.data(), offset.data()};
            rc = rslam_set_posterior(ctx_, &lay, x_k_k.data(), p_k_k.data());
            if (rc) throw Error(rc, "rslam_set_posterior");
        }
        rc = rslam_ekf_prediction(ctx_, 1.0, std_a, std_alpha);
        if (rc) throw Error(rc, "rslam_ekf_prediction");
        prior_resident = true;
    }
    bool prior_resident = false;
    // to be called before p_k_k / p_k_km1 are freed or re-allocated without a change of dimension (pin_covariance only)
    void covariance_buffers_reallocated()
    {
        const int rc = rslam_unpin_host_buffers(ctx_);
        if (rc) throw Error(rc, "rslam_unpin_host_buffers");
    }

    // Partial update using low-innovation inliers (ExtendKF.cpp:559-596).  The device
    // already ran both updates inside Tracking::ransac_hypotheses; this publishes nothing
    // new (the reference's intermediate x_k_k/p_k_k is overwritten two calls later).
    void ekf_update_li_inliers() {}
    // Partial update using high-innovation inliers (ExtendKF.cpp:640-678): publish x_k_k,
    // and p_k_k when something on the host needs it (Map management reads it each frame).
    void ekf_update_hi_inliers(bool fetch_covariance = true)
    {
        const int n = x_k_km1.rows();
        x_k_k.resize(n);
        int rc = rslam_fetch_state(ctx_, x_k_k.data());
        if (rc) throw Error(rc, "rslam_fetch_state");
        if (fetch_covariance) {
            p_k_k.resize(n, n);
            rc = rslam_fetch_cov(ctx_, p_k_k.data());
            if (rc) throw Error(rc, "rslam_fetch_cov");
        }
    }

private:
    rslam_ctx* ctx_ = nullptr;
    int n_draws_;
};

// Tracking methods on the hot path (Tracking.h:19-48)
class Tracking {
public:
    explicit Tracking(ExtendKF* m_ExtendKF) : mT_ExtendKF(m_ExtendKF) {}

    // First part of Tracking::search_IC_matches (Tracking.cpp:35-44): h_i, Jacobians, S_i.
    // The patch warp and NCC matching (Tracking.cpp:46-69) stay on the host and consume
    // features_info[i].h / .S exactly as before.
    void search_IC_matches_predict()
    {
        ExtendKF& k = *mT_ExtendKF;
        const int L = (int)k.features_info.size();
        type_.resize(L); offset_.resize(L);
        int off = 13;
        for (int i = 0; i < L; ++i) {
            const bool id = (k.features_info[i].type == "inversedepth");
            type_[i] = id ? RSLAM_FEAT_INVERSE_DEPTH : RSLAM_FEAT_CARTESIAN;
            offset_[i] = off; off += id ? 6 : 3;
        }
        rslam_layout lay{off, L, type_.data(), offset_.data()};
        std::vector<double> h(2 * (size_t)L + 1), S(4 * (size_t)L + 1);
        std::vector<uint8_t> vis((size_t)L + 1);
        const int rc = k.prior_resident
            ? rslam_predict(k.ctx(), &lay, nullptr, nullptr, h.data(), vis.data(), S.data())
            : rslam_predict(k.ctx(), &lay, k.x_k_km1.data(), k.p_k_km1.data(), h.data(), vis.data(), S.data());
        if (rc) throw Error(rc, "rslam_predict");
        for (int i = 0; i < L; ++i) {
            Feature& f = k.features_info[i];
            if (!vis[i]) continue;                      // "if(hi.rows() != 0)", ExtendKF.cpp:77
            f.h.resize(2); f.h(0) = h[2 * i]; f.h(1) = h[2 * i + 1];
            f.S.resize(2, 2);
            for (int q = 0; q < 4; ++q) f.S.v[q] = S[4 * i + q];
        }
    }

    // The loop over pred_patch_fc in Tracking::search_IC_matches (Tracking.cpp:46-65, :164-278): every
    // predicted feature's 13 x 13 patch, warped on the device from the feature store.  upload_records = true
    // (re)sends every feature's initialisation record first; afterwards Map::add_a_feature /
    // delete_a_feature keep the store in step.  With fetch = false the patches stay on the device for
    // matching(image, true).
    void pred_patches(bool upload_records, bool fetch = true)
    {
        ExtendKF& k = *mT_ExtendKF;
        const int L = (int)k.features_info.size();
        int rc;
        if (upload_records) {
            std::vector<double> uv(2 * (size_t)L + 1), R(9 * (size_t)L + 1), r(3 * (size_t)L + 1), pf(1681 * (size_t)L + 1);
            for (int i = 0; i < L; ++i) {
                const Feature& f = k.features_info[i];
                std::memcpy(&uv[2 * (size_t)i], f.uv_when_initialized.data(), sizeof(double) * 2);
                std::memcpy(&R[9 * (size_t)i], f.R_wc_when_initialized.data(), sizeof(double) * 9);
                std::memcpy(&r[3 * (size_t)i], f.r_wc_when_initialized.data(), sizeof(double) * 3);
                std::memcpy(&pf[1681 * (size_t)i], f.patch_when_initialized.data(), sizeof(double) * 1681);
            }
            rc = rslam_set_feature_records(k.ctx(), L, uv.data(), R.data(), r.data(), pf.data());
            if (rc) throw Error(rc, "rslam_set_feature_records");
        }
        std::vector<double> patches(fetch ? 169 * (size_t)L + 1 : 1);
        rc = rslam_predict_patches(k.ctx(), fetch ? patches.data() : nullptr, nullptr);
        if (rc) throw Error(rc, "rslam_predict_patches");
        if (fetch)
            for (int i = 0; i < L; ++i) {
                MatrixXd& pm = k.features_info[i].patch_when_matching;
                pm.resize(13, 13);
                std::memcpy(pm.data(), &patches[169 * (size_t)i], sizeof(double) * 169);
            }
    }

    // Tracking::matching (Tracking.cpp:279-351): the NCC search on the device, on the h / S that
    // search_IC_matches_predict left there.  image = the frame's cv::Mat data (uint8, nRows x nCols,
    // row-major); the predicted patches come from pred_patch_fc, which stays on the host.
    void matching(const uint8_t* image, bool patches_on_device = false)
    {
        ExtendKF& k = *mT_ExtendKF;
        const int L = (int)k.features_info.size();
        std::vector<double> patches((size_t)L * 169 + 1, 0.0), z(2 * (size_t)L + 1);
        std::vector<uint8_t> ic((size_t)L + 1);
        for (int i = 0; i < L; ++i) {
            const MatrixXd& pm = k.features_info[i].patch_when_matching;
            if (pm.rows() == 13 && pm.cols() == 13) std::memcpy(&patches[(size_t)i * 169], pm.data(), sizeof(double) * 169);
        }
        const int rc = rslam_match(k.ctx(), image, patches_on_device ? nullptr : patches.data(), z.data(), ic.data(), nullptr);
        if (rc) throw Error(rc, "rslam_match");
        for (int i = 0; i < L; ++i) {
            Feature& f = k.features_info[i];
            if (!ic[i]) continue;
            f.individually_compatible = true;                       // Tracking.cpp:345
            f.z.resize(2); f.z(0) = z[2 * i]; f.z(1) = z[2 * i + 1];   // :346
        }
    }

    // Tracking::ransac_hypotheses (Tracking.cpp:352-539).  One device call runs the RANSAC
    // stage, both EKF updates and the rescue (System.cpp:120-129) with P resident in HBM.
    // The reference draws rand()/RAND_MAX per iteration (ExtendKF.cpp:230); the draws are
    // generated up front from the same std::rand stream.
    void ransac_hypotheses()
    {
        ExtendKF& k = *mT_ExtendKF;
        const int L = (int)k.features_info.size();
        std::vector<double> z(2 * (size_t)L + 1, 0.0), draws((size_t)k.n_draws());
        std::vector<uint8_t> ic((size_t)L + 1, 0);
        li_.assign((size_t)L + 1, 0); hi_.assign((size_t)L + 1, 0);
        for (int i = 0; i < L; ++i) {
            const Feature& f = k.features_info[i];
            ic[i] = f.individually_compatible ? 1 : 0;
            if (ic[i]) { z[2 * i] = f.z(0); z[2 * i + 1] = f.z(1); }
        }
        if (!replay_draws.empty()) draws = replay_draws;          // deterministic replay (tests)
        else for (double& d : draws) d = (double)std::rand() / ((double)RAND_MAX + 1.0);
        std::vector<double> x_new((size_t)k.x_k_km1.rows());
        const int rc = rslam_ransac_update(k.ctx(), z.data(), ic.data(), draws.data(), (int)draws.size(), x_new.data(),
                                           nullptr /* P stays resident */, li_.data(), hi_.data(), &best_hyp,
                                           &best_support, &hyps_evaluated);
        if (rc) throw Error(rc, "rslam_ransac_update");
        for (int i = 0; i < L; ++i) k.features_info[i].low_innovation_inlier = li_[i] != 0;
    }

    // Tracking::rescue_hi_inliers (Tracking.cpp:574-597): flags were produced on the device.
    void rescue_hi_inliers()
    {
        ExtendKF& k = *mT_ExtendKF;
        for (size_t i = 0; i < k.features_info.size(); ++i) k.features_info[i].high_innovation_inlier = hi_[i] != 0;
    }

    int32_t best_hyp = -1, best_support = 0, hyps_evaluated = 0;
    std::vector<double> replay_draws;    // when set, used instead of the std::rand stream

private:
    ExtendKF* mT_ExtendKF;
    std::vector<uint8_t> type_, li_, hi_;
    std::vector<int32_t> offset_;
};

// Map methods that edit the filter state (Map.h:24-37), run on the posterior that the update
// left in HBM: with these, System::TrackRunning never moves p_k_k across PCIe
// (ExtendKF::ekf_prediction(false, ...) follows).  Image work -- FAST corners, patches, the
// random box of Map::initialize_a_features (Map.cpp:232-250) -- stays on the host.
class Map {
public:
    explicit Map(ExtendKF* m_ExtendKF) : mM_ExtendKF(m_ExtendKF) {}

    // Map::delete_a_feature (Map.cpp:69-104); featToDelete counts from 1 as in Map.cpp:21.  The caller
    // erased features_info[featToDelete - 1] beforehand, as Map::map_management does (Map.cpp:27-28).
    void delete_a_feature(int featToDelete)
    {
        const int rc = rslam_map_delete_feature(mM_ExtendKF->ctx(), featToDelete - 1);
        if (rc) throw Error(rc, "rslam_map_delete_feature");
    }

    // Map::inversedepth_2_cartesian (Map.cpp:105-196): at most one feature per call
    int inversedepth_2_cartesian(double linearity_index_threshold = 0.1)
    {
        int32_t converted = -1;
        const int rc = rslam_map_convert(mM_ExtendKF->ctx(), linearity_index_threshold, &converted, nullptr);
        if (rc) throw Error(rc, "rslam_map_convert");
        if (converted >= 0) mM_ExtendKF->features_info[converted].type = "cartesian";     // Map.cpp:191
        return converted;
    }

    // First step of Map::initialize_a_features (Map.cpp:221-229): where every feature is expected,
    // for the occupancy test of the sampling box (Map.cpp:252-261)
    std::vector<std::array<double, 2>> predicted_positions()
    {
        ExtendKF& k = *mM_ExtendKF;
        const int L = (int)k.features_info.size();
        std::vector<double> h(2 * (size_t)L + 1);
        std::vector<uint8_t> vis((size_t)L + 1);
        const int rc = rslam_map_predict(k.ctx(), h.data(), vis.data());
        if (rc) throw Error(rc, "rslam_map_predict");
        std::vector<std::array<double, 2>> h_pred;
        for (int i = 0; i < L; ++i) if (vis[i]) h_pred.push_back({h[2 * i], h[2 * i + 1]});
        return h_pred;
    }

    // Fourth step of Map::initialize_a_features (Map.cpp:271-312): hinv, the covariance of the new
    // feature (add_a_feature_covariance_inverse_depth, Map.cpp:339-400) and its features_info entry
    // new_feature carries the image-side fields the host filled (patch_when_initialized, uv_when_initialized,
    // R_wc / r_wc_when_initialized, Map.cpp:286-292); when all four are set its record goes to the feature store
    void add_a_feature(const double uv[2], Feature new_feature = Feature(), int initial_rho = 1, int std_rho = 1)
    {
        int rc = rslam_map_add_feature(mM_ExtendKF->ctx(), uv, initial_rho, std_rho);
        if (rc) throw Error(rc, "rslam_map_add_feature");
        new_feature.type = "inversedepth";
        if (new_feature.patch_when_initialized.rows() == 41 && new_feature.R_wc_when_initialized.rows() == 3 &&
            new_feature.r_wc_when_initialized.rows() == 3 && new_feature.uv_when_initialized.rows() == 2) {
            rc = rslam_append_feature_record(mM_ExtendKF->ctx(), new_feature.uv_when_initialized.data(),
                                             new_feature.R_wc_when_initialized.data(), new_feature.r_wc_when_initialized.data(),
                                             new_feature.patch_when_initialized.data());
            if (rc) throw Error(rc, "rslam_append_feature_record");
        }
        mM_ExtendKF->features_info.push_back(new_feature);
    }

private:
    ExtendKF* mM_ExtendKF;
};

}  // namespace ransac_slam_hip
