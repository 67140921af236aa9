// rslam_api.hip -- context management and the C ABI of include/rslam.h.
//
// Data layout in HBM (all FP64, column-major):
//   x_pred[NP], P_pred[NP x NP]        prior, NP = round_up(n, 64), zero padded
//   H13[L][2][13]                      compact Jacobians (7 pose + 6 feature columns)
//   W[NP x 2m]                         P H^T of the m matched features (2 columns each)
//   A[(RP + NP + 64) x RP]             stacked system [S; P H^T; nu^T] of one update,
//                                      RP = round_up(2m, 64); becomes [L; Y; u^T]
//   P[NP x NP]                         posterior covariance (LI update out of place,
//                                      HI update in place)
// Frame scalars (best hypothesis, inlier counts, block counts, status) live in
// d_sel[] on the device: no kernel launch depends on a host read-back, so a whole
// frame is one stream-ordered (or hipGraph-replayed) launch sequence.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <chrono>

#include "../../include/rslam.h"
#include "kernels.h"

using namespace rslam;

#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { ctx_last_hip_error = (int)e_; return RSLAM_ERR_HIP; } } while (0)
static thread_local int ctx_last_hip_error = 0;

namespace {

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;   // elements
    int ensure(size_t n) {
        if (n <= cap && p) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        if (n == 0) n = 1;
        if (hipMalloc((void**)&p, n * sizeof(T)) != hipSuccess) return -1;
        cap = n;
        return 1;     // (re)allocated
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

enum { EV_START = 0, EV_PREDICT, EV_PHT, EV_SCORE, EV_SELECT, EV_LI_FACTOR0, EV_LI_FACTOR1, EV_LI_RANK0,
       EV_LI_RANK1, EV_LI_END, EV_RESCUE, EV_HI_FACTOR0, EV_HI_FACTOR1, EV_HI_RANK0, EV_HI_RANK1, EV_HI_END, EV_COUNT };

}  // namespace

struct rslam_ctx {
    rslam_config cfg;
    Cam cam;
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // launch-per-step route: the update counts the host sizes its launch sequences from arrive in page-locked, host-mapped memory
    // (kernels.h HostCounts): [0..2] LI {count, blocks, seq}, [4..6] HI
    int32_t* h_counts = nullptr; int32_t* d_counts = nullptr; int32_t count_seq = 0;
    // frame shape
    int n = 0, NP = 0, L = 0, m = 0, H = 0, words = 0, RP = 0, ldA = 0;
    int m_id = 0, m_euc = 0;
    bool have_state = false, have_meas = false, predicted = false, pht_done = false, dedup_done = false;
    int masks_all = 0;          // this frame's scoring kept the inlier mask of every hypothesis (1: d_masks, by hypothesis; 2: d_posmask, by position)
    bool have_post = false;      // (d_x2, d_P) hold a posterior x_k_k / p_k_k
    std::vector<uint8_t> h_type, h_vis;
    std::vector<int32_t> h_off;
    // device buffers
    DevBuf<uint8_t> d_type, d_vis, d_hash, d_hash2, d_ic, d_li, d_hi, d_mtype;
    DevBuf<int32_t> d_tile_order;
    // rank update of large maps on 128 x 128 macro tiles (rank_macro.hip): tile lists for the current nT
    DevBuf<int32_t> d_macro_order, d_small_order;
    int macro_nT = 0, n_macro = 0, n_small = 0;
    bool macro_attr = false;
    bool li_defer_host = false;                  // this frame's LI update (stand-alone path) leaves its covariance deferred
    int tile_order_nT = 0;
    DevBuf<int32_t> d_off, d_mfeat, d_moff, d_mith, d_miph, d_mzsrc, d_rank_of, d_pos, d_nhyp,
                    d_sup, d_possup, d_lilist, d_hilist, d_sel;
    DevBuf<uint64_t> d_masks, d_posmask;
    DevBuf<double> d_xpred, d_Ppred, d_h, d_h2, d_H13, d_H13b, d_S, d_S2, d_z, d_wv, d_W, d_A, d_Y, d_Linv,
                   d_x1, d_x2, d_P, d_T, d_probe, d_FQ, d_mapcoef, d_lin, d_patches, d_corr, d_sc, d_hctx, d_Y1, d_Gd;
    DevBuf<uint8_t> d_image;
    DevBuf<double> d_stage;               // drop-in API: the caller's n x n covariance as it crosses PCIe (one linear transfer)
    // feature store: initialisation records of Map::initialize_a_features (Map.cpp:286-292), one slot per feature
    DevBuf<double> d_rec;                 // slot * 14: uv(2) R(9 col-major) r(3)
    DevBuf<float> d_rec_patch;            // slot * 1681: patch_when_initialized as float32, row-major 41 x 41
    DevBuf<int32_t> d_slot, d_xyz_src, d_pstatus;
    std::vector<int32_t> h_slot, free_slots;
    int store_cap = 0;
    bool patches_valid = false;           // d_patches holds the output of rslam_predict_patches for the current prediction
    DevBuf<int32_t> d_first;
    DevBuf<int32_t> d_sup_local, d_sup_all;   // rslam_shard_frame: this rank's slice of the supports, the gathered list
    DevBuf<unsigned long long> d_shard_key;   // rslam_shard_frame_allreduce: this rank's key, the reduced key
    DevBuf<int32_t> d_sweep_flags;        // hand-over flags of the persistent factor sweep (zeroed by the sweeps themselves, set by set)
    // timing
    int timing = 0;
    hipEvent_t ev[EV_COUNT];
    bool ev_ok = false;
    rslam_stage_times times;
    // graph
    // hipGraph slots: 0 = whole frame, 1 = predict + score of a hypothesis slice, 2 = update stage
    hipGraph_t graph[3] = {nullptr, nullptr, nullptr};
    hipGraphExec_t graph_exec[3] = {nullptr, nullptr, nullptr};
    bool graph_valid[3] = {false, false, false};
    int g1_hb = -1, g1_he = -1;
    const void* g1_sup = nullptr;
    const void* g2_sup = nullptr;
    int g2_masks = -1;
    const int32_t* last_sup = nullptr;
    int graph_captures = 0;
    int last_raw_status = 0;
    int last_wait_first = 0;              // SEL_WAIT_FIRST of the last bounded wait that ran out (diagnosis)
    int last_wait_polls = 0;              // SEL_WAIT_POLLS of the same wait: its polls << 16 | elapsed microseconds
    // Reference-faithful mode: the consensus set is the hypothesis' own feature (Q1) -- one inlier, two by coincidence -- and the
    // consensus launch does that low-innovation update itself (kernels.h LiSmallArgs): the persistent route's launch sequence
    // then has NO low-innovation sweep.  A frame with any other count says so out of band (sel[SEL_LI_NEED]; raw code -40);
    // rslam_sync re-runs its update stage with the sweep in the sequence, and the context keeps it there from then on.
    bool li_skip = false;
    int li_shape_reruns = 0;
    // The persistent sweep needs its whole grid resident at once.  When a bounded wait runs out (somebody else is holding
    // CUs of this GPU) the update stage is re-run with the launch-per-step sweep, and so are the next frames for a while.
    int steps_frames_left = 0;
    int sweep_fallbacks = 0;
    int consecutive_fallbacks = 0;
    bool k10_riders_first = false;     // a Jnorm wait of the stand-alone rank update timed out once: riders in front from now on
    int k10_reruns = 0;
    int k10_inject = 0;                // fault injection (diagnostic variant of the library only)
    void* checked_comm = nullptr;      // rslam_shard_frame: the communicator whose size / rank have been checked, with the (rank, world) it was checked for
    int32_t checked_rank = -1, checked_world = -1;
    // drop-in API: the caller's covariance buffers (p_k_km1 in, p_k_k out) are page-locked on first use so that the two
    // 26 MB transfers of a frame run at the PCIe rate instead of through the runtime's pageable staging
    // (ONE table for both directions: a caller that swaps its in / out buffers every frame finds both registered)
    struct HostReg { const void* p = nullptr; size_t bytes = 0; bool ok = false; unsigned long long used = 0; } reg[4];
    unsigned long long reg_clock = 0;
    int reg_n = -1;                    // the state dimension the registrations were made for (a resize re-allocates the caller's matrices)
    int rep_status = 0, rep_front = 0, rep_sticky = 0;   // status words of the frame in flight once read_status has taken them off the device
    bool frame_checked = true;         // read_status has (not) looked at the update stage in flight yet
    // Staged route of large systems (staged_kernels.hip): three side streams with disjoint CU masks -- the S stage (pivot chain
    // of the innovation covariance), the group inverses, the R stage (everything with n rows) -- joined to the context's
    // stream by events; created on first use, absent (ok == false) when the runtime refuses CU masks
    struct Staged {
        bool tried = false, ok = false;
        hipStream_t sS = nullptr, sI = nullptr, sR = nullptr;
        int cus_S = 0, cus_I = 0;
        static constexpr int MAX_GROUPS = 16;
        hipEvent_t e0 = nullptr, eGrp[MAX_GROUPS] = {}, eInv[MAX_GROUPS] = {}, eR[MAX_GROUPS] = {};
        DevBuf<double> d_M, d_Mt;      // group inverses L_gg^-1 (RP x RP, the groups' diagonal blocks) and their transposes
        int updates = 0;               // updates that took the route (rslam_update_mode reports it)
    } staged;
};

// bookkeeping of every path that puts an update stage on the stream (eager or graph replay)
static void unpin_host_buffers(rslam_ctx* c);
static void staged_release(rslam_ctx* c);
static int staged_env_int(const char* name, int dflt);
static bool sweep_is_persistent(const rslam_ctx* c)
{
    if (c->steps_frames_left > 0) return false;      // a hand-over of the persistent sweep timed out recently (see read_status)
    SystemDims d; d.n = c->n; d.NP = c->NP; d.RP = c->RP; d.ldA = c->ldA;
    return sweep_persistent_eligible(d);
}

static void mark_update_enqueued(rslam_ctx* c, const int32_t* d_sup)
{
    // (the persistent sweep is one launch sized for the largest inlier count, the launch-per-step sweep is sized by the
    //  host from this frame's own counts: no launch sequence can turn out too short)
    c->last_sup = d_sup;
    c->frame_checked = false;
    c->rep_status = c->rep_front = c->rep_sticky = 0;
    c->have_post = true;
}

static void invalidate_graph(rslam_ctx* c)
{
    for (int k = 0; k < 3; ++k) {
        if (c->graph_exec[k]) { (void)hipGraphExecDestroy(c->graph_exec[k]); c->graph_exec[k] = nullptr; }
        if (c->graph[k]) { (void)hipGraphDestroy(c->graph[k]); c->graph[k] = nullptr; }
        c->graph_valid[k] = false;
    }
}

extern "C" int rslam_destroy(rslam_ctx* c);

// Largest squared undistorted radius at which the scoring kernel's six Newton steps (camera_model.h distort_fm_score) are at
// the fixed point of ExtendKF::distort_fm's iteration: found by running the iteration on a geometric grid of radii (host
// arithmetic; the device's slope reciprocals differ by 2^-26, which moves the contraction by that much, not the fixed point),
// with a 10 % safety factor on the radius.  Pairs beyond it take the reference's ten-step sequence.
static double score_fast_radius2(double k1, double k2, double dx, double dy, int nRows, int nCols)
{
    auto step = [&](double rd, double ru) {
        const double v2 = rd * rd, v4 = v2 * v2;
        const double f = rd + k1 * (v2 * rd) + k2 * (v4 * rd) - ru, fp = 1 + 3 * k1 * v2 + 5 * k2 * v4;
        return rd - f / fp;
    };
    const double diag = sqrt((double)nCols * dx * (double)nCols * dx + (double)nRows * dy * (double)nRows * dy);
    double good = 0.0;
    for (double ru = 1e-3 * diag; ru < 64.0 * diag; ru *= 1.02) {
        const double ru2 = ru * ru;
        double rd = ru / (1 + k1 * ru2 + k2 * ru2 * ru2);
        for (int k = 0; k < 6; ++k) rd = step(rd, ru);
        double ref = rd;
        for (int k = 0; k < 60; ++k) ref = step(ref, ru);
        if (!(fabs(rd - ref) <= 1e-14 * fabs(ref)) || !(ref > 0.0)) break;      // (also a model whose iteration goes astray)
        good = ru;
    }
    good *= 0.9;
    return good * good;
}

extern "C" const char* rslam_version(void) { return "rslam-hip 0.2 (gfx950)"; }

extern "C" const char* rslam_error_string(int code)
{
    switch (code) {
    case RSLAM_OK: return "ok";
    case RSLAM_ERR_ARG: return "invalid argument";
    case RSLAM_ERR_NO_DEVICE: return "no usable HIP device (there is no CPU fallback)";
    case RSLAM_ERR_HIP: return "HIP runtime error";
    case RSLAM_ERR_STATE: return "call order violated";
    case RSLAM_ERR_REF_ASSERT: return "input on which the reference hits an Eigen assertion (Tracking.cpp:498)";
    case RSLAM_ERR_NOT_SPD: return "innovation covariance not positive definite";
    case RSLAM_ERR_IC_NOT_VISIBLE: return "individually compatible flag on a feature that is not visible";
    case RSLAM_ERR_COMM: return "RCCL could not be loaded or a collective failed";
    default: return "unknown error";
    }
}

extern "C" int rslam_create(const rslam_config* cfg, int device, rslam_ctx** out)
{
    if (!cfg || !out) return RSLAM_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return RSLAM_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return RSLAM_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return RSLAM_ERR_NO_DEVICE;
    rslam_ctx* c = new rslam_ctx();
    c->cfg = *cfg;
    c->cam.k1 = cfg->cam.k1; c->cam.k2 = cfg->cam.k2; c->cam.Cx = cfg->cam.Cx; c->cam.Cy = cfg->cam.Cy;
    c->cam.f = cfg->cam.f; c->cam.dx = cfg->cam.dx; c->cam.dy = cfg->cam.dy;
    c->cam.inv_dx = 1.0 / cfg->cam.dx; c->cam.inv_dy = 1.0 / cfg->cam.dy; c->cam.f_ku = cfg->cam.f * (1.0 / cfg->cam.dx);
    c->cam.nRows = cfg->cam.nRows; c->cam.nCols = cfg->cam.nCols;
    c->cam.ru2_fast = score_fast_radius2(cfg->cam.k1, cfg->cam.k2, cfg->cam.dx, cfg->cam.dy, cfg->cam.nRows, cfg->cam.nCols);
    c->device = device;
    c->li_skip = cfg->compat != 0;
#if defined(RSLAM_DEBUG)
    if (const char* e = getenv("RSLAM_LI_SKIP")) c->li_skip = atoi(e) != 0;     // tests: the guard of the sequence without a low-innovation sweep
#endif
    // every failure path below goes through rslam_destroy, which releases whatever exists so far
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { c->own_stream = nullptr; (void)rslam_destroy(c); return RSLAM_ERR_HIP; }
    c->stream = c->own_stream;
    if (init_kernel_attributes() != 0 || init_kernel_attributes2() != 0) { (void)rslam_destroy(c); return RSLAM_ERR_HIP; }
    int n_ev = 0;
    for (; n_ev < EV_COUNT; ++n_ev) if (hipEventCreate(&c->ev[n_ev]) != hipSuccess) break;
    c->ev_ok = (n_ev == EV_COUNT);
    if (!c->ev_ok) for (int i = 0; i < n_ev; ++i) (void)hipEventDestroy(c->ev[i]);
    memset(&c->times, 0, sizeof(c->times));
    if (c->d_sel.ensure(SEL_COUNT) < 0 || c->d_T.ensure(32) < 0 || c->d_sweep_flags.ensure(2 * SWEEP_FLAG_INTS) < 0) { (void)rslam_destroy(c); return RSLAM_ERR_HIP; }
    if (hipMemsetAsync(c->d_sweep_flags.p, 0, sizeof(int32_t) * 2 * SWEEP_FLAG_INTS, c->stream) != hipSuccess ||
        hipMemsetAsync(c->d_sel.p, 0, sizeof(int32_t) * SEL_COUNT, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) { (void)rslam_destroy(c); return RSLAM_ERR_HIP; }
    *out = c;
    return RSLAM_OK;
}

extern "C" int rslam_destroy(rslam_ctx* c)
{
    if (!c) return RSLAM_ERR_ARG;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    invalidate_graph(c);
    c->d_type.release(); c->d_vis.release(); c->d_hash.release(); c->d_hash2.release(); c->d_ic.release();
    c->d_li.release(); c->d_hi.release(); c->d_mtype.release();
    c->d_off.release(); c->d_mfeat.release(); c->d_moff.release(); c->d_mith.release(); c->d_miph.release();
    c->d_mzsrc.release(); c->d_rank_of.release(); c->d_pos.release(); c->d_nhyp.release(); c->d_sup.release();
    c->d_possup.release(); c->d_lilist.release(); c->d_hilist.release(); c->d_sel.release();
    c->d_masks.release(); c->d_posmask.release();
    c->d_xpred.release(); c->d_Ppred.release(); c->d_h.release(); c->d_h2.release(); c->d_H13.release();
    c->d_H13b.release(); c->d_S.release(); c->d_S2.release(); c->d_z.release(); c->d_wv.release();
    c->d_W.release(); c->d_A.release(); c->d_Y.release(); c->d_Linv.release(); c->d_x1.release(); c->d_x2.release();
    c->d_P.release(); c->d_T.release(); c->d_probe.release(); c->d_FQ.release(); c->d_tile_order.release();
    c->d_mapcoef.release(); c->d_lin.release(); c->d_first.release(); c->d_sweep_flags.release();
    c->d_sup_local.release(); c->d_sup_all.release(); c->d_shard_key.release(); c->d_macro_order.release(); c->d_small_order.release();
    c->d_patches.release(); c->d_corr.release(); c->d_image.release(); c->d_stage.release(); c->d_sc.release(); c->d_hctx.release(); c->d_Y1.release(); c->d_Gd.release();
    c->d_rec.release(); c->d_rec_patch.release(); c->d_slot.release(); c->d_xyz_src.release(); c->d_pstatus.release();
    unpin_host_buffers(c);
    staged_release(c);
    if (c->ev_ok) for (int i = 0; i < EV_COUNT; ++i) (void)hipEventDestroy(c->ev[i]);
    if (c->h_counts) (void)hipHostFree(c->h_counts);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return RSLAM_OK;
}

extern "C" int rslam_set_stream(rslam_ctx* c, void* hip_stream)
{
    if (!c) return RSLAM_ERR_ARG;
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    invalidate_graph(c);
    return RSLAM_OK;
}

extern "C" int rslam_enable_timing(rslam_ctx* c, int on)
{
    if (!c) return RSLAM_ERR_ARG;
    c->timing = on ? 1 : 0;
    return RSLAM_OK;
}

// ------------------------------------------------------------------------
// uploads
// ------------------------------------------------------------------------
// validate a layout, size the per-frame buffers, upload the layout tables
static int set_layout(rslam_ctx* c, const rslam_layout* lay)
{
    if (!lay || lay->n < 13 || lay->L < 0 || (lay->L > 0 && (!lay->type || !lay->offset))) return RSLAM_ERR_ARG;
    const int n = lay->n, L = lay->L;
    for (int i = 0; i < L; ++i) {
        if (lay->type[i] > 1) return RSLAM_ERR_ARG;
        const int w = lay->type[i] == RSLAM_FEAT_INVERSE_DEPTH ? 6 : 3;
        if (lay->offset[i] < 13 || lay->offset[i] + w > n) return RSLAM_ERR_ARG;
    }
    HIPCHK(hipSetDevice(c->device));
    const int NP = round_up(n, 64);
    if (n != c->n || L != c->L) invalidate_graph(c);
    c->n = n; c->NP = NP; c->L = L;
    c->h_type.assign(lay->type, lay->type + L);
    c->h_off.assign(lay->offset, lay->offset + L);
    c->h_vis.assign((size_t)L, 0);
    int re = 0, r;
#define ENS(buf, cnt) do { r = (buf).ensure(cnt); if (r < 0) return RSLAM_ERR_HIP; re |= r; } while (0)
    ENS(c->d_type, L); ENS(c->d_off, L); ENS(c->d_vis, L); ENS(c->d_hash, L); ENS(c->d_hash2, L);
    ENS(c->d_ic, L); ENS(c->d_li, L); ENS(c->d_hi, L); ENS(c->d_rank_of, L);
    ENS(c->d_xpred, NP); ENS(c->d_x1, NP); ENS(c->d_x2, NP); ENS(c->d_FQ, 338); ENS(c->d_Y1, 4 * (size_t)NP); ENS(c->d_Gd, 34 * (size_t)L);
    ENS(c->d_Ppred, (size_t)NP * NP); ENS(c->d_P, (size_t)NP * NP);
    ENS(c->d_h, 2 * (size_t)L); ENS(c->d_h2, 2 * (size_t)L); ENS(c->d_H13, 26 * (size_t)L); ENS(c->d_H13b, 26 * (size_t)L);
    ENS(c->d_S, 4 * (size_t)L); ENS(c->d_S2, 4 * (size_t)L); ENS(c->d_z, 2 * (size_t)L);
    if (re) { invalidate_graph(c); c->have_post = false; c->have_state = false; }
    hipStream_t s = c->stream;
    if (L > 0) {
        HIPCHK(hipMemcpyAsync(c->d_type.p, lay->type, L, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(c->d_off.p, lay->offset, sizeof(int32_t) * L, hipMemcpyHostToDevice, s));
    }
    return RSLAM_OK;
}

// Page-lock a caller buffer the drop-in API moves every frame (same pointer and size as last time: nothing to do).  A
// registration that fails -- the caller has pinned it already, the range is not registrable -- is not an error: the copy
// then takes the pageable path as before.  Only when the caller asked for it (rslam_config.reserved & RSLAM_PIN_HOST_COV:
// it promises the buffers' lifetime, include/rslam.h).
static void unpin_host_buffers(rslam_ctx* c)
{
    for (auto& r : c->reg) {
        // (a buffer the caller has freed in the meantime fails to unregister: not an error, and no sticky error is left behind)
        if (r.ok && hipHostUnregister(const_cast<void*>(r.p)) != hipSuccess) (void)hipGetLastError();
        r = rslam_ctx::HostReg{};
    }
}

static void pin_host_buffer(rslam_ctx* c, const void* p, size_t bytes)
{
    if (!(c->cfg.reserved & RSLAM_PIN_HOST_COV)) return;
    // a change of the state dimension means the caller's matrices were re-allocated (Map::map_management resizes x_k_k /
    // p_k_k): every registration is stale then, even one whose address and size come back later
    if (c->reg_n != c->n) { unpin_host_buffers(c); c->reg_n = c->n; }
    rslam_ctx::HostReg* slot = nullptr;
    for (auto& r : c->reg) {
        if (r.p == p && r.bytes == bytes) { r.used = ++c->reg_clock; return; }     // known (registered, or known not to be registrable)
        if (!slot || r.used < slot->used) slot = &r;                                  // least recently used (an empty one first)
    }
    if (slot->ok && hipHostUnregister(const_cast<void*>(slot->p)) != hipSuccess) (void)hipGetLastError();
    slot->p = p; slot->bytes = bytes; slot->used = ++c->reg_clock;
    slot->ok = (hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterDefault) == hipSuccess);
    if (!slot->ok) (void)hipGetLastError();            // (leave no sticky error behind)
}

// host (x, P) -> padded device buffers
static int upload_xp(rslam_ctx* c, const double* x, const double* P, double* d_x, double* d_Pm)
{
    hipStream_t s = c->stream;
    const int n = c->n, NP = c->NP;
    HIPCHK(hipMemsetAsync(d_x, 0, sizeof(double) * NP, s));
    HIPCHK(hipMemcpyAsync(d_x, x, sizeof(double) * n, hipMemcpyHostToDevice, s));
    pin_host_buffer(c, P, sizeof(double) * (size_t)n * n);
    if (NP != n) {
        // one linear transfer, then the leading dimension n -> NP and the zero padding on the device
        if (c->d_stage.ensure((size_t)n * n) < 0) return RSLAM_ERR_HIP;
        HIPCHK(hipMemcpyAsync(c->d_stage.p, P, sizeof(double) * (size_t)n * n, hipMemcpyHostToDevice, s));
        launch_repitch(s, c->d_stage.p, n, d_Pm, NP, n, NP, n, NP);
        HIPCHK(hipGetLastError());
    } else {
        HIPCHK(hipMemcpyAsync(d_Pm, P, sizeof(double) * (size_t)n * n, hipMemcpyHostToDevice, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    return RSLAM_OK;
}

static int upload_state(rslam_ctx* c, const rslam_layout* lay, const double* x_pred, const double* P_pred)
{
    if (!x_pred || !P_pred) return RSLAM_ERR_ARG;
    int rc = set_layout(c, lay);
    if (rc) return rc;
    rc = upload_xp(c, x_pred, P_pred, c->d_xpred.p, c->d_Ppred.p);
    if (rc) return rc;
    c->have_state = true; c->have_meas = false; c->predicted = false; c->pht_done = false; c->dedup_done = false; c->masks_all = 0;
    return RSLAM_OK;
}

static const int32_t* tile_order(rslam_ctx* c, int NP);

// Builds the gather tables that replace Converter::find/select/repmat
// (Converter.cpp:210-287 as used at Tracking.cpp:361-397,413-415,443-448).
static int upload_measurements(rslam_ctx* c, const double* z, const uint8_t* ic, const double* draws, int n_draws,
                               bool check_visible)
{
    if (!c->have_state) return RSLAM_ERR_STATE;
    if (!z || !ic || (n_draws > 0 && !draws) || n_draws < 0) return RSLAM_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    const int L = c->L;
    std::vector<int32_t> mfeat, moff, mith, miph, mzsrc, rank_of((size_t)L, -1), id_list, euc_list;
    std::vector<uint8_t> mtype;
    for (int i = 0; i < L; ++i) {
        if (!ic[i]) continue;
        if (check_visible && !c->h_vis[i]) return RSLAM_ERR_IC_NOT_VISIBLE;   // matching() needs h, Tracking.cpp:293
        rank_of[i] = (int32_t)mfeat.size();
        mfeat.push_back(i);
        (c->h_type[i] == RSLAM_FEAT_INVERSE_DEPTH ? id_list : euc_list).push_back(i);
    }
    const int m = (int)mfeat.size(), m_id = (int)id_list.size(), m_euc = (int)euc_list.size();
    if (m >= 4096) return RSLAM_ERR_ARG;
    if (c->cfg.compat && m_euc > 0 && m_euc != m_id) return RSLAM_ERR_REF_ASSERT;     // Q2
    moff.resize(m); mith.resize(m); miph.resize(m); mzsrc.resize(m); mtype.resize(m);
    int jj = 0, je = 0;
    for (int j = 0; j < m; ++j) {
        const int f = mfeat[j];
        moff[j] = c->h_off[f]; mtype[j] = c->h_type[f];
        if (mtype[j] == RSLAM_FEAT_INVERSE_DEPTH) {
            if (c->cfg.compat) {        // Q1: anglesi mapped onto ri_v (Tracking.cpp:448)
                const int a = 2 * jj, b = 2 * jj + 1;
                mith[j] = c->h_off[id_list[a / 3]] + a % 3;
                miph[j] = c->h_off[id_list[b / 3]] + b % 3;
            } else { mith[j] = moff[j] + 3; miph[j] = moff[j] + 4; }
            mzsrc[j] = f; ++jj;
        } else {
            mith[j] = miph[j] = 0;
            mzsrc[j] = c->cfg.compat ? id_list[je] : f;   // Q2: z_id used for Cartesian residuals (Tracking.cpp:498)
            ++je;
        }
    }
    const int H = n_draws;
    const int words = (m + 63) / 64;
    const int RP = round_up(2 * m, 64);
    const int ldA = RP + c->NP + 64;
    if (m != c->m || H != c->H || RP != c->RP) invalidate_graph(c);
    c->m = m; c->H = H; c->words = words; c->RP = RP; c->ldA = ldA; c->m_id = m_id; c->m_euc = m_euc;
    // hypothesis -> matched rank: floor(t * N)-th IC feature (Tracking.cpp:412-415; Q3 guard)
    std::vector<int32_t> pos((size_t)H, 0);
    for (int i = 0; i < H; ++i) {
        int p = (int)floor(draws[i] * (double)m);
        if (p >= m) p = m - 1;
        if (p < 0) p = 0;
        pos[i] = p;
    }
    // n_hyp after an improvement, tabulated with the host libm (Tracking.cpp:531-532)
    std::vector<int32_t> nhyp((size_t)m + 1, 0);
    for (int sidx = 1; sidx <= m; ++sidx) {
        const double epsilon = 1 - ((double)sidx / (double)m);
        nhyp[sidx] = (int32_t)ceil((log(1 - c->cfg.p_success)) / (log(1 - (1 - epsilon))));
    }
    int re = 0, r;
    ENS(c->d_mfeat, m); ENS(c->d_moff, m); ENS(c->d_mith, m); ENS(c->d_miph, m); ENS(c->d_mzsrc, m); ENS(c->d_mtype, m);
    ENS(c->d_pos, H); ENS(c->d_nhyp, (size_t)m + 1); ENS(c->d_sup, H); ENS(c->d_possup, m);
    ENS(c->d_masks, (size_t)H * (words ? words : 1)); ENS(c->d_posmask, (size_t)m * (words ? words : 1));
    ENS(c->d_lilist, m); ENS(c->d_hilist, m);
    ENS(c->d_sc, 4 * (size_t)m); ENS(c->d_hctx, 16 * (size_t)(m ? m : 1)); ENS(c->d_wv, 2 * (size_t)m); ENS(c->d_W, (size_t)c->NP * 2 * (m ? m : 1));
    ENS(c->d_A, (size_t)ldA * (RP ? RP : 1)); ENS(c->d_Y, (size_t)ldA * (RP ? RP : 1)); ENS(c->d_Linv, (size_t)64 * 64 * (RP / 64 ? RP / 64 : 1));
    if (re) invalidate_graph(c);
#undef ENS
    hipStream_t s = c->stream;
#define H2D(dst, vec) do { if (!(vec).empty()) HIPCHK(hipMemcpyAsync((dst).p, (vec).data(), sizeof((vec)[0]) * (vec).size(), hipMemcpyHostToDevice, s)); } while (0)
    H2D(c->d_mfeat, mfeat); H2D(c->d_moff, moff); H2D(c->d_mith, mith); H2D(c->d_miph, miph); H2D(c->d_mzsrc, mzsrc);
    H2D(c->d_mtype, mtype); H2D(c->d_rank_of, rank_of); H2D(c->d_pos, pos); H2D(c->d_nhyp, nhyp);
#undef H2D
    if (L > 0) {
        HIPCHK(hipMemcpyAsync(c->d_z.p, z, sizeof(double) * 2 * L, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(c->d_ic.p, ic, L, hipMemcpyHostToDevice, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    (void)tile_order(c, c->NP);          // never inside a graph capture
    c->have_meas = true; c->pht_done = false; c->dedup_done = false; c->masks_all = 0;
    return RSLAM_OK;
}

// ------------------------------------------------------------------------
// stage enqueue
// ------------------------------------------------------------------------
static inline void mark(rslam_ctx* c, int ev) { if (c->timing && c->ev_ok) (void)hipEventRecord(c->ev[ev], c->stream); }

static ScoreTables tables(rslam_ctx* c)
{
    ScoreTables t;
    t.feat = c->d_mfeat.p; t.off = c->d_moff.p; t.type = c->d_mtype.p;
    t.ith = c->d_mith.p; t.iph = c->d_miph.p; t.zsrc = c->d_mzsrc.p; t.sc = c->d_sc.p; t.hctx = c->d_hctx.p;
    return t;
}

static int enqueue_predict(rslam_ctx* c)
{
    if (!c->have_state) return RSLAM_ERR_STATE;
    hipStream_t s = c->stream;
    mark(c, EV_START);
    // no previous h: Map.cpp:51 resets it every frame; the kernel also zeroes the frame scalars
    launch_predict(s, c->cam, c->d_xpred.p, c->d_Ppred.p, c->NP, c->L, c->d_type.p, c->d_off.p, nullptr, nullptr,
                   c->d_h.p, c->d_hash.p, c->d_vis.p, c->d_H13.p, c->d_S.p, 1.0 /* features_info[i].R = I, Map.cpp:310 */,
                   c->d_sel.p);
    mark(c, EV_PREDICT);
    c->predicted = true; c->pht_done = false; c->dedup_done = false; c->masks_all = 0; c->patches_valid = false;
    return RSLAM_OK;
}

static int enqueue_score(rslam_ctx* c, int hb, int he, int32_t* d_sup)
{
    if (!c->predicted || !c->have_meas) return RSLAM_ERR_STATE;
    if (hb < 0 || he > c->H || hb > he || !d_sup) return RSLAM_ERR_ARG;
    hipStream_t s = c->stream;
    if (!c->pht_done) {
        launch_pht(s, c->d_Ppred.p, c->NP, c->d_mfeat.p, c->m, nullptr, c->d_H13.p, c->d_off.p, c->d_type.p,
                   c->d_W.p, c->NP, c->d_S.p, c->d_z.p, c->d_h.p, c->d_hash.p, c->d_wv.p, c->d_sel.p + SEL_STATUS_FRONT,
                   c->d_xpred.p, c->d_mith.p, c->d_miph.p, c->d_sc.p, nullptr, nullptr, c->d_hctx.p);
        c->pht_done = true;
    }
    mark(c, EV_PHT);
    if (c->m > 0 && he > hb) {
        if (c->cfg.dedup) {
            if (!c->dedup_done) {
                launch_score(s, c->cam, c->d_xpred.p, c->d_W.p, c->NP, c->d_wv.p, tables(c), c->d_z.p, c->m, c->words,
                             nullptr, c->m, c->cfg.sigma_z, c->d_possup.p, c->d_posmask.p);
                c->dedup_done = true;
            }
            c->masks_all = 2;
            launch_map_support(s, c->d_possup.p, c->d_pos.p, hb, he, d_sup);
        } else {
            launch_score(s, c->cam, c->d_xpred.p, c->d_W.p, c->NP, c->d_wv.p, tables(c), c->d_z.p, c->m, c->words,
                         c->d_pos.p + hb, he - hb, c->cfg.sigma_z, d_sup + hb, c->d_masks.p + (size_t)hb * c->words);
            c->masks_all = (hb == 0 && he == c->H) ? 1 : 0;
        }
    } else if (he > hb) {
        HIPCHK(hipMemsetAsync(d_sup + hb, 0, sizeof(int32_t) * (he - hb), s));
    }
    mark(c, EV_SCORE);
    return RSLAM_OK;
}

// XCD-aware tile order of the covariance rank update for the current NP (built once per size)
static const int32_t* tile_order(rslam_ctx* c, int NP)
{
    const int nT = NP / 64;
    if (nT <= 0) return nullptr;
    // (round 2 kept the plain row-major order beyond 48 tile rows -- 3 % faster with the K loop of the time; with the LDS-DMA
    //  loop the XCD-aware order wins there too: C5 frame 2.353 -> 2.320 ms, PMC: the row-major pass fetched 4.2 GB per launch)
    if (c->tile_order_nT != nT) {
        std::vector<int32_t> order;
        make_rank_update_order(nT, order);
        if (c->d_tile_order.ensure(order.size()) < 0) return nullptr;
        if (hipMemcpy(c->d_tile_order.p, order.data(), sizeof(int32_t) * order.size(), hipMemcpyHostToDevice) != hipSuccess)
            return nullptr;
        c->tile_order_nT = nT;
        invalidate_graph(c);
    }
    return c->d_tile_order.p;
}



// Tile lists of the macro-tile rank update for the current map size (built once per size); false: not for this size
static bool macro_lists(rslam_ctx* c)
{
    const int nT = c->NP / 64;
    if (nT < 32) return false;                             // small maps: the fused persistent sweep or the 64 x 64 form
    if (staged_env_int("RSLAM_NO_MACRO", 0)) return false;   // (measurement; diagnostic variant only)
    if (!c->macro_attr) {
        if (init_macro_kernel_attributes() != 0) { (void)hipGetLastError(); return false; }
        c->macro_attr = true;
    }
    if (c->macro_nT != nT) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return false; }
        std::vector<int32_t> macro, small;
        make_macro_order(nT, prop.multiProcessorCount, macro, small);
        if (c->d_macro_order.ensure(macro.size()) < 0 || c->d_small_order.ensure(small.size()) < 0) return false;
        if (!macro.empty() && hipMemcpy(c->d_macro_order.p, macro.data(), sizeof(int32_t) * macro.size(), hipMemcpyHostToDevice) != hipSuccess) return false;
        if (!small.empty() && hipMemcpy(c->d_small_order.p, small.data(), sizeof(int32_t) * small.size(), hipMemcpyHostToDevice) != hipSuccess) return false;
        c->macro_nT = nT; c->n_macro = (int)macro.size(); c->n_small = (int)small.size();
    }
    return c->n_macro > 0;
}

// One pass P' = sym(P) - Y(:, k0 .. k0 + K) Y(..)^T of an update whose width the host knows (large-map route), on stream x:
// macro tiles for whole rounds + the 64 x 64 form for the rest when the map is large enough, else the 64 x 64 form alone.
// `xu` (nullable): the x update of the LAST pass -- it rides in the 64 x 64 launch, or runs as a launch of its own in front of
// the macro tiles (whose first block column then finds Jnorm written).
static int enqueue_rank_pass(rslam_ctx* c, hipStream_t x, const double* Pin, double* Pout, const double* Ycols, long ldy, int K,
                             int slot_k, int slot_nblk, const double* Tq, const XuArgs* xu, const MatArgs* mat, int mirror_flag)
{
    int32_t* sel = c->d_sel.p;
    const int32_t* order = (c->tile_order_nT == c->NP / 64) ? c->d_tile_order.p : nullptr;
    XuArgs plain{};
    plain.mirror_known = mirror_flag;
    plain.token = (slot_k == SEL_K_LI) ? 1 : 2;
    // (macro tiles pay from about 20 column blocks on: their epilogue -- four quadrants behind a 128 x 128 x K loop, one
    //  workgroup per compute unit with nothing to overlap it -- is 25 us per tile whatever K; measured at C5: K = 1600 -20 us,
    //  K = 1152 + 512 (the two updates of the corrected mode) +22 us against the 64 x 64 form)
    if (K >= 64 * staged_env_int("RSLAM_MACRO_MIN_BLOCKS", 20) && K % 64 == 0 && macro_lists(c)) {
        // The x update (K9: riders) rides in the 64 x 64 launch of the partial round, which therefore goes FIRST: its ~400 tile
        // pairs leave a fifth of the device's workgroup slots to the riders, and Jnorm is written when the macro launch's first
        // block column reaches its epilogue (until round 5's last day the riders were a launch of their own in front: 18 us).
        const bool riders = xu && xu->groups > 0;
        const bool ride_small = riders && c->n_small > 0 && staged_env_int("RSLAM_RIDERS_IN_SMALL", 1);
        if (riders && !ride_small) {
            XuArgs alone = *xu;
            alone.riders_only = 1; alone.Y1out = nullptr; alone.defer_flag = nullptr;
            launch_rank_update(x, c->NP, Pin, c->NP, Ycols, ldy, sel, slot_nblk, K, Pout, c->NP, order, nullptr, slot_k, &alone, nullptr);
        }
        if (ride_small) {
            XuArgs with = *xu;
            with.mirror_known = mirror_flag;
            launch_rank_update(x, c->NP, Pin, c->NP, Ycols, ldy, sel, slot_nblk, K, Pout, c->NP, c->d_small_order.p, Tq, slot_k, &with, mat,
                               c->n_small);
        }
        launch_rank_update_macro(x, Pin, c->NP, Ycols, ldy, K, Pout, c->NP, c->d_macro_order.p, c->n_macro, sel, slot_k, Tq,
                                 mirror_flag, plain.token, mat);
        if (c->n_small > 0 && !ride_small)
            launch_rank_update(x, c->NP, Pin, c->NP, Ycols, ldy, sel, slot_nblk, K, Pout, c->NP, c->d_small_order.p, Tq, slot_k, &plain, mat,
                               c->n_small);
    } else {
        XuArgs with = xu ? *xu : plain;
        with.mirror_known = mirror_flag;
        launch_rank_update(x, c->NP, Pin, c->NP, Ycols, ldy, sel, slot_nblk, K, Pout, c->NP, order, Tq, slot_k, &with, mat);
    }
    HIPCHK(hipGetLastError());
    return RSLAM_OK;
}

// ------------------------------------------------------------------------
// Staged route of large systems (see staged_kernels.hip)
// ------------------------------------------------------------------------
static void staged_release(rslam_ctx* c)
{
    rslam_ctx::Staged& st = c->staged;
    for (hipStream_t* sp : {&st.sS, &st.sI, &st.sR}) if (*sp) { (void)hipStreamSynchronize(*sp); (void)hipStreamDestroy(*sp); *sp = nullptr; }
    if (st.e0) { (void)hipEventDestroy(st.e0); st.e0 = nullptr; }
    for (int g = 0; g < rslam_ctx::Staged::MAX_GROUPS; ++g) {
        if (st.eGrp[g]) { (void)hipEventDestroy(st.eGrp[g]); st.eGrp[g] = nullptr; }
        if (st.eInv[g]) { (void)hipEventDestroy(st.eInv[g]); st.eInv[g] = nullptr; }
        if (st.eR[g]) { (void)hipEventDestroy(st.eR[g]); st.eR[g] = nullptr; }
    }
    st.d_M.release(); st.d_Mt.release();
    st.ok = false;
}

static int staged_env_int(const char* name, int dflt)
{
#if defined(RSLAM_DEBUG)
    const char* v = getenv(name);          // measurement switches exist in the diagnostic variant only
    if (v && *v) return atoi(v);
#else
    (void)name;
#endif
    return dflt;
}

// Streams with disjoint CU masks.  The first cus_S bits of a mask are compute units spread evenly over the XCDs and shader
// engines (scripts/probes/cu_mask.hip: 32 bits = 4 CUs of each XCD), the next cus_I bits likewise, the R stage gets the rest.
static bool staged_init(rslam_ctx* c)
{
    rslam_ctx::Staged& st = c->staged;
    if (st.tried) return st.ok;
    st.tried = true;
    if (staged_env_int("RSLAM_NO_STAGED", 0)) return false;
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return false; }
    const int cus = prop.multiProcessorCount;
    st.cus_S = staged_env_int("RSLAM_STAGED_CUS_S", 32);
    st.cus_I = staged_env_int("RSLAM_STAGED_CUS_I", 8);
    if (cus < 128 || st.cus_S < 8 || st.cus_I < 8 || st.cus_S + st.cus_I > cus / 2) return false;
    const int words = (cus + 31) / 32;
    std::vector<uint32_t> mS(words, 0), mI(words, 0), mR(words, 0);
    for (int i = 0; i < cus; ++i) {
        std::vector<uint32_t>& m = i < st.cus_S ? mS : (i < st.cus_S + st.cus_I ? mI : mR);
        m[i / 32] |= 1u << (i % 32);
    }
    bool ok = hipExtStreamCreateWithCUMask(&st.sS, (uint32_t)words, mS.data()) == hipSuccess
           && hipExtStreamCreateWithCUMask(&st.sI, (uint32_t)words, mI.data()) == hipSuccess
           && hipExtStreamCreateWithCUMask(&st.sR, (uint32_t)words, mR.data()) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&st.e0, hipEventDisableTiming) == hipSuccess;
    for (int g = 0; ok && g < rslam_ctx::Staged::MAX_GROUPS; ++g)
        ok = hipEventCreateWithFlags(&st.eGrp[g], hipEventDisableTiming) == hipSuccess
          && hipEventCreateWithFlags(&st.eInv[g], hipEventDisableTiming) == hipSuccess
          && hipEventCreateWithFlags(&st.eR[g], hipEventDisableTiming) == hipSuccess;
    ok = ok && init_staged_kernel_attributes() == 0;
    if (!ok) { (void)hipGetLastError(); staged_release(c); return false; }
    st.ok = true;
    return true;
}

// Group boundaries of the blocked substitution of an update of nblk column blocks (groups of four diagonal blocks; a short
// last group joins the one before).  Empty = the route is not taken.
static std::vector<int> staged_groups(int nblk)
{
    std::vector<int> b;
    // Measured in round 5 (scripts/ab_c5.py, C5): with the launch-per-step S stage the route is SLOWER than the plain
    // launch-per-step sweep (2.47 against 2.10 ms per frame: the S stage alone takes 630-700 us on its compute units -- 25
    // pivot chains of ~19 us in 38 launches -- and the last group's inverse and solve stand behind it), so the product never
    // takes it; the diagnostic variant does when RSLAM_STAGED_MIN_BLOCKS says so (tests hold it to the oracle).  It needs the
    // one-launch S stage (the persistent chain at 13 us per block) to pay: NOTEBOOK.md, round 5.
    const int min_blocks = staged_env_int("RSLAM_STAGED_MIN_BLOCKS", 1 << 20);
    const int gs = staged_env_int("RSLAM_STAGED_GROUP", 4);
    if (nblk < min_blocks || nblk < 4 || gs < 1) return b;
    // the LAST group is short (its inverse and its solve stand between the end of the S stage and the rank update): the
    // groups are counted from the end, the first one takes the remainder
    const int tail = staged_env_int("RSLAM_STAGED_TAIL", 2);
    std::vector<int> rev;
    int k = nblk - tail;
    rev.push_back(nblk);
    for (; k > gs / 2; k -= gs) rev.push_back(k);
    rev.push_back(0);
    b.assign(rev.rbegin(), rev.rend());
    if ((int)b.size() - 1 > rslam_ctx::Staged::MAX_GROUPS) b.clear();
    return b;
}

// One EKF update of a large system on the staged route.  Everything is stream-ordered; the host enqueues each group's
// inverse / solve / update right behind the S-stage launch that finishes the group (a stream only starts what has been
// enqueued: with the whole S stage enqueued first the other streams' first launches came ~400 us late).  The context's
// stream waits for the side streams before the rank update, so what follows it is ordered behind the whole update.
static int enqueue_staged_update(rslam_ctx* c, const std::vector<int>& grp, const SystemDims& d, int slot_k, int slot_nblk, int nblk,
                                 const double* x_in, double* x_out, const double* Pin, double* Pout,
                                 int ev_f1, int ev_r0)
{
    rslam_ctx::Staged& st = c->staged;
    hipStream_t s = c->stream;
    int32_t* sel = c->d_sel.p;
    const long ldm = d.RP;
    if (st.d_M.ensure((size_t)d.RP * d.RP) < 0 || st.d_Mt.ensure((size_t)d.RP * d.RP) < 0) return RSLAM_ERR_HIP;
    const int G = (int)grp.size() - 1;
    double* A = c->d_A.p;
    double* Ys = c->d_Y.p;
    HIPCHK(hipEventRecord(st.e0, s));                                   // the system is assembled (prepare_system)
    HIPCHK(hipStreamWaitEvent(st.sS, st.e0, 0));
    int next = 1;                                                       // next group boundary to announce
    hipError_t err = hipSuccess;
    auto chk = [&](hipError_t e) { if (e != hipSuccess && err == hipSuccess) err = e; };
    // ---- S stage: the sweep of the innovation covariance alone, on its own compute units; behind the launch that finishes a
    //      group: its inverse (stream sI) and, behind that, Y_g = W_g M_g^T and the update of every later column block (stream sR)
    launch_s_stage(st.sS, d, sel, slot_k, slot_nblk, nblk, A, Ys, c->d_Linv.p, sel + SEL_STATUS, st.cus_S,
                   [&](int done) {
                       while (next <= G && grp[next] <= done) {
                           const int g = next - 1, b0 = grp[g], nb = grp[g + 1] - grp[g];
                           chk(hipEventRecord(st.eGrp[g], st.sS));
                           chk(hipStreamWaitEvent(st.sI, st.eGrp[g], 0));
                           launch_group_inverse(st.sI, b0, nb, Ys, d.ldA, c->d_Linv.p, st.d_M.p, st.d_Mt.p, ldm);
                           chk(hipEventRecord(st.eInv[g], st.sI));
                           chk(hipStreamWaitEvent(st.sR, st.eInv[g], 0));
                           launch_staged_Y(st.sR, d, b0, nb, A, Ys, st.d_M.p, ldm);
                           launch_staged_update(st.sR, d, b0, nb, nblk, A, Ys);
                           ++next;
                       }
                   });
    HIPCHK(err);
    HIPCHK(hipGetLastError());
    if (next <= G) return RSLAM_ERR_HIP;                                // (every group was announced: the S stage covers nblk blocks)
    HIPCHK(hipEventRecord(st.eR[0], st.sR));
    // ---- the covariance: one pass over the whole width on every compute unit, K9 with it, K11 its epilogue
    HIPCHK(hipStreamWaitEvent(s, st.eR[0], 0));
    if (ev_f1 >= 0) mark(c, ev_f1);
    if (ev_r0 >= 0) mark(c, ev_r0);
    double* Tq = c->d_T.p + (slot_k == SEL_K_LI ? 0 : 16);
    XuArgs xu{};
    xu.token = (slot_k == SEL_K_LI) ? 1 : 2;
    xu.groups = c->NP / 16;
    xu.d = d; xu.A = Ys; xu.x_in = x_in; xu.x_out = x_out; xu.T = Tq; xu.compat = c->cfg.compat;
    xu.flag = sel + SEL_XU_FLAG;
    xu.riders_first = c->k10_riders_first ? 1 : 0; xu.inject = c->k10_inject;
    const MatArgs mat{sel + SEL_LI_DEFER, c->d_Ppred.p, c->NP, c->d_Y1.p, c->NP, c->d_T.p};
    const int rc = enqueue_rank_pass(c, s, Pin, Pout, Ys + d.RP, d.ldA, 64 * nblk, slot_k, slot_nblk, Tq, &xu,
                                     slot_k == SEL_K_HI ? &mat : nullptr, 0);
    if (rc) return rc;
    ++st.updates;
    return RSLAM_OK;
}

static int enqueue_one_update(rslam_ctx* c, const int32_t* list, int slot_k, int slot_nblk, int host_blocks, const double* Wsrc,
                              const double* H13,
                              const double* z_h, const double* x_in, double* x_out, const double* Pin, double* Pout,
                              int ev_f0, int ev_f1, int ev_r0, int ev_r1)
{
    hipStream_t s = c->stream;
    SystemDims d; d.n = c->n; d.NP = c->NP; d.RP = c->RP; d.ldA = c->ldA;
    int32_t* sel = c->d_sel.p;
    // The persistent sweep assembles the stacked system itself; the launch-per-step sequence of large systems has a pass
    // of its own for that.
    const bool persistent = sweep_is_persistent(c);
    SysSrc src{list, H13, c->d_off.p, c->d_type.p, c->d_z.p, z_h, Wsrc, c->d_rank_of.p};
    if (!persistent)
        launch_prepare_system(s, d, list, sel, slot_k, slot_nblk, H13, c->d_off.p, c->d_type.p, c->d_z.p, z_h, c->d_A.p,
                              Wsrc, c->d_rank_of.p, nullptr);
    if (ev_f0 >= 0) mark(c, ev_f0);
    const double* Ysys = c->d_A.p;        // the system whose lower rows hold Y = P H^T L^-T and u^T after the sweep
    // K9 / K10 / K11 ride INSIDE the persistent sweep launch when the compute units it leaves idle can hold every tile pair
    // of P (kernels.hip, "tile workers"): the rank update then runs under the pivot chain instead of behind it.
    const bool fused = persistent && sweep_fused_eligible(d);
    const int32_t* order = (c->tile_order_nT == c->NP / 64) ? c->d_tile_order.p : nullptr;
    WorkerArgs wk{};
    if (fused) {
        wk.Pin = Pin; wk.ldp = c->NP; wk.Pout = Pout; wk.ldo = c->NP; wk.tile_order = order; wk.nT = c->NP / 64;
        wk.x_in = x_in; wk.x_out = x_out; wk.T = c->d_T.p; wk.compat = c->cfg.compat;
        wk.token = (slot_k == SEL_K_LI) ? 1 : 2;          // sel[] is zeroed at the start of a frame (predict_kernel)
        wk.xu_flag = sel + SEL_XU_FLAG;
        wk.li_done_slot = (slot_k == SEL_K_HI) ? SEL_NBLK_LI : -1;
        // Jnorm of the LI update at d_T, of the HI update behind it: the HI pass may still need the LI one (deferred covariance)
        wk.T = c->d_T.p + (slot_k == SEL_K_LI ? 0 : 16);
        wk.defer_flag = sel + SEL_LI_DEFER; wk.Y1 = c->d_Y1.p; wk.ldy1 = c->NP; wk.Ppred = c->d_Ppred.p; wk.T_li = c->d_T.p;
    }
    // Large systems whose block count the host knows (launch-per-step route): S stage and R stage side by side
    if (!persistent && c->RP > 0 && host_blocks <= c->RP / 64 && !(slot_k == SEL_K_LI && c->li_defer_host)) {
        const std::vector<int> grp = staged_groups(host_blocks);
        if (!grp.empty() && staged_init(c)) {
            const int rc = enqueue_staged_update(c, grp, d, slot_k, slot_nblk, host_blocks, x_in, x_out, Pin, Pout, ev_f1, ev_r0);
            if (rc) return rc;
            if (ev_r1 >= 0) mark(c, ev_r1);
            return RSLAM_OK;
        }
    }
    Ysys = launch_factor_sweep(s, d, sel, slot_k, slot_nblk, host_blocks, c->d_A.p, c->d_Y.p, c->d_Linv.p, sel + SEL_STATUS,
                               persistent ? c->d_sweep_flags.p : nullptr, persistent ? &src : nullptr, fused ? &wk : nullptr);
    if (ev_f1 >= 0) mark(c, ev_f1);
    if (c->RP <= 0) HIPCHK(hipMemcpyAsync(x_out, x_in, sizeof(double) * c->NP, hipMemcpyDeviceToDevice, s));
    if (ev_r0 >= 0) mark(c, ev_r0);
    if (!fused) {
        // K9 (x_k_k = x + Y u, quaternion normalisation, Jnorm) rides in the rank-update launch
        XuArgs xu{};
        xu.groups = c->RP > 0 ? c->NP / 16 : 0;
        // (Jnorm of the LI update at d_T, of the HI update behind it: the HI pass may still need the LI one -- deferred covariance)
        double* Tq = c->d_T.p + (slot_k == SEL_K_LI ? 0 : 16);
        xu.d = d; xu.A = Ysys; xu.x_in = x_in; xu.x_out = x_out; xu.T = Tq; xu.compat = c->cfg.compat;
        xu.token = (slot_k == SEL_K_LI) ? 1 : 2;              // sel[] is zeroed at the start of a frame (predict_kernel)
        xu.flag = sel + SEL_XU_FLAG;
        xu.riders_first = c->k10_riders_first ? 1 : 0; xu.inject = c->k10_inject;
        if (slot_k == SEL_K_LI && c->li_defer_host && xu.groups > 0) {
            // rank <= 4 (the host has read the count: enqueue_update): the x update alone; P_li stays implicit until the HI pass
            xu.riders_only = 1; xu.Y1out = c->d_Y1.p; xu.ldy1 = c->NP; xu.defer_flag = sel + SEL_LI_DEFER;
        }
        const MatArgs mat{sel + SEL_LI_DEFER, c->d_Ppred.p, c->NP, c->d_Y1.p, c->NP, c->d_T.p};
        if (!persistent && !xu.riders_only && c->RP > 0 && host_blocks > 0 && host_blocks <= c->RP / 64) {
            // the host knows this update's width (launch-per-step route): large maps take the macro-tile form
            const int rc = enqueue_rank_pass(c, s, Pin, Pout, Ysys + c->RP, c->ldA, 64 * host_blocks, slot_k, slot_nblk, Tq, &xu,
                                             slot_k == SEL_K_HI ? &mat : nullptr, 0);
            if (rc) return rc;
        } else {
            launch_rank_update(s, c->NP, Pin, c->NP, Ysys + c->RP, c->ldA, sel, slot_nblk, c->RP > 0 ? -1 : 0, Pout, c->NP,
                               order, c->RP > 0 ? Tq : nullptr, slot_k, &xu, slot_k == SEL_K_HI ? &mat : nullptr);
        }
    }
    if (ev_r1 >= 0) mark(c, ev_r1);
    return RSLAM_OK;
}

// The launch-per-step route's counts, host side (kernels.h HostCounts): page-locked, host-mapped memory, allocated on first use
static HostCounts host_counts_slot(rslam_ctx* c, int base)
{
#if defined(RSLAM_DEBUG)
    if (getenv("RSLAM_NO_HOST_COUNTS")) return HostCounts{nullptr, 0};          // tests: the copy + synchronise path
#endif
    if (!c->h_counts) {
        void* h = nullptr; void* d = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped) == hipSuccess && h) {
            memset(h, 0, 64);
            if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess && d) { c->h_counts = (int32_t*)h; c->d_counts = (int32_t*)d; }
            else (void)hipHostFree(h);
        }
        (void)hipGetLastError();
    }
    if (!c->h_counts) return HostCounts{nullptr, 0};
    return HostCounts{c->d_counts + base, ++c->count_seq};
}
// spin until the kernel that decides the count has written `seq` behind it; false: no mapped memory, or nothing within 50 ms
// (a device busy with somebody else's work): the caller then reads sel[] with a copy and a stream synchronisation
static bool host_counts_wait(rslam_ctx* c, int base, const HostCounts& hc, int32_t* count, int32_t* blocks)
{
    if (!hc.p || !c->h_counts) return false;
    int32_t* p = c->h_counts + base;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n(p + 2, __ATOMIC_ACQUIRE) != hc.seq) {
        if ((++spins & 4095u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) return false;
    }
    *count = __atomic_load_n(p + 0, __ATOMIC_RELAXED); *blocks = __atomic_load_n(p + 1, __ATOMIC_RELAXED);
    return true;
}

static int enqueue_update(rslam_ctx* c, const int32_t* d_sup)
{
    if (!c->predicted || !c->have_meas) return RSLAM_ERR_STATE;
    if (!d_sup) return RSLAM_ERR_ARG;
    hipStream_t s = c->stream;
    int32_t* sel = c->d_sel.p;
    mark_update_enqueued(c, d_sup);
    if (!c->pht_done) {   // update without a local score pass (supports came from elsewhere)
        launch_pht(s, c->d_Ppred.p, c->NP, c->d_mfeat.p, c->m, nullptr, c->d_H13.p, c->d_off.p, c->d_type.p,
                   c->d_W.p, c->NP, c->d_S.p, c->d_z.p, c->d_h.p, c->d_hash.p, c->d_wv.p, c->d_sel.p + SEL_STATUS_FRONT,
                   c->d_xpred.p, c->d_mith.p, c->d_miph.p, c->d_sc.p, nullptr, nullptr, c->d_hctx.p);
        c->pht_done = true;
    }
    const bool persistent = sweep_is_persistent(c);
    int blocks_li = 1 << 20, blocks_hi = 1 << 20;          // launch-per-step route: read from the device below
    const HostCounts hc_li = persistent ? HostCounts{nullptr, 0} : host_counts_slot(c, 0);
    // K5 consensus (Tracking.cpp:507-537)
    // (the winner's inlier mask is the one the scoring launch kept, where this context scored every hypothesis of the frame;
    //  with supports from elsewhere -- other ranks -- the winner is scored again)
    // A low-innovation update of one or two inliers (compat mode: Q1) is done by the consensus launch itself (kernels.h
    // LiSmallArgs) wherever its covariance would be deferred: the fused persistent sweep and the launch-per-step route.
    SystemDims d; d.n = c->n; d.NP = c->NP; d.RP = c->RP; d.ldA = c->ldA;
#if defined(RSLAM_DEBUG)
    const bool no_defer = getenv("RSLAM_NO_LI_DEFER") != nullptr;       // measurement: the covariance of every LI update is streamed
    const bool no_li_small = getenv("RSLAM_NO_LI_SMALL") != nullptr;    // measurement / tests: the update as a launch sequence of its own
#else
    const bool no_defer = false, no_li_small = false;
#endif
    const bool li_small = !no_defer && !no_li_small && c->RP > 0 && c->m > 0 && !(sweep_exp_mask() & (4 | 8 | 512)) /* test switches: the shared route, the in-LDS pipeline, no deferral */ &&
                          (persistent ? sweep_fused_eligible(d) : true);
    // no low-innovation sweep in this frame's sequence (li_skip).  (The launch-per-step route keeps reading the low-innovation
    // count and simply enqueues nothing for one or two inliers: it has to read the high-innovation count anyway, and not
    // waiting for this one made no measurable difference at C5, scripts/ab_front.py 1000 1000 -- its frames are the sum of
    // their kernels either way.)
    const bool li_skipped = li_small && persistent && c->li_skip;
    const LiSmallArgs ls{SysSrc{c->d_lilist.p, c->d_H13.p, c->d_off.p, c->d_type.p, c->d_z.p, c->d_h.p, c->d_W.p, c->d_rank_of.p},
                         c->NP, c->d_xpred.p, c->d_x1.p, c->d_Y1.p, c->NP, c->d_T.p, c->cfg.compat,
                         sel + SEL_XU_FLAG, sel + SEL_LI_DEFER, sel + SEL_STATUS,
                         li_skipped ? 1 : 0, persistent ? c->d_sweep_flags.p + SWEEP_FLAG_INTS : nullptr, SWEEP_FLAG_INTS};
    launch_best_mask(s, c->cam, c->d_xpred.p, c->d_W.p, c->NP, c->d_wv.p, tables(c), c->d_z.p, c->m, c->d_pos.p,
                     c->cfg.sigma_z, c->L, sel, c->d_li.p, c->d_lilist.p, d_sup, c->H, c->d_nhyp.p,
                     c->cfg.adaptive, c->cfg.n_hyp_init,
                     c->masks_all == 1 ? c->d_masks.p : c->masks_all == 2 ? c->d_posmask.p : nullptr, c->words, c->masks_all == 2, hc_li,
                     li_small ? &ls : nullptr);
    mark(c, EV_SELECT);
    // Systems too large for the persistent sweep (more 16-row strips than compute units, e.g. 1000 landmarks) run one launch
    // sequence per block step, and how many steps an update needs is only known on the device.  The host reads that one
    // integer here and enqueues exactly that many: no sizing from the previous frame, no overflow, no re-run, no hipGraph
    // re-capture.  Such frames are therefore not captured into graphs.  (The count arrives in host-mapped memory, HostCounts: a
    // copy + stream synchronisation cost 50 us of idle device per read in the kernel trace, and the HI count is known before the
    // second P H^T has run -- the host enqueues the HI sweep under it.)
    c->li_defer_host = false;
    if (!persistent) {
        int32_t cnt[SEL_NBLK_LI - SEL_K_LI + 1] = {0};            // sel[SEL_K_LI .. SEL_NBLK_LI]
        if (!host_counts_wait(c, 0, hc_li, &cnt[0], &cnt[SEL_NBLK_LI - SEL_K_LI])) {
            HIPCHK(hipMemcpyAsync(cnt, sel + SEL_K_LI, sizeof(cnt), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
        }
        blocks_li = cnt[SEL_NBLK_LI - SEL_K_LI];
        // a low-innovation update of rank <= 4 (compat mode: Q1 leaves one or two inliers) does not stream P: kernels.h MatArgs
        c->li_defer_host = !no_defer && cnt[0] >= 1 && 2 * cnt[0] <= 4 && c->RP > 0;
    }
    // low-innovation update (ExtendKF.cpp:559-596)
    int rc = RSLAM_OK;
    if (li_skipped || (!persistent && li_small && c->li_defer_host)) {
        // (done inside the consensus launch: nothing to enqueue)
        mark(c, EV_LI_FACTOR0); mark(c, EV_LI_FACTOR1); mark(c, EV_LI_RANK0); mark(c, EV_LI_RANK1);
    } else {
        rc = enqueue_one_update(c, c->d_lilist.p, SEL_K_LI, SEL_NBLK_LI, blocks_li, c->d_W.p, c->d_H13.p, c->d_h.p, c->d_xpred.p, c->d_x1.p,
                                c->d_Ppred.p, c->d_P.p, EV_LI_FACTOR0, EV_LI_FACTOR1, EV_LI_RANK0, EV_LI_RANK1);
    }
    if (rc) return rc;
    mark(c, EV_LI_END);
    // rescue (Tracking.cpp:574-597): re-predict at x_k_k; invisible features keep their stale h
    // (P_li may be deferred -- a rank <= 4 low-innovation update keeps Y1 and its Jnorm aside instead of streaming P: the readers
    //  of P_li between here and the HI pass form its entries themselves, kernels.h DeferArgs)
    DeferArgs da{sel + SEL_LI_DEFER, c->d_Ppred.p, c->NP, c->d_Y1.p, c->NP, c->d_T.p, c->d_Gd.p};
    // The gate nu' S^-1 nu < chi2 (:584-595) has no launch of its own: the prediction decides every feature's flag, the
    // workgroups of the second P H^T find their feature and the count from the flags (kernels.h GateArgs / GateList)
    // (large maps keep the gate as a launch: there the second P H^T has tens of thousands of workgroups that would each scan
    //  the flags -- measured at C5, 1000 landmarks: 25 us per frame more than the 6 us launch)
#if defined(RSLAM_DEBUG)
    static const bool gate_env = getenv("RSLAM_GATE_APART") != nullptr;        // measurement: rescue_gate_kernel as a launch
#else
    const bool gate_env = false;
#endif
    const bool gate_apart = gate_env || c->L > 512;
    const GateArgs ga{c->d_ic.p, c->d_li.p, c->d_z.p, c->cfg.chi2_gate, c->d_hi.p};
    launch_predict(s, c->cam, c->d_x1.p, c->d_P.p, c->NP, c->L, c->d_type.p, c->d_off.p, c->d_h.p, c->d_hash.p,
                   c->d_h2.p, c->d_hash2.p, nullptr, c->d_H13b.p, c->d_S2.p,
                   c->cfg.compat ? 0.0 : 1.0 /* Q7: no +R at Tracking.cpp:589 */, nullptr, &da, gate_apart ? nullptr : &ga);
    const HostCounts hc_hi = (!persistent && gate_apart) ? host_counts_slot(c, 4) : HostCounts{nullptr, 0};
    if (gate_apart)
        launch_rescue_gate(s, c->L, c->d_ic.p, c->d_li.p, c->d_hash2.p, c->d_S2.p, c->d_z.p, c->d_h2.p, c->cfg.chi2_gate,
                           c->d_hi.p, c->d_hilist.p, sel, hc_hi);
    mark(c, EV_RESCUE);
    // high-innovation update (ExtendKF.cpp:640-678): P H^T at the new linearisation, written straight into A
    const GateList gl{c->d_hi.p, c->L, c->d_hilist.p, sel};
    if (c->RP > 0)
        launch_pht(s, c->d_P.p, c->NP, c->d_hilist.p, c->m, sel + SEL_K_HI, c->d_H13b.p, c->d_off.p, c->d_type.p,
                   c->d_A.p + c->RP, c->ldA, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &da,
                   gate_apart ? nullptr : &gl);
    if (!persistent) {
        int32_t nb = 0, k_hi = 0;
        if (!host_counts_wait(c, 4, hc_hi, &k_hi, &nb)) {
            HIPCHK(hipMemcpyAsync(&nb, sel + SEL_NBLK_HI, sizeof(int32_t), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
        }
        blocks_hi = nb;
    }
    rc = enqueue_one_update(c, c->d_hilist.p, SEL_K_HI, SEL_NBLK_HI, blocks_hi, nullptr, c->d_H13b.p, c->d_h2.p, c->d_x1.p, c->d_x2.p,
                            c->d_P.p, c->d_P.p, EV_HI_FACTOR0, EV_HI_FACTOR1, EV_HI_RANK0, EV_HI_RANK1);
    if (rc) return rc;
    mark(c, EV_HI_END);
    return RSLAM_OK;
}

static int read_status_raw(rslam_ctx* c, int32_t* sel)
{
    HIPCHK(hipMemcpyAsync(sel, c->d_sel.p, sizeof(int32_t) * SEL_COUNT, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipGetLastError());
    return RSLAM_OK;
}

// Synchronise and return the device-side status of the frame (and of earlier frames nobody synchronised on); a bounded
// wait that ran out inside a persistent launch is handled here: re-run from the kept inputs, launch-per-step for a while.
static int read_status(rslam_ctx* c, int32_t* sel_host)
{
    int32_t sel[SEL_COUNT];
    int rc = read_status_raw(c, sel);
    if (rc) return rc;
    // statuses of frames that were enqueued behind each other without a sync in between (predict_kernel folds the word it
    // resets into SEL_STICKY): reported with this frame, then cleared
    int sticky = sel[SEL_STICKY];
    if (sticky != 0) HIPCHK(hipMemsetAsync(c->d_sel.p + SEL_STICKY, 0, sizeof(int32_t), c->stream));
    if (sel[SEL_WAIT_FIRST] != 0) {
        c->last_wait_first = sel[SEL_WAIT_FIRST];
        c->last_wait_polls = sel[SEL_WAIT_POLLS];
        HIPCHK(hipMemsetAsync(c->d_sel.p + SEL_WAIT_FIRST, 0, sizeof(int32_t), c->stream));
        HIPCHK(hipMemsetAsync(c->d_sel.p + SEL_WAIT_POLLS, 0, sizeof(int32_t), c->stream));
    }
    // "this sequence lacked the low-innovation sweep a frame needed" arrives out of band (SEL_LI_NEED: bit 0 this frame, bit 1 an
    // earlier frame of an unsynchronised run), not as a value competing in the min-folded status words
    const int li_need = sel[SEL_LI_NEED];
    if (li_need != 0) HIPCHK(hipMemsetAsync(c->d_sel.p + SEL_LI_NEED, 0, sizeof(int32_t), c->stream));
    if (li_need & 2) {
        // an earlier frame of this unsynchronised run needed the sweep (nobody used its posterior): the sequence gets it back.
        // That frame's rescue stage and high-innovation pass ran on a posterior that was never made: a "not positive definite"
        // or a timed-out wait they folded into the sticky word says nothing about the data or the device and goes with them;
        // any other code of the run (an input error of the prediction stage) stays and is reported below as ever.
        c->li_skip = false;
        invalidate_graph(c);
        if (sticky == RSLAM_ERR_NOT_SPD || sticky <= -30) sticky = 0;
    }
    if (sticky <= -30 && sticky != -39) {
        // An earlier frame of this unsynchronised run hit a bounded wait of the persistent sweep (somebody else held compute
        // units).  Nobody used its posterior -- every consumer settles the frame in flight first -- so it is not an error of
        // this call; but the environment is hostile: take the launch-per-step path for a while, as for a timeout seen directly.
        c->last_raw_status = sticky;
        ++c->sweep_fallbacks;
        if (c->steps_frames_left == 0) { c->steps_frames_left = 64 << (c->consecutive_fallbacks < 6 ? c->consecutive_fallbacks : 6); ++c->consecutive_fallbacks; invalidate_graph(c); }
        sticky = 0;
    }
    if (c->frame_checked) {
        // this frame has been reported before: its status words were taken off the device then (so that the next frame's
        // reset does not report them a second time) and are remembered here
        if (c->rep_status < sel[SEL_STATUS]) sel[SEL_STATUS] = c->rep_status;
        if (c->rep_front < sel[SEL_STATUS_FRONT]) sel[SEL_STATUS_FRONT] = c->rep_front;
        if (c->rep_sticky < sticky) sticky = c->rep_sticky;
    }
    if (li_need & 1) {
        // The frame has other than one or two low-innovation inliers and its sequence had no sweep for them (li_skip): re-run
        // the update stage with the sweep, and keep it in the sequence of this context.  What the rest of that sequence -- a
        // rescue stage and a high-innovation pass on a posterior that was never made -- put into the status word goes with it.
        // (raw code -40, for rslam_last_raw_status: the name this case had when it was a status value)
        if (!c->last_sup || !c->predicted || !c->have_meas) { c->last_raw_status = -40; c->li_skip = false; invalidate_graph(c); c->frame_checked = true; return RSLAM_ERR_HIP; }
        c->last_raw_status = -40;
        c->li_skip = false;
        ++c->li_shape_reruns;
        invalidate_graph(c);
        const int timing = c->timing; c->timing = 0;
        HIPCHK(hipMemsetAsync(c->d_sel.p + SEL_STATUS, 0, sizeof(int32_t), c->stream));
        HIPCHK(hipMemsetAsync(c->d_sel.p + SEL_XU_FLAG, 0, sizeof(int32_t), c->stream));
        HIPCHK(hipMemsetAsync(c->d_sweep_flags.p, 0, sizeof(int32_t) * 2 * SWEEP_FLAG_INTS, c->stream));
        rc = enqueue_update(c, c->last_sup);
        c->timing = timing;
        if (rc) return rc;
        rc = read_status_raw(c, sel);
        if (rc) return rc;
    }
    if (sel[SEL_STATUS] == -39 && c->last_sup && c->predicted && c->have_meas && !c->k10_riders_first) {
        // The first block column of the stand-alone rank update waited for Jnorm in vain: the riders that produce it had been
        // placed behind the tiles and did not become resident (somebody else holds compute units).  Re-run the update stage
        // with the riders in front -- dispatched first, they cannot be locked out by the tiles -- and keep that order.
        c->last_raw_status = sel[SEL_STATUS];
        c->k10_riders_first = true;
        ++c->k10_reruns;
        invalidate_graph(c);
        const int timing = c->timing; c->timing = 0;
        HIPCHK(hipMemsetAsync(c->d_sel.p + SEL_STATUS, 0, sizeof(int32_t), c->stream));
        HIPCHK(hipMemsetAsync(c->d_sel.p + SEL_XU_FLAG, 0, sizeof(int32_t), c->stream));
        rc = enqueue_update(c, c->last_sup);
        c->timing = timing;
        if (rc) return rc;
        rc = read_status_raw(c, sel);
        if (rc) return rc;
    }
    // (codes -31..-38 and the chain's per-block -36-10k: waits of the persistent sweep and its tile workers; -39 is the
    //  rider hand-over of the stand-alone rank update, handled below)
    const bool sweep_timeout = sel[SEL_STATUS] <= -30 && sel[SEL_STATUS] != -39;
    if (sweep_timeout && c->last_sup && sweep_is_persistent(c)) {
        if (!c->predicted || !c->have_meas) {     // the frame's inputs are gone (a new prior was installed unchecked): nothing to re-run from
            c->last_raw_status = sel[SEL_STATUS];
            c->frame_checked = true;
            return RSLAM_ERR_HIP;
        }
        // A bounded wait of the persistent sweep ran out: its workgroups were not all resident (another user of the GPU), not
        // an error of the data.  Re-run the update stage with the launch-per-step sweep at full length, and keep to it for
        // the next frames.
        c->last_raw_status = sel[SEL_STATUS];
        // back off exponentially: 64, 128, ... 4096 frames on the launch-per-step sweep while the timeouts keep coming
        c->steps_frames_left = 64 << (c->consecutive_fallbacks < 6 ? c->consecutive_fallbacks : 6);
        ++c->consecutive_fallbacks;
        ++c->sweep_fallbacks;
        invalidate_graph(c);
        const int timing = c->timing; c->timing = 0;
        HIPCHK(hipMemsetAsync(c->d_sel.p + SEL_STATUS, 0, sizeof(int32_t), c->stream));
        HIPCHK(hipMemsetAsync(c->d_sel.p + SEL_XU_FLAG, 0, sizeof(int32_t), c->stream));
        HIPCHK(hipMemsetAsync(c->d_sweep_flags.p, 0, sizeof(int32_t) * 2 * SWEEP_FLAG_INTS, c->stream));
        rc = enqueue_update(c, c->last_sup);
        c->timing = timing;
        if (rc) return rc;
        rc = read_status_raw(c, sel);
        if (rc) return rc;
    } else if (c->steps_frames_left > 0 && c->last_sup && !c->frame_checked) {
        if (--c->steps_frames_left == 0) invalidate_graph(c);     // back to the persistent sweep: new launch sequence
    } else if (c->steps_frames_left == 0 && c->last_sup && !c->frame_checked && sel[SEL_STATUS] > -30) {
        c->consecutive_fallbacks = 0;                             // a persistent sweep went through
    }
    if (sel[SEL_STATUS] != 0 || sel[SEL_STATUS_FRONT] != 0) {
        HIPCHK(hipMemsetAsync(c->d_sel.p + SEL_STATUS, 0, sizeof(int32_t), c->stream));
        HIPCHK(hipMemsetAsync(c->d_sel.p + SEL_STATUS_FRONT, 0, sizeof(int32_t), c->stream));
    }
    c->rep_status = sel[SEL_STATUS]; c->rep_front = sel[SEL_STATUS_FRONT]; c->rep_sticky = sticky;
    if (sel[SEL_STATUS_FRONT] < sel[SEL_STATUS]) sel[SEL_STATUS] = sel[SEL_STATUS_FRONT];
    if (sticky < sel[SEL_STATUS]) sel[SEL_STATUS] = sticky;
    if (sel[SEL_STATUS] <= -30) {                 // (still: which wait it was is kept for diagnosis)
        c->last_raw_status = sel[SEL_STATUS];
        sel[SEL_STATUS] = RSLAM_ERR_HIP;
    }
    if (sel_host) memcpy(sel_host, sel, sizeof(sel));
    c->frame_checked = true;
    return sel[SEL_STATUS];
}

static void collect_times(rslam_ctx* c)
{
    if (!c->timing || !c->ev_ok) return;
    auto el = [&](int a, int b) { float ms = 0; if (hipEventElapsedTime(&ms, c->ev[a], c->ev[b]) != hipSuccess) return 0.0; return (double)ms * 1e3; };
    rslam_stage_times& t = c->times;
    t.predict_us = el(EV_START, EV_PREDICT);
    t.pht_us = el(EV_PREDICT, EV_PHT);
    t.score_us = el(EV_PHT, EV_SCORE);
    t.select_us = el(EV_SCORE, EV_SELECT);
    t.update_li_us = el(EV_SELECT, EV_LI_END);
    t.rescue_us = el(EV_LI_END, EV_RESCUE);
    t.update_hi_us = el(EV_RESCUE, EV_HI_END);
    t.rank_update_li_us = el(EV_LI_RANK0, EV_LI_RANK1);
    t.factor_li_us = el(EV_LI_FACTOR0, EV_LI_FACTOR1);
    t.rank_update_hi_us = el(EV_HI_RANK0, EV_HI_RANK1);
    t.factor_hi_us = el(EV_HI_FACTOR0, EV_HI_FACTOR1);
    t.total_us = el(EV_START, EV_HI_END);
}

// ------------------------------------------------------------------------
// drop-in API
// ------------------------------------------------------------------------
extern "C" int rslam_predict(rslam_ctx* c, const rslam_layout* layout, const double* x_pred, const double* P_pred,
                             double* h, uint8_t* visible, double* S)
{
    if (!c) return RSLAM_ERR_ARG;
    int rc;
    if (!x_pred && !P_pred) {                      // prior left resident by rslam_ekf_prediction
        if (!c->have_state) return RSLAM_ERR_STATE;
        if (layout && (layout->n != c->n || layout->L != c->L)) return RSLAM_ERR_ARG;
        HIPCHK(hipSetDevice(c->device));
        c->have_meas = false;
    } else {
        rc = upload_state(c, layout, x_pred, P_pred);
        if (rc) return rc;
    }
    rc = enqueue_predict(c);
    if (rc) return rc;
    const int L = c->L;
    std::vector<double> hh(2 * (size_t)L + 1), SS(4 * (size_t)L + 1);
    if (L > 0) {
        HIPCHK(hipMemcpyAsync(hh.data(), c->d_h.p, sizeof(double) * 2 * L, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(SS.data(), c->d_S.p, sizeof(double) * 4 * L, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(c->h_vis.data(), c->d_vis.p, L, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < L; ++i) {
        if (visible) visible[i] = c->h_vis[i];
        if (!c->h_vis[i]) continue;
        if (h) { h[2 * i] = hh[2 * i]; h[2 * i + 1] = hh[2 * i + 1]; }
        if (S) memcpy(S + 4 * i, SS.data() + 4 * i, sizeof(double) * 4);
    }
    return RSLAM_OK;
}

// ------------------------------------------------------------------------
// Tracking::matching on the resident prediction
// ------------------------------------------------------------------------
extern "C" int rslam_match(rslam_ctx* c, const uint8_t* image, const double* patches, double* z, uint8_t* ic, double* corr)
{
    if (!c || !image || !z || !ic) return RSLAM_ERR_ARG;
    if (!c->have_state || !c->predicted) return RSLAM_ERR_STATE;
    if (!patches && !c->patches_valid) return RSLAM_ERR_STATE;      // NULL = the patches rslam_predict_patches left on the device
    HIPCHK(hipSetDevice(c->device));
    const int L = c->L;
    if (L == 0) return RSLAM_OK;
    const size_t npix = (size_t)c->cam.nRows * c->cam.nCols;
    if (c->d_image.ensure(npix) < 0 || c->d_corr.ensure(L) < 0) return RSLAM_ERR_HIP;
    if (patches && c->d_patches.ensure((size_t)L * 169) < 0) return RSLAM_ERR_HIP;
    hipStream_t s = c->stream;
    HIPCHK(hipMemcpyAsync(c->d_image.p, image, npix, hipMemcpyHostToDevice, s));
    if (patches) {
        HIPCHK(hipMemcpyAsync(c->d_patches.p, patches, sizeof(double) * (size_t)L * 169, hipMemcpyHostToDevice, s));
        c->patches_valid = false;
    }
    HIPCHK(hipMemsetAsync(c->d_z.p, 0, sizeof(double) * 2 * L, s));
    launch_match(s, c->cam, c->d_image.p, c->d_patches.p, L, c->d_h.p, c->d_hash.p, c->d_S.p, 0.80 /* Tracking.cpp:281 */,
                 5.9915 /* :283 */, c->d_z.p, c->d_ic.p, c->d_corr.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(z, c->d_z.p, sizeof(double) * 2 * L, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(ic, c->d_ic.p, L, hipMemcpyDeviceToHost, s));
    if (corr) HIPCHK(hipMemcpyAsync(corr, c->d_corr.p, sizeof(double) * L, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return RSLAM_OK;
}

// ------------------------------------------------------------------------
// Tracking::pred_patch_fc on the resident prior and the feature store it reads
// ------------------------------------------------------------------------
namespace {

template <typename T>
int grow_keep(DevBuf<T>& b, size_t old_count, size_t new_count, hipStream_t s)
{
    if (new_count <= b.cap && b.p) return 0;
    T* np = nullptr;
    if (hipMalloc((void**)&np, new_count * sizeof(T)) != hipSuccess) return -1;
    if (b.p && old_count) {
        if (hipMemcpyAsync(np, b.p, old_count * sizeof(T), hipMemcpyDeviceToDevice, s) != hipSuccess) { (void)hipFree(np); return -1; }
        if (hipStreamSynchronize(s) != hipSuccess) { (void)hipFree(np); return -1; }
    }
    if (b.p) (void)hipFree(b.p);
    b.p = np; b.cap = new_count;
    return 1;
}

int store_write(rslam_ctx* c, int slot, const double* uv, const double* R, const double* r, const double* patch)
{
    double rec[14];
    rec[0] = uv[0]; rec[1] = uv[1];
    for (int k = 0; k < 9; ++k) rec[2 + k] = R[k];
    for (int k = 0; k < 3; ++k) rec[11 + k] = r[k];
    std::vector<float> pf(1681);
    for (int rr = 0; rr < 41; ++rr)
        for (int cc = 0; cc < 41; ++cc) pf[(size_t)rr * 41 + cc] = (float)patch[rr + 41 * cc];       // toCvMat_f, Converter.cpp:83-94
    HIPCHK(hipMemcpyAsync(c->d_rec.p + (size_t)slot * 14, rec, sizeof(rec), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_rec_patch.p + (size_t)slot * 1681, pf.data(), sizeof(float) * 1681, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));                       // the staging buffers are on this stack frame
    return RSLAM_OK;
}

int store_reserve(rslam_ctx* c, int slots)
{
    if (slots <= c->store_cap) return RSLAM_OK;
    int cap = c->store_cap ? c->store_cap : 64;
    while (cap < slots) cap *= 2;
    if (grow_keep(c->d_rec, (size_t)c->store_cap * 14, (size_t)cap * 14, c->stream) < 0) return RSLAM_ERR_HIP;
    if (grow_keep(c->d_rec_patch, (size_t)c->store_cap * 1681, (size_t)cap * 1681, c->stream) < 0) return RSLAM_ERR_HIP;
    for (int k = cap - 1; k >= c->store_cap; --k) c->free_slots.push_back(k);
    c->store_cap = cap;
    return RSLAM_OK;
}

}  // namespace

extern "C" int rslam_append_feature_record(rslam_ctx* c, const double* uv, const double* R_wc, const double* r_wc, const double* patch)
{
    if (!c || !uv || !R_wc || !r_wc || !patch) return RSLAM_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    int rc = store_reserve(c, (int)c->h_slot.size() + 1);
    if (rc) return rc;
    const int slot = c->free_slots.back();
    rc = store_write(c, slot, uv, R_wc, r_wc, patch);
    if (rc) return rc;
    c->free_slots.pop_back();
    c->h_slot.push_back(slot);
    c->patches_valid = false;
    return RSLAM_OK;
}

extern "C" int rslam_set_feature_records(rslam_ctx* c, int32_t L, const double* uv, const double* R_wc, const double* r_wc, const double* patches)
{
    if (!c || L < 0 || (L > 0 && (!uv || !R_wc || !r_wc || !patches))) return RSLAM_ERR_ARG;
    for (int32_t s : c->h_slot) c->free_slots.push_back(s);
    c->h_slot.clear();
    for (int i = 0; i < L; ++i) {
        const int rc = rslam_append_feature_record(c, uv + 2 * (size_t)i, R_wc + 9 * (size_t)i, r_wc + 3 * (size_t)i, patches + 1681 * (size_t)i);
        if (rc) return rc;
    }
    return RSLAM_OK;
}

extern "C" int rslam_predict_patches(rslam_ctx* c, double* patches, int32_t* status)
{
    if (!c) return RSLAM_ERR_ARG;
    if (!c->have_state || !c->predicted) return RSLAM_ERR_STATE;
    const int L = c->L;
    if ((int)c->h_slot.size() != L) return RSLAM_ERR_STATE;        // one record per feature (append after rslam_map_add_feature)
    if (L == 0) return RSLAM_OK;
    HIPCHK(hipSetDevice(c->device));
    if (c->d_slot.ensure(L) < 0 || c->d_xyz_src.ensure(L) < 0 || c->d_pstatus.ensure(L) < 0 || c->d_patches.ensure((size_t)L * 169) < 0)
        return RSLAM_ERR_HIP;
    // search_IC_matches refreshes XYZ_w for inverse-depth features only (Tracking.cpp:52-61): in compat mode a
    // Cartesian feature is warped with the point of the last inverse-depth feature before it
    std::vector<int32_t> src((size_t)L);
    int last_id = -1;
    for (int i = 0; i < L; ++i) {
        if (c->h_type[i] == RSLAM_FEAT_INVERSE_DEPTH) { last_id = i; src[i] = i; }
        else src[i] = c->cfg.compat ? last_id : i;
    }
    hipStream_t s = c->stream;
    HIPCHK(hipMemcpyAsync(c->d_slot.p, c->h_slot.data(), sizeof(int32_t) * L, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_xyz_src.p, src.data(), sizeof(int32_t) * L, hipMemcpyHostToDevice, s));
    launch_pred_patches(s, c->cam, c->cfg.compat, L, c->d_type.p, c->d_off.p, c->d_xyz_src.p, c->d_xpred.p, c->d_h.p, c->d_hash.p,
                        c->d_slot.p, c->d_rec.p, c->d_rec_patch.p, c->d_patches.p, c->d_pstatus.p);
    HIPCHK(hipGetLastError());
    if (patches) HIPCHK(hipMemcpyAsync(patches, c->d_patches.p, sizeof(double) * (size_t)L * 169, hipMemcpyDeviceToHost, s));
    if (status) HIPCHK(hipMemcpyAsync(status, c->d_pstatus.p, sizeof(int32_t) * L, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));                               // src / h_slot staging and the outputs
    c->patches_valid = true;
    return RSLAM_OK;
}

// ------------------------------------------------------------------------
// EKF prediction on the resident posterior
// ------------------------------------------------------------------------
extern "C" int rslam_set_posterior(rslam_ctx* c, const rslam_layout* layout, const double* x_kk, const double* P_kk)
{
    if (!c || !x_kk || !P_kk) return RSLAM_ERR_ARG;
    int rc = set_layout(c, layout);
    if (rc) return rc;
    rc = upload_xp(c, x_kk, P_kk, c->d_x2.p, c->d_P.p);
    if (rc) return rc;
    c->have_post = true; c->have_state = false; c->have_meas = false; c->predicted = false;
    return RSLAM_OK;
}

static int settle_posterior(rslam_ctx* c);

extern "C" int rslam_ekf_prediction(rslam_ctx* c, double delta_t, double std_a, double std_alpha)
{
    if (!c) return RSLAM_ERR_ARG;
    // a frame whose factor sweep may have been enqueued too short must be checked (and re-run) before its
    // posterior becomes the next prior: the next predict_kernel would erase the overflow flag
    const int rc = settle_posterior(c);
    if (rc) return rc;
    hipStream_t s = c->stream;
    const size_t nn = (size_t)c->NP * c->NP;
    HIPCHK(hipMemcpyAsync(c->d_xpred.p, c->d_x2.p, sizeof(double) * c->NP, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_Ppred.p, c->d_P.p, sizeof(double) * nn, hipMemcpyDeviceToDevice, s));
    launch_ekf_prediction(s, c->n, c->NP, c->d_x2.p, c->d_P.p, delta_t, std_a, std_alpha, c->d_xpred.p, c->d_Ppred.p, c->d_FQ.p);
    HIPCHK(hipGetLastError());
    c->have_state = true; c->have_meas = false; c->predicted = false; c->pht_done = false; c->dedup_done = false; c->masks_all = 0;
    return RSLAM_OK;
}

// ------------------------------------------------------------------------
// Map::map_management state surgery on the resident posterior
// ------------------------------------------------------------------------
// A frame whose update stage is still in flight must have completed (and possibly been re-run, see
// read_status) before its posterior is edited, propagated or fetched.  Frames whose sweep cannot
// overflow (persistent sweep, or caps at full length) need no host round trip.
static int settle_posterior(rslam_ctx* c)
{
    if (!c->have_post) return RSLAM_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    // An update stage that nobody has looked at yet is checked before its posterior is used: a shortened launch-per-step
    // sweep may have to be re-run, and a bounded wait of the persistent sweep may have run out (another user of the GPU),
    // in which case the posterior in the buffers is not the frame's -- read_status re-runs the stage while the frame's
    // inputs are still in place (the next predict_kernel would also erase the status word).
    if (c->have_meas && c->last_sup && !c->frame_checked) {
        const int st = read_status(c, nullptr);
        if (st < 0) return st;
    }
    return RSLAM_OK;
}

namespace {

// Runs the congruence into the scratch pair (d_xpred, d_Ppred), swaps it with the posterior
// pair and installs the new layout.  mode 0 delete / 1 convert / 2 insert.
int apply_map_edit(rslam_ctx* c, int mode, int cut, int special, int shift, int sp_base, int sp_cnt,
                   const std::vector<uint8_t>& type, int o, double ud, double vd, double rho0, double std_rho)
{
    const int L2 = (int)type.size();
    std::vector<int32_t> off((size_t)L2);
    int n2 = 13;
    for (int i = 0; i < L2; ++i) { off[i] = n2; n2 += type[i] == RSLAM_FEAT_INVERSE_DEPTH ? 6 : 3; }
    const int NP2 = round_up(n2, 64);
    if (c->d_xpred.ensure(NP2) < 0 || c->d_Ppred.ensure((size_t)NP2 * NP2) < 0 || c->d_mapcoef.ensure(MAP_COEF_DOUBLES) < 0)
        return RSLAM_ERR_HIP;
    hipStream_t s = c->stream;
    launch_map_state(s, mode, c->cam, c->d_x2.p, o, ud, vd, rho0, c->cfg.sigma_z, std_rho, c->d_mapcoef.p, c->d_xpred.p,
                     n2, NP2, cut, special, shift);
    launch_map_cov(s, c->d_P.p, c->NP, c->d_Ppred.p, NP2, n2, cut, special, shift, sp_base, sp_cnt, mode == 2, c->d_mapcoef.p);
    HIPCHK(hipGetLastError());
    std::swap(c->d_x2, c->d_xpred);
    std::swap(c->d_P, c->d_Ppred);
    invalidate_graph(c);                       // the captured launches hold the old pointers
    rslam_layout lay{n2, L2, type.data(), off.data()};
    const int rc = set_layout(c, &lay);        // (d_x2, d_P) are large enough: never reallocated here
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(s));
    c->have_post = true; c->have_state = false; c->have_meas = false; c->predicted = false;
    c->pht_done = false; c->dedup_done = false; c->masks_all = 0; c->last_sup = nullptr;
    return RSLAM_OK;
}

}  // namespace

extern "C" int rslam_get_layout(rslam_ctx* c, int32_t* n, int32_t* L, uint8_t* type, int32_t* offset)
{
    if (!c) return RSLAM_ERR_ARG;
    if (n) *n = c->n;
    if (L) *L = c->L;
    for (int i = 0; i < c->L; ++i) {
        if (type) type[i] = c->h_type[i];
        if (offset) offset[i] = c->h_off[i];
    }
    return RSLAM_OK;
}

extern "C" int rslam_map_delete_feature(rslam_ctx* c, int32_t feature)
{
    if (!c) return RSLAM_ERR_ARG;
    int rc = settle_posterior(c);
    if (rc) return rc;
    if (feature < 0 || feature >= c->L) return RSLAM_ERR_ARG;
    const int w = c->h_type[feature] == RSLAM_FEAT_INVERSE_DEPTH ? 6 : 3;
    std::vector<uint8_t> type(c->h_type);
    type.erase(type.begin() + feature);
    const bool has_store = ((int)c->h_slot.size() == c->L);
    rc = apply_map_edit(c, 0, c->h_off[feature], 0, w, 0, 0, type, 0, 0.0, 0.0, 0.0, 0.0);
    if (rc) return rc;
    if (has_store) {                                               // the feature store follows features_info.erase (Map.cpp:27)
        c->free_slots.push_back(c->h_slot[feature]);
        c->h_slot.erase(c->h_slot.begin() + feature);
    }
    return RSLAM_OK;
}

extern "C" int rslam_map_convert(rslam_ctx* c, double linearity_threshold, int32_t* converted, double* linearity)
{
    if (!c) return RSLAM_ERR_ARG;
    if (converted) *converted = -1;
    int rc = settle_posterior(c);
    if (rc) return rc;
    const int L = c->L;
    if (L == 0) return RSLAM_OK;
    if (c->d_lin.ensure(L) < 0 || c->d_first.ensure(1) < 0) return RSLAM_ERR_HIP;
    hipStream_t s = c->stream;
    HIPCHK(hipMemsetAsync(c->d_first.p, 0x7f, sizeof(int32_t), s));
    launch_map_linearity(s, c->d_x2.p, c->d_P.p, c->NP, L, c->d_type.p, c->d_off.p, linearity_threshold, c->d_lin.p, c->d_first.p);
    HIPCHK(hipGetLastError());
    int32_t first = 0;
    HIPCHK(hipMemcpyAsync(&first, c->d_first.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (linearity) HIPCHK(hipMemcpyAsync(linearity, c->d_lin.p, sizeof(double) * L, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (first < 0 || first >= L) return RSLAM_OK;            // nothing below the threshold
    std::vector<uint8_t> type(c->h_type);
    type[first] = RSLAM_FEAT_CARTESIAN;
    const int o = c->h_off[first];
    rc = apply_map_edit(c, 1, o, 3, 3, o, 6, type, o, 0.0, 0.0, 0.0, 0.0);
    if (rc) return rc;
    if (converted) *converted = first;                        // one feature per call, Map.cpp:192
    return RSLAM_OK;
}

extern "C" int rslam_map_add_feature(rslam_ctx* c, const double* uvd, double initial_rho, double std_rho)
{
    if (!c || !uvd) return RSLAM_ERR_ARG;
    int rc = settle_posterior(c);
    if (rc) return rc;
    std::vector<uint8_t> type(c->h_type);
    type.push_back(RSLAM_FEAT_INVERSE_DEPTH);
    return apply_map_edit(c, 2, c->n, 6, 0, 0, 13, type, 0, uvd[0], uvd[1], initial_rho, std_rho);
}

extern "C" int rslam_map_predict(rslam_ctx* c, double* h, uint8_t* visible)
{
    if (!c) return RSLAM_ERR_ARG;
    int rc = settle_posterior(c);
    if (rc) return rc;
    const int L = c->L;
    if (L == 0) return RSLAM_OK;
    hipStream_t s = c->stream;
    // predict_camera_measurements(x_k_k), Map.cpp:221; h was reset at Map.cpp:51, so no stale values
    launch_predict(s, c->cam, c->d_x2.p, c->d_P.p, c->NP, L, c->d_type.p, c->d_off.p, nullptr, nullptr,
                   c->d_h2.p, c->d_hash2.p, c->d_vis.p, c->d_H13b.p, c->d_S2.p, 0.0, nullptr);
    HIPCHK(hipGetLastError());
    std::vector<double> hh(2 * (size_t)L);
    HIPCHK(hipMemcpyAsync(hh.data(), c->d_h2.p, sizeof(double) * 2 * L, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(c->h_vis.data(), c->d_vis.p, L, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int i = 0; i < L; ++i) {
        if (visible) visible[i] = c->h_vis[i];
        if (h && c->h_vis[i]) { h[2 * i] = hh[2 * i]; h[2 * i + 1] = hh[2 * i + 1]; }
    }
    return RSLAM_OK;
}

extern "C" int rslam_fetch_prior(rslam_ctx* c, double* x_pred, double* P_pred)
{
    if (!c) return RSLAM_ERR_ARG;
    if (!c->have_state) return RSLAM_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    if (x_pred) HIPCHK(hipMemcpyAsync(x_pred, c->d_xpred.p, sizeof(double) * c->n, hipMemcpyDeviceToHost, c->stream));
    if (P_pred) HIPCHK(hipMemcpy2DAsync(P_pred, sizeof(double) * c->n, c->d_Ppred.p, sizeof(double) * c->NP, sizeof(double) * c->n,
                                        c->n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return RSLAM_OK;
}

extern "C" int rslam_fetch_cov(rslam_ctx* c, double* P)
{
    if (!c || !P) return RSLAM_ERR_ARG;
    const int rc = settle_posterior(c);          // RSLAM_ERR_STATE unless a posterior exists
    if (rc) return rc;
    pin_host_buffer(c, P, sizeof(double) * (size_t)c->n * c->n);
    if (c->NP != c->n) {
        if (c->d_stage.ensure((size_t)c->n * c->n) < 0) return RSLAM_ERR_HIP;
        launch_repitch(c->stream, c->d_P.p, c->NP, c->d_stage.p, c->n, c->n, c->n, c->n, c->n);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(P, c->d_stage.p, sizeof(double) * (size_t)c->n * c->n, hipMemcpyDeviceToHost, c->stream));
    } else {
        HIPCHK(hipMemcpyAsync(P, c->d_P.p, sizeof(double) * (size_t)c->n * c->n, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return RSLAM_OK;
}

extern "C" int rslam_unpin_host_buffers(rslam_ctx* c)
{
    if (!c) return RSLAM_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));      // no transfer from / to a registered buffer is in flight
    unpin_host_buffers(c);
    return RSLAM_OK;
}

extern "C" int rslam_fetch_state(rslam_ctx* c, double* x)
{
    if (!c || !x) return RSLAM_ERR_ARG;
    const int rc = settle_posterior(c);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(x, c->d_x2.p, sizeof(double) * c->n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return RSLAM_OK;
}

extern "C" int rslam_fetch_results(rslam_ctx* c, double* x_new, uint8_t* li, uint8_t* hi, int32_t* best_hyp,
                                   int32_t* best_support, int32_t* hyps_evaluated, int32_t* n_li, int32_t* n_hi)
{
    if (!c) return RSLAM_ERR_ARG;
    if (!c->have_state) return RSLAM_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    int32_t sel[SEL_COUNT];
    const int status = read_status(c, sel);      // first: it may re-run the update stage (sweep sizing)
    collect_times(c);
    if (x_new) HIPCHK(hipMemcpyAsync(x_new, c->d_x2.p, sizeof(double) * c->n, hipMemcpyDeviceToHost, c->stream));
    if (li && c->L) HIPCHK(hipMemcpyAsync(li, c->d_li.p, c->L, hipMemcpyDeviceToHost, c->stream));
    if (hi && c->L) HIPCHK(hipMemcpyAsync(hi, c->d_hi.p, c->L, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (best_hyp) *best_hyp = sel[SEL_BEST_HYP];
    if (best_support) *best_support = sel[SEL_BEST_SUPPORT];
    if (hyps_evaluated) *hyps_evaluated = sel[SEL_HYPS_EVALUATED];
    if (n_li) *n_li = sel[SEL_K_LI];
    if (n_hi) *n_hi = sel[SEL_K_HI];
    return status;
}

extern "C" int rslam_ransac_update(rslam_ctx* c, const double* z, const uint8_t* ic, const double* draws,
                                   int32_t n_draws, double* x_new, double* P_new, uint8_t* li, uint8_t* hi,
                                   int32_t* best_hyp, int32_t* best_support, int32_t* hyps_evaluated)
{
    if (!c) return RSLAM_ERR_ARG;
    if (!c->predicted) return RSLAM_ERR_STATE;
    int rc = upload_measurements(c, z, ic, draws, n_draws, true);
    if (rc) return rc;
    rc = enqueue_score(c, 0, c->H, c->d_sup.p);
    if (rc) return rc;
    rc = enqueue_update(c, c->d_sup.p);
    if (rc) return rc;
    rc = rslam_fetch_results(c, x_new, li, hi, best_hyp, best_support, hyps_evaluated, nullptr, nullptr);
    if (rc) return rc;
    if (P_new) return rslam_fetch_cov(c, P_new);
    return RSLAM_OK;
}

extern "C" int rslam_timings(rslam_ctx* c, rslam_stage_times* out)
{
    if (!c || !out) return RSLAM_ERR_ARG;
    *out = c->times;
    return RSLAM_OK;
}

// ------------------------------------------------------------------------
// resident API
// ------------------------------------------------------------------------
extern "C" int rslam_load_frame(rslam_ctx* c, const rslam_layout* layout, const double* x_pred, const double* P_pred,
                                const double* z, const uint8_t* ic, const double* draws, int32_t n_draws)
{
    if (!c) return RSLAM_ERR_ARG;
    int rc = upload_state(c, layout, x_pred, P_pred);
    if (rc) return rc;
    return upload_measurements(c, z, ic, draws, n_draws, false);
}

extern "C" int rslam_load_measurements(rslam_ctx* c, const double* z, const uint8_t* ic, const double* draws, int32_t n_draws)
{
    if (!c) return RSLAM_ERR_ARG;
    if (!c->have_state) return RSLAM_ERR_STATE;
    // the posterior of a frame still in flight reads the measurement buffers: let it finish (and be checked) first
    if (c->have_post && !c->frame_checked) { const int st = read_status(c, nullptr); if (st < 0) return st; }
    return upload_measurements(c, z, ic, draws, n_draws, false);
}

extern "C" int rslam_get_counters(rslam_ctx* c, int32_t* graph_captures, int32_t* sweep_reruns)
{
    if (!c) return RSLAM_ERR_ARG;
    if (graph_captures) *graph_captures = c->graph_captures;
    if (sweep_reruns) *sweep_reruns = c->sweep_fallbacks + c->k10_reruns + c->li_shape_reruns;
    return RSLAM_OK;
}

extern "C" int rslam_step_predict(rslam_ctx* c)
{
    if (!c) return RSLAM_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    return enqueue_predict(c);
}

extern "C" int rslam_step_score(rslam_ctx* c, int32_t hyp_begin, int32_t hyp_end, int32_t* d_supports)
{
    if (!c) return RSLAM_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    return enqueue_score(c, hyp_begin, hyp_end, d_supports);
}

extern "C" int rslam_step_update(rslam_ctx* c, const int32_t* d_supports)
{
    if (!c) return RSLAM_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    return enqueue_update(c, d_supports);
}

static int enqueue_frame(rslam_ctx* c)
{
    int rc = enqueue_predict(c);
    if (rc) return rc;
    rc = enqueue_score(c, 0, c->H, c->d_sup.p);
    if (rc) return rc;
    return enqueue_update(c, c->d_sup.p);
}

// capture `enqueue` into graph slot k (once) and replay it
template <typename F>
static int replay(rslam_ctx* c, int k, F enqueue)
{
    if (!c->graph_valid[k]) {
        if (c->graph_exec[k]) { (void)hipGraphExecDestroy(c->graph_exec[k]); c->graph_exec[k] = nullptr; }
        if (c->graph[k]) { (void)hipGraphDestroy(c->graph[k]); c->graph[k] = nullptr; }
        HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        const int rc = enqueue();
        const hipError_t e = hipStreamEndCapture(c->stream, &c->graph[k]);
        if (rc) { if (c->graph[k]) { (void)hipGraphDestroy(c->graph[k]); c->graph[k] = nullptr; } return rc; }
        if (e != hipSuccess || !c->graph[k]) { ctx_last_hip_error = (int)e; return RSLAM_ERR_HIP; }
        HIPCHK(hipGraphInstantiate(&c->graph_exec[k], c->graph[k], nullptr, nullptr, 0));
        c->graph_valid[k] = true;
        ++c->graph_captures;
    }
    HIPCHK(hipGraphLaunch(c->graph_exec[k], c->stream));
    return RSLAM_OK;
}

extern "C" int rslam_step_frame(rslam_ctx* c, int32_t use_graph)
{
    if (!c) return RSLAM_ERR_ARG;
    if (!c->have_state || !c->have_meas) return RSLAM_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    if (!use_graph || c->timing || !sweep_is_persistent(c)) return enqueue_frame(c);      // (host-sized update stage: never captured)
    const int rc = replay(c, 0, [&]() { return enqueue_frame(c); });
    if (rc) return rc;
    c->predicted = true; c->pht_done = true; c->patches_valid = false;
    c->masks_all = (c->m > 0 && c->H > 0) ? (c->cfg.dedup ? 2 : 1) : 0;        // (what enqueue_score leaves: the replay ran it)
    mark_update_enqueued(c, c->d_sup.p);
    return RSLAM_OK;
}

// Multi-GPU frame in two replayed graphs with the caller's exchange of supports in between:
// phase 0 = predict + score of [hyp_begin, hyp_end), phase 1 = consensus + updates.
extern "C" int rslam_step_phase(rslam_ctx* c, int32_t phase, int32_t hyp_begin, int32_t hyp_end, int32_t* d_supports,
                                int32_t use_graph)
{
    if (!c || !d_supports || phase < 0 || phase > 1) return RSLAM_ERR_ARG;
    if (!c->have_state || !c->have_meas) return RSLAM_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    if (phase == 0) {
        auto work = [&]() { int rc = enqueue_predict(c); if (rc) return rc; return enqueue_score(c, hyp_begin, hyp_end, d_supports); };
        if (!use_graph || c->timing) return work();
        if (c->g1_hb != hyp_begin || c->g1_he != hyp_end || c->g1_sup != d_supports) {
            c->graph_valid[1] = false; c->g1_hb = hyp_begin; c->g1_he = hyp_end; c->g1_sup = d_supports;
        }
        const int rc = replay(c, 1, work);
        if (rc) return rc;
        c->predicted = true; c->pht_done = true; c->patches_valid = false;
        c->masks_all = (c->m > 0 && hyp_end > hyp_begin) ? (c->cfg.dedup ? 2 : (hyp_begin == 0 && hyp_end == c->H) ? 1 : 0) : 0;
        return RSLAM_OK;
    }
    if (!c->predicted) return RSLAM_ERR_STATE;
    auto work = [&]() { return enqueue_update(c, d_supports); };
    if (!use_graph || c->timing || !sweep_is_persistent(c)) return work();
    if (c->g2_sup != d_supports || c->g2_masks != c->masks_all) { c->graph_valid[2] = false; c->g2_sup = d_supports; c->g2_masks = c->masks_all; }
    const int rc = replay(c, 2, work);
    if (rc) return rc;
    mark_update_enqueued(c, d_supports);
    return RSLAM_OK;
}

// ------------------------------------------------------------------------
// hypothesis-sharded frame with the exchange inside (RCCL bound at run time)
// ------------------------------------------------------------------------
namespace {
typedef int (*nccl_allgather_fn)(const void*, void*, size_t, int /* ncclDataType_t */, void* /* ncclComm_t */, hipStream_t);
typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int /* ncclDataType_t */, int /* ncclRedOp_t */, void* /* ncclComm_t */, hipStream_t);
typedef int (*nccl_comm_query_fn)(void* /* const ncclComm_t */, int*);
constexpr int NCCL_INT32 = 2;               // ncclInt32 / ncclInt, rccl.h
constexpr int NCCL_UINT64 = 5;              // ncclUint64, rccl.h
constexpr int NCCL_MAX = 2;                 // ncclMax, rccl.h

struct RcclBinding { nccl_allgather_fn allgather; nccl_allreduce_fn allreduce; nccl_comm_query_fn count, user_rank; };

const RcclBinding& bind_rccl()
{
    static const RcclBinding b = []() -> RcclBinding {
        // the RCCL the process already uses (the caller created its communicator with it), else the system one
        void* h = nullptr;
        auto find = [&](const char* name) -> void* {
            void* sym = dlsym(RTLD_DEFAULT, name);
            if (!sym) {
                if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
                if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
                if (h) sym = dlsym(h, name);
            }
            return sym;
        };
        RcclBinding r;
        r.allgather = reinterpret_cast<nccl_allgather_fn>(find("ncclAllGather"));
        r.allreduce = reinterpret_cast<nccl_allreduce_fn>(find("ncclAllReduce"));
        r.count = reinterpret_cast<nccl_comm_query_fn>(find("ncclCommCount"));
        r.user_rank = reinterpret_cast<nccl_comm_query_fn>(find("ncclCommUserRank"));
        return r;
    }();
    return b;
}
}  // namespace

static int shard_frame(rslam_ctx* c, void* nccl_comm, int32_t rank, int32_t world, int32_t use_graph, bool by_allreduce);

extern "C" int rslam_shard_frame(rslam_ctx* c, void* nccl_comm, int32_t rank, int32_t world, int32_t use_graph)
{
    return shard_frame(c, nccl_comm, rank, world, use_graph, false);
}

// The same frame with north_star's literal collective: ONE ncclAllReduce(MAX) of an 8-byte key instead of the all-gather of the
// supports (kernels.hip shard_key_kernel).  Only without the adaptive stop (adaptive = 0: every hypothesis is evaluated, so
// the consensus is the earliest strict maximum and nothing else of the list matters): RSLAM_ERR_ARG otherwise.
extern "C" int rslam_shard_frame_allreduce(rslam_ctx* c, void* nccl_comm, int32_t rank, int32_t world, int32_t use_graph)
{
    if (c && c->cfg.adaptive) return RSLAM_ERR_ARG;
    return shard_frame(c, nccl_comm, rank, world, use_graph, true);
}

static int shard_frame(rslam_ctx* c, void* nccl_comm, int32_t rank, int32_t world, int32_t use_graph, bool by_allreduce)
{
    if (!c || world < 1 || rank < 0 || rank >= world || (!nccl_comm && world > 1)) return RSLAM_ERR_ARG;
    if (!c->have_state || !c->have_meas) return RSLAM_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    const int H = c->H;
    const int chunk = (H + world - 1) / world > 0 ? (H + world - 1) / world : 1;
    const int begin = rank * chunk < H ? rank * chunk : H;
    const int end = begin + chunk < H ? begin + chunk : H;
    int r1 = c->d_sup_local.ensure((size_t)chunk), r2 = c->d_sup_all.ensure((size_t)chunk * world);
    if (r1 < 0 || r2 < 0) return RSLAM_ERR_HIP;
    if (r1 > 0 || r2 > 0) {                    // new buffers: the graphs hold the old pointers; a short last slice leaves a tail
        invalidate_graph(c);
        HIPCHK(hipMemsetAsync(c->d_sup_local.p, 0, sizeof(int32_t) * chunk, c->stream));
        HIPCHK(hipMemsetAsync(c->d_sup_all.p, 0, sizeof(int32_t) * (size_t)chunk * world, c->stream));
    }
    if (by_allreduce && c->d_shard_key.ensure(2) < 0) return RSLAM_ERR_HIP;
    nccl_allgather_fn allgather = nullptr;
    nccl_allreduce_fn allreduce = nullptr;
    if (nccl_comm) {
        const RcclBinding& rccl = bind_rccl();
        allgather = rccl.allgather;
        allreduce = rccl.allreduce;
        if (by_allreduce ? !allreduce : !allgather) return RSLAM_ERR_COMM;
        if (c->checked_comm != nccl_comm || c->checked_rank != rank || c->checked_world != world) {
            // a communicator of another size, or this process under another index in it, would leave the all-gather hanging
            // (or scatter the slices wrongly): checked once per (communicator, rank, world), before anything of the frame is
            // enqueued -- a refused call leaves the context as it was
            int cnt = -1, me = -1;
            if (!rccl.count || !rccl.user_rank || rccl.count(nccl_comm, &cnt) != 0 || rccl.user_rank(nccl_comm, &me) != 0) return RSLAM_ERR_COMM;
            if (cnt != world || me != rank) return RSLAM_ERR_COMM;
            c->checked_comm = nccl_comm; c->checked_rank = rank; c->checked_world = world;
        }
    }
    // phase 0 indexes the support array by global hypothesis id
    int rc = rslam_step_phase(c, 0, begin, end, c->d_sup_local.p - begin, use_graph);
    if (rc) return rc;
    int32_t* full = c->d_sup_local.p;
    if (by_allreduce) {
        // slice -> key, MAX over the ranks (8 bytes on the wire), key -> one-hot list: the consensus replay of phase 1 then finds
        // the earliest strict maximum of the WHOLE list, as on the gathered supports (Tracking.cpp:507-537)
        unsigned long long* key = c->d_shard_key.p;
        launch_shard_key(c->stream, c->d_sup_local.p, begin, end - begin, key);
        const unsigned long long* reduced = key;
        if (nccl_comm) {
            if (allreduce(key, key + 1, 1, NCCL_UINT64, NCCL_MAX, nccl_comm, c->stream) != 0) return RSLAM_ERR_COMM;
            reduced = key + 1;
        }
        launch_shard_expand(c->stream, reduced, c->d_sup_all.p, H);
        HIPCHK(hipGetLastError());
        full = c->d_sup_all.p;
    } else if (nccl_comm) {
        if (allgather(c->d_sup_local.p, c->d_sup_all.p, (size_t)chunk, NCCL_INT32, nccl_comm, c->stream) != 0) return RSLAM_ERR_COMM;
        full = c->d_sup_all.p;
    }
    return rslam_step_phase(c, 1, 0, H, full, use_graph);
}

extern "C" int rslam_sync(rslam_ctx* c)
{
    if (!c) return RSLAM_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    const int status = read_status(c, nullptr);
    collect_times(c);
    return status;
}

extern "C" int rslam_fetch_prediction(rslam_ctx* c, double* h, uint8_t* visible, double* S)
{
    if (!c) return RSLAM_ERR_ARG;
    if (!c->predicted) return RSLAM_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    const int L = c->L;
    if (L > 0) {
        if (h) HIPCHK(hipMemcpyAsync(h, c->d_h.p, sizeof(double) * 2 * L, hipMemcpyDeviceToHost, c->stream));
        if (S) HIPCHK(hipMemcpyAsync(S, c->d_S.p, sizeof(double) * 4 * L, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(c->h_vis.data(), c->d_vis.p, L, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    if (visible && L > 0) memcpy(visible, c->h_vis.data(), L);
    return RSLAM_OK;
}

extern "C" int rslam_fetch_supports(rslam_ctx* c, int32_t* supports, uint64_t* masks, int32_t* n_mask_words)
{
    if (!c) return RSLAM_ERR_ARG;
    if (!c->have_meas) return RSLAM_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    if (n_mask_words) *n_mask_words = c->words;
    if (supports && c->H) HIPCHK(hipMemcpyAsync(supports, c->d_sup.p, sizeof(int32_t) * c->H, hipMemcpyDeviceToHost, c->stream));
    if (masks && c->H && c->words) {
        if (c->cfg.dedup) {
            // expand per-position masks to per-hypothesis masks on the host
            std::vector<uint64_t> pm((size_t)c->m * c->words);
            std::vector<int32_t> pos((size_t)c->H);
            HIPCHK(hipMemcpyAsync(pm.data(), c->d_posmask.p, sizeof(uint64_t) * pm.size(), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(pos.data(), c->d_pos.p, sizeof(int32_t) * c->H, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            for (int i = 0; i < c->H; ++i)
                memcpy(masks + (size_t)i * c->words, pm.data() + (size_t)pos[i] * c->words, sizeof(uint64_t) * c->words);
        } else {
            HIPCHK(hipMemcpyAsync(masks, c->d_masks.p, sizeof(uint64_t) * (size_t)c->H * c->words, hipMemcpyDeviceToHost, c->stream));
        }
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return RSLAM_OK;
}

// ------------------------------------------------------------------------
// kernel-level entry points
// ------------------------------------------------------------------------
extern "C" int rslam_k_rank_update(rslam_ctx* c, int32_t n, int32_t r, const double* dA, int32_t lda,
                                   const double* dY, int32_t ldy, double* dC, int32_t ldc)
{
    if (!c || !dA || !dY || !dC || n <= 0 || r < 0) return RSLAM_ERR_ARG;
    const int NP = round_up(n, 64), K = round_up(r, TG_KC_HOST);
    if (lda < NP || ldc < NP || ldy < NP) return RSLAM_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    launch_rank_update(c->stream, NP, dA, lda, dY, ldy, c->d_sel.p, 0, K, dC, ldc, tile_order(c, NP), nullptr, 0, nullptr);
    HIPCHK(hipGetLastError());
    return RSLAM_OK;
}

// K10 on its own: `reps` back-to-back launches of the stand-alone rank update on context-owned buffers of the given shape
// (P: round_up(n, 64)^2, Y: round_up(n, 64) x round_up(r, 32), constant fill: the kernel's time does not depend on the
// values), bracketed by hipEvents on the context's stream.  us_per_launch = mean duration of one launch.
extern "C" int rslam_k_rank_update_time(rslam_ctx* c, int32_t n, int32_t r, int32_t reps, double* us_per_launch)
{
    if (!c || !us_per_launch || n <= 0 || r <= 0 || reps <= 0) return RSLAM_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    const int NP = round_up(n, 64), K = round_up(r, TG_KC_HOST);
    double *P = nullptr, *Y = nullptr;
    HIPCHK(hipMalloc((void**)&P, sizeof(double) * (size_t)NP * NP));
    if (hipMalloc((void**)&Y, sizeof(double) * (size_t)NP * K) != hipSuccess) { (void)hipFree(P); return RSLAM_ERR_HIP; }
    hipStream_t s = c->stream;
    (void)hipMemsetAsync(P, 0, sizeof(double) * (size_t)NP * NP, s);
    (void)hipMemsetAsync(Y, 0, sizeof(double) * (size_t)NP * K, s);
    const int32_t* order = tile_order(c, NP);
    for (int i = 0; i < 3; ++i) launch_rank_update(s, NP, P, NP, Y, NP, c->d_sel.p, 0, K, P, NP, order, nullptr, 0, nullptr);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a, s);
    for (int i = 0; i < reps; ++i) launch_rank_update(s, NP, P, NP, Y, NP, c->d_sel.p, 0, K, P, NP, order, nullptr, 0, nullptr);
    (void)hipEventRecord(b, s);
    const hipError_t e = hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    (void)hipFree(P); (void)hipFree(Y);
    if (e != hipSuccess) return RSLAM_ERR_HIP;
    *us_per_launch = (double)ms * 1e3 / reps;
    return RSLAM_OK;
}

extern "C" int rslam_k_gemm_nt(rslam_ctx* c, int32_t m, int32_t n, int32_t k, double alpha, const double* dA,
                               int32_t lda, const double* dB, int32_t ldb, double beta, double* dC, int32_t ldc)
{
    if (!c || !dA || !dB || !dC) return RSLAM_ERR_ARG;
    if (m <= 0 || n <= 0 || k <= 0 || (m % 64) || (n % 64) || (k % TG_KC_HOST)) return RSLAM_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    launch_gemm_nt(c->stream, m, n, k, alpha, dA, lda, dB, ldb, beta, dC, ldc);
    HIPCHK(hipGetLastError());
    return RSLAM_OK;
}

// waves_per_simd 1..8; mode in the high bits: waves_per_simd + 16 * mode (0: 4 accumulators of 16x16x4,
// 1: 8 accumulators, 2: 4x4x4_4b); cycles_per_mfma and clock_mhz (s_memtime / s_memrealtime) may be NULL
extern "C" int rslam_k_mfma_f64_probe(rslam_ctx* c, int32_t waves_and_mode, double* tflops, double* cycles_per_mfma,
                                      double* clock_mhz)
{
    const int waves_per_simd = waves_and_mode & 15, mode = waves_and_mode >> 4;
    if (!c || !tflops || waves_per_simd < 1 || waves_per_simd > 8 || mode < 0 || mode > 2) return RSLAM_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, c->device));
    const int blocks = prop.multiProcessorCount * waves_per_simd, iters = 20000;
    if (c->d_probe.ensure((size_t)blocks * 256 + 8) < 0) return RSLAM_ERR_HIP;
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(c->d_probe.p + (size_t)blocks * 256);
    launch_mfma_probe(c->stream, blocks, 2000, mode, c->d_probe.p, nullptr);       // warm-up
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    HIPCHK(hipEventRecord(a, c->stream));
    launch_mfma_probe(c->stream, blocks, iters, mode, c->d_probe.p, stamps);
    HIPCHK(hipEventRecord(b, c->stream));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    unsigned long long st[2] = {0, 0};
    HIPCHK(hipMemcpy(st, stamps, sizeof(st), hipMemcpyDeviceToHost));
    const double flops = (double)blocks * 4.0 * (double)iters * 4.0 * (mode == 2 ? 512.0 : 2048.0);   // per MFMA
    *tflops = flops / ((double)ms * 1e-3) * 1e-12;
    // one SIMD executed waves_per_simd * 4 * iters MFMAs during st[0] shader cycles
    if (cycles_per_mfma) *cycles_per_mfma = (double)st[0] / ((double)iters * 4.0 * waves_per_simd);
    if (clock_mhz) *clock_mhz = st[1] ? (double)st[0] / (double)st[1] * 100.0 : 0.0;
    return RSLAM_OK;
}

extern "C" int rslam_k_mfma_f64_peak(rslam_ctx* c, double* tflops)
{
    return rslam_k_mfma_f64_probe(c, 2, tflops, nullptr, nullptr);
}

// d[64] = one v_mfma_f64_4x4x4_4b_f64(a[64], b[64], c[64]) with the given CBSZ/ABID (host pointers)
extern "C" int rslam_k_mfma4_raw(rslam_ctx* c, int32_t cbsz, int32_t abid, const double* a, const double* b,
                                 const double* cc, double* d)
{
    if (!c || !a || !b || !cc || !d) return RSLAM_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    if (c->d_probe.ensure(256) < 0) return RSLAM_ERR_HIP;
    double* p = c->d_probe.p;
    HIPCHK(hipMemcpy(p, a, 512, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p + 64, b, 512, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p + 128, cc, 512, hipMemcpyHostToDevice));
    launch_mfma4_raw(c->stream, cbsz, abid, p, p + 64, p + 128, p + 192);
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(d, p + 192, 512, hipMemcpyDeviceToHost));
    return RSLAM_OK;
}

// which bounded wait of the last frame ran out (0: none): the raw device-side code that rslam_sync folds into RSLAM_ERR_HIP
extern "C" int rslam_last_raw_status(rslam_ctx* c) { return c ? c->last_raw_status : 0; }
extern "C" int rslam_last_wait_detail(rslam_ctx* c) { return c ? c->last_wait_first : 0; }
extern "C" int rslam_last_wait_polls(rslam_ctx* c) { return c ? c->last_wait_polls : 0; }
// how the update stage of the loaded frame shape runs: 0 launch-per-step sweep + stand-alone rank update, 1 persistent sweep +
// stand-alone rank update, 2 persistent sweep with the x / covariance update inside its launch
extern "C" int rslam_update_mode(rslam_ctx* c)
{
    if (!c) return RSLAM_ERR_ARG;
    SystemDims d; d.n = c->n; d.NP = c->NP; d.RP = c->RP; d.ldA = c->ldA;
    if (!sweep_is_persistent(c)) return c->staged.updates > 0 ? 3 : 0;
    return sweep_fused_eligible(d) ? 2 : 1;
}

#if defined(RSLAM_DEBUG)
// ------------------------------------------------------------------------
// Diagnostic variant of the library only (librslam_hip_dbg.so, -DRSLAM_DEBUG): value-level probes, time stamps and fault
// injection for the tests and scripts of this repository.  None of this is compiled into the product library.
// ------------------------------------------------------------------------
// the squared residuals score_kernel compares with sigma_z^2, for every hypothesised position of the resident frame
// (out: host, m * m, row = matched rank of the hypothesised feature); needs a frame whose scoring stage has been enqueued
extern "C" int rslam_debug_score_residuals(rslam_ctx* c, double* out, int32_t* m_out)
{
    if (!c) return RSLAM_ERR_ARG;
    if (m_out) *m_out = c->m;
    if (!out) return RSLAM_OK;
    if (!c->predicted || !c->have_meas || !c->pht_done) return RSLAM_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    const int m = c->m;
    if (m == 0) return RSLAM_OK;
    if (c->d_probe.ensure((size_t)m * m) < 0) return RSLAM_ERR_HIP;
    launch_score_residuals(c->stream, c->cam, c->d_xpred.p, c->d_W.p, c->NP, c->d_wv.p, tables(c), c->d_z.p, m, c->d_probe.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, c->d_probe.p, sizeof(double) * (size_t)m * m, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return RSLAM_OK;
}

// distort_fm_score (scoring kernel) and distort_fm (everything else; the reference's ten Newton steps) on n undistorted pixels
extern "C" int rslam_debug_distort(rslam_ctx* c, int32_t n, const double* uv, double* out_score, double* out_ref)
{
    if (!c || n <= 0 || !uv || !out_score || !out_ref) return RSLAM_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    if (c->d_probe.ensure((size_t)6 * n) < 0) return RSLAM_ERR_HIP;
    double* d = c->d_probe.p;
    HIPCHK(hipMemcpyAsync(d, uv, sizeof(double) * 2 * n, hipMemcpyHostToDevice, c->stream));
    launch_distort_probe(c->stream, c->cam, n, d, d + 2 * (size_t)n, d + 4 * (size_t)n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_score, d + 2 * (size_t)n, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(out_ref, d + 4 * (size_t)n, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return RSLAM_OK;
}

// RSLAM_SWEEP_EXP switches from the host, -1 = environment.  PROCESS-WIDE, like the two stamp buffers below (one per
// process, the last stamped launch wins): the tests and scripts drive one context at a time.
extern "C" int rslam_debug_set_sweep_exp(int mask) { rslam::set_sweep_exp_mask(mask); return RSLAM_OK; }
// fault injection, per context: riders of the stand-alone rank update that sit behind the tiles never publish Jnorm
extern "C" int rslam_debug_set_k10_inject(rslam_ctx* c, int on) { if (!c) return RSLAM_ERR_ARG; c->k10_inject = on ? 1 : 0; invalidate_graph(c); return RSLAM_OK; }

// time stamps of the persistent factor sweep, see scripts/sweep_stamps.py
namespace rslam { int debug_sweep_stamps(unsigned long long* out, int enable); }
extern "C" int rslam_debug_sweep_stamps(rslam_ctx* c, unsigned long long* out, int enable)
{
    if (!c) return RSLAM_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    invalidate_graph(c);                      // captured launches hold the old debug pointer
    return rslam::debug_sweep_stamps(out, enable) == 0 ? RSLAM_OK : RSLAM_ERR_HIP;
}

namespace rslam { int debug_k10_stamps(unsigned long long* out, int enable); }
extern "C" int rslam_debug_k10_stamps(rslam_ctx* c, unsigned long long* out, int enable)
{
    if (!c) return RSLAM_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    invalidate_graph(c);                      // captured launches hold the old debug pointer
    return rslam::debug_k10_stamps(out, enable) == 0 ? RSLAM_OK : RSLAM_ERR_HIP;
}

#if defined(CD_TIMELINE)
namespace rslam { int debug_read_cd_log(unsigned long long* out); }
extern "C" int rslam_debug_cd_log(rslam_ctx* c, unsigned long long* out)
{
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return rslam::debug_read_cd_log(out) == 0 ? RSLAM_OK : RSLAM_ERR_HIP;
}
#endif
#if defined(CD_STAMPS) || defined(CD_SPINS)
namespace rslam { int debug_read_cd_stamps(unsigned long long* out, int reset); }
extern "C" int rslam_debug_cd_stamps(rslam_ctx* c, unsigned long long* out, int reset)
{
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return rslam::debug_read_cd_stamps(out, reset) == 0 ? RSLAM_OK : RSLAM_ERR_HIP;
}
#endif
#endif   // RSLAM_DEBUG

extern "C" int rslam_k_hbm_copy_peak(rslam_ctx* c, int64_t bytes, double* gbps)
{
    if (!c || !gbps || bytes < 1024) return RSLAM_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    const long n = (long)(bytes / 16) * 2;
    double *src = nullptr, *dst = nullptr;
    HIPCHK(hipMalloc((void**)&src, n * sizeof(double)));
    if (hipMalloc((void**)&dst, n * sizeof(double)) != hipSuccess) { (void)hipFree(src); return RSLAM_ERR_HIP; }
    (void)hipMemsetAsync(src, 0, n * sizeof(double), c->stream);
    launch_copy_probe(c->stream, src, dst, n);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a, c->stream);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) launch_copy_probe(c->stream, src, dst, n);
    (void)hipEventRecord(b, c->stream);
    (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    (void)hipFree(src); (void)hipFree(dst);
    *gbps = 2.0 * (double)n * 8.0 * reps / ((double)ms * 1e-3) * 1e-9;
    return RSLAM_OK;
}
