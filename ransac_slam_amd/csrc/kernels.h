// kernels.h -- launch interface of the HIP kernels (internal; the public
// boundary is include/rslam.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include <functional>
#include "camera_model.h"

namespace rslam {

// d_sel[] slots (device-side frame scalars)
enum SelSlot {
    SEL_BEST_HYP = 0, SEL_BEST_SUPPORT = 1, SEL_HYPS_EVALUATED = 2,
    SEL_K_LI = 3, SEL_K_HI = 4, SEL_STATUS = 5, SEL_NBLK_LI = 6, SEL_NBLK_HI = 7, SEL_XU_FLAG = 8,
    SEL_WAIT_FIRST = 9,      // diagnosis: the FIRST bounded wait of the persistent sweep that ran out since the host last looked
                             // (code | workgroup << 8 | what it needed << 20; the status word keeps the smallest code only, and one
                             // wait that runs out makes others run out behind it); kept across the frame reset, cleared by the host
    SEL_STATUS_FRONT = 10,   // status of the prediction / scoring stage (kept when only the update stage is re-run)
    SEL_STICKY = 11,         // smallest status of the frames whose status word the next frame's reset has overwritten unread
    SEL_LI_DEFER = 12,       // != 0: the covariance of the (rank <= 4) low-innovation update has not been written: it is
                             // P_li = J (sym(P_pred) - Y1 Y1^T) J^T with Y1 (n x 4) and J (Jnorm of that update) kept aside,
                             // and every reader of P_li until the high-innovation pass has written P forms it on the fly
    SEL_LI_NEED = 13,        // out of band, not a status code (the status words are min-folded: a code there competes with others):
                             // bit 0 = THIS frame has other than zero, one or two low-innovation inliers and its launch sequence
                             // had no low-innovation sweep (rslam_api.hip li_skip); the frame reset turns it into bit 1 = "an
                             // earlier, unsynchronised frame had"; cleared by the host
    SEL_WAIT_POLLS = 14,     // diagnosis, written with SEL_WAIT_FIRST by the wait that ran out first: its own polls (16 bits,
                             // saturating) << 16 | elapsed wall clock in microseconds (16 bits, saturating) -- few polls in a
                             // long time = the wave was not running (queue eviction), many = the hand-over really was late
    SEL_COUNT = 16
};

constexpr int TG_KC_HOST = 32;   // K granularity of the MFMA tile engine (tile_gemm.h TG_KC)

// The deferred low-innovation covariance (SEL_LI_DEFER): what a reader of P_li needs to form its entries itself.
// P_li(a, b) = sum J(a, a') M(a', b') J(b, b'),  M = 1/2 (P_pred + P_pred^T) - Y1 Y1^T,  J = I but for rows / columns 3..6,
// where it is the 4 x 4 Jnorm T1 (column-major) of the low-innovation update (ExtendKF.cpp:608-609,629-634).
struct DeferArgs {
    const int32_t* flag;      // sel + SEL_LI_DEFER (nullptr: this launch never reads a deferred covariance)
    const double* Ppred; long ldp;
    const double* Y1; long ldy;     // n x 4, zero columns beyond the update's rank
    const double* T1;
    double* Gd;               // L x 34: per feature G = H J (2 x 13, rows at 0 and 17) and D = G Y1(cols, :) (2 x 4, at 13 and 30),
                              // written by the rescue prediction, read by the second P H^T
};

// The rescue gate (Tracking.cpp:584-595) without a launch of its own: the rescue prediction decides every feature's flag
// (GateArgs: the group that has just produced S_i and h_i evaluates nu' S^-1 nu < chi2 for it) and every workgroup of the
// second P H^T finds its feature -- the c-th flagged one -- and the count from the L flag bytes itself (GateList: a ballot scan,
// two barriers per 256 features); its first workgroups write the ordered list and the counts the HI sweep reads.
struct GateArgs { const uint8_t* ic; const uint8_t* li; const double* z; double chi2; uint8_t* hi; };
struct GateList { const uint8_t* flags; int L; int32_t* list_out; int32_t* sel; };

struct ScoreTables {          // per matched feature (rank j in feature order), m entries each
    const int32_t* feat;      // feature index
    const int32_t* off;       // state offset of the feature
    const uint8_t* type;      // RSLAM_FEAT_*
    const int32_t* ith;       // state index read as theta (Q1 in compat mode)
    const int32_t* iph;       // state index read as phi
    const int32_t* zsrc;      // feature whose z is compared (Q2 in compat mode)
    const double* sc;         // 4 per matched feature: sin, cos of x[ith], sin, cos of x[iph] at the prior (written by launch_pht)
    const double* hctx;       // 16 per matched feature: camera pose (7) and rotation (9) of the hypothesis "this feature alone"
                              // (Tracking.cpp:420-448), written by launch_pht beside the innovation solve
};

// h_in / has_h_in: previous prediction (nullable); sel_reset: frame scalars to zero (nullable)
void launch_predict(hipStream_t s, const Cam& cam, const double* x, const double* P, int NP, int L,
                    const uint8_t* type, const int32_t* off, const double* h_in, const uint8_t* has_h_in,
                    double* h, uint8_t* has_h, uint8_t* vis, double* H13, double* S, double radd, int32_t* sel_reset,
                    const DeferArgs* defer = nullptr /* P is P_li: possibly deferred (rescue prediction) */,
                    const GateArgs* gate = nullptr /* with defer: the rescue flags hi[] in the same launch */);

// out[:, 2c+p] = P[:, cols(list[c])] * H13[list[c]][p]^T for c < count; with wv != nullptr also
// wv[2c..] = S_f^-1 (z_f - h_f) (K3)
void launch_pht(hipStream_t s, const double* P, int NP, const int32_t* list, int max_count,
                const int32_t* d_count /* nullable */, const double* H13, const int32_t* off,
                const uint8_t* type, double* out, long ldo, const double* S, const double* z, const double* h,
                const uint8_t* has_h, double* wv, int32_t* status,
                const double* x = nullptr, const int32_t* ith = nullptr, const int32_t* iph = nullptr, double* sc = nullptr /* with wv:
                the angle table of ScoreTables::sc */, const DeferArgs* defer = nullptr /* P is P_li: possibly deferred */,
                const GateList* gl = nullptr /* with defer: list / count come from the rescue flags (no launch_rescue_gate) */,
                double* hctx = nullptr /* with wv: ScoreTables::hctx */);

// The launch-per-step route (systems too large for the persistent sweep) sizes its launch sequence on the host from the
// frame's own counts.  Besides sel[] the kernel that decides a count writes {count, blocks} and then `seq` into page-locked,
// host-mapped memory (system-scope stores): the host spins on seq instead of paying a copy and a stream synchronisation.
struct HostCounts { int32_t* p; int32_t seq; };      // p[0] = count, p[1] = blocks, p[2] = seq (nullptr: sel[] only)

// K5 (launch_best_mask): what the consensus and the winner's inlier list read and write
struct SelectArgs {
    const int32_t* pos; int L; int32_t* sel; uint8_t* li; int32_t* list; const int32_t* sup; int H;
    const int32_t* nhyp_table; int adaptive; int n_hyp_init;
    // the inlier masks this frame's scoring launch(es) wrote for EVERY hypothesis (row = hypothesis, or its position when
    // mask_by_pos), `words` 64-bit words per row; nullptr: the winner is scored again
    const uint64_t* masks; int words; int mask_by_pos;
    HostCounts host;          // launch-per-step route: the update's counts also go to host-visible memory
};

void launch_score(hipStream_t s, const Cam& cam, const double* x, const double* W, int NP,
                  const double* wv, const ScoreTables& tab, const double* z, int m, int words,
                  const int32_t* pos_list /* nullable = identity */, int n_entries,
                  double threshold, int32_t* sup_out, uint64_t* masks_out);

// diagnostics for the value-level tests of the scoring arithmetic: squared residuals of every (position, feature) pair;
// distort_fm_score next to the ten-step distort_fm on caller-supplied undistorted pixels
void launch_score_residuals(hipStream_t s, const Cam& cam, const double* x, const double* W, int NP, const double* wv,
                            const ScoreTables& tab, const double* z, int m, double* out /* m * m */);
void launch_distort_probe(hipStream_t s, const Cam& cam, int n, const double* uv, double* out_score, double* out_ref);

void launch_map_support(hipStream_t s, const int32_t* possup, const int32_t* pos, int hb, int he,
                        int32_t* sup);

// K5: consensus replay over the full support list, then re-score the winning hypothesis and
// scatter its mask to li[], build list/count
// the consensus exchange as one 8-byte MAX all-reduce (rslam_shard_frame_allreduce): a slice folded into
// support << 32 | (0xFFFFFFFF - index), and the reduced key expanded into a one-hot support list of H entries
void launch_shard_key(hipStream_t s, const int32_t* sup_local, int begin, int n, unsigned long long* key);
void launch_shard_expand(hipStream_t s, const unsigned long long* key, int32_t* sup_all, int H);
void launch_best_mask(hipStream_t s, const Cam& cam, const double* x, const double* W, int NP,
                      const double* wv, const ScoreTables& tab, const double* z, int m,
                      const int32_t* pos, double threshold, int L, int32_t* sel, uint8_t* li,
                      int32_t* list, const int32_t* sup, int H, const int32_t* nhyp_table, int adaptive, int n_hyp_init,
                      const uint64_t* masks = nullptr /* every hypothesis' inlier mask of this frame, else the winner is scored again */,
                      int words = 0, int mask_by_pos = 0, HostCounts host = HostCounts{nullptr, 0},
                      const struct LiSmallArgs* li_small = nullptr /* the rank <= 4 low-innovation update inside this launch */);

void launch_rescue_gate(hipStream_t s, int L, const uint8_t* ic, const uint8_t* li, const uint8_t* has_h,
                        const double* S, const double* z, const double* h, double chi2,
                        uint8_t* hi, int32_t* list, int32_t* sel, HostCounts host = HostCounts{nullptr, 0});

struct SystemDims { int n, NP, RP, ldA; };   // stacked matrix A: rows [0,RP) S | [RP,RP+NP) W | RP+NP: nu^T (+63 pad)

// Wsrc != nullptr: the P*H^T columns are gathered from the matched-feature matrix first (LI pass)
void launch_prepare_system(hipStream_t s, const SystemDims& d, const int32_t* list, const int32_t* sel,
                           int slot_k, int slot_nblk, const double* H13, const int32_t* off,
                           const uint8_t* type, const double* z, const double* h, double* A,
                           const double* Wsrc, const int32_t* rank_of, int32_t* sweep_flags /* SWEEP_FLAG_INTS, zeroed here; nullable */);
constexpr int SWEEP_FLAG_INTS = 112 + 256 + 128;   // hand-over flags of the persistent factor sweep (kernels.hip SweepFlags)
bool sweep_persistent_eligible(const SystemDims& d);
// The covariance side of an update done INSIDE the persistent sweep launch (K9, K10, K11 without launches of their own):
// the compute units the sweep leaves idle run tile workers that keep P - Y Y^T of their lower-triangle tile pairs in MFMA
// accumulators and consume each 64-column block of Y as the P H^T strips publish it; the strips accumulate x + Y u.
struct WorkerArgs {
    const double* Pin; long ldp; double* Pout; long ldo;   // Pout == nullptr: not fused (the caller launches the rank update)
    const int32_t* tile_order; int nT;                     // XCD-aware tile list (nullable) and tiles per side
    const double* x_in; double* x_out; double* T;          // K9: x_k_k, Jnorm (4 x 4)
    int compat; int token; int32_t* xu_flag;               // token published in *xu_flag when x_k_k and Jnorm are out
    int li_done_slot;                                      // sel[] slot that tells the HI pass whether the LI pass wrote P (mirror tiles equal), -1: never
    // deferred tiny low-innovation update (SEL_LI_DEFER): the LI launch (token 1) writes Y1 and sets *defer_flag instead of
    // streaming P; the HI launch (token 2) starts its tiles from P_pred, Y1 and T_li when the flag is up
    int32_t* defer_flag; double* Y1; long ldy1; const double* Ppred; const double* T_li;
};
bool sweep_fused_eligible(const SystemDims& d);            // enough idle compute units for every tile pair of P
#if defined(RSLAM_DEBUG)
void set_sweep_exp_mask(int mask);     // -1 = environment (RSLAM_SWEEP_EXP); diagnostics and fault injection (diagnostic variant only)
#endif
int sweep_exp_mask();                  // always 0 in the product build
// What the persistent sweep builds its stacked system [S; P H^T; nu^T] from (it has no prepare_system pass): the
// arguments of launch_prepare_system.  Wsrc != nullptr: P H^T columns come from the matched-feature matrix (LI pass),
// else they are already in rows [RP, RP+NP) of A (HI pass: launch_pht wrote them there).
struct SysSrc {
    const int32_t* list; const double* H13; const int32_t* off; const uint8_t* type;
    const double* z; const double* h; const double* Wsrc; const int32_t* rank_of;
};   // the whole sweep in one launch (no dependence on the previous frame's counts)
// A low-innovation update of rank <= 4 (the reference-faithful mode: the consensus set is the hypothesis' own feature, Q1) is
// one 4 x 4 factor and one row solve per state entry; its covariance is deferred (SEL_LI_DEFER).  The consensus launch does it
// itself right behind the inlier list (li_small_update): Y1, x_k_k, Jnorm, *xu_flag = 1, *defer_flag = 1.  The persistent
// sweep's low-innovation launch then returns at once (it finds *defer_flag set); the launch-per-step route enqueues nothing.
struct LiSmallArgs {
    SysSrc src;                       // list = the consensus launch's own inlier list
    int NP;
    const double* x_in; double* x_out; double* Y1; long ldy1; double* T; int compat;
    int32_t* xu_flag; int32_t* defer_flag; int32_t* status;
    // must != 0: the launch sequence has NO low-innovation sweep behind this launch (reference-faithful mode: the caller
    // counts on one or two inliers): any other count is reported out of band (sel[SEL_LI_NEED]; raw code -40 on the host) and
    // the host re-runs the update stage with the sweep in the sequence.  clear_flags (nullable): the hand-over flag set that sweep would have cleared for the next one.
    int must; int32_t* clear_flags; int n_clear;
};
// returns the buffer (A or Ystore, same shape) whose rows [RP, RP + NP] hold Y and u^T afterwards
double* launch_factor_sweep(hipStream_t s, const SystemDims& d, const int32_t* sel,
                            int slot_k, int slot_nblk, int host_blocks /* launch-per-step route: the update's block count as the host has read it */, double* A, double* Ystore, double* Linv,
                            int32_t* status_sel, int32_t* flags /* 2 * SWEEP_FLAG_INTS zeroed ints, or nullptr: never the persistent sweep */,
                            const SysSrc* src /* with flags: the sweep assembles the system itself (no launch_prepare_system) */,
                            const WorkerArgs* wk = nullptr /* persistent sweep only: x and covariance update inside the launch */);
// K9 riding in the rank-update launch: the first `groups` workgroups compute x_k_k = x + Y u (16 rows each),
// group 0 the quaternion normalisation and Jnorm, published through *flag = token (sel[SEL_XU_FLAG])
struct XuArgs {
    int groups;                 // 0 = no x update in this launch
    SystemDims d;
    const double* A;            // the system whose rows [RP, RP + NP] hold Y and u^T
    const double* x_in; double* x_out; double* T;
    int compat; int token; int32_t* flag;
    int riders_first;           // != 0: the x update in front of the tiles whatever the occupancy says (re-run after a timed-out Jnorm wait)
    int inject;                 // fault injection (tests): riders placed behind the tiles never publish Jnorm
    // deferred low-innovation covariance on the stand-alone path (rank <= 4, decided by the host from the frame's counts): the
    // launch is the x update alone -- the riders also keep the four columns of Y aside and group 0 sets *defer_flag -- and the
    // covariance is materialised by the HI pass (MatArgs): one stream over P less (134 us at C5)
    int riders_only; double* Y1out; long ldy1; int32_t* defer_flag;
    int mirror_known;           // != 0: Pin holds exactly mirrored tile pairs (a later pass of a staged update): 1/2 (P + P^T) = P
};
// What the tiles of the stand-alone rank update need to start from a deferred P_li (SEL_LI_DEFER) instead of Pin:
// M = 1/2 (P_pred + P_pred^T) - Y1 Y1^T, then the Jnorm congruence of the low-innovation update (T_li) on rows / columns 3..6
struct MatArgs { const int32_t* flag; const double* Ppred; long ldp; const double* Y1; long ldy1; const double* T_li; };
// C = sym(Pin) - Y Y^T on the lower-triangle tile pairs (K from sel[slot_nblk]*64);
// K == 0: C = Pin exactly (ExtendKF.cpp:635-638 pass-through)
void launch_rank_update(hipStream_t s, int NP, const double* Pin, long ldp, const double* Y, long ldy,
                        const int32_t* sel, int slot_nblk, int fixed_k, double* Pout, long ldo,
                        const int32_t* tile_order /* nullable: row-major triangle */,
                        const double* Tq /* nullable: 4 x 4 Jnorm, applied to rows/columns 3..6 (K11) when sel[slot_k] != 0 */,
                        int slot_k, const XuArgs* xu /* nullable */, const MatArgs* mat = nullptr /* HI pass: P_li may be deferred */,
                        int n_tiles = -1 /* >= 0: only the first n_tiles entries of tile_order */);
// rank_macro.hip: the same update on 128 x 128 macro tiles (large maps; whole rounds of one workgroup per compute unit, the
// rest of the triangle is left to launch_rank_update with the `small` list)
void make_macro_order(int nT, int cus, std::vector<int32_t>& macro, std::vector<int32_t>& small);
void launch_rank_update_macro(hipStream_t s, const double* Pin, long ldp, const double* Y, long ldy, int K, double* Pout, long ldo,
                              const int32_t* macro_order, int n_macro, const int32_t* sel, int slot_k, const double* Tq,
                              int mirror_flag, int token, const MatArgs* mat);
int init_macro_kernel_attributes();
void make_rank_update_order(int nT, std::vector<int32_t>& order);   // XCD-aware (bi << 16 | bj) per block

// ---- the staged route of large systems (staged_kernels.hip; S stage in kernels.hip) ----
void launch_s_stage(hipStream_t s, const SystemDims& d, const int32_t* sel, int slot_k, int slot_nblk, int steps,
                    double* A, double* Ystore, double* Linv, int32_t* status_sel, int cus, const std::function<void(int)>& progress);
void launch_group_inverse(hipStream_t s, int b0, int nb, const double* Lp, long ldl, const double* Linv, double* M, double* Mt, long ldm);
void launch_staged_update(hipStream_t s, const SystemDims& d, int b0, int nb, int nblk, double* A, const double* Ystore);
void launch_staged_Y(hipStream_t s, const SystemDims& d, int b0, int nb, const double* A, double* Ystore, const double* M, long ldm);
int init_staged_kernel_attributes();

// x_pred[0:13], FQ (338 doubles) and the 13-row/column strips of P_pred; the caller copies
// x[13:] and P beforehand
void launch_ekf_prediction(hipStream_t s, int n, int NP, const double* x_kk, const double* P_kk, double dt,
                           double std_a, double std_alpha, double* x_pred, double* P_pred, double* FQ);
// map_kernels.hip: Map::map_management state surgery on the resident posterior
constexpr int MAP_COEF_DOUBLES = 160;
void launch_map_state(hipStream_t s, int mode, const Cam& cam, const double* x_old, int o, double ud, double vd,
                      double rho0, double std_z, double std_rho, double* coef, double* x_new, int n_new, int NP_new,
                      int cut, int special, int shift);
void launch_map_cov(hipStream_t s, const double* P, int ld_old, double* Pn, int ld_new, int n_new, int cut, int special,
                    int shift, int sp_base, int sp_cnt, int add_r, const double* coef);
void launch_map_linearity(hipStream_t s, const double* x, const double* P, int NP, int L, const uint8_t* type,
                          const int32_t* off, double threshold, double* out, int32_t* first);
// match_kernels.hip: Tracking::matching (NCC search) on the resident prediction
void launch_match(hipStream_t s, const Cam& cam, const uint8_t* image, const double* patches, int L, const double* h,
                  const uint8_t* has_h, const double* S, double corr_threshold, double chi2, double* z, uint8_t* ic,
                  double* corr);
void launch_pred_patches(hipStream_t s, const Cam& cam, int compat, int L, const uint8_t* type, const int32_t* off,
                         const int32_t* xyz_src, const double* x, const double* h, const uint8_t* has_h,
                         const int32_t* slot, const double* rec, const float* rec_patch, double* out, int32_t* status);
// dst (rows_dst x cols_dst, ld_dst) = src (rows_src x cols_src, ld_src), zero beyond the source: the change of leading
// dimension of the covariance around a linear PCIe transfer (drop-in API)
void launch_repitch(hipStream_t s, const double* src, long ld_src, double* dst, long ld_dst, int rows_src, int rows_dst, int cols_src, int cols_dst);
int init_kernel_attributes();    // raise the dynamic-LDS limit of the MFMA kernels (80 KiB)
int init_kernel_attributes2();
void launch_gemm_nt(hipStream_t s, int M, int N, int K, double alpha, const double* A, long lda,
                    const double* B, long ldb, double beta, double* C, long ldc);
void launch_mfma_probe(hipStream_t s, int blocks, int iters, int mode, double* out, unsigned long long* stamps);
void launch_mfma4_raw(hipStream_t s, int cbsz, int abid, const double* a, const double* b, const double* c, double* d);
void launch_copy_probe(hipStream_t s, const double* src, double* dst, long n);

}  // namespace rslam
