// kernels.hip -- HIP kernels of the 1-point-RANSAC EKF update for gfx950.
//
// K1  predict_kernel        h_i, visibility, compact Jacobians, S_i
// K2  pht_kernel            P*H^T exploiting the 13 structurally non-zero columns (HBM-bound)
// K3  (in pht_kernel)       w_i = S_i^-1 (z_i - h_i)
// K4  score_kernel          hypothesis x feature inlier scoring (wave ballot/popcount)
// K5  best_mask_kernel      replay of the sequential best/adaptive-n_hyp scan, inlier set of the winner
// K6  prepare_system_kernel stacked system [S; P*H^T; nu^T]
// K8  chol_diag_kernel, sweep_step_kernel (large systems: + panel_kernel, trail_stream2_kernel)
//                           blocked right-looking Cholesky sweep, one launch per block step
// K9  xupdate_rows          x + Y u, quaternion normalisation, Jnorm (first workgroups of K10's launch)
// K10 rank_update_kernel    P - Y Y^T with symmetrisation (MFMA, lower-triangle tile pairs)
// K11 (K10's epilogue)      Jnorm congruence on rows/cols 3..6
// K12 rescue_gate_kernel    chi-square gate of the high-innovation candidates
#include "kernels.h"
#include "tile_gemm.h"
#include "rank_common.h"
#include <vector>
#include <type_traits>
#include <cstddef>
#include <cstdio>
#include <cstdlib>

namespace rslam {

// ---------------------------------------------------------------------------
// K1: measurement prediction, Jacobians, innovation covariance per feature
// Replaces ExtendKF::predict_camera_measurements (ExtendKF.cpp:56-90),
// Tracking::calculate_derivatives (Tracking.cpp:540-573) and the S_i loop of
// Tracking::search_IC_matches (Tracking.cpp:39-44) / rescue (:589).
// ---------------------------------------------------------------------------
// 16 lanes per feature: the scalar camera/Jacobian arithmetic is done redundantly by the
// group (latency-bound either way), the 13 x 13 gather of P for S_i is split one column
// per lane and reduced with shuffles.
// Two waves per four features: the kernel is ONE instruction stream of ~1900 vector instructions per feature (it runs at
// 75 waves on 1024 SIMDs: its duration is that stream).  Wave 1 runs the part of the Jacobian that does not need the
// predicted pixel (jacobian_core, ~860 instructions) while wave 0 runs the prediction (~830) and hands it over through LDS.
// GATE (rescue prediction): the group also decides the feature's rescue flag hi[i] (GateArgs) from the S, h it has just produced.
template <bool CAN_DEFER, bool GATE>      // (false: the frame's first prediction -- P is the prior as it stands: none of the deferral code)
__global__ void __launch_bounds__(128)
predict_kernel(Cam cam, const double* __restrict__ x, const double* __restrict__ P, int NP, int L,
               const uint8_t* __restrict__ type, const int32_t* __restrict__ off,
               const double* __restrict__ h_in, const uint8_t* __restrict__ has_h_in,   // nullable: no previous h
               double* __restrict__ h, uint8_t* __restrict__ has_h, uint8_t* vis, double* __restrict__ H13,
               double* __restrict__ S, double radd, int32_t* __restrict__ sel_reset /* nullable */, DeferArgs da, GateArgs ga)
{
    __shared__ double sB[4][26];
    if (sel_reset && blockIdx.x == 0 && threadIdx.x < SEL_COUNT) {   // new frame: the frame scalars start from zero ...
        // ... except that a status nobody has read yet (frames enqueued back to back without rslam_sync) is folded into a
        // sticky slot first: an error of an unchecked frame is reported by the next sync instead of being erased here
        const int old = sel_reset[threadIdx.x];
        const int st = min(__shfl(old, SEL_STATUS, 16), __shfl(old, SEL_STATUS_FRONT, 16));
        sel_reset[threadIdx.x] = ((int)threadIdx.x == SEL_STICKY) ? min(old, st)
                               : ((int)threadIdx.x == SEL_WAIT_FIRST || (int)threadIdx.x == SEL_WAIT_POLLS) ? old
                               : ((int)threadIdx.x == SEL_LI_NEED) ? (old ? 2 : 0) : 0;
    }
    const int grp = (threadIdx.x & 63) >> 4;
    const int i = blockIdx.x * 4 + grp;
    const int sub = threadIdx.x & 15;
    const bool second = threadIdx.x >= 64;
    const bool live = i < L;
    const bool is_id = live && (type[i] == 0);
    const int o = live ? off[i] : 0;
    if (second) {
        if (live) {
            double Bc[26];
            jacobian_core(cam, x, o, is_id, Bc);
            if (sub == 0) {
#pragma unroll
                for (int k = 0; k < 26; ++k) sB[grp][k] = Bc[k];
            }
        }
        __syncthreads();
        return;
    }
    const int w = is_id ? 13 : 10;
    // lane `sub` owns column jj = sub of (H P): its 13 entries of P are requested first and arrive under the camera model
    double pc[13];
    // P_li may be deferred (DeferArgs): S = H P_li H^T = G sym(P_pred) G^T - D D^T with G = H J (the Jacobian rows with their
    // compact columns 3..6 mixed by the low-innovation update's Jnorm) and D = G Y1(cols, :) (2 x 4): the same gather, from
    // P_pred, a quadratic form with G instead of H (it does not see the asymmetric part of P_pred but in the cross term,
    // which is averaged below), and a rank-4 correction.  G and D are also what the second P H^T needs per feature: stored.
    // (the flag is itself a load: nothing below waits for it -- both candidate gathers, Jnorm and the Y1 entries are requested
    //  at once and arrive under the camera model; the choice is made when they are used)
    int dflag = 0;
    if (CAN_DEFER && da.flag) dflag = *da.flag;
#pragma unroll
    for (int kk = 0; kk < 13; ++kk) pc[kk] = (live && sub < w && kk < w) ? P[col_index(o, kk) + (long)col_index(o, sub) * NP] : 0.0;
    double pd[13], y1[4] = {0.0, 0.0, 0.0, 0.0}, T1[16];
    if (CAN_DEFER && da.flag) {
#pragma unroll
        for (int kk = 0; kk < 13; ++kk) pd[kk] = (live && sub < w && kk < w) ? da.Ppred[col_index(o, kk) + (long)col_index(o, sub) * da.ldp] : 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) T1[q] = da.T1[q];
        if (live && sub < w) {
#pragma unroll
            for (int c = 0; c < 4; ++c) y1[c] = da.Y1[col_index(o, sub) + (long)c * da.ldy];
        }
    }
    bool cand = false;
    double gz0 = 0.0, gz1 = 0.0;
    if (GATE && live) { cand = ga.ic[i] && !ga.li[i]; gz0 = ga.z[2 * i]; gz1 = ga.z[2 * i + 1]; }
    double u = 0, v = 0;
    const bool visible = live && predict_feature(cam, x, o, is_id, u, v);
    const bool had = live && has_h_in && (has_h_in[i] != 0);
    const bool have = visible || had;
    double hu_ = u, hv_ = v;
    if (!visible && had) { hu_ = h_in[2 * i]; hv_ = h_in[2 * i + 1]; }    // stale h (ExtendKF.cpp:77-78)
    if (live && sub == 0) {
        if (have) { h[2 * i] = hu_; h[2 * i + 1] = hv_; }
        has_h[i] = have ? 1 : 0;
        if (vis) vis[i] = visible ? 1 : 0;
    }
    __syncthreads();                         // the other wave's half of the Jacobian is in LDS
    if (!have) {
        if (GATE && live && sub == 0) ga.hi[i] = 0;
        return;
    }
    const bool deferred = CAN_DEFER && dflag != 0;                   // (uniform)
    if (CAN_DEFER && deferred) {
#pragma unroll
        for (int kk = 0; kk < 13; ++kk) pc[kk] = pd[kk];
    }
    double Bc[26], Hc[26];
#pragma unroll
    for (int k = 0; k < 26; ++k) Bc[k] = sB[grp][k];
    jacobian_finish(cam, hu_, hv_, Bc, Hc);
    if (sub == 0) {
#pragma unroll
        for (int k = 0; k < 26; ++k) H13[26 * i + k] = Hc[k];
    }
    double d0[4] = {0.0, 0.0, 0.0, 0.0}, d1[4] = {0.0, 0.0, 0.0, 0.0};
    if (deferred) {
        // G = H J: G[3+q] = sum_i H[3+i] J(3+i, 3+q) = sum_i H[3+i] T1[i + 4 q]   (T1 column-major)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            double m[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                m[q] = T1[4 * q] * Hc[13 * p + 3] + T1[1 + 4 * q] * Hc[13 * p + 4] + T1[2 + 4 * q] * Hc[13 * p + 5] + T1[3 + 4 * q] * Hc[13 * p + 6];
#pragma unroll
            for (int q = 0; q < 4; ++q) Hc[13 * p + 3 + q] = m[q];
        }
        // D = G Y1(cols, :): lane `sub` contributes its own column, summed over the 16-lane group
        double g0s = 0.0, g1s = 0.0;
#pragma unroll
        for (int kk = 0; kk < 13; ++kk) if (kk == sub) { g0s = Hc[kk]; g1s = Hc[13 + kk]; }
#pragma unroll
        for (int c = 0; c < 4; ++c) { d0[c] = g0s * y1[c]; d1[c] = g1s * y1[c]; }
#pragma unroll
        for (int dd = 8; dd >= 1; dd >>= 1)
#pragma unroll
            for (int c = 0; c < 4; ++c) { d0[c] += __shfl_xor(d0[c], dd, 16); d1[c] += __shfl_xor(d1[c], dd, 16); }
        if (sub == 0 && da.Gd) {
            double* gd = da.Gd + 34L * i;
#pragma unroll
            for (int k = 0; k < 13; ++k) { gd[k] = (k < w) ? Hc[k] : 0.0; gd[17 + k] = (k < w) ? Hc[13 + k] : 0.0; }
#pragma unroll
            for (int c = 0; c < 4; ++c) { gd[13 + c] = d0[c]; gd[30 + c] = d1[c]; }
        }
    }
    double t0 = 0, t1 = 0, hj0 = 0, hj1 = 0;
    if (sub < w) {
#pragma unroll
        for (int kk = 0; kk < 13; ++kk) {
            if (kk < w) {
                t0 += Hc[kk] * pc[kk];
                t1 += Hc[13 + kk] * pc[kk];
            }
            if (kk == sub) { hj0 = Hc[kk]; hj1 = Hc[13 + kk]; }
        }
    }
    double s00 = t0 * hj0, s10 = t1 * hj0, s01 = t0 * hj1, s11 = t1 * hj1;
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) {
        s00 += __shfl_xor(s00, d, 16); s10 += __shfl_xor(s10, d, 16);
        s01 += __shfl_xor(s01, d, 16); s11 += __shfl_xor(s11, d, 16);
    }
    if (deferred) {
        const double sx = 0.5 * (s10 + s01);
        s00 -= (d0[0] * d0[0] + d0[1] * d0[1]) + (d0[2] * d0[2] + d0[3] * d0[3]);
        s11 -= (d1[0] * d1[0] + d1[1] * d1[1]) + (d1[2] * d1[2] + d1[3] * d1[3]);
        s10 = s01 = sx - ((d0[0] * d1[0] + d0[1] * d1[1]) + (d0[2] * d1[2] + d0[3] * d1[3]));
    }
    if (sub == 0) {
        S[4 * i + 0] = s00 + radd; S[4 * i + 1] = s10; S[4 * i + 2] = s01; S[4 * i + 3] = s11 + radd;
        if (GATE) {                              // nu' S^-1 nu < chi2 (Tracking.cpp:589-593), as rescue_gate_kernel evaluates it
            bool rescue = false;
            if (cand) {
                double Si[4] = { s00 + radd, s10, s01, s11 + radd }, Sinv[4];
                inv2_lu(Si, Sinv);
                const double n0 = gz0 - hu_, n1 = gz1 - hv_;
                const double t0g = n0 * Sinv[0] + n1 * Sinv[1];
                const double t1g = n0 * Sinv[2] + n1 * Sinv[3];
                rescue = (t0g * n0 + t1g * n1) < ga.chi2;
            }
            ga.hi[i] = rescue ? 1 : 0;
        }
    }
}

void launch_predict(hipStream_t s, const Cam& cam, const double* x, const double* P, int NP, int L,
                    const uint8_t* type, const int32_t* off, const double* h_in, const uint8_t* has_h_in,
                    double* h, uint8_t* has_h, uint8_t* vis, double* H13, double* S, double radd, int32_t* sel_reset, const DeferArgs* defer,
                    const GateArgs* gate)
{
    DeferArgs da{}; if (defer) da = *defer;
    GateArgs ga{}; if (gate) ga = *gate;
    if (L <= 0) {
        // an empty map: only the frame scalars are reset -- by the kernel itself (one workgroup, no feature), so that a status
        // nobody has read is folded into the sticky slot here too
        if (sel_reset) predict_kernel<false, false><<<dim3(1), dim3(128), 0, s>>>(cam, x, P, NP, 0, type, off, h_in, has_h_in, h, has_h, vis, H13, S, radd, sel_reset, da, ga);
        return;
    }
    const dim3 grid((L + 3) / 4), block(128);
#define PREDICT_ARGS cam, x, P, NP, L, type, off, h_in, has_h_in, h, has_h, vis, H13, S, radd, sel_reset, da, ga
    if (defer && gate) predict_kernel<true, true><<<grid, block, 0, s>>>(PREDICT_ARGS);
    else if (defer) predict_kernel<true, false><<<grid, block, 0, s>>>(PREDICT_ARGS);
    else predict_kernel<false, false><<<grid, block, 0, s>>>(PREDICT_ARGS);
#undef PREDICT_ARGS
}

// ---------------------------------------------------------------------------
// K2: out[:, 2c+p] = sum_k P[:, col_k] * H13[f][p][k] over the 13 (10) non-zero
// columns of H_f (Tracking.cpp:42,420-421 do this as dense 2 x n products).
// HBM-bound: every column of P that belongs to a listed feature is streamed once.
// K3 rides along (first row block of each column): w_j = S_j^-1 (z_j - h_j); the hypothesis
// state of Tracking.cpp:420-422 is then xi = x + (P H_j^T) w_j, the gain K = P H^T S^-1 is
// never materialised.
// ---------------------------------------------------------------------------
// pose (7) and rotation (9, rotcw = q2r(q)^T, column-major) of the state x + c0 w0 + c1 w1 restricted to the camera rows
__device__ __forceinline__ void hyp_context(const double* __restrict__ x, const double (&c0)[7], const double (&c1)[7], double w0, double w1,
                                            double* __restrict__ out16)
{
    double pose[7];
#pragma unroll
    for (int a = 0; a < 7; ++a) pose[a] = x[a] + (c0[a] * w0 + c1[a] * w1);
    double Rq[9];
    q2r(pose + 3, Rq);
#pragma unroll
    for (int a = 0; a < 7; ++a) out16[a] = pose[a];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) out16[7 + a + 3 * b] = Rq[b + 3 * a];
}

struct InnovArgs {            // all nullable together
    const double* S; const double* z; const double* h; const uint8_t* has_h; double* wv; int32_t* status;
    const double* x; const int32_t* ith; const int32_t* iph; double* sc;      // angle table of the scoring kernel (nullable)
    double* hctx;             // per matched feature: the hypothesis' camera pose (7) and rotation (9), ScoreTables::hctx (nullable)
};

template <bool CAN_DEFER>
__global__ void __launch_bounds__(256)
pht_kernel(const double* __restrict__ P, int NP, const int32_t* __restrict__ list, int max_count,
           const int32_t* __restrict__ d_count, const double* __restrict__ H13,
           const int32_t* __restrict__ off, const uint8_t* __restrict__ type, double* __restrict__ out, long ldo,
           InnovArgs iv, DeferArgs da, GateList gl)
{
    const int c = blockIdx.y;
    int count, f;
    if (CAN_DEFER && gl.flags) {
        // list and count from the rescue flags: this workgroup's feature is the c-th flagged one.  Every wave scans all L flag
        // bytes by itself -- eight loads in flight per round of 512 features, ballots, no LDS, no barrier (a workgroup-wide
        // scan with two barriers per 256 features cost a dependent load per chunk: +47 us on the 24 000 workgroups of C5)
        const int lane = threadIdx.x & 63;
        int running = 0;
        f = -1;
        for (int base = 0; base < gl.L; base += 512) {
            bool fl[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = base + 64 * u + lane; fl[u] = i < gl.L && gl.flags[i] != 0; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned long long bal = __ballot(fl[u]);
                const int cnt = __popcll(bal);
                if (f < 0 && running + cnt > c) {               // (uniform) the c-th flagged feature is one of these 64
                    const bool me = fl[u] && __popcll(bal & ((1ull << lane) - 1ull)) == c - running;
                    f = base + 64 * u + (__ffsll((unsigned long long)__ballot(me)) - 1);
                }
                running += cnt;
            }
        }
        count = running;
        if (blockIdx.x == 0 && c == 0 && threadIdx.x == 0) {
            const int nblk = (2 * count + 63) / 64;
            gl.sel[SEL_K_HI] = count;
            gl.sel[SEL_NBLK_HI] = nblk;
        }
        if (c >= count) return;
        if (blockIdx.x == 0 && threadIdx.x == 0) gl.list_out[c] = f;
    } else {
        count = d_count ? *d_count : max_count;
        if (c >= count) return;
        f = list[c];
    }
    if (iv.sc && blockIdx.x == 0 && threadIdx.x == 1) {
        double sv, cv;
        sincos(iv.x[iv.ith[c]], &sv, &cv); iv.sc[4 * c] = sv; iv.sc[4 * c + 1] = cv;
        sincos(iv.x[iv.iph[c]], &sv, &cv); iv.sc[4 * c + 2] = sv; iv.sc[4 * c + 3] = cv;
    }
    double w0 = 0.0, w1 = 0.0;              // (thread 0 of the first row block: the innovation solve of this feature)
    if (iv.wv && blockIdx.x == 0 && threadIdx.x == 0) {
        if (!iv.has_h[f]) {                     // matching() only produces z where h exists (Tracking.cpp:293)
            atomicMin(iv.status, -7);           // RSLAM_ERR_IC_NOT_VISIBLE
        } else {
            double Si[4] = { iv.S[4 * f], iv.S[4 * f + 1], iv.S[4 * f + 2], iv.S[4 * f + 3] }, Sinv[4];
            inv2_lu(Si, Sinv);
            const double n0 = iv.z[2 * f] - iv.h[2 * f], n1 = iv.z[2 * f + 1] - iv.h[2 * f + 1];
            w0 = Sinv[0] * n0 + Sinv[2] * n1;
            w1 = Sinv[1] * n0 + Sinv[3] * n1;
        }
        iv.wv[2 * c] = w0; iv.wv[2 * c + 1] = w1;
    }
    const int o = off[f];
    const double* Hf = H13 + 26 * (long)f;
    // Two consecutive rows per thread: thirteen 16-byte loads in flight per lane (a column of P is a contiguous run of rows,
    // NP is a multiple of 64 and every base is 16-byte aligned) and half as many waves to schedule for the same bytes -- the
    // kernel is a latency chain (list entry -> layout -> columns) on top of a stream of 35.6 MB at C3.
    const int row = 2 * (blockIdx.x * 256 + threadIdx.x);
    if (row >= NP) return;
    const int w = (type[f] == 0) ? 13 : 10;
    d2 a0 = {0.0, 0.0}, a1 = {0.0, 0.0};
    d2 pv[13];
    if (CAN_DEFER && da.flag && *da.flag != 0) {
        // P is the deferred P_li = J M J^T (DeferArgs): P_li H^T = J (M G^T) with M G^T = P_pred(:, cols) G^T - Y1 D^T; G = H J and
        // D = G Y1(cols, :) per feature come from the rescue prediction (da.Gd).  P_pred's columns are read as they are (it is
        // symmetric to rounding: an uploaded prior exactly, one left by rslam_ekf_prediction to ~1e-16 relative; the transposed
        // entries would be a strided gather per row); the tile workers of the HI pass symmetrise what they write.
        const double* gd = da.Gd + 34L * f;
        // state rows 3..6 (J from the left): rows 2, 3 are thread 1's, 4, 5 thread 2's, 6, 7 thread 3's of the first row block
        const bool first = blockIdx.x == 0 && threadIdx.x < 64;
        double t1a[4] = {0.0, 0.0, 0.0, 0.0}, t1b[4] = {0.0, 0.0, 0.0, 0.0};      // rows of T1 for this thread's two state rows
        if (first && threadIdx.x >= 1 && threadIdx.x <= 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (row >= 3 && row <= 6) t1a[q] = da.T1[(row - 3) + 4 * q];             // (requested with everything else)
                if (row + 1 >= 3 && row + 1 <= 6) t1b[q] = da.T1[(row + 1 - 3) + 4 * q];
            }
        }
#pragma unroll
        for (int k = 0; k < 13; ++k) pv[k] = (k < w) ? *reinterpret_cast<const d2*>(da.Ppred + row + (long)col_index(o, k) * da.ldp) : (d2){0.0, 0.0};
        d2 y[4];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) y[cc] = *reinterpret_cast<const d2*>(da.Y1 + row + (long)cc * da.ldy);
#pragma unroll
        for (int k = 0; k < 13; ++k) { a0 += pv[k] * gd[k]; a1 += pv[k] * gd[17 + k]; }
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) { a0 -= y[cc] * gd[13 + cc]; a1 -= y[cc] * gd[30 + cc]; }
        if (first) {
            // rows 3, 4, 5, 6 of both result columns, as they stand before the mix
            const double p3 = __shfl(a0.y, 1), p4 = __shfl(a0.x, 2), p5 = __shfl(a0.y, 2), p6 = __shfl(a0.x, 3);
            const double q3 = __shfl(a1.y, 1), q4 = __shfl(a1.x, 2), q5 = __shfl(a1.y, 2), q6 = __shfl(a1.x, 3);
            if (row >= 3 && row <= 6) {
                a0.x = t1a[0] * p3 + t1a[1] * p4 + t1a[2] * p5 + t1a[3] * p6;
                a1.x = t1a[0] * q3 + t1a[1] * q4 + t1a[2] * q5 + t1a[3] * q6;
            }
            if (row + 1 >= 3 && row + 1 <= 6) {
                a0.y = t1b[0] * p3 + t1b[1] * p4 + t1b[2] * p5 + t1b[3] * p6;
                a1.y = t1b[0] * q3 + t1b[1] * q4 + t1b[2] * q5 + t1b[3] * q6;
            }
        }
        *reinterpret_cast<d2*>(out + row + (long)(2 * c) * ldo) = a0;
        *reinterpret_cast<d2*>(out + row + (long)(2 * c + 1) * ldo) = a1;
        return;
    }
#pragma unroll
    for (int k = 0; k < 13; ++k) pv[k] = (k < w) ? *reinterpret_cast<const d2*>(P + row + (long)col_index(o, k) * NP) : (d2){0.0, 0.0};     // all thirteen in flight at once
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        if (k < w) { a0 += pv[k] * Hf[k]; a1 += pv[k] * Hf[13 + k]; }
    }
    *reinterpret_cast<d2*>(out + row + (long)(2 * c) * ldo) = a0;
    *reinterpret_cast<d2*>(out + row + (long)(2 * c + 1) * ldo) = a1;
    // The hypothesis "this feature alone" as the scoring kernel needs it (Tracking.cpp:420-448): camera pose x_i[0:7] = x +
    // (P H^T) w and its rotation.  Every workgroup of the scoring launch used to rebuild it in all of its lanes -- two
    // dependent loads and ~180 of its ~350 vector instructions per wave; the rows it needs, 0..6 of this feature's two
    // columns, are in the registers of lanes 0..3 right here, w in lane 0.
    if (iv.hctx && blockIdx.x == 0 && threadIdx.x < 64) {
        double c0[7], c1[7];
#pragma unroll
        for (int a = 0; a < 7; ++a) {
            c0[a] = __shfl((a & 1) ? a0.y : a0.x, a >> 1);
            c1[a] = __shfl((a & 1) ? a1.y : a1.x, a >> 1);
        }
        if (threadIdx.x == 0) hyp_context(iv.x, c0, c1, w0, w1, iv.hctx + 16L * c);
    }
}

void launch_pht(hipStream_t s, const double* P, int NP, const int32_t* list, int max_count,
                const int32_t* d_count, const double* H13, const int32_t* off, const uint8_t* type,
                double* out, long ldo, const double* S, const double* z, const double* h, const uint8_t* has_h,
                double* wv, int32_t* status, const double* x, const int32_t* ith, const int32_t* iph, double* sc, const DeferArgs* defer,
                const GateList* gl, double* hctx)
{
    if (max_count <= 0) return;
    InnovArgs iv{S, z, h, has_h, wv, status, x, ith, iph, sc, hctx};
    DeferArgs da{}; if (defer) da = *defer;
    GateList g{}; if (gl && defer) g = *gl;
    const dim3 grid(NP / 512 + (NP % 512 ? 1 : 0), max_count);          // two rows per thread
    if (defer) pht_kernel<true><<<grid, dim3(256), 0, s>>>(P, NP, list, max_count, d_count, H13, off, type, out, ldo, iv, da, g);
    else pht_kernel<false><<<grid, dim3(256), 0, s>>>(P, NP, list, max_count, d_count, H13, off, type, out, ldo, iv, da, g);
}

// ---------------------------------------------------------------------------
// K4: hypothesis scoring (Tracking.cpp:422-503).  One workgroup per hypothesis,
// one lane per matched feature.  xi entries are formed on the fly from x and the
// two P*H^T columns of the hypothesised feature; lanes j, j+1, ... read
// consecutive 48-byte state groups of those columns (coalesced).
// ---------------------------------------------------------------------------
struct HypCtx {
    double pose[7];
    double rot[9];     // rotcw = q2r(xi[3:7])^T, column-major
    double w0, w1;
    const double* c0;  // P*H^T column for pixel row 0
    const double* c1;
};

__device__ __forceinline__ void hyp_setup(const double* __restrict__ x, const double* __restrict__ W, int NP,
                                          const double* __restrict__ wv, int p, HypCtx& hc, const double* __restrict__ hctx)
{
    hc.c0 = W + (long)(2 * p) * NP;
    hc.c1 = hc.c0 + NP;
    hc.w0 = wv[2 * p]; hc.w1 = wv[2 * p + 1];
    // pose and rotation of the hypothesis: tabulated per matched feature by the launch that made its P H^T columns
    // (pht_kernel, hyp_context: the same expressions on the same values)
    const double* t = hctx + 16L * p;
#pragma unroll
    for (int a = 0; a < 7; ++a) hc.pose[a] = t[a];
#pragma unroll
    for (int a = 0; a < 9; ++a) hc.rot[a] = t[7 + a];
}

// squared residual |z_j - h_j(x_i)|^2 of matched feature j under the hypothesis hc (Tracking.cpp:425-476,480-503)
__device__ __forceinline__ double score_residual2(const Cam& cam, const double* __restrict__ x, const HypCtx& hc,
                                                  const ScoreTables& tab, const double* __restrict__ z, int j)
{
    const int o = tab.off[j];
#define XI(idx) (x[(idx)] + (hc.c0[(idx)] * hc.w0 + hc.c1[(idx)] * hc.w1))
    double v[3];
    const double r0 = XI(o), r1 = XI(o + 1), r2 = XI(o + 2);
    if (tab.type[j] == 0) {
        const int ith = tab.ith[j], iph = tab.iph[j];
        const double rho = XI(o + 5);
        // the angles are the prior's plus the hypothesis' correction: tabulated sin / cos and a short series in the
        // correction (two full sincos per pair were a third of this kernel); a large correction takes the long way
        const double dth = hc.c0[ith] * hc.w0 + hc.c1[ith] * hc.w1, dph = hc.c0[iph] * hc.w0 + hc.c1[iph] * hc.w1;
        double st, ct, sp, cp;
        if (fabs(dth) <= 0.125 && fabs(dph) <= 0.125) {
            sincos_delta(tab.sc[4 * j], tab.sc[4 * j + 1], dth, st, ct);
            sincos_delta(tab.sc[4 * j + 2], tab.sc[4 * j + 3], dph, sp, cp);
        } else {
            sincos(x[ith] + dth, &st, &ct);
            sincos(x[iph] + dph, &sp, &cp);
        }
        v[0] = (r0 - hc.pose[0]) * rho + cp * st;
        v[1] = (r1 - hc.pose[1]) * rho + (-sp);
        v[2] = (r2 - hc.pose[2]) * rho + cp * ct;
    } else {
        v[0] = r0 - hc.pose[0]; v[1] = r1 - hc.pose[1]; v[2] = r2 - hc.pose[2];
    }
#undef XI
    const double h0 = hc.rot[0] * v[0] + hc.rot[3] * v[1] + hc.rot[6] * v[2];
    const double h1 = hc.rot[1] * v[0] + hc.rot[4] * v[1] + hc.rot[7] * v[2];
    const double h2 = hc.rot[2] * v[0] + hc.rot[5] * v[1] + hc.rot[8] * v[2];
    // Tracking.cpp:471: ku on both axes.  One full-precision reciprocal instead of two divisions; the residual is compared
    // squared (sqrt is monotonic; the decisions carry a 1e-9 margin audit, see distort_fm_score)
    const double rh2 = rcp_nr2(h2);
    const double ui = cam.f_ku * (h0 * rh2) + cam.Cx;
    const double vi = cam.f_ku * (h1 * rh2) + cam.Cy;
    double ud, vd;
    distort_fm_score(cam, ui, vi, ud, vd);
    const int zf = tab.zsrc[j];
    const double n0 = z[2 * zf] - ud, n1 = z[2 * zf + 1] - vd;
    return n0 * n0 + n1 * n1;
}

__device__ __forceinline__ bool score_pair(const Cam& cam, const double* __restrict__ x, const HypCtx& hc,
                                           const ScoreTables& tab, const double* __restrict__ z, int j, double thr)
{
    return score_residual2(cam, x, hc, tab, z, j) < thr * thr;
}

// diagnostic (tests): the residuals the scoring kernel compares with the threshold, for every hypothesised position
__global__ void __launch_bounds__(256)
score_residual_kernel(Cam cam, const double* __restrict__ x, const double* __restrict__ W, int NP,
                      const double* __restrict__ wv, ScoreTables tab, const double* __restrict__ z, int m, double* __restrict__ out)
{
    const int p = blockIdx.x;
    HypCtx hc;
    hyp_setup(x, W, NP, wv, p, hc, tab.hctx);
    for (int j = threadIdx.x; j < m; j += blockDim.x) out[(long)p * m + j] = score_residual2(cam, x, hc, tab, z, j);
}

void launch_score_residuals(hipStream_t s, const Cam& cam, const double* x, const double* W, int NP, const double* wv,
                            const ScoreTables& tab, const double* z, int m, double* out)
{
    if (m <= 0) return;
    score_residual_kernel<<<dim3(m), dim3(256), 0, s>>>(cam, x, W, NP, wv, tab, z, m, out);
}

// diagnostic (tests): distort_fm_score (six steps, raw reciprocal slopes) beside distort_fm (the reference's ten divisions)
__global__ void distort_probe_kernel(Cam cam, int n, const double* __restrict__ uv, double* __restrict__ out_score, double* __restrict__ out_ref)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a, b;
    distort_fm_score(cam, uv[2 * i], uv[2 * i + 1], a, b);
    out_score[2 * i] = a; out_score[2 * i + 1] = b;
    distort_fm(cam, uv[2 * i], uv[2 * i + 1], a, b);
    out_ref[2 * i] = a; out_ref[2 * i + 1] = b;
}

void launch_distort_probe(hipStream_t s, const Cam& cam, int n, const double* uv, double* out_score, double* out_ref)
{
    if (n <= 0) return;
    distort_probe_kernel<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(cam, n, uv, out_score, out_ref);
}

template <bool ROT>
__global__ void __launch_bounds__(1024)
score_kernel(Cam cam, const double* __restrict__ x, const double* __restrict__ W, int NP,
             const double* __restrict__ wv, ScoreTables tab, const double* __restrict__ z, int m, int words,
             const int32_t* __restrict__ pos_list, double thr, int32_t* __restrict__ sup_out,
             uint64_t* __restrict__ masks_out)
{
    __shared__ int s_count;
    const int e = blockIdx.x;
    const int p = pos_list ? pos_list[e] : e;
    if (threadIdx.x == 0) s_count = 0;
    HypCtx hc;
    hyp_setup(x, W, NP, wv, p, hc, tab.hctx);
    __syncthreads();
    int local = 0;
    // Which wave takes which 64 features of a pass rotates with the workgroup: where the last pass is partial (300 features in
    // workgroups of four waves: the fifth chunk) the extra chunk falls on a different wave -- and with it on a different SIMD --
    // from one workgroup to the next.  (Round 6: as workgroups of FIVE waves the hardware put two waves of every workgroup on
    // the same SIMD of its compute unit: that SIMD carried 40 % of the instructions and capped the residency at three workgroups.)
    // (ROT: workgroups of exactly four waves)
    const int chunk = ROT ? (((int)(threadIdx.x >> 6) + (int)blockIdx.x) & 3) : (int)(threadIdx.x >> 6);
    for (int base = 0; base < m; base += blockDim.x) {
        const int j = base + 64 * chunk + (int)(threadIdx.x & 63);
        const bool inl = (j < m) && score_pair(cam, x, hc, tab, z, j, thr);
        const unsigned long long bal = __ballot(inl);
        if ((threadIdx.x & 63) == 0) {
            const int word = j >> 6;
            if (word < words) {
                if (masks_out) masks_out[(long)e * words + word] = bal;
                local += __popcll(bal);
            }
        }
    }
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(&s_count, local);
    __syncthreads();
    if (threadIdx.x == 0) sup_out[e] = s_count;
}

static inline int score_block_size(int m)
{
    int bs = ((m + 63) / 64) * 64;
    if (bs < 64) bs = 64;
    if (bs > 1024) bs = 1024;
    return bs;
}
// The scoring launch for chunk counts that score_spare_kernel (below) does not take: multiples of four, and more than eight.
// A workgroup of one pass per wave is the fastest form as long as the whole list is resident at once: there a wave with two
// passes would be the launch's critical path (+0.5 us measured at 1000 x 300).
// A launch of MORE waves than the device holds (> SCORE_RESIDENT_WAVES: 4000 hypotheses on one GPU, the x16 grid of bench.py,
// 1000 hypotheses x 16 waves at 1000 landmarks) with more than four waves' worth of features runs as workgroups of FOUR waves
// that take the features in passes of 256, the extra chunk of a partial last pass rotating over the waves (score_kernel<true>):
// one wave per SIMD and seven workgroups resident, where sixteen waves per workgroup leave room for one workgroup per compute
// unit.  1 000 x 1000 (C5): 34 us against 38.5; 4 000 x 1000: 83-92 against 110.  (Five chunks in this form: 16 000 x 300 in 92 us
// against 104 as five-wave workgroups -- and 81 with spare waves, which is what five chunks get: profiles/r06_score_variants.txt.)
constexpr long SCORE_RESIDENT_WAVES = 8192;
static bool score_one_pass_only()                           // (diagnostic variant of the library -- tests: the forms against each other)
{
#if defined(RSLAM_DEBUG)
    return getenv("RSLAM_SCORE_ONE_PASS") != nullptr;
#else
    return false;
#endif
}
static inline int score_launch_block_size(int m, int n_entries)
{
    const int bs = score_block_size(m);
    if (score_one_pass_only()) return bs;
    return (bs > 256 && (long)n_entries * (bs / 64) > SCORE_RESIDENT_WAVES) ? 256 : bs;
}

// One pass per wave WITHOUT the fifth wave on the first wave's SIMD: C = 4 a + b chunks of 64 features (b = 1..3) are launched as
// 4 (a + 1) waves; of the last four -- one per SIMD -- b take a chunk, WHICH ones rotates with the workgroup, the others leave
// at once (s_barrier waits for the surviving waves only: a wave that has ended does not count).  The hardware deals a workgroup's waves to the four SIMDs of
// its compute unit in turn, starting at the same SIMD every time: five waves put waves 0 and 4 of EVERY workgroup on it -- 2 of
// 5 passes, 1.6 times the average, and 3.25 of 7 possible waves resident per SIMD in a long launch (scripts/pmc_score.sh).
// C3's scoring launch 8.9 us against 10.2 as a kernel; 4 000 hypotheses 27.3 against 32.7; 16 000: 81 against 104 = 5.6-5.8 TB/s
// of the nominal 96 B per pair, 0.70-0.72 of the roof (round 5: 0.56); bit-identical (profiles/r06_score_variants.txt).
__global__ void __launch_bounds__(1024)
score_spare_kernel(Cam cam, const double* __restrict__ x, const double* __restrict__ W, int NP,
                   const double* __restrict__ wv, ScoreTables tab, const double* __restrict__ z, int m, int words,
                   const int32_t* __restrict__ pos_list, int C, double thr, int32_t* __restrict__ sup_out,
                   uint64_t* __restrict__ masks_out)
{
    __shared__ int s_count;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int a4 = C & ~3, b = C & 3;
    int chunk = wave;
    if (wave >= a4) {
        const int k = (wave - a4 - (int)blockIdx.x) & 3;
        if (k >= b) return;
        chunk = a4 + k;
    }
    const int e = blockIdx.x;
    const int p = pos_list ? pos_list[e] : e;
    // (the workgroup's count is kept by the wave that holds chunk 0 -- with fewer than four chunks wave 0 may be one that left)
    const bool leader = chunk == 0 && (threadIdx.x & 63) == 0;
    if (leader) s_count = 0;
    HypCtx hc;
    hyp_setup(x, W, NP, wv, p, hc, tab.hctx);
    __syncthreads();
    const int j = 64 * chunk + (int)(threadIdx.x & 63);
    const bool inl = (j < m) && score_pair(cam, x, hc, tab, z, j, thr);
    const unsigned long long bal = __ballot(inl);
    if ((threadIdx.x & 63) == 0 && chunk < words) {
        if (masks_out) masks_out[(long)e * words + chunk] = bal;
        const int cnt = __popcll(bal);
        if (cnt) atomicAdd(&s_count, cnt);
    }
    __syncthreads();
    if (leader) sup_out[e] = s_count;
}

void launch_score(hipStream_t s, const Cam& cam, const double* x, const double* W, int NP,
                  const double* wv, const ScoreTables& tab, const double* z, int m, int words,
                  const int32_t* pos_list, int n_entries, double threshold, int32_t* sup_out,
                  uint64_t* masks_out)
{
    if (n_entries <= 0 || m <= 0) return;
    const int C = score_block_size(m) / 64;              // chunks of 64 features
    if (m <= 64 * C && C >= 3 && C <= 8 && (C % 4) != 0 && !score_one_pass_only()) {
        score_spare_kernel<<<dim3(n_entries), dim3(64 * ((C & ~3) + 4)), 0, s>>>(cam, x, W, NP, wv, tab, z, m, words, pos_list, C, threshold,
                                                                            sup_out, masks_out);
        return;
    }
    const int bs = score_launch_block_size(m, n_entries);
    if (bs != score_block_size(m)) score_kernel<true><<<dim3(n_entries), dim3(bs), 0, s>>>(cam, x, W, NP, wv, tab, z, m, words, pos_list, threshold, sup_out, masks_out);
    else score_kernel<false><<<dim3(n_entries), dim3(bs), 0, s>>>(cam, x, W, NP, wv, tab, z, m, words, pos_list, threshold, sup_out, masks_out);
}

__global__ void map_support_kernel(const int32_t* __restrict__ possup, const int32_t* __restrict__ pos,
                                   int hb, int he, int32_t* __restrict__ sup)
{
    const int i = hb + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < he) sup[i] = possup[pos[i]];
}

void launch_map_support(hipStream_t s, const int32_t* possup, const int32_t* pos, int hb, int he, int32_t* sup)
{
    if (he <= hb) return;
    map_support_kernel<<<dim3((he - hb + 255) / 256), dim3(256), 0, s>>>(possup, pos, hb, he, sup);
}

// ---------------------------------------------------------------------------
// The consensus exchange of the hypothesis-sharded frame as ONE 8-byte MAX all-reduce (north_star's "all-reduce for
// the consensus inlier count"; rslam_shard_frame_allreduce, adaptive = 0 only): a rank's slice is folded into
//     key = support << 32 | (0xFFFFFFFF - hypothesis index)
// so that the maximum over all ranks is the LARGEST support and, among equal supports, the SMALLEST index -- the earliest
// strict maximum the sequential loop of Tracking.cpp:507-537 keeps (`support > max_hypothesis_support`).  Behind the
// all-reduce every rank writes the one-hot list {winner: its support, everybody else: 0} and replays the consensus on it as
// on a gathered list: same winner, same support, and with adaptive = 0 the same evaluated count; the winner's inlier mask is
// recomputed locally (one scoring workgroup: best_mask_body scores the winner again wherever this context did not score
// every hypothesis itself).  An empty slice contributes key 0, which loses to every scored hypothesis.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
shard_key_kernel(const int32_t* __restrict__ sup /* this rank's slice: entry 0 = hypothesis `begin` */, int begin, int n,
                 unsigned long long* __restrict__ key)
{
    __shared__ unsigned long long s_k[16];
    unsigned long long k = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const unsigned long long cand = ((unsigned long long)(unsigned)sup[i] << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)(begin + i));
        k = cand > k ? cand : k;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(k, d);
        k = o > k ? o : k;
    }
    if ((threadIdx.x & 63) == 0) s_k[threadIdx.x >> 6] = k;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) k = s_k[w] > k ? s_k[w] : k;
        *key = k;
    }
}

__global__ void shard_expand_kernel(const unsigned long long* __restrict__ key, int32_t* __restrict__ sup_all, int H)
{
    const unsigned long long k = *key;
    const int h = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull)), sv = (int)(k >> 32);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < H) sup_all[i] = (i == h) ? sv : 0;
}

void launch_shard_key(hipStream_t s, const int32_t* sup_local, int begin, int n, unsigned long long* key)
{
    shard_key_kernel<<<dim3(1), dim3(n > 256 ? 1024 : 256), 0, s>>>(sup_local, begin, n > 0 ? n : 0, key);
}

void launch_shard_expand(hipStream_t s, const unsigned long long* key, int32_t* sup_all, int H)
{
    if (H <= 0) return;
    shard_expand_kernel<<<dim3((H + 255) / 256), dim3(256), 0, s>>>(key, sup_all, H);
}

// ---------------------------------------------------------------------------
// K5: replay of the sequential scan of Tracking.cpp:403,507-537 from the
// complete support list.  Only strict prefix-maximum records can change the
// loop state, so the block finds the records in parallel (prefix-max scan) and
// one lane replays the loop over them.
// ---------------------------------------------------------------------------
constexpr int SEL_MAX_THREADS = 1024;
constexpr int SEL_MAX_RECORDS = 4096;

// block-wide scans on wave shuffles: two barriers each instead of two per doubling step
__device__ __forceinline__ int block_excl_max(int v, int* s_w)           // max over the threads before this one (values >= 0)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d, 64); if (lane >= d) inc = max(inc, o); }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before = max(before, s_w[w]);
    const int prev = __shfl_up(inc, 1, 64);
    __syncthreads();
    return lane > 0 ? max(before, prev) : before;
}
__device__ __forceinline__ int block_incl_sum(int v, int* s_w, int* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    int before = 0, all = 0;
    for (int w = 0; w < nw; ++w) { if (w < wave) before += s_w[w]; all += s_w[w]; }
    __syncthreads();
    *total = all;
    return before + inc;
}

// all threads of the block; blockDim.x <= SEL_MAX_THREADS and a multiple of 64; result in sel[0..2]
__device__ void select_consensus(const int32_t* __restrict__ sup, int H, const int32_t* __restrict__ nhyp_table,
                                 int adaptive, int n_hyp_init, int32_t* __restrict__ sel,
                                 int* s_max, int* s_cnt, int* s_rec, int* s_nh, int m /* the table has m + 1 entries */)
{
    // (a support is a count of matched features, 0 .. m; the list may come from the caller -- other ranks -- and indexes the table:
    //  a value outside that range must not become an address)
    auto nhyp_of = [&](int support) { return nhyp_table[support < 0 ? 0 : support > m ? m : support]; };
    const int t = threadIdx.x, nt = blockDim.x;
    const int chunk = (H + nt - 1) / nt;
    const int lo = min(H, t * chunk), hi = min(H, lo + chunk);
    // (the kernel is one workgroup's chain of dependent steps: the thread's first support -- its only one up to 1024
    //  hypotheses -- is loaded once, and the n_hyp table entries of all records at once, not one per step of the replay)
    const int v_first = lo < hi ? sup[lo] : 0;
    int mx = 0;
    for (int i = lo; i < hi; ++i) mx = max(mx, i == lo ? v_first : sup[i]);
    const int before = block_excl_max(mx, s_max);       // max of everything before the chunk (supports >= 0)
    int run = before;
    int cnt = 0;
    for (int i = lo; i < hi; ++i) { const int v = i == lo ? v_first : sup[i]; if (v > run) { run = v; ++cnt; } }
    int total = 0;
    int wpos = block_incl_sum(cnt, s_max, &total) - cnt;
    run = before;
    for (int i = lo; i < hi; ++i) {
        const int v = i == lo ? v_first : sup[i];
        if (v > run) { run = v; if (wpos < SEL_MAX_RECORDS) { s_rec[wpos] = i; if (wpos < SEL_MAX_THREADS) s_cnt[wpos] = v; } ++wpos; }
    }
    __syncthreads();
    if (adaptive && t < min(total, SEL_MAX_THREADS)) s_nh[t] = nhyp_of(s_cnt[t]);
    if (t == 0) s_max[0] = total;
    __syncthreads();
    if (t == 0) {
        const int nrec = min(s_max[0], SEL_MAX_RECORDS);
        int n_hyp = adaptive ? n_hyp_init : H;
        int best = 0, besti = -1, evaluated = 0, last = 0;
        bool done = false;
        for (int k = 0; k < nrec && !done; ++k) {
            const int i = s_rec[k];
            if (i >= n_hyp || i >= H) break;           // loop ended before reaching this record
            best = k < SEL_MAX_THREADS ? s_cnt[k] : sup[i]; besti = i; last = i + 1;   // (the record's support sits beside its index)
            if (adaptive) {
                n_hyp = k < SEL_MAX_THREADS ? s_nh[k] : nhyp_of(best);
                if (n_hyp == 0 || i > n_hyp) { evaluated = i + 1; done = true; }   // the two breaks, :533,:536
            }
        }
        // otherwise the for-condition i < n_hyp ends the loop; iteration `last-1` always ran
        if (!done) evaluated = max(last, min(n_hyp, H));
        if (evaluated < 0) evaluated = 0;
        sel[SEL_BEST_HYP] = besti;
        sel[SEL_BEST_SUPPORT] = best;
        sel[SEL_HYPS_EVALUATED] = evaluated;
        s_max[1] = besti;                               // for the caller, without a trip through memory
    }
    __syncthreads();
}

// ordered compaction helper: every thread contributes flag; returns its output
// slot (valid when flag) and advances *running (shared) by the block total.
__device__ __forceinline__ int block_compact(bool flag, int* s_wave, int* running)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned long long bal = __ballot(flag);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(bal);
    __syncthreads();
    int base = *running;
    for (int w = 0; w < wave; ++w) base += s_wave[w];
    int total = 0;
    for (int w = 0; w < nw; ++w) total += s_wave[w];
    __syncthreads();
    if (threadIdx.x == 0) *running += total;
    __syncthreads();
    return base + before;
}

// Winner's inlier set (Tracking.cpp:507-529) -> li[] flags, ordered feature list,
// k and the number of 64-column blocks of the stacked system.
// count / blocks / seq to the host (HostCounts): the sequence number last, behind a system-scope fence
__device__ __forceinline__ void host_counts_store(const HostCounts& hc, int count, int blocks)
{
    if (!hc.p) return;
    __hip_atomic_store(hc.p + 0, count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(hc.p + 1, blocks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(hc.p + 2, hc.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ void quat_jnorm(double* xq, int compat, double* T);      // (defined with the sweep's x update)

// ---------------------------------------------------------------------------
// The low-innovation update of rank <= 4 inside the consensus launch (LiSmallArgs).  In the reference-faithful mode the
// consensus set is the hypothesis' own feature (Q1): the update that follows (ExtendKF.cpp:559-634) is a 2 x 2 (at most 4 x 4)
// system, one row solve per state entry and the quaternion normalisation -- the covariance part is deferred anyway
// (SEL_LI_DEFER).  The workgroup that has just written the inlier list does it on the spot: S from the two (four) P H^T
// columns and the Jacobian rows, the factor in registers, Y1 = (P H^T) L^-T, x_k_k = x + Y1 u, Jnorm -- what the register
// route of the persistent sweep's strips does (sweep_strip, r_total <= 4) on ~115 workgroups.  This is ONE workgroup: the
// factor is kept as reciprocals (v_rsq_f64 + two Newton steps: an IEEE division or square root is ~40 instructions, four
// per row and a dozen per factor made the first version of this 11 us long) and a thread requests all of its rows at once.
// All threads of the workgroup; blockDim.x a multiple of 64; k = number of inliers (1 or 2).
// ---------------------------------------------------------------------------
__device__ __forceinline__ double li_rsqrt(double d)
{
    double y = __builtin_amdgcn_rsq(d);              // ~2^-26 relative
    const double h = 0.5 * d;
    y = y * (1.5 - h * y * y);                       // two Newton steps -> full double
    y = y * (1.5 - h * y * y);
    return y;
}

template <bool FOUR>          // two inliers: four columns
__device__ __forceinline__ void li_small_update_t(const LiSmallArgs& ls, int f0, int j0, int f1, int j1 /* the inliers and their matched-feature indices */)
{
    constexpr int r_total = FOUR ? 4 : 2;
    const int l = threadIdx.x & 63;
    const SysSrc& src = ls.src;
    // (column c of P H^T for inlier i is column 2 rank_of[f_i] + (c & 1) of the matched-feature matrix, and rank_of[feat[j]] = j)
    const double* w0 = src.Wsrc + (long)(2 * j0) * ls.NP;
    const double* w2 = src.Wsrc + (long)(2 * j1) * ls.NP;
    const double* wcol[4] = { w0, w0 + ls.NP, w2, w2 + ls.NP };
    constexpr bool four = FOUR;
    // the rows of this thread (first round) are requested before anything else: they need no more than the column pointers
    constexpr int LR = FOUR ? 2 : 4;                                 // rows per thread and round (128 registers per thread: the launch bound is 1024)
    double q0[LR], q1[LR], q2[LR], q3[LR], xi[LR];
    auto request = [&](int base) {
#pragma unroll
        for (int j = 0; j < LR; ++j) {
            const int i = base + j * (int)blockDim.x + (int)threadIdx.x;
            const bool ok = i < ls.NP;
            q0[j] = ok ? wcol[0][i] : 0.0; q1[j] = ok ? wcol[1][i] : 0.0;
            q2[j] = (ok && four) ? wcol[2][i] : 0.0; q3[j] = (ok && four) ? wcol[3][i] : 0.0;
            xi[j] = ok ? ls.x_in[i] : 0.0;
        }
    };
    request(0);
    // the innovation (requested beside the entries of S)
    double nu4[4];
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
        const int f = c4 < 2 ? f0 : f1;
        nu4[c4] = c4 < r_total ? src.z[2 * f + (c4 & 1)] - src.h[2 * f + (c4 & 1)] : 0.0;
    }
    // S(a, c), lower triangle authoritative: entry e = 4 a + c in lanes 0..15 of every wave (each wave factors for itself)
    double v;
    {
        const int e = l & 15, ea = e >> 2, ec = e & 3;
        const int a = ea >= ec ? ea : ec, c = ea >= ec ? ec : ea;
        v = (a == c) ? 1.0 : 0.0;
        if (a < r_total) {
            const int fa = a < 2 ? f0 : f1;
            const int fo = src.off[fa], fw = (src.type[fa] == 0) ? 13 : 10;
            const double* hf = src.H13 + 26L * fa + 13 * (a & 1);
            const double* wc = c == 0 ? wcol[0] : c == 1 ? wcol[1] : c == 2 ? wcol[2] : wcol[3];
            double sacc = 0;
#pragma unroll
            for (int q = 0; q < 13; ++q) if (q < fw) sacc += hf[q] * wc[col_index(fo, q)];
            v += sacc;
        }
    }
    const double s00 = __shfl(v, 0), s10 = __shfl(v, 4), s11 = __shfl(v, 5), s20 = __shfl(v, 8), s21 = __shfl(v, 9),
                 s22 = __shfl(v, 10), s30 = __shfl(v, 12), s31 = __shfl(v, 13), s32 = __shfl(v, 14), s33 = __shfl(v, 15);
    // L by columns, its diagonal as reciprocals i_jj = 1 / l_jj
    const double i00 = li_rsqrt(s00), l10 = s10 * i00, l20 = s20 * i00, l30 = s30 * i00;
    const double d1 = s11 - l10 * l10, i11 = li_rsqrt(d1), l21 = (s21 - l20 * l10) * i11, l31 = (s31 - l30 * l10) * i11;
    const double d2v = s22 - l20 * l20 - l21 * l21, i22 = li_rsqrt(d2v), l32 = (s32 - l30 * l20 - l31 * l21) * i22;
    const double d3 = s33 - l30 * l30 - l31 * l31 - l32 * l32, i33 = li_rsqrt(d3);
    if (threadIdx.x == 0 && !(s00 > 0.0 && d1 > 0.0 && d2v > 0.0 && d3 > 0.0 && i33 > 1.0e-300)) atomicMin(ls.status, -6);   // RSLAM_ERR_NOT_SPD
    // u^T = nu^T L^-T
    const double u0 = nu4[0] * i00, u1 = (nu4[1] - u0 * l10) * i11, u2 = (nu4[2] - u0 * l20 - u1 * l21) * i22,
                 u3 = (nu4[3] - u0 * l30 - u1 * l31 - u2 * l32) * i33;
    for (int base = 0; base < ls.NP; base += LR * (int)blockDim.x) {
        if (base) request(base);
#pragma unroll
        for (int j = 0; j < LR; ++j) {
            const int i = base + j * (int)blockDim.x + (int)threadIdx.x;
            const bool ok = i < ls.NP;                               // (NP is a multiple of 64: uniform per wave)
            const double x0 = q0[j] * i00, x1 = (q1[j] - x0 * l10) * i11, x2 = (q2[j] - x0 * l20 - x1 * l21) * i22,
                         x3 = (q3[j] - x0 * l30 - x1 * l31 - x2 * l32) * i33;
            const double xn = xi[j] + (((x0 * u0 + x1 * u1) + x2 * u2) + x3 * u3);
            if (ok) {
                ls.Y1[i] = x0; ls.Y1[i + ls.ldy1] = x1; ls.Y1[i + 2 * ls.ldy1] = x2; ls.Y1[i + 3 * ls.ldy1] = x3;
            }
            if (base == 0 && j == 0 && threadIdx.x < 64) {
                // rows 3..6: quaternion normalisation and Jnorm (ExtendKF.cpp:613-627; Q6)
                double q[4] = { __shfl(xn, 3), __shfl(xn, 4), __shfl(xn, 5), __shfl(xn, 6) };
                if (l < 3 || l > 6) ls.x_out[l] = xn;
                if (l == 0) {
                    quat_jnorm(q, ls.compat, ls.T);
                    for (int jj = 0; jj < 4; ++jj) ls.x_out[3 + jj] = q[jj];
                }
            } else if (ok) {
                ls.x_out[i] = xn;
            }
        }
    }
    if (threadIdx.x == 0) {
        *ls.xu_flag = 1;            // (token of the low-innovation update; its readers are later launches)
        *ls.defer_flag = 2;         // P_li = J (sym(P_pred) - Y1 Y1^T) J^T stays implicit (SEL_LI_DEFER; 2: ... and the update is done)
    }
}

__device__ void li_small_update(const LiSmallArgs& ls, int k, int f0, int j0, int f1, int j1)
{
    if (k > 1) li_small_update_t<true>(ls, f0, j0, f1, j1);
    else li_small_update_t<false>(ls, f0, j0, f1, j1);
}

__device__ void best_mask_body(const Cam& cam, const double* __restrict__ x, const double* __restrict__ W, int NP,
                               const double* __restrict__ wv, const ScoreTables& tab, const double* __restrict__ z, int m,
                               double thr, const SelectArgs& sa, int* s_wave, int* s_running, int* s_max, int* s_cnt, int* s_rec, int* s_nh,
                               const LiSmallArgs& ls, bool with_ls /* (no pointer to the by-value kernel argument: it would be copied to scratch) */)
{
    int32_t* __restrict__ sel = sa.sel;
    uint8_t* __restrict__ li = sa.li;
    int32_t* __restrict__ list = sa.list;
    // what the winner's scoring needs and does not depend on who wins: requested before the consensus
    const int j0 = threadIdx.x;
    int feat0 = -1;
    if (j0 < m) feat0 = tab.feat[j0];
    select_consensus(sa.sup, sa.H, sa.nhyp_table, sa.adaptive, sa.n_hyp_init, sel, s_max, s_cnt, s_rec, s_nh, m);   // K5
    if (threadIdx.x == 0) *s_running = 0;
    for (int i = threadIdx.x; i < sa.L; i += blockDim.x) li[i] = 0;
    __syncthreads();
    const int best = s_max[1];                          // (= sel[SEL_BEST_HYP], left there by select_consensus)
    if (best >= 0 && sa.masks) {
        // the scoring launch of this frame kept every hypothesis' inlier mask (same arithmetic, same inputs): one word per wave
        const uint64_t* mrow = sa.masks + (long)(sa.mask_by_pos ? sa.pos[best] : best) * sa.words;
        for (int base = 0; base < m; base += blockDim.x) {
            const int j = base + threadIdx.x;
            const int word = j >> 6;
            uint64_t bits = 0;
            if (word < sa.words) bits = mrow[word];
            const bool inl = (j < m) && ((bits >> (j & 63)) & 1ull);
            const int slot = block_compact(inl, s_wave, s_running);
            if (inl) {
                const int f = base == 0 ? feat0 : tab.feat[j]; li[f] = 1; list[slot] = f;
                if (slot < 2) { s_nh[2 * slot] = f; s_nh[2 * slot + 1] = j; }     // (li_small_update: the first two inliers and their columns of W)
            }
        }
    } else if (best >= 0) {
        HypCtx hc;
        hyp_setup(x, W, NP, wv, sa.pos[best], hc, tab.hctx);
        for (int base = 0; base < m; base += blockDim.x) {
            const int j = base + threadIdx.x;
            const bool inl = (j < m) && score_pair(cam, x, hc, tab, z, j, thr);
            const int slot = block_compact(inl, s_wave, s_running);
            if (inl) {
                li[tab.feat[j]] = 1; list[slot] = tab.feat[j];
                if (slot < 2) { s_nh[2 * slot] = tab.feat[j]; s_nh[2 * slot + 1] = j; }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nblk = (2 * *s_running + 63) / 64;
        sel[SEL_K_LI] = *s_running;
        sel[SEL_NBLK_LI] = nblk;
        sel[SEL_XU_FLAG] = 0;                        // Jnorm hand-over of this update stage's rank-update launches (tokens 1, 2)
        sel[SEL_LI_DEFER] = 0;                       // (an update stage that is re-run starts without a deferred covariance)
        sel[SEL_K_HI] = 0; sel[SEL_NBLK_HI] = 0;     // written by the second P H^T (GateList); no such launch without matched features
        host_counts_store(sa.host, *s_running, nblk);
    }
    // a low-innovation update of one or two inliers: here and now (the frame scalars above are out: *defer_flag is written behind them)
    const int k_li = *s_running;                     // (uniform)
    if (with_ls) {
        if (ls.clear_flags) for (int i = threadIdx.x; i < ls.n_clear; i += blockDim.x) ls.clear_flags[i] = 0;
        if (k_li >= 1 && k_li <= 2) {
            __syncthreads();
            li_small_update(ls, k_li, s_nh[0], s_nh[1], s_nh[k_li > 1 ? 2 : 0], s_nh[k_li > 1 ? 3 : 1]);
        } else if (ls.must && k_li == 0) {
            // no low-innovation inlier at all (the winner's own feature fell outside the threshold): update() is the identity
            // (ExtendKF.cpp:635-638).  In the deferred form: Y1 = 0, Jnorm = I -- every reader of P_li forms sym(P_pred), and
            // the prior is symmetric (exactly when uploaded, to rounding when rslam_ekf_prediction left it)
            for (int i = threadIdx.x; i < ls.NP; i += blockDim.x) {
                ls.x_out[i] = ls.x_in[i];
                ls.Y1[i] = 0.0; ls.Y1[i + ls.ldy1] = 0.0; ls.Y1[i + 2 * ls.ldy1] = 0.0; ls.Y1[i + 3 * ls.ldy1] = 0.0;
            }
            if (threadIdx.x < 16) ls.T[threadIdx.x] = ((threadIdx.x & 3) == (threadIdx.x >> 2)) ? 1.0 : 0.0;
            if (threadIdx.x == 0) { *ls.xu_flag = 1; *ls.defer_flag = 2; }
        } else if (ls.must && threadIdx.x == 0) {
            // the sequence was enqueued without the low-innovation sweep this frame needs: said out of band (SEL_LI_NEED), not as
            // a status code -- whatever the rest of this sequence reports about the posterior it works on is not to outrank it
            atomicOr(ls.status + (SEL_LI_NEED - SEL_STATUS), 1);
        }
    }
}

__global__ void __launch_bounds__(1024)
best_mask_kernel(Cam cam, const double* __restrict__ x, const double* __restrict__ W, int NP,
                 const double* __restrict__ wv, ScoreTables tab, const double* __restrict__ z, int m,
                 double thr, SelectArgs sa, LiSmallArgs ls, int with_li_small)
{
    __shared__ int s_wave[16];
    __shared__ int s_running;
    __shared__ int s_max[SEL_MAX_THREADS];
    __shared__ int s_cnt[SEL_MAX_THREADS];
    __shared__ int s_rec[SEL_MAX_RECORDS];
    __shared__ int s_nh[SEL_MAX_THREADS];
    best_mask_body(cam, x, W, NP, wv, tab, z, m, thr, sa, s_wave, &s_running, s_max, s_cnt, s_rec, s_nh, ls, with_li_small != 0);
}

void launch_best_mask(hipStream_t s, const Cam& cam, const double* x, const double* W, int NP,
                      const double* wv, const ScoreTables& tab, const double* z, int m,
                      const int32_t* pos, double threshold, int L, int32_t* sel, uint8_t* li,
                      int32_t* list, const int32_t* sup, int H, const int32_t* nhyp_table, int adaptive, int n_hyp_init,
                      const uint64_t* masks, int words, int mask_by_pos, HostCounts host, const LiSmallArgs* li_small)
{
    int bs = score_block_size(m);
    if (bs < 256) bs = 256;          // the consensus scan wants a few waves even for tiny maps
    const SelectArgs sa{pos, L, sel, li, list, sup, H, nhyp_table, adaptive, n_hyp_init, masks, words, mask_by_pos, host};
    LiSmallArgs ls{};
    if (li_small) {
        ls = *li_small; ls.src.list = list;
        if (bs < 512) bs = 512;      // (the update's rows: four per thread and round -- C3's 1856 in one)
    }
    best_mask_kernel<<<dim3(1), dim3(bs), 0, s>>>(cam, x, W, NP, wv, tab, z, m, threshold, sa, ls, li_small ? 1 : 0);
}

// ---------------------------------------------------------------------------
// K12: Tracking::rescue_hi_inliers gate (Tracking.cpp:584-595): candidates are
// individually compatible and not low-innovation inliers; nu' S^-1 nu < chi2.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
rescue_gate_kernel(int L, const uint8_t* __restrict__ ic, const uint8_t* __restrict__ li,
                   const uint8_t* __restrict__ has_h, const double* __restrict__ S,
                   const double* __restrict__ z, const double* __restrict__ h, double chi2,
                   uint8_t* __restrict__ hi, int32_t* __restrict__ list, int32_t* __restrict__ sel, HostCounts host)
{
    __shared__ int s_wave[16];
    __shared__ int s_running;
    if (threadIdx.x == 0) s_running = 0;
    __syncthreads();
    for (int base = 0; base < L; base += blockDim.x) {
        const int i = base + threadIdx.x;
        bool flag = false;
        if (i < L && ic[i] && !li[i] && has_h[i]) {
            double Si[4] = { S[4 * i], S[4 * i + 1], S[4 * i + 2], S[4 * i + 3] }, Sinv[4];
            inv2_lu(Si, Sinv);
            const double n0 = z[2 * i] - h[2 * i], n1 = z[2 * i + 1] - h[2 * i + 1];
            const double t0 = n0 * Sinv[0] + n1 * Sinv[1];
            const double t1 = n0 * Sinv[2] + n1 * Sinv[3];
            flag = (t0 * n0 + t1 * n1) < chi2;
        }
        if (i < L) hi[i] = flag ? 1 : 0;
        const int slot = block_compact(flag, s_wave, &s_running);
        if (flag) list[slot] = i;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nblk = (2 * s_running + 63) / 64;
        sel[SEL_K_HI] = s_running;
        sel[SEL_NBLK_HI] = nblk;
        host_counts_store(host, s_running, nblk);
    }
}

void launch_rescue_gate(hipStream_t s, int L, const uint8_t* ic, const uint8_t* li, const uint8_t* has_h,
                        const double* S, const double* z, const double* h, double chi2,
                        uint8_t* hi, int32_t* list, int32_t* sel, HostCounts host)
{
    int bs = ((L + 63) / 64) * 64;
    if (bs < 64) bs = 64;
    if (bs > 1024) bs = 1024;
    rescue_gate_kernel<<<dim3(1), dim3(bs), 0, s>>>(L, ic, li, has_h, S, z, h, chi2, hi, list, sel, host);
}

// ---------------------------------------------------------------------------
// K6: stacked system A = [S ; P H^T ; nu^T] of the flagged features
// (ExtendKF.cpp:565-594 stack z, h, H; :602 S = H P H^T + I).
// ---------------------------------------------------------------------------
// STAGED: the P H^T column of the workgroup goes through LDS (dynamic, NP doubles): the 13-term dots of its S entries gather
// from there instead of from global memory (C5: 97 -> 3x µs for the HI system; the dependent gathers were the kernel)
template <bool STAGED>
__global__ void __launch_bounds__(256)
prepare_system_kernel(SystemDims d, const int32_t* __restrict__ list, const int32_t* __restrict__ sel,
                      int slot_k, int slot_nblk, const double* __restrict__ H13, const int32_t* __restrict__ off,
                      const uint8_t* __restrict__ type,
                      const double* __restrict__ z, const double* __restrict__ h, double* __restrict__ A,
                      const double* __restrict__ Wsrc /* nullable */, const int32_t* __restrict__ rank_of,
                      int32_t* __restrict__ sweep_flags /* nullable */)
{
    extern __shared__ __attribute__((aligned(16))) double s_col[];
    const int c = blockIdx.x;
    if (sweep_flags && c == 0) for (int i = threadIdx.x; i < SWEEP_FLAG_INTS; i += blockDim.x) sweep_flags[i] = 0;   // hand-over flags of the persistent sweep
    if (c >= 64 * sel[slot_nblk]) return;
    const int r = 2 * sel[slot_k];
    double* col = A + (long)c * d.ldA;
    double* wcol = col + d.RP;                  // P*H^T column c
    if (Wsrc && c < r) {                        // low-innovation pass: gather it from the matched-feature P*H^T
        const double* src = Wsrc + (long)(2 * rank_of[list[c >> 1]] + (c & 1)) * d.NP;
        for (int a = threadIdx.x; a < d.NP; a += 256) { const double v = src[a]; wcol[a] = v; if (STAGED) s_col[a] = v; }
        __syncthreads();                        // the block re-reads its own column below
    } else if (STAGED && c < r) {
        for (int a = threadIdx.x; a < d.NP; a += 256) s_col[a] = wcol[a];
        __syncthreads();
    }
    const double* gsrc = STAGED ? s_col : wcol;
    for (int a = threadIdx.x; a < d.RP; a += 256) {
        double v = (a == c) ? 1.0 : 0.0;        // + R = I (ExtendKF.cpp:594); identity on the padding
        if (c < r && a < r) {
            const int fa = list[a >> 1];
            const double* Hf = H13 + 26 * (long)fa + 13 * (a & 1);
            const int o = off[fa];
            const int w = (type[fa] == 0) ? 13 : 10;
            double sacc = 0;
            for (int k = 0; k < w; ++k) sacc += Hf[k] * gsrc[col_index(o, k)];
            v += sacc;
        }
        col[a] = v;
    }
    if (c >= r) for (int a = threadIdx.x; a < d.NP; a += 256) col[d.RP + a] = 0.0;
    if (threadIdx.x < 64) {
        double v = 0.0;
        if (threadIdx.x == 0 && c < r) {
            const int f = list[c >> 1];
            v = z[2 * f + (c & 1)] - h[2 * f + (c & 1)];
        }
        col[d.RP + d.NP + threadIdx.x] = v;
    }
}

void launch_prepare_system(hipStream_t s, const SystemDims& d, const int32_t* list, const int32_t* sel,
                           int slot_k, int slot_nblk, const double* H13, const int32_t* off,
                           const uint8_t* type, const double* z, const double* h, double* A,
                           const double* Wsrc, const int32_t* rank_of, int32_t* sweep_flags)
{
    if (d.RP <= 0) return;
    const size_t bytes = sizeof(double) * (size_t)d.NP;
    if (bytes <= 64 * 1024)
        prepare_system_kernel<true><<<dim3(d.RP), dim3(256), bytes, s>>>(d, list, sel, slot_k, slot_nblk, H13, off, type, z, h, A, Wsrc, rank_of,
                                                                      sweep_flags);
    else
        prepare_system_kernel<false><<<dim3(d.RP), dim3(256), 0, s>>>(d, list, sel, slot_k, slot_nblk, H13, off, type, z, h, A, Wsrc, rank_of,
                                                                    sweep_flags);
}

// ---------------------------------------------------------------------------
// K8: blocked right-looking Cholesky sweep over the stacked matrix.  After the
// sweep rows [0,RP) hold L (S = L L^T), rows [RP,RP+NP) hold Y = P H^T L^-T and
// row RP+NP holds u^T = nu^T L^-T, so that K S K^T = Y Y^T and K nu = Y u
// (ExtendKF.cpp:603,606,608 use an explicit PartialPivLU inverse of S instead).
// ---------------------------------------------------------------------------
// Diagonal-block factorisation: one workgroup factors the 64 x 64 block in
// registers.  Thread (i = t & 63, g = t >> 6) owns row i, columns 16g..16g+15 of
// ONE array that holds A(i,c) until column c has been eliminated and (L^-1)(i,c)
// afterwards (the row operations that build L are applied to the identity at the
// same time, so L^-1 is ready when the last pivot is done and the panel solve
// becomes an MFMA product).  Per pivot the only shared data is one 64-entry
// vector X (column j of the trailing matrix below the diagonal, row j of the
// partial inverse left of it), double-buffered in LDS: one barrier per pivot.
__device__ __forceinline__ double rsqrt_f64(double d)
{
    double y = __builtin_amdgcn_rsq(d);              // ~2^-26 relative
    const double h = 0.5 * d;
    y = y * (1.5 - h * y * y);                       // two Newton steps -> full double
    y = y * (1.5 - h * y * y);
    return y;
}

// ---- 64 x 64 diagonal block: Cholesky factor and its inverse ---------------------------
// The serial pivot chain of these blocks bounds the whole sweep.  It is cut four-fold (one
// iteration eliminates a 4 x 4 pivot block) and software-pipelined across the ten waves of one
// workgroup so that only what the NEXT pivot block needs sits on the chain.  Measured on
// MI355X: a wave64 instruction costs ~4-5 cycles of its SIMD whatever it is (FP64 FMA, select,
// address arithmetic; dependent or not), an LDS round trip ~100, v_rsq_f64 ~20 -- so the chain
// is bound by the NUMBER of instructions the panel wave executes per pivot block, and everything
// that is not strictly needed for the next pivot lives in other waves:
//   wave 0 (panel wave, lane = row)   reads the 4 columns of pivot block s ("strip", one update
//          behind), applies the rank-4 update of block s-1 to them itself and runs a
//          lane-parallel right-looking Cholesky over the 4 columns: the pivot-row lanes end up
//          holding L4, the other lanes their row of the panel X = T(:,p) L4^-T; L4 entries and
//          pivots travel by readlane.  ~110 instructions per block.
//   wave 1 (inverse wave, lane = column) follows one block behind: rows p of the running
//          inverse, Ms = L4^-1 M(p,:), with L4 read from the panel and the pivot reciprocals
//          published by wave 0;
//   waves 2..6 (T waves) hold the trailing matrix in MFMA accumulator tiles (16 x 16, the ten
//          lower tiles, two per wave); in iteration s they apply T -= X(s-1) X(s-1)^T (ONE
//          v_mfma_f64_16x16x4_f64 per tile) and publish the strip of block s+1;
//   waves 7..9 (M waves) hold the running inverse the same way and apply M -= X(s-2) Ms(s-2).
// Rows at and above the current pivot carry don't-care values in X (they only reach tile
// entries that are never read again), so nothing is masked.  No barrier inside the chain:
// panels and strips are multi-buffered in LDS and handed over through LDS flags (producer:
// data, release fence, flag; consumer: poll, acquire fence).  L and L^-1 are collected in LDS
// and written once at the end (a global store inside the loop would put a vmcnt(0) wait on
// the chain).
constexpr int CD_LD = 65;
constexpr int CD_TW = 4;            // T waves
constexpr int CD_MW = 2;            // M waves
constexpr int CD_THREADS = 64 * (2 + CD_TW + CD_MW);
constexpr int CD_SPIN_LIMIT = 1 << 13;       // ~1 ms of LDS polls; the waves of the pipeline are never more than a pivot step (~0.6 us) apart
constexpr int CD_OPLD = 72;         // leading dimension of the operands staged for the in-kernel panel row

struct CdShared {
    double Lf[64 * CD_LD];       // L, column-major
    double Mf[64 * CD_LD];       // L^-1, column-major
    double Xs[4][4 * 64];        // [k][row]: panel of block s (buffer s & 3)
    double Ms[2][4 * 64];        // [k][col]: L4^-1 M(p,:) of block s (buffer s & 1)
    double Tst[2][4 * 64];       // [k][row]: columns of pivot block s (buffer s & 1), updates <= s-2 applied
    double Mst[2][4 * 64];       // [k][col]: rows of pivot block s of the running inverse (buffer s & 1), updates <= s-2
    double Rs[2][4];             // reciprocal square roots of the pivots of block s (buffer s & 1)
    int flags[16];               // 0: blocks finished by the panel wave; 1: by the inverse wave; 2..6: iterations of T wave; 7..9: of M wave
    int timeout;
};

#if defined(CD_STAMPS) || defined(CD_SPINS)
__device__ unsigned long long g_cd_stamps[16];
#endif
// CD_SPINS: non-intrusive view of who waits for whom in the pivot pipeline -- polls that found their flag not yet up, counted in
// registers and added to g_cd_stamps once per diagonal block: [0] extra looks of the panel wave, [1] its steps, [2] polls of T
// wave 0's wait for the panel, [3] its steps, [4] / [5] M wave 0, [6] / [7] inverse wave
#if defined(CD_SPINS)
#define CD_SPIN_ADD(slot, v) atomicAdd(&g_cd_stamps[slot], (unsigned long long)(v))
#else
#define CD_SPIN_ADD(slot, v)
#endif
// CD_TIMELINE: shader-clock time line of the pivot pipeline, [who][block & 7][step][slot]: who 0 = panel wave (slots: first
// look issued, flags up + data landed, panel posted), who 1 = T wave 0 (step begins, panel flag seen, operands landed, strip
// posted); written with fire-and-forget global stores by lane 0; each stamp drains the wave's LDS queue (~60 cycles).
#if defined(CD_TIMELINE)
__device__ unsigned long long g_cd_log[2][8][16][4];
#define CD_TL(who, blk, step, slot) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                                         if ((threadIdx.x & 63) == 0) g_cd_log[who][(blk) & 7][step][slot] = t_; } while (0)
#else
#define CD_TL(who, blk, step, slot)
#endif
#if !defined(CD_ASK_ADAPTIVE)     // (asking ahead only when the step did not have to wait: measured 1 % slower than always asking)
#define CD_ASK_IF(cond)
#else
#define CD_ASK_IF(cond) && (cond)
#endif
#if defined(CD_TL_M)
#define CD_TL_M_ON 1
#else
#define CD_TL_M_ON 0
#endif
#if defined(CD_STAMPS)   // diagnostic build: where do the cycles of a pivot step go (never in the product build)
#define CD_STAMP(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#define CD_ACC_T(slot, a, b, tid) do { if (threadIdx.x == (tid)) g_cd_stamps[slot] += (b) - (a); } while (0)
#define CD_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#else
#define CD_STAMP(var)
#define CD_ACC_T(slot, a, b, tid)
#define CD_PIN4(a, b, c, d)
#endif

// wait until panel flag >= need_panel, inverse flag >= need_inv, every T-wave flag >= need_t and
// every M-wave flag >= need_m; wave-uniform, bounded so that a logic error ends in an error
// code instead of a hung queue
template <int SLEEP>
__device__ __forceinline__ int cd_wait(CdShared& sh, int need_panel, int need_inv, int need_t, int need_m)
{
    const int li = threadIdx.x & 15;
    const int need = li == 0 ? need_panel : li == 1 ? need_inv : li < 2 + CD_TW ? need_t : li < 2 + CD_TW + CD_MW ? need_m : -(1 << 30);
    int spins = 0;
    while (true) {
        // (an atomic load keeps the LDS address space; a volatile access through the reference turns into FLAT)
        const int v = __hip_atomic_load(&sh.flags[li], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (__all(v >= need)) break;
        if (++spins > CD_SPIN_LIMIT) { sh.timeout = 1; break; }
        if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
    }
    // (no hardware fence: see cd_post -- the hand-over is LDS only, served in order; the poll's value has been waited for)
#if defined(CD_HW_FENCES)     // measurement build: the workgroup-scope fences that stood here
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#else
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
#endif
    return spins;
}

// Publish: what this wave wrote to LDS, then the flag.  Everything handed over between the waves of the pipeline lives in
// LDS, and LDS executes a wave's operations in the order they were issued: the flag store cannot overtake the data stores in
// front of it, and a consumer that has SEEN the flag issues its data reads behind that.  So neither side needs a hardware
// fence -- a compiler barrier keeps the program order.  The workgroup-scope release / acquire fences that stood here lower to
// s_waitcnt vmcnt(0) lgkmcnt(0): a wait for the LDS round trip on the publishing side (on the panel wave: on the chain) and,
// worse, for every outstanding GLOBAL access of the wave -- the T waves of the persistent sweep keep hand-over polls and
// operand loads in flight across pivot steps, and each such wait stalled their step, and with it the chain, for a memory latency.
// INVARIANT of the fence-free form: everything that cd_post publishes has been written with plain ds_write by the posting
// wave itself.  An operand staged by LDS-DMA (global_load ... lds, tracked by vmcnt, not ordered with ds_write) in front of a
// cd_post would race silently -- no such transfer feeds this pipeline (the tile engine's DMA buffers are never handed over
// through cd_post).  The in-order argument is a property of the gfx950 LDS path: other targets take the fenced form.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(CD_HW_FENCES)
#error "cd_post / cd_wait without hardware fences are validated on gfx950 only: build other targets with -DCD_HW_FENCES"
#endif
__device__ __forceinline__ void cd_post(CdShared& sh, int flag, int value)
{
#if defined(CD_HW_FENCES)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#else
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
#endif
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(&sh.flags[flag], value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(d) to ~2^-48: v_rsq_f64 (2^-24) and ONE Newton step.  The second step of rsqrt_f64 would
// cost three more instructions per pivot on the chain; the factor is accurate to ~4e-15 instead
// of 1e-16, far inside the parity tolerance of the state and covariance.
__device__ __forceinline__ double rsqrt_chain(double d)
{
    const double y = __builtin_amdgcn_rsq(d);
    const double e = fma(-d * y, y, 1.0);                // 1 - d y^2
    return fma(0.5 * y, e, y);                           // y (1 + e/2)
}

// lower-triangle 16 x 16 tiles in row-major order
__device__ constexpr int cd_tr(int idx) { return idx < 1 ? 0 : idx < 3 ? 1 : idx < 6 ? 2 : 3; }
__device__ constexpr int cd_tc(int idx) { return idx - (idx < 1 ? 0 : idx < 3 ? 1 : idx < 6 ? 3 : 6); }

// Panel wave: the pivot chain.  KEEP_L = false: L itself is not collected (persistent sweep: nobody reads it).
template <bool KEEP_L = true>
__device__ __forceinline__ void cd_panel_wave(CdShared& sh, int n_piv4, int blk = 0)
{
    const int l = threadIdx.x & 63;
    __builtin_amdgcn_s_setprio(3);                          // the chain wins every issue arbitration on its SIMD
    double xp0 = 0.0, xp1 = 0.0, xp2 = 0.0, xp3 = 0.0;     // own row of the previous panel
    int dbg_looks = 0, dbg_steps = 0;
    struct SpinOut { int& a; int& b; __device__ ~SpinOut() { if ((threadIdx.x & 63) == 0) { CD_SPIN_ADD(0, a); CD_SPIN_ADD(1, b); } } } spin_out{dbg_looks, dbg_steps};
    // flag li must reach s + need_off: T waves s, inverse wave and M waves s - 1, nothing else
    // (Round 4 also built the pipeline TWO updates deep -- the T waves publish the columns of block j+2 in their step j, this
    //  wave applies the last two rank-4 updates itself: 32 FMAs, sixteen more broadcast reads.  Correct, and no faster: the panel
    //  wave stops waiting (0.8 extra looks per step instead of 3.3) but its own step grows, and with ~90 KB of LDS traffic per
    //  step -- half of it broadcast reads that return 1 KB for 16 bytes of data -- the LDS pipe of the compute unit is what
    //  every wave's round trips queue behind.  HI launch 140-145 us against 142.)
    const int need_off = ((threadIdx.x & 15) >= 2 && (threadIdx.x & 15) < 2 + CD_TW) ? 0
                       : ((threadIdx.x & 15) == 1 || ((threadIdx.x & 15) >= 2 + CD_TW && (threadIdx.x & 15) < 2 + CD_TW + CD_MW)) ? -1 : -(1 << 30);
#pragma unroll 1
    for (int sb = 0; sb < 4; ++sb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s = 4 * sb + q;
            const int p0 = 16 * sb + 4 * q;
            if (s >= n_piv4) return;                      // uniform: only identity padding is left
            CD_STAMP(s0);
            // Needs: strip of block s (T waves, iteration s-1); panel buffer s & 3 free: inverse wave past block
            // s-3, M waves past iteration s-2; Rs buffer free: inverse wave past block s-2.
            // The flags and the data are read in ONE batch: LDS serves a wave's requests in order, so data read
            // after a flag that says "ready" is the published data; only if a flag was not ready yet (it almost
            // never is not: the other waves have a whole iteration of slack) everything is read again.
            const double* Tc = sh.Tst[q & 1];
            const double* Xp = sh.Xs[(q + 3) & 3];        // panel s-1 (zeros for s = 0)
            double a0, a1, a2, a3, u[16];
            {
                const int li = threadIdx.x & 15;
                const int need = s + need_off;
                // (these sixteen u values are what lanes p0..p0+3 of THIS wave hold in xp0..xp3; taking them by readlane --
                //  32 v_readlane_b32 under the latency of the other reads -- measured 1 us per block SLOWER than the broadcast
                //  LDS reads in round 2 and again, with the rebuilt pipeline, in round 4: 138.4 / 141.4 against 138.3 / 136.3 us)
                int v;
#define CD_PANEL_LOADS() do {                                                                                              \
                    v = __hip_atomic_load(&sh.flags[li], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);                  \
                    __atomic_signal_fence(__ATOMIC_SEQ_CST);    /* compiler: keep the data reads behind the flag read */     \
                    a0 = Tc[l]; a1 = Tc[64 + l]; a2 = Tc[128 + l]; a3 = Tc[192 + l];                                       \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                            \
                        _Pragma("unroll") for (int k = 0; k < 4; ++k) u[4 * j + k] = Xp[j * 64 + p0 + k];                    \
                    __atomic_signal_fence(__ATOMIC_SEQ_CST);                                                               \
                } while (0)
                CD_TL(0, blk, s, 0);
#if !defined(CD_NO_PEEL_POLL)
                // The first look is straight-line code: a loop header would carry the s_waitcnt lgkmcnt(0) its back edge needs
                // (the re-issued reads write the same registers), i.e. a wait for this wave's own panel stores of the step
                // before, on the chain, in every step.  Only a flag that is not up yet enters the polling loop.
                CD_PANEL_LOADS();
                ++dbg_steps;
                if (!__all(v >= need)) {
#if defined(CD_POLL_ALL)
                    int spins = 0;
                    while (true) {
                        CD_PANEL_LOADS();
                        ++dbg_looks;
                        if (__all(v >= need)) break;
                        if (++spins > CD_SPIN_LIMIT) { sh.timeout = 1; break; }
                    }
#else
                    // not up yet: poll the flags ALONE (one 4-byte read per look), then read the data once more.  Every look
                    // that re-reads the whole batch moves 10 KB through the LDS pipe, and the counters of a -DCD_SPINS build
                    // say the first look fails in most steps, three to four times in a row: 30-40 KB per step of traffic
                    // that delays the LDS reads of the very waves the chain is waiting for.
                    int spins = 0;
                    while (true) {
                        v = __hip_atomic_load(&sh.flags[li], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        ++dbg_looks;
                        if (__all(v >= need)) break;
                        if (++spins > CD_SPIN_LIMIT) { sh.timeout = 1; break; }
                    }
                    __atomic_signal_fence(__ATOMIC_SEQ_CST);
                    CD_PANEL_LOADS();
#endif
                }
#else
                int spins = 0;
                while (true) {
                    CD_PANEL_LOADS();
                    if (__all(v >= need)) break;
                    if (++spins > CD_SPIN_LIMIT) { sh.timeout = 1; break; }
                }
#endif
#undef CD_PANEL_LOADS
            }
            CD_STAMP(s1);
            CD_TL(0, blk, s, 1);
            __builtin_amdgcn_sched_barrier(0);
            // strip row l: T(l, p0+k) -= sum_j X_prev(l, j) X_prev(p0+k, j)
            a0 = fma(-xp0, u[0], a0); a1 = fma(-xp0, u[1], a1); a2 = fma(-xp0, u[2], a2); a3 = fma(-xp0, u[3], a3);
            a0 = fma(-xp1, u[4], a0); a1 = fma(-xp1, u[5], a1); a2 = fma(-xp1, u[6], a2); a3 = fma(-xp1, u[7], a3);
            a0 = fma(-xp2, u[8], a0); a1 = fma(-xp2, u[9], a1); a2 = fma(-xp2, u[10], a2); a3 = fma(-xp2, u[11], a3);
            a0 = fma(-xp3, u[12], a0); a1 = fma(-xp3, u[13], a1); a2 = fma(-xp3, u[14], a2); a3 = fma(-xp3, u[15], a3);
            CD_PIN4(a0, a1, a2, a3);
            CD_STAMP(sa);
            // right-looking Cholesky over the 4 columns, lane = row
            const double r0 = rsqrt_chain(readlane_f64(a0, p0));
            xp0 = a0 * r0;
            a1 = fma(-xp0, readlane_f64(xp0, p0 + 1), a1);
            const double r1 = rsqrt_chain(readlane_f64(a1, p0 + 1));
            xp1 = a1 * r1;
            a2 = fma(-xp1, readlane_f64(xp1, p0 + 2), fma(-xp0, readlane_f64(xp0, p0 + 2), a2));
            const double r2 = rsqrt_chain(readlane_f64(a2, p0 + 2));
            xp2 = a2 * r2;
            a3 = fma(-xp2, readlane_f64(xp2, p0 + 3), fma(-xp1, readlane_f64(xp1, p0 + 3), fma(-xp0, readlane_f64(xp0, p0 + 3), a3)));
            const double r3 = rsqrt_chain(readlane_f64(a3, p0 + 3));
            xp3 = a3 * r3;
            CD_PIN4(xp0, xp1, xp2, xp3);
            CD_STAMP(sc);
            // rows of the pivot block hold L4 itself (x_k is then l_{row,k}); what lands above the
            // diagonal of L is never read
            double* Xc = sh.Xs[q];
            Xc[l] = xp0; Xc[64 + l] = xp1; Xc[128 + l] = xp2; Xc[192 + l] = xp3;
            if (KEEP_L) {
                sh.Lf[(p0 + 0) * CD_LD + l] = xp0; sh.Lf[(p0 + 1) * CD_LD + l] = xp1;
                sh.Lf[(p0 + 2) * CD_LD + l] = xp2; sh.Lf[(p0 + 3) * CD_LD + l] = xp3;
            }
            if (l == 0) { double* R = sh.Rs[q & 1]; R[0] = r0; R[1] = r1; R[2] = r2; R[3] = r3; }
            CD_STAMP(sd);
            cd_post(sh, 0, s + 1);
            CD_TL(0, blk, s, 2);
            CD_STAMP(s2);
            CD_ACC_T(0, s0, s1, 0); CD_ACC_T(1, s1, sa, 0); CD_ACC_T(5, sa, sc, 0); CD_ACC_T(7, sc, sd, 0); CD_ACC_T(2, sd, s2, 0);
            CD_ACC_T(6, s0, s2, 0);
        }
    }
}

// 16-byte write-through store (agent scope: served by the memory side, see "Memory protocol" of the persistent sweep)
__device__ __forceinline__ void st_coh2(double* p, double a, double b)
{
    typedef double d2v __attribute__((ext_vector_type(2)));
    const d2v v = {a, b};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}

// What the inverse wave of the persistent sweep does with its rows of L^-1 besides collecting them in LDS (CdInvOut::Lglob
// != nullptr): it writes them to global memory AS THEY BECOME FINAL, one pivot step behind the chain (write-through stores,
// never waited for inside the loop), and posts the flag the strips wait for itself, right after its last step -- the
// strips learn of L^-1(k) ~0.3 us after the chain of block k has ended instead of ~2.5 us (stores at the top of the next
// block, X product, barrier, flag), so their hand-over of the next row block reaches the chain workgroup early in its
// chain instead of late, and the sweep's last solve starts that much earlier behind the last block.
struct CdInvOut {
    double* Lglob;            // L^-1(k): [row + 64 col], lower triangle, identity beyond the rows that exist
    int32_t* flag; int value; // *flag = value once every row has reached memory
};

// Inverse wave: pivot rows of the running inverse, one block behind the panel wave.  Returns (CHECK only) whether a
// pivot was not a positive finite number: where L is not collected this wave, which is off the chain, looks at the
// reciprocal square roots the panel wave publishes.
template <bool CHECK = false>
__device__ __forceinline__ bool cd_inverse_wave(CdShared& sh, int n_piv4, const CdInvOut io = CdInvOut{nullptr, nullptr, 0})
{
    const int l = threadIdx.x & 63;
    bool bad = false;
    double mp0 = 0.0, mp1 = 0.0, mp2 = 0.0, mp3 = 0.0;     // own column of Ms of the previous block
    if (io.Lglob && n_piv4 < 16) {
        // a last block of fewer than 64 rows: identity beyond the pivots that exist (column l: rows 4 n_piv4 .. 63)
        for (int row = 4 * n_piv4; row < 64; row += 2) st_coh2(io.Lglob + row + 64 * l, row == l ? 1.0 : 0.0, row + 1 == l ? 1.0 : 0.0);
    }
    auto finish = [&]() {
        if (io.Lglob) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (l == 0) __hip_atomic_store(io.flag, io.value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
#pragma unroll 1
    for (int sb = 0; sb < 4; ++sb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s = 4 * sb + q;                     // block
            const int p0 = 16 * sb + 4 * q;
            if (s >= n_piv4) { finish(); return bad; }
            { const int sp = cd_wait<1>(sh, s + 1, 0, 0, s + 1);           // panel s (L4, reciprocals); strip of block s (M waves, iteration s)
              if (l == 0) { CD_SPIN_ADD(6, sp); CD_SPIN_ADD(7, 1); } (void)sp; }
            const double* Mc = sh.Mst[q & 1];
            const double* Xp = sh.Xs[(q + 3) & 3];        // panel s-1
            const double* Xc = sh.Xs[q];                  // panel s: rows p0..p0+3 are L4
            const double* R = sh.Rs[q & 1];
            double a0 = Mc[l], a1 = Mc[64 + l], a2 = Mc[128 + l], a3 = Mc[192 + l];
            double u[16];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) u[4 * j + k] = Xp[j * 64 + p0 + k];
            const double l10 = Xc[p0 + 1], l20 = Xc[p0 + 2], l30 = Xc[p0 + 3];
            const double l21 = Xc[64 + p0 + 2], l31 = Xc[64 + p0 + 3], l32 = Xc[128 + p0 + 3];
            const double q0 = R[0], q1 = R[1], q2 = R[2], q3 = R[3];
            __builtin_amdgcn_sched_barrier(0);           // every LDS read of the block is in flight before the arithmetic starts
            if (CHECK) bad = bad || !(q0 > 1e-300 && q0 < 1e300 && q1 > 1e-300 && q1 < 1e300 && q2 > 1e-300 && q2 < 1e300 && q3 > 1e-300 && q3 < 1e300);
            {   // strip column l: M(p0+k, l) -= sum_j X_prev(p0+k, j) Ms_prev(j, l)
                const double ms[4] = {mp0, mp1, mp2, mp3};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a0 = fma(-u[4 * j + 0], ms[j], a0); a1 = fma(-u[4 * j + 1], ms[j], a1);
                    a2 = fma(-u[4 * j + 2], ms[j], a2); a3 = fma(-u[4 * j + 3], ms[j], a3);
                }
            }
            // entries right of the pivot columns come out as exact zeros (M is lower triangular)
            mp0 = a0 * q0;
            mp1 = fma(-l10, mp0, a1) * q1;
            mp2 = fma(-l21, mp1, fma(-l20, mp0, a2)) * q2;
            mp3 = fma(-l32, mp2, fma(-l31, mp1, fma(-l30, mp0, a3))) * q3;
            double* Mo = sh.Ms[q & 1];
            Mo[l] = mp0; Mo[64 + l] = mp1; Mo[128 + l] = mp2; Mo[192 + l] = mp3;
            // column l of rows p0..p0+3 of L^-1 (final)
            sh.Mf[l * CD_LD + p0 + 0] = mp0; sh.Mf[l * CD_LD + p0 + 1] = mp1;
            sh.Mf[l * CD_LD + p0 + 2] = mp2; sh.Mf[l * CD_LD + p0 + 3] = mp3;
            cd_post(sh, 1, s + 1);
            // (entries right of the diagonal are exact zeros: the running inverse is lower triangular throughout)
            if (io.Lglob) { st_coh2(io.Lglob + p0 + 64 * l, mp0, mp1); st_coh2(io.Lglob + p0 + 2 + 64 * l, mp2, mp3); }
        }
    }
    finish();
    return bad;
}

// T wave B: tiles B, B + CD_TW, ... (< 10) of the trailing matrix.
constexpr int CD_TT = (10 + CD_TW - 1) / CD_TW;     // most tiles a T wave owns
template <int B>
__device__ __forceinline__ void cd_t_wave(CdShared& sh, int n_piv4, const d4 (&acc_in)[CD_TT], int pending)
{
    const int l = threadIdx.x & 63, lr = l >> 4, lc = l & 15;
    constexpr int NT = (10 - B + CD_TW - 1) / CD_TW;
    d4 acc[NT];                                       // loaded by the caller
#pragma unroll
    for (int o = 0; o < NT; ++o) acc[o] = acc_in[o];
    if (pending) {
        __syncthreads();                 // (A) the previous panel row is staged in Lf
        const double* Xg = sh.Lf;
        // the update of a third tile (T waves 0 and 1 own three) is computed by the M wave of the same number, idle
        // until the chain starts, and handed over through the (still unused) Ms buffers
        constexpr int NP_ = NT > 2 ? 2 : NT;
#pragma unroll
        for (int o = 0; o < NP_; ++o) {
            const int tr = cd_tr(B + CD_TW * o), tc = cd_tc(B + CD_TW * o);
#pragma unroll 4
            for (int kk = 0; kk < 64; kk += 4)
                acc[o] = __builtin_amdgcn_mfma_f64_16x16x4f64(Xg[(kk + lr) * CD_LD + 16 * tr + lc],
                                                              -Xg[(kk + lr) * CD_LD + 16 * tc + lc], acc[o], 0, 0, 0);
        }
        __syncthreads();                 // (B) Lf may be cleared
        if (NT > 2) {
            static_assert(CD_TT <= 3 && CD_MW >= 2, "third tiles go to M waves 0 and 1");
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) acc[NT - 1][reg] += sh.Ms[B][reg * 64 + l];
        }
    }
    // strip of pivot block 0
#pragma unroll
    for (int o = 0; o < NT; ++o) {
        if (cd_tc(B + CD_TW * o) == 0 && lc < 4) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) sh.Tst[0][lc * 64 + 16 * cd_tr(B + CD_TW * o) + lr + 4 * reg] = acc[o][reg];
        }
    }
    __syncthreads();                     // (C) strips, cleared collectors and flags are visible
#pragma unroll 1
    for (int sb = 0; sb < 4; ++sb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s = 4 * sb + q;
            const int p0 = 16 * sb + 4 * q;
            if (s >= n_piv4) return;
            CD_STAMP(s0);
            if (s > 0) {
                cd_wait<1>(sh, s, 0, 0, 0);                  // panel s-1 (which also means the panel wave is done with strip buffer (s+1) & 1)
                CD_STAMP(s1);
                const double* Xp = sh.Xs[(q + 3) & 3];
                double a[NT], b[NT];
#pragma unroll
                for (int o = 0; o < NT; ++o) {
                    a[o] = Xp[lr * 64 + 16 * cd_tr(B + CD_TW * o) + lc];
                    b[o] = Xp[lr * 64 + 16 * cd_tc(B + CD_TW * o) + lc];
                }
#pragma unroll
                for (int o = 0; o < NT; ++o) acc[o] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[o], -b[o], acc[o], 0, 0, 0);   // T -= X X^T
                CD_ACC_T(3, s0, s1, 128);
            }
            // strip of block s+1 (updates <= s-1 applied)
            if (p0 + 4 < 64) {
                const int p1 = p0 + 4, bb = p1 >> 4, o1 = p1 & 15;
                double* To = sh.Tst[(q + 1) & 1];
#pragma unroll
                for (int o = 0; o < NT; ++o) {
                    if (cd_tc(B + CD_TW * o) == bb && lc >= o1 && lc < o1 + 4) {
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) To[(lc - o1) * 64 + 16 * cd_tr(B + CD_TW * o) + lr + 4 * reg] = acc[o][reg];
                    }
                }
            }
            cd_post(sh, 2 + B, s + 1);
            CD_STAMP(s2);
            CD_ACC_T(4, s0, s2, 128);
        }
    }
}

// M wave C: tiles C, C + CD_MW, ... (< 10) of the running inverse.
template <int C>
__device__ __forceinline__ void cd_m_wave(CdShared& sh, int n_piv4, int pending)
{
    const int l = threadIdx.x & 63, lr = l >> 4, lc = l & 15;
    constexpr int NT = (10 - C + CD_MW - 1) / CD_MW;
    d4 acc[NT];
#pragma unroll
    for (int o = 0; o < NT; ++o)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int idx = C + CD_MW * o;
            acc[o][reg] = (16 * cd_tr(idx) + lr + 4 * reg == 16 * cd_tc(idx) + lc) ? 1.0 : 0.0;
        }
    if (pending) {
        __syncthreads();                                  // (A)
        // pending update of trailing tile 2 * CD_TW + C on behalf of T wave C (see cd_t_wave)
        constexpr int idx = 2 * CD_TW + C;
        if (idx < 10) {
            const double* Xg = sh.Lf;
            d4 pacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
            for (int kk = 0; kk < 64; kk += 4)
                pacc = __builtin_amdgcn_mfma_f64_16x16x4f64(Xg[(kk + lr) * CD_LD + 16 * cd_tr(idx) + lc],
                                                            -Xg[(kk + lr) * CD_LD + 16 * cd_tc(idx) + lc], pacc, 0, 0, 0);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) sh.Ms[C][reg * 64 + l] = pacc[reg];
        }
        __syncthreads();                                  // (B)
    }
    __syncthreads();                                      // (C)
#pragma unroll 1
    for (int sb = 0; sb < 4; ++sb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s = 4 * sb + q;
            const int p0 = 16 * sb + 4 * q;
            if (s >= n_piv4) return;
            { const int sp = cd_wait<1>(sh, 0, s - 1, 0, 0);                  // Ms of block s-2; the inverse wave is done with strip buffer s & 1
              if (C == 0 && l == 0) { CD_SPIN_ADD(4, sp); CD_SPIN_ADD(5, 1); } (void)sp; }
            if (s > 1) {
                const double* Xq = sh.Xs[(q + 2) & 3];       // panel s-2
                const double* Mq = sh.Ms[q & 1];             // Ms of block s-2
                // (rows of the inverse above the pivot block are final: a tile whose rows all are -- 16 tr + 15 <= 4 (s-2) + 3 --
                //  gets no more updates)
#pragma unroll
                for (int o = 0; o < NT; ++o) {
                    const int idx = C + CD_MW * o;
                    if (sb <= cd_tr(idx) || (sb == cd_tr(idx) + 1 && q == 0)) {
#if !defined(ABL_NO_M_MFMA)
                        acc[o] = __builtin_amdgcn_mfma_f64_16x16x4f64(Xq[lr * 64 + 16 * cd_tr(idx) + lc], -Mq[lr * 64 + 16 * cd_tc(idx) + lc],
                                                                      acc[o], 0, 0, 0);          // M -= X (L4^-1 M(p,:))
#else
                        acc[o][0] += Xq[lr * 64 + 16 * cd_tr(idx) + lc] * Mq[lr * 64 + 16 * cd_tc(idx) + lc];
#endif
                    }
                }
            }
            // strip of block s (updates <= s-2 applied): rows p0 + lr of a tile in block row b0 = register q
            const int b0 = p0 >> 4;
            double* Mo = sh.Mst[q & 1];
#pragma unroll
            for (int o = 0; o < NT; ++o) {
                const int idx = C + CD_MW * o;
                if (cd_tr(idx) == b0) Mo[lr * 64 + 16 * cd_tc(idx) + lc] = acc[o][q];
            }
            cd_post(sh, 2 + CD_TW + C, s + 1);
        }
    }
}

// Global reads of one diagonal block, issued before anything else so that they can fly while the frame scalars
// (sel[]) are still being fetched: the block's own tiles (T waves) and, for the single-launch step, the operands
// of its panel row.
struct CdPre { d4 tacc[CD_TT]; double a[64 / (CD_THREADS / 64)], l[64 / (CD_THREADS / 64)]; };

__device__ __forceinline__ void cd_preload(CdPre& pre, const double* __restrict__ A, long ldA, int step,
                                           const double* __restrict__ Linv, int pending)
{
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const double* tile = A + (long)step * 64 + (long)step * 64 * ldA;
#pragma unroll
    for (int o = 0; o < CD_TT; ++o) pre.tacc[o] = (d4){0.0, 0.0, 0.0, 0.0};
    if (wave >= 2 && wave < 2 + CD_TW) {
        const int l = t & 63, lr = l >> 4, lc = l & 15;
#pragma unroll
        for (int o = 0; o < CD_TT; ++o) {
            const int idx = (wave - 2) + CD_TW * o;
            if (idx >= 10) continue;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = 16 * cd_tr(idx) + lr + 4 * reg, col = 16 * cd_tc(idx) + lc;
                // the lower triangle of the global tile is authoritative; mirror it
                pre.tacc[o][reg] = (row >= col) ? tile[row + (long)col * ldA] : tile[col + (long)row * ldA];
            }
        }
    }
    if (pending == 2) {
        const double* Ag = A + (long)step * 64 + (long)(step - 1) * 64 * ldA;
        const double* Lg = Linv + (long)(step - 1) * 64 * 64;
        const int row = t & 63, g = t >> 6;
#pragma unroll
        for (int q = 0; q < 64 / (CD_THREADS / 64); ++q) {
            const int m = g + (CD_THREADS / 64) * q;
            pre.a[q] = Ag[row + (long)m * ldA];
            pre.l[q] = Lg[row + 64 * m];
        }
    }
}

// pending: 0 nothing; 1 lookahead (the panel X = A(k,k-1) is in global memory, T -= X X^T is applied here);
//          2 single-launch block step (X is formed here from A(k,k-1) and L^-1(k-1), both preloaded from global memory).
// (The persistent sweep has its own block loop: cdp_role.)
// LOCAL: nothing is read from or written to global memory: the block comes in `pre.tacc`, and L^-1 is left complete in
//          sh.Mf (zero above the diagonal, identity beyond the rows that exist) behind a barrier (single-block systems,
//          factored redundantly by every strip workgroup of the persistent sweep; pending must be 0).
template <bool LOCAL = false>
__device__ __forceinline__ void cd_factor_block(CdShared& sh, double* __restrict__ A, long ldA, int step,
                                                const int32_t* __restrict__ sel, int slot_k, double* __restrict__ Linv,
                                                int32_t* __restrict__ status, int pending, const CdPre& pre,
                                                const double* __restrict__ Xsrc = nullptr /* pending == 1: the buffer (shape of A) that holds
                                                                                              the solved panel; default A itself */)
{
    // rows/columns at and beyond r = 2k are identity padding (prepare_system_kernel): the pivot
    // chain stops after the last real row, L and L^-1 are the identity there
    const int r_here = min(64, max(0, 2 * sel[slot_k] - 64 * step));
    const int n_piv4 = (r_here + 3) >> 2;
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    double* tile = LOCAL ? nullptr : A + (long)step * 64 + (long)step * 64 * ldA;
    double* Lout = LOCAL ? nullptr : Linv + (long)step * 64 * 64;
    CD_STAMP(pr0);
    const d4 (&tacc)[CD_TT] = pre.tacc;
    if (pending == 2) {
        // Single-launch block step: the panel row of THIS block for the previous column, X = A(k,k-1) Linv(k-1)^T,
        // is formed here (nobody else needs it: the tile workgroups of the same launch recompute the S-row panels
        // they use).  Operands staged behind Lf, over members that are initialised afterwards.
        double* Aop = sh.Mf;                         // [m][CD_OPLD] : A(k,k-1)(row, m)
        double* Lop = sh.Mf + 64 * CD_OPLD;          // [m][CD_OPLD] : Linv(k-1)(c, m)
        const int row = t & 63, g = t >> 6;
#pragma unroll
        for (int q = 0; q < 64 / (CD_THREADS / 64); ++q) {
            const int m = g + (CD_THREADS / 64) * q;
            Aop[m * CD_OPLD + row] = pre.a[q];
            Lop[m * CD_OPLD + row] = pre.l[q];
        }
        __syncthreads();
        CD_STAMP(pr1);
        CD_ACC_T(8, pr0, pr1, 0);
        if (wave < 8) {
            // L^-1 is lower triangular: column tile tc of X only needs k < 16 (tc + 1).  Waves pair the column tiles
            // {0,3} and {1,2} of one row tile: 20 MFMAs each instead of 32.
            const int l = t & 63, lr = l >> 4, lc = l & 15;
            const int tr = wave >> 1;
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int tc = (wave & 1) ? 1 + o : 3 * o;
                d4 acc = {0.0, 0.0, 0.0, 0.0};
                for (int kk = 0; kk < 16 * (tc + 1); kk += 4)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Aop[(kk + lr) * CD_OPLD + 16 * tr + lc],
                                                               Lop[(kk + lr) * CD_OPLD + 16 * tc + lc], acc, 0, 0, 0);
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) sh.Lf[(16 * tc + lc) * CD_LD + 16 * tr + lr + 4 * reg] = acc[reg];
            }
        }
        __syncthreads();
        CD_STAMP(pr2);
        CD_ACC_T(9, pr1, pr2, 0);
    }
    CD_STAMP(pr3);
    {
        const int row = t & 63, g = t >> 6;
        if (pending == 1) {
            // Lookahead: the trailing update of step-1 for THIS tile, A(k,k) -= X X^T with X = A(k,k-1)
            // (already solved by panel(step-1)), is applied by the T waves so that the trailing-update
            // kernel of the previous step can run elsewhere while this block is factored.
            const double* Xg = (Xsrc ? Xsrc : A) + (long)step * 64 + (long)(step - 1) * 64 * ldA;
            for (int c = g; c < 64; c += CD_THREADS / 64) sh.Lf[c * CD_LD + row] = Xg[row + (long)c * ldA];
        }
        if (t < 16) sh.flags[t] = 0;
        if (t == 16) sh.timeout = 0;
        if (g < 4) sh.Xs[3][g * 64 + row] = 0.0;                                         // "panel" -1
        else if (g < 8) { sh.Mst[0][(g - 4) * 64 + row] = 0.0; sh.Mst[1][(g - 4) * 64 + row] = 0.0; }
    }
    if (wave == 0 || wave == 1) {
        if (pending) { __syncthreads(); __syncthreads(); }          // (A), (B)
        {   // L, L^-1 collectors cleared
            const int row = t & 63, g = t >> 6;      // g = 0, 1
            for (int c = g; c < 64; c += 2) { sh.Lf[c * CD_LD + row] = 0.0; sh.Mf[c * CD_LD + row] = 0.0; }
        }
        __syncthreads();                                             // (C)
        CD_STAMP(pr4);
        CD_ACC_T(10, pr3, pr4, 0);
        if (wave == 0) cd_panel_wave(sh, n_piv4);
        else (void)cd_inverse_wave(sh, n_piv4);
        CD_STAMP(pr5);
        CD_ACC_T(11, pr4, pr5, 0);
    } else {
        switch (wave) {
        case 2: cd_t_wave<0>(sh, n_piv4, tacc, pending); break;
        case 3: cd_t_wave<1>(sh, n_piv4, tacc, pending); break;
        case 4: cd_t_wave<2>(sh, n_piv4, tacc, pending); break;
        case 5: cd_t_wave<3>(sh, n_piv4, tacc, pending); break;
        case 6: cd_m_wave<0>(sh, n_piv4, pending); break;
        default: cd_m_wave<1>(sh, n_piv4, pending); break;
        }
    }
    __syncthreads();
    bool bad = false;
    {
        const int i = t & 63, g = t >> 6;
        const int done = 4 * n_piv4;                 // pivots processed; beyond them L = L^-1 = I
        for (int c = g; c < 64; c += CD_THREADS / 64) {
            const bool pad = (c >= done);
            const double lv = pad ? ((i == c) ? 1.0 : 0.0) : sh.Lf[c * CD_LD + i];
            const double mv = (i >= done) ? ((i == c) ? 1.0 : 0.0) : sh.Mf[c * CD_LD + i];
            if (i == c && !(lv > 0.0 && lv < 1.0e300)) bad = true;      // a non-positive pivot turns the diagonal into NaN / 0 / inf
            if constexpr (LOCAL) {
                sh.Mf[c * CD_LD + i] = (i >= c) ? mv : 0.0;
            } else {
                Lout[i + 64 * c] = (i >= c) ? mv : 0.0;
                if (i >= c) tile[i + (long)c * ldA] = lv;
            }
        }
    }
    if constexpr (LOCAL) __syncthreads();
    CD_STAMP(pr6);
    CD_ACC_T(12, pr0, pr6, 0);
    if (t == 0) { CD_ACC_T(13, pr0, pr0 + 1, 0); }
    if (bad) atomicMin(status, -6);                              // RSLAM_ERR_NOT_SPD
    if (t == 0 && sh.timeout) atomicMin(status, -3);             // RSLAM_ERR_HIP: hand-over protocol broke (never expected)
}

__global__ void __launch_bounds__(CD_THREADS)
chol_diag_kernel(double* __restrict__ A, long ldA, int step, const int32_t* __restrict__ sel, int slot_nblk,
                 int slot_k, double* __restrict__ Linv, int32_t* __restrict__ status, int pending)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    CdPre pre;
    cd_preload(pre, A, ldA, step, Linv, pending);                 // speculative: flies with the sel[] fetch
    if (step >= sel[slot_nblk]) return;
    cd_factor_block(*reinterpret_cast<CdShared*>(lds), A, ldA, step, sel, slot_k, Linv, status, pending, pre);
}

// rows of block b participate in step `step` of a sweep with nblk column blocks?
__device__ __forceinline__ bool row_block_active(int b, int step, int nblk, int rp_blocks)
{
    if (b <= step) return false;
    if (b < rp_blocks && b >= nblk) return false;   // padding rows of S
    return true;
}

// coalesced write of an LDS tile Cs[col][row] to a column-major global tile
__device__ __forceinline__ void store_tile(const double* Cs, double* C, long ldc)
{
    const int row = threadIdx.x & 63, g = threadIdx.x >> 6;
#pragma unroll 4
    for (int q = 0; q < 16; ++q) {
        const int c = g + 4 * q;
        C[row + (long)c * ldc] = Cs[c * TS_LD + row];
    }
}

__global__ void __launch_bounds__(256)
panel_kernel(double* __restrict__ A, long ldA, int step, const int32_t* __restrict__ sel, int slot_nblk,
             const double* __restrict__ Linv, int rp_blocks, double* __restrict__ out /* same shape as A; may be A itself */)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int nblk = sel[slot_nblk];
    if (step >= nblk) return;
    const int b = blockIdx.x;
    if (!row_block_active(b, step, nblk, rp_blocks)) return;
    double* tile = A + (long)b * 64 + (long)step * 64 * ldA;
    TgAcc acc;
    tg_zero(acc);
    tile_gemm_nt(tile, ldA, Linv + (long)step * 64 * 64, 64, 64, lds, acc);   // X * Linv^T
    tg_acc_to_lds(acc, lds, 1.0);
    __syncthreads();
    store_tile(lds, out + (long)b * 64 + (long)step * 64 * ldA, ldA);
}

// One launch per block step k of the sweep:
//   workgroup 0      forms its own panel row X = A(k+1,k) Linv(k)^T, applies the step-k update to tile
//                    (k+1,k+1) and factors it (cd_factor_block, pending = 2): the serial chain of the sweep;
//   workgroup (i,j)  tile (i,j), j = k+1+jj: recomputes the two panel blocks it needs,
//                    Y_i = A(i,k) Linv(k)^T and Y_j = A(j,k) Linv(k)^T, as LDS-resident MFMA operands and
//                    applies A(i,j) -= Y_i Y_j^T; the jj = 0 column also stores Y_i for the P H^T / nu rows into
//                    a second buffer of the same shape (the S-row panels are never read again, so they are
//                    never written; A(i,k) itself is read by the other tiles of row i during the launch).
// The panel solve thus leaves the chain: a block step costs max(diagonal path, three K = 64 tile products)
// instead of panel launch + max(trailing, diagonal).  All workgroups have the diagonal block's shape
// (CD_THREADS); the tile workgroups retire their surplus waves at once.
__device__ __forceinline__ void step_tile(double* __restrict__ A, long ldA, int step, int nblk, int rp_blocks,
                                          int i, int j, const double* __restrict__ Linv_k, double* __restrict__ Yout, double* lds,
                                          bool store_s_rows = false /* also the panel blocks of the S rows below k+1 (a wide pass reads them) */)
{
    if (!row_block_active(i, step, nblk, rp_blocks)) return;
    const bool has_col = (j < nblk) && (i >= j) && !(i == step + 1 && j == step + 1);   // (k+1,k+1) belongs to workgroup 0
    // (store_s_rows: also tile (k+1, k), whose product the diagonal workgroup forms for itself and keeps in LDS: the staged
    //  route's group inverse reads every L(a, c) from the second buffer)
    const bool store_y = (j == step + 1) && (i >= rp_blocks || (store_s_rows && i >= step + 1));
    if (!has_col && !store_y) return;
    double* bufA = lds;                               // A(i,k), then Y_i
    double* bufB = lds + 2 * TG_OPER_DOUBLES;         // Linv(k)
    double* bufC = lds + 4 * TG_OPER_DOUBLES;         // A(j,k), then Y_j
    double* Ai = A + (long)i * 64 + (long)step * 64 * ldA;
    double* C = A + (long)i * 64 + (long)j * 64 * ldA;
    const int row = threadIdx.x & 63, g = threadIdx.x >> 6;
    const bool two = has_col && i != j;
    // every global read of the tile is issued before the first wait: one memory latency for the whole tile
    tg_fill64(Ai, ldA, bufA);
    tg_fill64(Linv_k, 64, bufB);
    if (two) tg_fill64(A + (long)j * 64 + (long)step * 64 * ldA, ldA, bufC);
    __syncthreads();
    TgAcc yi, yj;
    tg_zero(yi);
    tg_gemm64_lds(bufA, bufB, yi);                    // Y_i = A(i,k) Linv^T
    __builtin_amdgcn_sched_barrier(0);                // keep the two products' LDS prefetch windows apart (register pressure)
    if (two) {
        tg_zero(yj);
        tg_gemm64_lds(bufC, bufB, yj);                // Y_j = A(j,k) Linv^T
    }
    __builtin_amdgcn_sched_barrier(0);
    double cin[16];                                   // the tile to update: in flight under the last product
    if (has_col) {
#pragma unroll
        for (int q = 0; q < 16; ++q) cin[q] = C[row + (long)(g + 4 * q) * ldA];
    }
    __syncthreads();
    // the products become operands: a 64 x 64 result written with the operand leading dimension IS two operand buffers
    tg_acc_to_lds<TG_LD>(yi, bufA, 1.0);
    if (two) tg_acc_to_lds<TG_LD>(yj, bufC, 1.0);
    __syncthreads();
    const double* Yj = two ? bufC : bufA;
    if (store_y) {                                    // into the copy of the system: A(i,k) itself is still being read by the row's other tiles
        double* Yi = Yout + (long)i * 64 + (long)step * 64 * ldA;
#pragma unroll 4
        for (int q = 0; q < 16; ++q) { const int c = g + 4 * q; Yi[row + (long)c * ldA] = bufA[c * TG_LD + row]; }
    }
    if (!has_col) return;
    TgAcc acc;
    tg_zero(acc);
    tg_gemm64_lds(bufA, Yj, acc);                     // Y_i Y_j^T
    __syncthreads();
    tg_acc_to_lds(acc, lds, 1.0);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c = g + 4 * q;
        C[row + (long)c * ldA] = cin[q] - lds[c * TS_LD + row];
    }
}

__global__ void __launch_bounds__(CD_THREADS)
sweep_step_kernel(double* __restrict__ A, long ldA, int step, const int32_t* __restrict__ sel, int slot_nblk,
                  int slot_k, int rp_blocks, int row_blocks, double* __restrict__ Linv, double* __restrict__ Yout,
                  int32_t* __restrict__ status, int narrow /* large systems: 0 every trailing column; 1 column k+1 only, its panel
                  blocks stored for every row (grid 1 + row_blocks); 2 the same + the diagonal tile (k+2,k+2) (one workgroup more) */)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if (blockIdx.x == 0) {
        if (step + 1 >= rp_blocks) return;                        // no such block (and its tiles would be out of bounds)
        CdPre pre;
        cd_preload(pre, A, ldA, step + 1, Linv, 2);               // speculative: flies with the sel[] fetch
        if (step + 1 < sel[slot_nblk])
            cd_factor_block(*reinterpret_cast<CdShared*>(lds), A, ldA, step + 1, sel, slot_k, Linv, status, 2, pre);
        return;
    }
    const int nblk = sel[slot_nblk];
    if (step >= nblk) return;
    if (threadIdx.x >= 256) return;                               // before any barrier: the tile code is written for 4 waves
    const int b = blockIdx.x - 1;
    if (narrow && b >= row_blocks) {                               // (narrow == 2) panel k onto the diagonal tile after the next one
        step_tile(A, ldA, step, nblk, rp_blocks, step + 2, step + 2, Linv + (long)step * 64 * 64, Yout, lds);
        return;
    }
    step_tile(A, ldA, step, nblk, rp_blocks, b % row_blocks, b / row_blocks + step + 1, Linv + (long)step * 64 * 64, Yout, lds, narrow != 0);
}

// compile-time loop: an accumulator array must never be indexed dynamically (it would move to scratch)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// The trailing update of large systems: block steps in pairs.
//   narrow pass: panel k onto column block k+1 only -- and onto the diagonal tile after it, which the NEXT diagonal workgroup
//                needs complete but for its own panel.  This is sweep_step_kernel restricted to one column (narrow = 1, 2):
//                every tile forms the two panel blocks it needs itself and stores its own for the wide pass, so the first
//                step of a pair has no panel launch (launch_factor_sweep);
//   wide pass:   panels k, k+1 onto every column block from j_lo = k+2 on, K = 128 per tile and pass, as a stream
//                (trail_stream2_kernel): the tiles, row-major over the active rows, are dealt out in contiguous ranges to at most
//                (compute units - 1) workgroups whose two four-wave engines take alternate tiles through the LDS-DMA engine of
//                the rank update -- two tiles in flight per compute unit, no workgroup launch / drain per tile (the round-2 form,
//                one workgroup per tile and step with half of its waves retired, moved 2.7 TB/s at C5).
// Workgroup 0 of either factors the next diagonal block (cd_factor_block applies the panel in front of it itself).
struct TrailPass { int p0, j_lo; };      // first panel block, first column block of the pass

__device__ __forceinline__ void trail_decode(int t, const TrailPass& ps, int nS, int tri, int rp_blocks, int& i, int& j)
{
    if (t < tri) {                                   // S rows: row r = i - j_lo holds the tiles j = j_lo .. i
        int r = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while ((r + 1) * (r + 2) / 2 <= t) ++r;
        while (r * (r + 1) / 2 > t) --r;
        i = ps.j_lo + r; j = ps.j_lo + (t - r * (r + 1) / 2);
    } else {                                         // P H^T and nu rows: every column of the pass
        const int u = t - tri;
        i = rp_blocks + u / nS; j = ps.j_lo + u % nS;
    }
}

// The wide pass (two panels, K = 128) with its two engines half a tile apart: while one engine runs the four chunks of a
// tile, the other writes its previous tile back and fetches its next one, so the matrix pipes of the compute unit see one
// engine's MFMAs at a time, back to back, instead of both engines' followed by both engines' memory phases (the lock-step
// form above: 35 TFLOP/s at C5's first pass).  A slot = the three chunk barriers of the computing engine; the other engine
// matches them: the values of its next tile are requested first of all; (1) its accumulators are staged in the operand pair its
// tile finished with two chunks ago (an LDS-only barrier: the requests stay in flight), (2) a bare s_barrier -- its stores,
// the next tile and that tile's first operand chunk stay in flight -- and (3) the wait for them.
// (Measured the same, C5 frame 2.440 ms against 2.441: both engines in step as one software pipeline -- the last chunk of a
//  tile fetching the first chunk of the next, the write-back behind a bare s_barrier -- at 248 VGPRs; not kept.  Either way a
//  K = 128 tile is ~6.5 us of engine time: 40 TFLOP/s chip-wide against the 58 of the rank update's K = 1664 loop.)
__global__ void __launch_bounds__(CD_THREADS)
trail_stream2_kernel(double* __restrict__ A, long ldA, TrailPass ps, const int32_t* __restrict__ sel, int slot_nblk,
                     int slot_k, int rp_blocks, int row_blocks, double* __restrict__ Linv, int32_t* __restrict__ status,
                     const double* __restrict__ X /* the solved panels (shape of A) */)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if (blockIdx.x == 0) {
        if (ps.j_lo >= rp_blocks) return;
        CdPre pre;
        cd_preload(pre, A, ldA, ps.j_lo, Linv, 1);
        if (ps.j_lo < sel[slot_nblk])
            cd_factor_block(*reinterpret_cast<CdShared*>(lds), A, ldA, ps.j_lo, sel, slot_k, Linv, status, 1, pre, X);
        return;
    }
    const int nblk = sel[slot_nblk];
    if (ps.j_lo >= nblk) return;
    const int nS = nblk - ps.j_lo, nP = row_blocks - rp_blocks, tri = nS * (nS + 1) / 2;
    const int total = tri + nP * nS;
    // Workgroups are dealt round-robin over the 8 XCDs (speed only, never correctness): the ranges are handed out so that the
    // workgroups that share an L2 own CONSECUTIVE ranges -- a dozen neighbouring rows whose X_i panels and the pass's X_j panels
    // fit that L2 -- instead of every eighth one
    const int W = (int)gridDim.x - 1;
    int w;
    {
        const int bx = (int)blockIdx.x, g = bx & 7;              // bx = 1 .. W
        int off = 0;                                              // workgroups of the XCD groups in front of g (order 1, 2, .., 7, 0)
        for (int h = 1; h < 8 && (g == 0 || h < g); ++h) off += (W >= h) ? (W - h) / 8 + 1 : 0;
        w = off + (g == 0 ? (bx >> 3) - 1 : (bx >> 3));
    }
    const int per = (total + W - 1) / W;
    const int t0 = w * per, t1 = min(total, t0 + per);
    if (t0 >= t1) return;                                         // (uniform over the workgroup, before any barrier)
    const int n = t1 - t0;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), e = wave >> 2, wave4 = wave & 3;
    const int tid = threadIdx.x & 255, row = tid & 63, g = tid >> 6;
    const int n_mine = (n - e + 1) / 2;                           // this engine's tiles: t0 + e, t0 + e + 2, ...
    double* hbase = lds + (long)e * TD_LDS_DOUBLES;
    double* Ac = hbase;                                           // operand pair of chunks 0 and 2 of the tile in hand
    double* Bc = hbase + TD_OPER_DOUBLES;
    double* An = hbase + 2 * TD_OPER_DOUBLES;                     // ... of chunks 1 and 3
    double* Bn = hbase + 3 * TD_OPER_DOUBLES;
    const unsigned lo = td_lane_offset(ldA);
    const double* Xi = A;                                         // operands and target of the tile in hand
    const double* Xj = A;
    double* C = A;
    bool own = false;                                             // ... and whether it is this engine's to write
    double cin[16], cnx[16];
    TgAcc acc;
    tg_zero(acc);
    // tile r of this engine: where it is, and its values into `dst`
    auto fetch_tile = [&](int r, double (&dst)[16], const double*& xi, const double*& xj, double*& c, bool& mine) {
        const int t = t0 + 2 * r + e;
        int i, j;
        trail_decode(t, ps, nS, tri, rp_blocks, i, j);
        xi = X + 64L * i + (long)ps.p0 * 64 * ldA;
        xj = X + 64L * j + (long)ps.p0 * 64 * ldA;
        c = A + 64L * i + 64L * j * ldA;
        mine = t != 0;                                            // tile 0 = (j_lo, j_lo): the diagonal workgroup's own
        if (mine) {
#pragma unroll
            for (int q = 0; q < 16; ++q) dst[q] = c[row + (long)(g + 4 * q) * ldA];
        }
    };
    if (e == 0) {                                                 // (n >= 1: engine 0 always has a tile)
        fetch_tile(0, cin, Xi, Xj, C, own);
        td_issue_chunk_w<0>(Xi, ldA, Xj, ldA, 0, Ac, Bc, wave4);
    }
    __syncthreads();
    for (int s = 0; s <= n; ++s) {
        const int r = s >> 1;
        if ((s & 1) == e) {
            if (r < n_mine) {
                // ---- compute: four chunks, three barriers
                double a[TG_MI];
                BFrag b[TG_NI];
                tg_zero(acc);
                td_read_frags(td_frag_ptr(Ac, Bc, wave4), 0, a, b);
                td_compute_chunk_w<true, 0, 0>(Ac, Bc, acc, a, b, Xi, ldA, lo, Xj, ldA, lo, 1 * TG_KC, An, Bn, wave4);
                td_compute_chunk_w<true, 0, 0>(An, Bn, acc, a, b, Xi, ldA, lo, Xj, ldA, lo, 2 * TG_KC, Ac, Bc, wave4);
                td_compute_chunk_w<true, 0, 0>(Ac, Bc, acc, a, b, Xi, ldA, lo, Xj, ldA, lo, 3 * TG_KC, An, Bn, wave4);
                td_compute_chunk_w<false, 0, 0>(An, Bn, acc, a, b, Xi, ldA, lo, Xj, ldA, lo, 0, Ac, Bc, wave4);
            } else {
                __syncthreads(); __syncthreads(); __syncthreads();
            }
        } else {
            // ---- write the tile of the previous slot back, fetch the one of the next slot
            const bool finish = s >= 1 && ((s - 1) >> 1) < n_mine;
            const int rn = (s + 1) >> 1;
            const bool more = rn < n_mine;
            double* Cw = C;
            const bool ownw = own;
            if (more) fetch_tile(rn, cnx, Xi, Xj, C, own);        // the next tile's values: requested first, they have the whole slot
            if (finish) tg_acc_to_lds_w(acc, Ac, 1.0, wave4);     // (Ac, Bc): read for the last time two chunks ago
            // (1) staged; every wave of the engine is through its last chunk.  Bare: the loads above stay in flight
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (more) td_issue_chunk_w<0>(Xi, ldA, Xj, ldA, 0, An, Bn, wave4);
            if (finish && ownw) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int c = g + 4 * q;
                    Cw[row + (long)c * ldA] = cin[q] - Ac[c * TS_LD + row];
                }
            }
            __builtin_amdgcn_s_barrier();                         // (2) matches the computing engine's second chunk: nothing to wait for here
            __syncthreads();                                      // (3) stores out, next tile and its first chunk in
            if (more) {
#pragma unroll
                for (int q = 0; q < 16; ++q) cin[q] = cnx[q];
                double* sw = Ac; Ac = An; An = sw;
                sw = Bc; Bc = Bn; Bn = sw;
            }
        }
    }
}

static_assert(TG_KC == 32, "trail_stream2_kernel: a 64-column panel is two chunks");
static_assert(2 * TD_OPER_DOUBLES >= TS_DOUBLES, "trail_stream2_kernel stages a tile over one operand pair");

// ---------------------------------------------------------------------------
// K8, persistent form: the whole factor sweep of one update in ONE launch (systems whose workgroups are all
// resident at once; launch_factor_sweep decides).  Launch boundaries are replaced by flags in global memory:
//   workgroup 0        the pivot chain: diagonal blocks 0, 1, ... back to back (cd_chain_persistent).
//   strip workgroups   one per 16-row strip of the stacked matrix [S; P H^T; nu^T].  A strip keeps its rows of ALL
//                      column blocks in MFMA accumulators for the whole sweep (16 x 64 nblk doubles over 8 waves):
//                      the system is read once and Y written once.  Step k (right-looking), as soon as L^-1(k) is out:
//                      X = strip(:,k) L^-T(k) (final: stored), then strip(:,j) -= X L(j,k)^T for j > k, with the
//                      panel blocks L(j,k) that the strips of the S rows j publish (buffer Ypanel + counter
//                      panel_cnt[k]).  The strips of S row block b hand the tiles (b,b-1), (b,b) with every update
//                      but the last to the chain (in place + counter row_ready[b]) while the chain factors block b-1.
// Products are computed transposed (D = L_tile X^T) so that accumulator loads/stores and the operand fetch of the
// panel tiles are 128-byte row segments of the column-major matrix.
// Memory protocol: an MI355X has eight L2 caches (one per XCD) that are not coherent with each other inside a
// kernel.  Agent-scope fences would make them so by writing back / invalidating a whole L2 on every hand-over
// (buffer_wbl2 / buffer_inv sc1: measured 3-5 us per fence while 160 workgroups stream their operands).  Instead
// every datum that crosses workgroups is written and read with agent-scope relaxed atomics -- global_store /
// global_load ... sc1, which go to the memory side -- the producer waits for its stores (s_waitcnt vmcnt(0)) before
// the flag, the consumer polls the flag with the same kind of load.  Data read from the previous kernel (the
// prepared system) and results read by the next kernel (Y) use ordinary accesses.
// Forward progress: a workgroup only waits for workgroup 0 and for strips of S rows, which have lower block indices
// and are dispatched first; the host only takes this path when the whole grid is resident at once.  Every wait is
// bounded (status -3) so that a scheduling surprise ends in an error code, never in a hung queue.
// ---------------------------------------------------------------------------
// optional time stamps (100 MHz wall clock) of the persistent sweep: dbg[who][block/step][slot], see scripts/sweep_stamps.py
constexpr int SWD_WHO = 6, SWD_K = 16, SWD_SLOT = 8;      // who 5: tile worker 0
constexpr int SWD_WG = 256;                                // behind the [who][k][slot] stamps: start and end of every workgroup
constexpr int SWD_TOTAL = SWD_WHO * SWD_K * SWD_SLOT + 2 * SWD_WG;
__device__ __forceinline__ void sw_stamp(unsigned long long* dbg, int who, int k, int slot, int tid = 0)
{
    if (dbg && (int)threadIdx.x == tid && k < SWD_K) dbg[(who * SWD_K + k) * SWD_SLOT + slot] = wall_clock64();
}

#ifndef SW_STAMP_STRIP
#define SW_STAMP_STRIP 8             // the S strip whose steps scripts/sweep_stamps.py shows (first strip of row block 2)
#endif
constexpr int SW_MAX_BLOCKS = 32;
struct SweepFlags {
    int32_t linv_ready;                 // diagonal blocks whose L^-1 is published
    int32_t xrow_ready;                 // k: the chain has published its panel blocks L(1,0) .. L(k,k-1)
    int32_t tiles01;                    // P H^T / nu strips that have written their share of tiles (0,0), (1,0), (1,1)
    int32_t pad[13];
    int32_t panel_cnt[SW_MAX_BLOCKS];   // [k]: S strips that have published their rows of panel k (row blocks k+2 ..)
    int32_t row_ready[SW_MAX_BLOCKS];   // [b]: strips of S row block b that have handed tiles (b,b-1), (b,b) over
    int32_t row_cnt[SW_MAX_BLOCKS];     // [b]: panel rows published by the strips of S row block b, all steps (4 per step)
    int32_t y_flag[256];                // [strip]: column blocks of Y (u^T) this P H^T (nu) strip has published -- one word per
                                        // producer, no atomics: the tile workers poll the strips of their own row blocks
    int32_t tiles01s[4 * 32];           // tiles01 in four shards on cache lines of their own ([32 s], strip & 3): ~120 strips finish
                                        // their share within a microsecond of each other, and adds to ONE address serialise
};
static_assert(SWEEP_FLAG_INTS <= CD_THREADS, "one thread per flag word clears the other set");
static_assert(sizeof(SweepFlags) == sizeof(int32_t) * SWEEP_FLAG_INTS, "flag block size");
// Every wait on another workgroup is bounded in TIME (100 MHz wall clock): the longest legitimate wait is one diagonal block
// of the chain (~15 us) -- the budget is ~70 times that, so that a launch whose workgroups are not all resident (another user
// of the device) costs about one frame before the host falls back to the launch-per-step route, not the ~30 ms (140 frames)
// of the spin-count bound that stood here until round 3.
// ... AND in polls the wave itself has made (round 5): the wall clock keeps running while the waves of the launch are off the
// device -- the driver evicts a process's queues for milliseconds at a time when somebody else on the host maps or moves
// memory -- and when they come back every wait in progress finds its millisecond gone at once.  The round-5 soaks saw one
// expiry per ~100 000 frames that fits nothing else (NOTEBOOK.md, round 5: status -37; SEL_WAIT_FIRST: a tile worker's wait
// for the Y blocks of step 1 ran out FIRST, although the strips that publish them wait for the chain with bounds of their own
// that started earlier).  A poll is a round trip to the L2 (>= 0.5 us): 768 of them are most of a millisecond of a wave that
// is RUNNING, which is what the bound is about.
constexpr unsigned long long SW_WAIT_TICKS = 100ull * 1000;          // 1 ms
constexpr int SW_WAIT_MIN_POLLS = 768;
struct SwDeadline {
    unsigned long long t0; int n;
    __device__ __forceinline__ SwDeadline() : t0(wall_clock64()), n(0) {}
    __device__ __forceinline__ bool expired() { return ((++n & 7) == 0) && n >= SW_WAIT_MIN_POLLS && (wall_clock64() - t0 > SW_WAIT_TICKS); }
    __device__ __forceinline__ unsigned long long elapsed() const { return wall_clock64() - t0; }
};

// data that crosses workgroups inside the launch (see "Memory protocol" above)
__device__ __forceinline__ double ld_coh(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_coh(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_flag(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wait_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// a bounded wait ran out: the status word (smallest code wins) and, for diagnosis, who was first (SEL_WAIT_FIRST; `status` is
// &sel[SEL_STATUS] in every launch of the persistent sweep)
__device__ __forceinline__ void sw_timed_out(int32_t* status, int value, int need, int polls = 0, unsigned long long ticks = 0)
{
    atomicMin(status, value);
    // (the code field saturates: the chain's per-block codes -(36 + 10 k) leave 8 bits from k = 22 on)
    const int code = -value > 255 ? 255 : -value;
    const int first = atomicCAS(status + (SEL_WAIT_FIRST - SEL_STATUS), 0, (code & 0xff) | ((int)(blockIdx.x & 0xfff) << 8) | ((need & 0x7ff) << 20));
    if (first == 0) {
        // this wait is the first: how it ran out -- its own polls and the wall clock it saw go by (100 MHz ticks -> us)
        const unsigned long long us = ticks / 100;
        status[SEL_WAIT_POLLS - SEL_STATUS] = (int)(((unsigned)(polls > 0xffff ? 0xffff : polls) << 16) | (unsigned)(us > 0xffff ? 0xffff : us));
    }
}

// All threads of the workgroup.  Returns true when *flag >= need; false when the spin bound was hit (status -3) or an
// earlier wait of this workgroup had failed (*abort, in LDS): the caller then leaves at once, so that a broken
// hand-over costs one bound per workgroup, not one per wait.
// Timeouts are reported as status -(30 + code) (the host maps everything <= -30 to RSLAM_ERR_HIP and keeps the raw value
// for diagnosis): 1 L^-1 flag, 2 chain's panel row, 3 sibling strips, 4 all S strips, 5 hand-over (chain), 6 LDS pipeline
// of the chain workgroup, 7 Y blocks (tile workers), 8 u^T / Jnorm (x update).
__device__ __forceinline__ bool sw_wait(const int32_t* flag, int need, int32_t* status, int* abort, int code = 0)
{
    if (threadIdx.x == 0) {
        // two polls in flight, half a memory latency apart: the flag is seen ~half a latency earlier than with one
        SwDeadline dl;
        int a = ld_flag(flag);
        while (true) {
            __builtin_amdgcn_s_sleep(4);
            int b = ld_flag(flag);
            if (a >= need) break;
            __builtin_amdgcn_s_sleep(4);
            a = ld_flag(flag);
            if (b >= need) break;
            if (dl.expired()) { sw_timed_out(status, -(30 + code), need, dl.n, dl.elapsed()); *abort = 1; break; }
        }
    }
    __syncthreads();
    return *abort == 0;
}

// Two flags at once (one memory round trip instead of two when both are already up, as they usually are): returns when
// *fa >= na and *fb >= nb.
__device__ __forceinline__ bool sw_wait2(const int32_t* fa, int na, const int32_t* fb, int nb, int32_t* status, int* abort, int code)
{
    if (threadIdx.x == 0) {
        SwDeadline dl;
        while (true) {
            const int a = ld_flag(fa), b = ld_flag(fb);
            if (a >= na && b >= nb) break;
            if (dl.expired()) { sw_timed_out(status, -(30 + code), 64 * na + nb, dl.n, dl.elapsed()); *abort = 1; break; }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    __syncthreads();
    return *abort == 0;
}

// all threads: the coherent stores of this workgroup have reached memory, then *flag += 1
__device__ __forceinline__ void sw_post_add(int32_t* flag)
{
    wait_stores();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// LDS of the persistent sweep: the chain workgroup's staging areas behind CdShared (see cdp_role)
constexpr int CDP_AOP_DOUBLES = 64 * CD_OPLD;          // A(k,k-1): [m][CD_OPLD]
constexpr int CDP_TPRE_DOUBLES = 40 * 64;              // tile (k,k): lower 16 x 16 tiles in accumulator layout [(idx, reg)][lane]
constexpr int CDP_PEND_DOUBLES = 6 * 4 * 64;           // products of the six trailing tiles outside the first tile column (cdp_t_wave)
constexpr size_t CDP_OFF_AOP = (sizeof(CdShared) + 15) / 16 * 2;     // in doubles
constexpr size_t SWP_CHAIN_LDS_BYTES = sizeof(double) * (CDP_OFF_AOP + CDP_AOP_DOUBLES + CDP_TPRE_DOUBLES + CDP_PEND_DOUBLES) + 16;
static_assert(SWP_CHAIN_LDS_BYTES <= 160 * 1024, "the chain workgroup's LDS must fit one compute unit");
constexpr size_t SWP_LDS_BYTES = SWP_CHAIN_LDS_BYTES > sizeof(double) * 2 * TD_LDS_DOUBLES ? SWP_CHAIN_LDS_BYTES : sizeof(double) * 2 * TD_LDS_DOUBLES;   // tile workers: two engines
static_assert(SWP_LDS_BYTES >= sizeof(CdShared), "trail_stream_kernel: the diagonal workgroup shares the launch's LDS size");


// Eight waves: wave = 4 g + w.  Group g owns the column blocks j = 2 jj + g, wave w the columns 16 w .. 16 w + 15 of each.
// column c of P H^T for the update at hand (row index = state index): gathered from the matched-feature matrix (LI pass)
// or already in the P H^T rows of A (HI pass)
__device__ __forceinline__ const double* sys_wcol(const SysSrc& s, const double* A, long ldA, int NP, int RP, int c)
{
    return s.Wsrc ? s.Wsrc + (long)(2 * s.rank_of[s.list[c >> 1]] + (c & 1)) * NP : A + RP + (long)c * ldA;
}

// S(a,c) = H_a (P H^T)_c + [a == c]  (ExtendKF.cpp:602 with R = I, :594; identity on the padding): what
// prepare_system_kernel writes, computed where it is needed.  fa/Hf/o/w describe row a (fixed per lane).
__device__ __forceinline__ double sys_S(const SysSrc& s, const double* A, long ldA, int NP, int RP, int r, int a, int c,
                                        const double (&Hf)[13], int o, int w)
{
    double v = (a == c) ? 1.0 : 0.0;
    if (a < r && c < r) {
        const double* wc = sys_wcol(s, A, ldA, NP, RP, c);
        double sacc = 0;
#pragma unroll
        for (int k = 0; k < 13; ++k) if (k < w) sacc += Hf[k] * wc[col_index(o, k)];
        v += sacc;
    }
    return v;
}

// ---- K9 inside the sweep (fused launches, WorkerArgs): x_k_k = x + Y u --------------------------------------------------
// The P H^T strips hold the rows of Y, the nu strip produces u^T block by block.  Thread (row = t & 15, pair = t >> 4)
// of a P H^T strip accumulates Y(row, 2 pair .. 2 pair + 1) u over the column blocks, ONE STEP BEHIND the sweep (the X of
// step k is still in the strip's LDS when step k + 1 begins, and by then the nu strip has long published u^T of block k:
// no wait is ever exposed); after the last block the 32 partial sums of a row are added in a fixed order.  The strip of
// state rows 0..15 also normalises the quaternion, writes Jnorm (ExtendKF.cpp:613-627; Q6) and publishes the token the
// tile workers of the first block column wait for -- long before they reach their epilogue.
struct XAcc {
    double part;                // this thread's partial sum
    int row, pair;
};

__device__ __forceinline__ void xacc_add(XAcc& xa, const double* Xs /* [64][16] */, const double* uT /* row RP + NP of Ypanel */,
                                         long ldA, int kblk)
{
    const double x0 = Xs[(2 * xa.pair) * 16 + xa.row], x1 = Xs[(2 * xa.pair + 1) * 16 + xa.row];
    const double u0 = ld_coh(uT + (64L * kblk + 2 * xa.pair) * ldA), u1 = ld_coh(uT + (64L * kblk + 2 * xa.pair + 1) * ldA);
    xa.part = fma(x1, u1, fma(x0, u0, xa.part));
}

// quaternion normalisation and Jnorm from the updated pose (one thread); x[3..6] are overwritten with the unit quaternion
__device__ __forceinline__ void quat_jnorm(double* xq /* x_k_k + 3, four entries, generic pointer */, int compat, double* T)
{
    const double qr = xq[0], qx = xq[1], qy = xq[2], qz = xq[3];
    const double q2 = qr * qr + qx * qx + qy * qy + qz * qz;
    const double nrm = sqrt(q2);
    xq[0] = qr / nrm; xq[1] = qx / nrm; xq[2] = qy / nrm; xq[3] = qz / nrm;
    const double scale = compat ? (1.0 / q2) : (1.0 / (q2 * nrm));
    const double rows[16] = {
        qx*qx+qy*qy+qz*qz, -qr*qx,           -qr*qy,           -qr*qz,
        -qx*qr,            qr*qr+qy*qy+qz*qz, -qx*qy,          -qx*qz,
        -qy*qr,            -qy*qx,           qr*qr+qx*qx+qz*qz, -qy*qz,
        -qz*qr,            -qz*qx,           -qz*qy,           qr*qr+qx*qx+qy*qy };
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) st_coh(T + i + 4 * j, scale * rows[4 * i + j]);
}

// all threads of a P H^T strip: reduce the partial sums, write the strip's 16 rows of x_k_k; the strip of rows 0..15 also
// publishes Jnorm + token.  scratch: 32 * 16 + 16 doubles of LDS nobody else is using.
__device__ __forceinline__ void xacc_finish(const XAcc& xa, int first_row, const WorkerArgs& wk, double* scratch)
{
    const int t = threadIdx.x;
    scratch[xa.pair * 16 + xa.row] = xa.part;
    __syncthreads();
    double* xs = scratch + 32 * 16;
    if (t < 16) {
        double ssum = 0;
#pragma unroll
        for (int q = 0; q < 32; ++q) ssum += scratch[q * 16 + t];
        const double v = wk.x_in[first_row + t] + ssum;
        xs[t] = v;
        if (first_row != 0 || t < 3 || t > 6) wk.x_out[first_row + t] = v;
    }
    if (first_row != 0) return;
    __syncthreads();
    if (t == 0) {
        double q[4] = { xs[3], xs[4], xs[5], xs[6] };
        quat_jnorm(q, wk.compat, wk.T);
        for (int i = 0; i < 4; ++i) wk.x_out[3 + i] = q[i];
        wait_stores();
        __hip_atomic_store(wk.xu_flag, wk.token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int NJ>
__device__ __forceinline__ void sweep_strip(double* A, long ldA, int rp_blocks, int nblk, int strip, int r_total, int NP,
                                            const SysSrc& src, const double* Linv, double* Ypanel, SweepFlags* fl,
                                            int32_t* status, double* lds, unsigned long long* dbg,
                                            bool single, const int32_t* sel, int slot_k, bool tiny_off, const WorkerArgs& wk, int exp_mask)
{
    constexpr int NH = (NJ + 1) / 2;                        // column blocks per group
    const int b = strip >> 2;                               // 64-row block of the strip
    int who = -1;
    if (dbg) who = (strip == SW_STAMP_STRIP) ? 1 : (strip == 4 * (nblk - 1)) ? 2 : (strip == 4 * rp_blocks) ? 3 : (strip == (int)(ldA / 16) - 4) ? 4 : -1;
    if (who < 0) dbg = nullptr;
    const bool is_s = b < rp_blocks;
    // padding rows of S; row blocks 0 and 1 are the chain's first two diagonal blocks, assembled by the lower strips together
    // (the caller turns those workgroups into tile workers when the launch is fused)
    if (is_s && (b >= nblk || b <= 1)) return;
    if (b == (int)(ldA / 64) - 1 && (strip & 3) != 0) return;   // below nu^T there is only zero padding
    const int RP = 64 * rp_blocks;
    const bool is_nu = (b == (int)(ldA / 64) - 1);
    const bool fused = wk.Pout != nullptr;                  // x and covariance update inside this launch
    const bool xrows = fused && !is_s && !is_nu;            // this strip holds 16 rows of Y: it accumulates their x + Y u
    const bool publish = fused && !is_s && !(exp_mask & 32);   // (exp_mask & 32: fault injection -- Y blocks are never announced)
    const int nu_strip = (int)(ldA / 16) - 4;
    const int first_row = 16 * (strip - 4 * rp_blocks);     // state row of this strip's first row (P H^T strips)
    const double* uT = Ypanel + RP + NP;                    // u^T = nu^T L^-T: row RP + NP of the second buffer
    const int ncols = is_s ? b + 1 : nblk;                  // column blocks held
    // steps k = 0 .. nsteps - 1, each followed by updates except the last of a P H^T strip.  S row block b stops after
    // step b-2: its step b-1 -- the panel block L(b,b-1) and the update of tile (b,b) -- is the chain's own prologue of
    // diagonal block b, and the chain publishes that panel block itself (flag xrow_ready).
    const int nsteps = is_s ? b - 1 : nblk;
    const int nupd = is_s ? b - 1 : nblk - 1;
    const int t = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(t >> 6), g = wv >> 2, w = wv & 3;
    const int l = t & 63, ln = l & 15, lq = l >> 4;
    double* Sk = lds;                  // [64][16]: the strip's column block k before the solve (B operand of the X product)
    double* Xs = lds + 64 * 16;        // [64][16]: X, the solved column block (B operand of the updates)
    int* abort = reinterpret_cast<int*>(lds + 2 * 64 * 16);
    double* xscratch = lds + 2 * 64 * 16 + 2;              // 32 * 16 + 16 doubles for xacc_finish
    XAcc xa; xa.part = 0.0; xa.row = t & 15; xa.pair = t >> 4;
    bool alive = true;
    if (t == 0) *abort = 0;            // (the first wait has a barrier before anybody reads it)
    if (!is_s && !single) {
        // First of all, together with the other P H^T / nu strips: the tiles the chain starts with, (0,0) and for its second
        // diagonal block (1,0), (1,1) -- 3 x 4096 entries of S over ~NP/16 workgroups, one entry per thread, so that the chain
        // waits for one gather + one hand-over instead of for a strip that gathers a whole 16 x 64 piece through its CU's
        // address unit (~6 us more on the first block of every sweep).
        const int nlow = NP / 16 + 1, p = strip - 4 * rp_blocks < nlow ? strip - 4 * rp_blocks : nlow - 1;   // (the nu strip is the last)
        const int E = (nblk >= 2 ? 3 : 1) * 4096;
        for (int e = (int)((long)p * E / nlow) + t; e < (int)((long)(p + 1) * E / nlow); e += CD_THREADS) {
            const int tile = e >> 12, a = (e & 63) + (tile >= 1 ? 64 : 0), c = ((e >> 6) & 63) + (tile == 2 ? 64 : 0);
            double Hf[13];
            int fo = 0, fw = 0;
            if (a < r_total) {
                const int fa = src.list[a >> 1];
                fo = src.off[fa]; fw = (src.type[fa] == 0) ? 13 : 10;
#pragma unroll
                for (int k = 0; k < 13; ++k) Hf[k] = src.H13[26L * fa + 13 * (a & 1) + k];
            } else {
#pragma unroll
                for (int k = 0; k < 13; ++k) Hf[k] = 0.0;
            }
            st_coh(A + a + (long)c * ldA, sys_S(src, A, ldA, NP, RP, r_total, a, c, Hf, fo, fw));
        }
#if defined(SW_TILES01_ONE)
        sw_post_add(&fl->tiles01);
#else
        sw_post_add(&fl->tiles01s[32 * (strip & 3)]);
#endif
    }
    // acc[jj][reg] of lane (ln, lq) = strip(ln, 64 (2 jj + g) + 16 w + lq + 4 reg); the strip assembles its rows of the
    // stacked system itself (there is no prepare_system pass in front of this kernel)
    double* base = A + 16L * strip + ln + (64L * g + 16L * w + lq) * ldA;
    d4 acc[NH];
    {
        const int row = 16 * strip + ln;                    // row of the stacked matrix
        double Hf[13];
        int fo = 0, fw = 0;
        if (is_s && row < r_total) {                        // S row a = row: its Jacobian row, fixed for the lane
            const int fa = src.list[row >> 1];
            fo = src.off[fa]; fw = (src.type[fa] == 0) ? 13 : 10;
#pragma unroll
            for (int k = 0; k < 13; ++k) Hf[k] = src.H13[26L * fa + 13 * (row & 1) + k];
        } else {
#pragma unroll
            for (int k = 0; k < 13; ++k) Hf[k] = 0.0;
        }
#if !defined(SW_GATHER_S)
        if (is_s) {
            // S rows (round 4): the entries S(a,c) = [a == c] + H_a (P H^T)_c need, of column c of P H^T, the seven pose rows and
            // the six (three) rows of feature a -- thirteen scattered 8-byte loads per entry and lane when gathered directly:
            // 312 per lane for a strip of the last row block, whose assembly then took 29 us and held back panel 0 of EVERY
            // strip (the step-0 update waits for all S strips: the hand-over of row block 3 reached the chain workgroup 3 us
            // after it needed it).  Now the four waves of a group stage, per column block, the 55 rows of P H^T this strip's
            // eight features can ask for (lane = row, sixteen columns per wave and block) in LDS, double-buffered, and the
            // dots read from there.  Same products in the same order: the same bits.  (-DSW_GATHER_S: the direct gather.)
            constexpr int WLD = 65;                             // (odd: the eight features' rows of a column on different banks)
            constexpr int WST = 55 * WLD;
            double* Wst = lds + 4096 + g * (2 * WST);           // two stages per group: block jj+1 is fetched under the dots of block jj
            // lane l < 55 of every wave fetches row `myrow` of P H^T: pose rows 0..6, then six (three) rows per feature of the
            // strip -- one load instruction per COLUMN (a column of P H^T is contiguous in rows: ~10 cache lines per instruction;
            // with a lane per column every load touched 64 lines and the vector L1 of the compute unit, one line per clock,
            // was what the assembly waited for)
            int myrow = -1;
            if (l < 7) myrow = l;
            else if (l < 55) {
                const int fi = (l - 7) / 6, i = (l - 7) - 6 * fi, a = 16 * strip + 2 * fi;
                if (a < r_total) { const int fa = src.list[a >> 1]; if (i < ((src.type[fa] == 0) ? 6 : 3)) myrow = src.off[fa] + i; }
            }
            const int sl = 7 + 6 * (ln >> 1);                   // first feature row of this lane's S row in the stage
            double wr[16];                                      // columns 16 w .. 16 w + 15 of the block, row myrow
            // LI pass: column c of P H^T is column 2 rank_of[list[c / 2]] + (c & 1) of the matched-feature matrix -- two dependent
            // loads per column: looked up once, lane i < 16 for column 16 w + i of every block of this group, broadcast by readlane
            int colv[NH];
#pragma unroll
            for (int jj = 0; jj < NH; ++jj) {
                const int c = 64 * (2 * jj + g) + 16 * w + (l & 15);
                colv[jj] = (src.Wsrc && 2 * jj + g < ncols && c < r_total) ? 2 * src.rank_of[src.list[c >> 1]] + (c & 1) : 0;
            }
            auto fetch = [&](auto JJ) {
                constexpr int jj = decltype(JJ)::value;
#if !defined(ABL_NO_ASSEMBLY)
                if constexpr (jj < NH) {
                    const int j = 2 * jj + g;
                    if (j < ncols) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int c = 64 * j + 16 * w + i;          // (wave-uniform)
                            const double* wc = src.Wsrc ? src.Wsrc + (long)__builtin_amdgcn_readlane(colv[jj], i) * NP : A + RP + (long)c * ldA;
                            wr[i] = (myrow >= 0 && c < r_total) ? wc[myrow] : 0.0;
                        }
                    }
                }
#endif
            };
            fetch(std::integral_constant<int, 0>{});
            static_for<0, NH>([&](auto JJ) {
                constexpr int jj = decltype(JJ)::value;
                const int j = 2 * jj + g;
                const bool active = j < ncols;
                double* Wb = Wst + (jj & 1) * WST;
                acc[jj] = (d4){0.0, 0.0, 0.0, 0.0};
                if (active && l < 55) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) Wb[l * WLD + 16 * w + i] = wr[i];
                }
                __syncthreads();                                // stage jj is complete (and stage jj-1 has been consumed by everybody)
                fetch(std::integral_constant<int, jj + 1>{});   // in flight under the dots below
                if (active) {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int cl = 16 * w + lq + 4 * reg, c = 64 * j + cl;
                        double v = (row == c) ? 1.0 : 0.0;
#if !defined(ABL_NO_ASSEMBLY)
                        if (row < r_total && c < r_total) {
                            double sacc = 0;
#pragma unroll
                            for (int k = 0; k < 13; ++k) if (k < fw) sacc += Hf[k] * Wb[(k < 7 ? k : sl + (k - 7)) * WLD + cl];
                            v += sacc;
                        }
#endif
                        acc[jj][reg] = v;
                    }
                }
            });
            __syncthreads();                                    // (the stages are free: the step loop reuses this LDS)
        } else
#endif
        static_for<0, NH>([&](auto JJ) {
            constexpr int jj = decltype(JJ)::value;
            acc[jj] = (d4){0.0, 0.0, 0.0, 0.0};
            if (2 * jj + g < ncols) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int c = 64 * (2 * jj + g) + 16 * w + lq + 4 * reg;
                    double v = 0.0;
                    if (is_s) v = sys_S(src, A, ldA, NP, RP, r_total, row, c, Hf, fo, fw);
                    else if (c < r_total) {
                        if (is_nu) { if (ln == 0) { const int f = src.list[c >> 1]; v = src.z[2 * f + (c & 1)] - src.h[2 * f + (c & 1)]; } }
                        else v = sys_wcol(src, A, ldA, NP, RP, c)[row - RP];
                    }
                    acc[jj][reg] = v;
                }
            }
        });
    }
    if (single) {
        // A system of one diagonal block (r <= 64: every LI update of the reference-faithful mode, where the consensus set
        // is the hypothesis' own feature): nothing crosses workgroups.  Every P H^T / nu strip assembles S itself (r^2
        // entries), factors it in its own LDS with the eight-wave pipeline of the chain workgroup -- the pivot chain stops
        // after r pivots -- and solves its rows: no hand-over hop (~3 us each) in either direction and no chain workgroup.
        // Same arithmetic as the shared route (lower triangle of S authoritative, same factor code): bit-identical Y.
        if (r_total <= 4 && !tiny_off) {
            // ... and a system of at most four rows (the rank-2 LI update of the reference-faithful mode) needs no pipeline
            // at all: the 4 x 4 lower triangle of S goes round wave 0 in shuffles, every lane factors it in registers
            // (sqrt / division at full precision) and solves its own row; the other columns of Y are zero.
            if (g == 0) {
                if (w == 0) {
                    const int e = l & 15, ea = e >> 2, ec = e & 3;
                    const int a = ea >= ec ? ea : ec, c = ea >= ec ? ec : ea;      // lower triangle authoritative
                    double v = (a == c) ? 1.0 : 0.0;
                    if (a < r_total) {
                        const int fa = src.list[a >> 1];
                        const int fo = src.off[fa], fw = (src.type[fa] == 0) ? 13 : 10;
                        const double* hf = src.H13 + 26L * fa + 13 * (a & 1);
                        const double* wc = sys_wcol(src, A, ldA, NP, RP, c);
                        double sacc = 0;
#pragma unroll
                        for (int k = 0; k < 13; ++k) if (k < fw) sacc += hf[k] * wc[col_index(fo, k)];
                        v += sacc;
                    }
                    const double s00 = __shfl(v, 0), s10 = __shfl(v, 4), s11 = __shfl(v, 5), s20 = __shfl(v, 8), s21 = __shfl(v, 9),
                                 s22 = __shfl(v, 10), s30 = __shfl(v, 12), s31 = __shfl(v, 13), s32 = __shfl(v, 14), s33 = __shfl(v, 15);
                    const double l00 = sqrt(s00), l10 = s10 / l00, l20 = s20 / l00, l30 = s30 / l00;
                    const double d1 = s11 - l10 * l10, l11 = sqrt(d1), l21 = (s21 - l20 * l10) / l11, l31 = (s31 - l30 * l10) / l11;
                    const double d2 = s22 - l20 * l20 - l21 * l21, l22 = sqrt(d2), l32 = (s32 - l30 * l20 - l31 * l21) / l22;
                    const double d3 = s33 - l30 * l30 - l31 * l31 - l32 * l32, l33 = sqrt(d3);
                    if (l == 0 && !(s00 > 0.0 && d1 > 0.0 && d2 > 0.0 && d3 > 0.0 && l33 < 1.0e300)) atomicMin(status, -6);   // RSLAM_ERR_NOT_SPD
                    // row ln of the strip: its four entries sit in lanes ln, ln + 16, ln + 32, ln + 48 (register 0)
                    const double q0 = __shfl(acc[0][0], ln), q1 = __shfl(acc[0][0], ln + 16), q2 = __shfl(acc[0][0], ln + 32), q3 = __shfl(acc[0][0], ln + 48);
                    const double x0 = q0 / l00, x1 = (q1 - x0 * l10) / l11, x2 = (q2 - x0 * l20 - x1 * l21) / l22,
                                 x3 = (q3 - x0 * l30 - x1 * l31 - x2 * l32) / l33;
                    acc[0][0] = lq == 0 ? x0 : lq == 1 ? x1 : lq == 2 ? x2 : x3;
                    // (low-innovation pass: Y1 kept aside for the deferred covariance, SEL_LI_DEFER)
                    if (xrows && wk.token == 1 && wk.Y1) wk.Y1[first_row + ln + (long)lq * wk.ldy1] = acc[0][0];
                    if (xrows) {
                        // K9 here: u^T = nu^T L^-T by the same substitution on the innovation row (every strip for itself: no hop)
                        double nu4[4];
#pragma unroll
                        for (int c4 = 0; c4 < 4; ++c4) {
                            nu4[c4] = 0.0;
                            if (c4 < r_total) { const int f = src.list[c4 >> 1]; nu4[c4] = src.z[2 * f + (c4 & 1)] - src.h[2 * f + (c4 & 1)]; }
                        }
                        const double u0 = nu4[0] / l00, u1 = (nu4[1] - u0 * l10) / l11, u2 = (nu4[2] - u0 * l20 - u1 * l21) / l22,
                                     u3 = (nu4[3] - u0 * l30 - u1 * l31 - u2 * l32) / l33;
                        const double xn = wk.x_in[first_row + ln] + (((x0 * u0 + x1 * u1) + x2 * u2) + x3 * u3);
                        if (first_row != 0) {
                            if (lq == 0) wk.x_out[first_row + ln] = xn;
                        } else {
                            double q[4] = { __shfl(xn, 3), __shfl(xn, 4), __shfl(xn, 5), __shfl(xn, 6) };
                            if (lq == 0 && (ln < 3 || ln > 6)) wk.x_out[ln] = xn;
                            if (l == 0) {
                                quat_jnorm(q, wk.compat, wk.T);
                                for (int i = 0; i < 4; ++i) wk.x_out[3 + i] = q[i];
                                wait_stores();
                                __hip_atomic_store(wk.xu_flag, wk.token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
                    }
                }
                double* dst = Ypanel + 16L * strip + ln + (16L * w + lq) * ldA;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const double v = (w == 0 && reg == 0) ? acc[0][0] : 0.0;
                    if (fused) st_coh(dst + (4L * reg) * ldA, v); else dst[(4L * reg) * ldA] = v;
                }
            }
            if (publish) {
                wait_stores();
                __syncthreads();
                if (t == 0) __hip_atomic_store(&fl->y_flag[strip], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        CdShared& sh = *reinterpret_cast<CdShared*>(lds);
        CdPre pre;
#pragma unroll
        for (int o = 0; o < CD_TT; ++o) pre.tacc[o] = (d4){0.0, 0.0, 0.0, 0.0};
        if (wv >= 2 && wv < 2 + CD_TW) {
            const int lr = l >> 4, lc = l & 15;
#pragma unroll
            for (int o = 0; o < CD_TT; ++o) {
                const int idx = (wv - 2) + CD_TW * o;
                if (idx >= 10) continue;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int row = 16 * cd_tr(idx) + lr + 4 * reg, col = 16 * cd_tc(idx) + lc;
                    const int a = row >= col ? row : col, c = row >= col ? col : row;
                    double v = (a == c) ? 1.0 : 0.0;
                    if (a < r_total) {
                        const int fa = src.list[a >> 1];
                        const int fo = src.off[fa], fw = (src.type[fa] == 0) ? 13 : 10;
                        const double* hf = src.H13 + 26L * fa + 13 * (a & 1);
                        const double* wc = sys_wcol(src, A, ldA, NP, RP, c);
                        double sacc = 0;
#pragma unroll
                        for (int k = 0; k < 13; ++k) if (k < fw) sacc += hf[k] * wc[col_index(fo, k)];
                        v += sacc;
                    }
                    pre.tacc[o][reg] = v;
                }
            }
        }
        cd_factor_block<true>(sh, nullptr, ldA, 0, sel, slot_k, nullptr, status, 0, pre);
        double* Sk1 = lds + CDP_OFF_AOP;                     // behind CdShared
        double* Xs1 = Sk1 + 64 * 16;                         // X in the layout of the general route ([64][16]), for the x update
        abort = reinterpret_cast<int*>(Xs1 + 64 * 16 + 32 * 16 + 16);   // (the first location was inside CdShared)
        if (t == 0) *abort = 0;
        if (g == 0) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) Sk1[(16 * w + lq + 4 * reg) * 16 + ln] = acc[0][reg];
        }
        __syncthreads();
        if (g == 0) {
            d4 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (q < 4 * (w + 1)) x = __builtin_amdgcn_mfma_f64_16x16x4f64(sh.Mf[(4 * q + lq) * CD_LD + 16 * w + ln], Sk1[(4 * q + lq) * 16 + ln], x, 0, 0, 0);
            double* dst = Ypanel + 16L * strip + ln + (16L * w + lq) * ldA;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                if (fused) st_coh(dst + (4L * reg) * ldA, x[reg]); else dst[(4L * reg) * ldA] = x[reg];
                Xs1[(16 * w + lq + 4 * reg) * 16 + ln] = x[reg];
            }
        }
        if (fused) {
            wait_stores();
            __syncthreads();
            if (publish && t == 0) __hip_atomic_store(&fl->y_flag[strip], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (xrows) {
                // one hop for u^T (this route is rare: a corrected-arithmetic LI set of at most 32 features)
                if (sw_wait(&fl->y_flag[nu_strip], 1, status, abort, 8)) {
                    xacc_add(xa, Xs1, uT, ldA, 0);
                    xacc_finish(xa, first_row, wk, Xs1 + 64 * 16);
                }
            }
        }
        return;
    }
    static_for<0, NJ>([&](auto K) {
        constexpr int k = decltype(K)::value;
        const bool mine = (g == (k & 1));                    // this group holds column block k
        // (round 6) What the solve of step k needs of this strip itself is ready BEFORE L^-1(k) is: the column block (final since
        // the update of step k-1: staged as the MFMA operand now) and, for the x update, the strip's X of step k-1 (still in Xs).
        // Both happen in front of the wait, whose barrier orders them: behind the flag only the loads of L^-1 and of u^T (consumed
        // behind the publication) and the MFMA chain are left -- the strips' step is a link of the hand-over loop that holds the
        // chain workgroup's block period together with that workgroup's own sequence (see cdp_t_wave)
        double xu0 = 0.0, xu1 = 0.0;
        if (alive && k < nsteps) {
            if (mine) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) Sk[(16 * w + lq + 4 * reg) * 16 + ln] = acc[k >> 1][reg];
            }
            if (xrows && k > 0) { xu0 = Xs[(2 * xa.pair) * 16 + xa.row]; xu1 = Xs[(2 * xa.pair + 1) * 16 + xa.row]; }
        }
        if (alive && k < nsteps) {
            sw_stamp(dbg, who, k, 0);
            // (P H^T strips of a fused launch also make sure u^T of the previous block is out: it has been for ~10 us)
            if (xrows && k > 0) alive = sw_wait2(&fl->linv_ready, k + 1, &fl->y_flag[nu_strip], k, status, abort, 1);
            else alive = sw_wait(&fl->linv_ready, k + 1, status, abort, 1);
            sw_stamp(dbg, who, k, 1);
        }
        if (alive && k < nsteps) {
            // (K9's share of step k-1, xacc_add in two halves: its u^T entries are requested first and consumed behind the publication)
            double uu0 = 0.0, uu1 = 0.0;
            if (xrows && k > 0) { uu0 = ld_coh(uT + (64L * (k - 1) + 2 * xa.pair) * ldA); uu1 = ld_coh(uT + (64L * (k - 1) + 2 * xa.pair + 1) * ldA); }
            const double* Lk = Linv + 64L * 64 * k;
            // L^-1 rows 16 w .. 16 w + 15 are zero right of column 16 w + 15: 4 (w + 1) MFMA steps
            double la[16];
            if (mine) {
#pragma unroll
                for (int q = 0; q < 16; ++q) la[q] = (q < 4 * (w + 1)) ? ld_coh(Lk + (16 * w + ln) + 64 * (4 * q + lq)) : 0.0;
            }
            if (mine) {
                d4 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    if (q < 4 * (w + 1)) x = __builtin_amdgcn_mfma_f64_16x16x4f64(la[q], Sk[(4 * q + lq) * 16 + ln], x, 0, 0, 0);
                // x[reg] = X(ln, 16 w + lq + 4 reg): final.  S rows: a panel block for the other strips; P H^T / nu rows: Y
                // and u^T for the next kernel -- or, in a fused launch, for the tile workers of this one (coherent stores +
                // flag).  Both go to the second buffer: in the HI pass the rows of A below S are the
                // P H^T that other strips may still be reading while they assemble their rows of S.
                double* dst = Ypanel + 16L * strip + ln + (64L * k + 16 * w + lq) * ldA;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    if (is_s || fused) st_coh(dst + (4L * reg) * ldA, x[reg]); else dst[(4L * reg) * ldA] = x[reg];
                    Xs[(16 * w + lq + 4 * reg) * 16 + ln] = x[reg];
                }
            }
            if (is_s) {
                wait_stores();
                __syncthreads();
                if (t == 0) {
                    __hip_atomic_fetch_add(&fl->row_cnt[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(&fl->panel_cnt[k], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else if (fused) {
                wait_stores();
                __syncthreads();
                if (k == 0 && (exp_mask & 1024) && !is_nu && t == 0) {
                    // fault injection: a hand-over that is merely LATE -- the first Y block is announced 2 ms of wall clock after
                    // it was stored, twice the time bound of the waits on it (tests/test_gpu_fused.py: with the waiters
                    // throttled below SW_WAIT_MIN_POLLS the frame must come out right WITHOUT a re-run)
                    const unsigned long long t_hold = wall_clock64();
                    while (wall_clock64() - t_hold < 2ull * SW_WAIT_TICKS) __builtin_amdgcn_s_sleep(64);
                }
                if (publish && t == 0) __hip_atomic_store(&fl->y_flag[strip], k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else __syncthreads();
            if (xrows && k > 0) xa.part = fma(xu1, uu1, fma(xu0, uu0, xa.part));
            sw_stamp(dbg, who, k, 2);
            if (k < nupd) {
                double xb[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) xb[q] = Xs[(4 * q + lq) * 16 + ln];
                // panel blocks L(j,k): j = k+1 from the chain, j >= k+2 from the strips of S row block j.  The step that
                // ends with the hand-over to the chain (S rows, k = b-2) only needs j = b-1 = k+1 and j = b, its own
                // row block: it waits for its three siblings, not for every S strip
                if (alive) {
                    if (is_s && k == b - 2) alive = sw_wait2(&fl->xrow_ready, k + 1, &fl->row_cnt[b], 4 * (k + 1), status, abort, 3);
                    else if (k + 2 < nblk) alive = sw_wait2(&fl->xrow_ready, k + 1, &fl->panel_cnt[k], 4 * (nblk - 2 - k), status, abort, 4);
                    else alive = sw_wait(&fl->xrow_ready, k + 1, status, abort, 2);
                }
                sw_stamp(dbg, who, k, 3);
                // column blocks j = 2 jj + g, k < j < ncols: a contiguous jj range whose lower end is known at compile
                // time per group; the operand rows of tile jj + 1 are in flight under the MFMAs of tile jj
                auto updates = [&](auto G) {
                    constexpr int gg = decltype(G)::value;
                    constexpr int lo = (k + 2 - gg) / 2;             // smallest jj with 2 jj + gg > k
                    const int hi = (ncols - gg + 1) / 2;             // jj < hi  <=>  2 jj + gg < ncols
                    const double* Lp = Ypanel + 64L * gg + (16 * w + ln) + (64L * k + lq) * ldA;   // + 128 jj: rows 16 w.. of panel block (j,k)
                    double a[2][16];
                    if (alive && lo < hi && lo < NH) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) a[0][q] = ld_coh(Lp + 128L * lo + (4L * q) * ldA);
                    }
                    static_for<lo, NH>([&](auto JJ) {
                        constexpr int jj = decltype(JJ)::value;
                        constexpr int cur = (jj - lo) & 1;
                        if (alive && jj < hi) {
                            if (jj + 1 < hi && jj + 1 < NH) {
#pragma unroll
                                for (int q = 0; q < 16; ++q) a[cur ^ 1][q] = ld_coh(Lp + 128L * (jj + 1) + (4L * q) * ldA);
                            }
#pragma unroll
                            for (int q = 0; q < 16; ++q) acc[jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[cur][q], xb[q], acc[jj], 0, 0, 0);
                        }
                    });
                };
                if (g == 0) updates(std::integral_constant<int, 0>{}); else updates(std::integral_constant<int, 1>{});
                sw_stamp(dbg, who, k, 4);
                if (alive && is_s && k == b - 2) {
                    // the chain factors block b after block b-1: hand tiles (b,b-1), (b,b) over, updates 0 .. b-2 applied
                    static_for<(k + 1) / 2, (k / 2 + 2 < NH ? k / 2 + 2 : NH)>([&](auto JJ) {
                        constexpr int jj = decltype(JJ)::value;
                        const int j = 2 * jj + g;
                        if (j == k + 1 || j == k + 2) {
#pragma unroll
                            for (int reg = 0; reg < 4; ++reg) st_coh(base + (128L * jj + 4 * reg) * ldA, acc[jj][reg]);
                        }
                    });
                    sw_post_add(&fl->row_ready[b]);
                    sw_stamp(dbg, who, k, 5);
                }
            }
        }
    });
    if (xrows && alive) {
        // the last block's share of x + Y u (its X is still in Xs), then the strip's rows of x_k_k
        if (sw_wait(&fl->y_flag[nu_strip], nblk, status, abort, 8)) {
            xacc_add(xa, Xs, uT, ldA, nblk - 1);
            xacc_finish(xa, first_row, wk, xscratch);
        }
    }
}

// ---- the chain workgroup of the persistent sweep ------------------------------------------------------------------
// Per diagonal block k the serial path is: [chain of block k-1 ends: L^-1(k-1) complete in LDS] -> X = A(k,k-1) L^-T(k-1)
// (8 waves, MFMA) -> the update T(k,k) -= X X^T of the FIRST tile column only -> strip of the first four pivots -> chain.
// Everything else is off that path:
//   * the inputs of block k -- tiles (k,k-1), (k,k) as the strips of row block k handed them over -- are fetched into
//     LDS by the T waves DURING the chain of block k-1: a flag load, and two iterations later the data loads, are
//     issued between pivot steps and consumed iterations later, so that no wave on the pipeline ever waits for memory
//     (if the hand-over comes too late, the fetch happens at the top of the block and its latency is exposed);
//   * the tile columns 1..3 of T -= X X^T are applied by the T waves during the first pivot steps (updates commute;
//     tile column c is first read at pivot 16 c);
//   * L^-1(k-1) is written to global memory right after the chain ends and its flag goes out after the X product;
//   * L itself is not kept: nobody reads it.
struct CdpNext {                 // the inputs of the next block, as the strips of its row block hand them over
    const double* a_base;        // A(k+1,k): row 0, column 0 of the block
    const double* tile;          // tile (k+1,k+1)
    long ldA;
    const int32_t* flag;         // hand-over counter of those tiles ...
    int need;                    // ... and the count that says they are all there
    int shards;                  // the counter is the sum of this many words, 32 ints apart (1, or 4: SweepFlags::tiles01s)
    double* Aop; double* Tpre;
    bool need_a;                 // A(k+1,k) exists (false at the sweep's first block: only its diagonal tile is fetched)
};
// the hand-over counter of the next block's inputs (all its shards requested at once; the sum when it is used)
__device__ __forceinline__ int cdp_count(const CdpNext& nx)
{
    int v = ld_flag(nx.flag);
    if (nx.shards == 4) v += ld_flag(nx.flag + 32) + ld_flag(nx.flag + 64) + ld_flag(nx.flag + 96);
    return v;
}

// A(k+1,k) goes from global memory straight into LDS (global_load_lds_dwordx4, sc1: the strips' stores are write-through): no
// registers, no ds_write -- the register-staged fetch cost the fetching wave 52 live doubles, and spilled.  Aop is the
// LDS-DMA image of tile_gemm.h: one transfer = k-rows m and m + 2 (64 rows each), segments TD_SEG doubles apart; the X
// product reads its A fragments (lane = (k mod 4, row)) from it without bank conflicts.
static_assert(16 * TD_STEP == CDP_AOP_DOUBLES, "A(k,k-1) as a 64-column LDS-DMA image");
__device__ __forceinline__ int cdp_aop_index(int m, int row) { return (m >> 2) * TD_STEP + (m & 1) * TD_SEG + ((m >> 1) & 1) * 64 + row; }

// transfers 2 part, 2 part + 1 of T wave B's eight of the 32 (part 0..3)
template <int B>
__device__ __forceinline__ void cdp_dma_issue(const CdpNext& nx, int part, unsigned lane_off)
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int sg = 8 * B + 2 * part + i;
        const long ku = 4 * (sg >> 1) + (sg & 1);             // wave-uniform: k-rows ku (lanes 0..31) and ku + 2 (lanes 32..63)
        const char* ga = reinterpret_cast<const char*>(nx.a_base + ku * nx.ldA) + lane_off;
#if defined(CDP_DMA_BUILTIN)   // measurement: the transfer as the compiler's builtin
        __builtin_amdgcn_global_load_lds((tg_glb_void*)ga, (tg_lds_void*)(nx.Aop + sg * TD_SEG), 16, 0, 16);
#else
        // Written out, so that the compiler does not know it is a transfer into LDS: it puts s_waitcnt vmcnt(0) in front of
        // every LDS read that may alias a transfer it knows of -- i.e. in front of the flag and operand reads of the very next
        // pivot step, a wait for global memory on a wave of the pivot pipeline, in each of the four steps the fetch is spread
        // over (the blocks that fetched inside the loop had pivot loops 1.2 us longer than those that did not).  Nothing reads
        // Aop before the next block's top, and cdp_finish waits for vmcnt(0) itself in front of the barrier there.
        const unsigned lds_addr = (unsigned)(size_t)(tg_lds_void*)(nx.Aop + sg * TD_SEG);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc1"
                     :: "v"(ga), "s"(lds_addr) : "memory", "m0");
#endif
    }
}

// The tile (k+1,k+1) lands in the accumulator layout of the T waves, [(tile, register)][lane]: eight-byte elements, mirrored
// above the diagonal -- through registers.  Chunk Q (0..3) of share B (0..3): 3/3/2/2 (tile, register) pairs.
constexpr __device__ int cdp_t0(int q) { return q < 2 ? 3 * q : 6 + 2 * (q - 2); }      // first pair of chunk q: 0, 3, 6, 8 (, 10)

template <int B, int Q>
__device__ __forceinline__ void cdp_issue(const CdpNext& nx, double (&pf)[10])
{
    const int l = threadIdx.x & 63, lr = l >> 4, lc = l & 15;     // (the lane offsets are loop invariant: the compiler keeps them)
#pragma unroll
    for (int i = cdp_t0(Q); i < cdp_t0(Q + 1); ++i) {
        const int p = B + 4 * i, idx = p >> 2, reg = p & 3;
        const int row = 16 * cd_tr(idx) + lr + 4 * reg, col = 16 * cd_tc(idx) + lc;
        pf[i] = ld_coh(nx.tile + ((row >= col) ? row + (long)col * nx.ldA : col + (long)row * nx.ldA));   // lower triangle is authoritative
    }
}

template <int B, int Q>
__device__ __forceinline__ void cdp_store(const CdpNext& nx, const double (&pf)[10])
{
    const int l = threadIdx.x & 63;
#pragma unroll
    for (int i = cdp_t0(Q); i < cdp_t0(Q + 1); ++i) nx.Tpre[(B + 4 * i) * 64 + l] = pf[i];
}

// q is a constant once the pivot-step loop is unrolled: the switches fold
template <int B>
__device__ __forceinline__ void cdp_issue_q(int q, const CdpNext& nx, double (&pf)[10])
{
    switch (q) { case 0: cdp_issue<B, 0>(nx, pf); break; case 1: cdp_issue<B, 1>(nx, pf); break;
                 case 2: cdp_issue<B, 2>(nx, pf); break; default: cdp_issue<B, 3>(nx, pf); break; }
}
template <int B>
__device__ __forceinline__ void cdp_store_q(int q, const CdpNext& nx, const double (&pf)[10])
{
    switch (q) { case 0: cdp_store<B, 0>(nx, pf); break; case 1: cdp_store<B, 1>(nx, pf); break;
                 case 2: cdp_store<B, 2>(nx, pf); break; default: cdp_store<B, 3>(nx, pf); break; }
}

// Top of a block (T wave B): make this wave's share of the block's inputs LDS-resident.  st: 4 = the transfers were issued
// during the last steps of the previous chain (the tile's values are in pf), anything else: poll and fetch now (kernel
// start, or a hand-over that came after the look: the latency is exposed).  The caller's barrier follows; the DMA
// transfers of this wave are waited for here.
template <int B>
__device__ __forceinline__ void cdp_finish(const CdpNext& nx, CdShared& sh, double (&pf)[10], int st, unsigned lane_off)
{
    if (st != 3) {
        if (st != 4) {
            if (nx.flag) {
                SwDeadline dl;
                while (cdp_count(nx) < nx.need) {
                    if (dl.expired()) { sh.timeout = 5; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            // (the sweep's first block has no panel row: nothing would read the transfers, and their landing would be waited for below)
            if (nx.need_a)
            { cdp_dma_issue<B>(nx, 0, lane_off); cdp_dma_issue<B>(nx, 1, lane_off); cdp_dma_issue<B>(nx, 2, lane_off); cdp_dma_issue<B>(nx, 3, lane_off); }
            cdp_issue<B, 0>(nx, pf); cdp_issue<B, 1>(nx, pf); cdp_issue<B, 2>(nx, pf); cdp_issue<B, 3>(nx, pf);
        }
        cdp_store<B, 0>(nx, pf); cdp_store<B, 1>(nx, pf); cdp_store<B, 2>(nx, pf); cdp_store<B, 3>(nx, pf);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the LDS-DMA transfers of this wave have landed
}

// Pend: LDS images (accumulator layout [reg][lane]) of the products X(tr,:) X(tc,:)^T of trailing tiles 2, 4, 7, 5 (slots 0..3),
// formed by the four waves that are not T waves while the T waves apply the first tile column (cdp_role), ready at barrier
// (C).  Tiles 8 and 9 (slots >= 4 mark them) are formed by their owners, T waves 0 and 1, a few MFMAs per step (steps 1..6).
__device__ constexpr int cdp_pend_slot(int idx) { return idx == 2 ? 0 : idx == 4 ? 1 : idx == 7 ? 2 : idx == 5 ? 3 : idx == 8 ? 4 : idx == 9 ? 5 : -1; }

// T wave B of the persistent chain (round 4: see DESIGN.md section 4, "what bounds the pivot chain").  Returns the state of the
// fetch of the next block's inputs (this wave's share: eight LDS-DMA transfers of A(k+1,k) and ten values of tile (k+1,k+1)):
// 4 = issued during steps 12..15, the tile's values are in pf and are stored at the top of the next block; 0 = the hand-over
// had not come by step 9: the next block fetches everything itself (cdp_finish).
template <int B>
__device__ __forceinline__ int cdp_t_wave(CdShared& sh, int n_piv4, bool pending, const double* Xb, const double* Tpre,
                                          const double* Pend, const CdpNext& nx, bool want_next, double (&pf)[10], unsigned lane_off,
                                          unsigned long long* stamp, int blk = 0)
{
    const int l = threadIdx.x & 63, lr = l >> 4, lc = l & 15;
    constexpr int NT = (10 - B + CD_TW - 1) / CD_TW;
    constexpr int EO = (cd_tc(B) == 0) ? 0 : 1;        // the tile of the first tile column (one per T wave)
    static_assert(cd_tc(B + CD_TW * EO) == 0, "every T wave owns one tile of tile column 0");
    d4 acc[NT];
#pragma unroll
    for (int o = 0; o < NT; ++o)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) acc[o][reg] = Tpre[((B + CD_TW * o) * 4 + reg) * 64 + l];
    auto pend = [&](auto O, int k0, int k1) {           // acc[O] -= X(tr,:) X(tc,:)^T over columns k0 .. k1-1
        constexpr int o = decltype(O)::value;
        constexpr int tr = cd_tr(B + CD_TW * o), tc = cd_tc(B + CD_TW * o);
#pragma unroll 4
        for (int kk = k0; kk < k1; kk += 4)
            acc[o] = __builtin_amdgcn_mfma_f64_16x16x4f64(Xb[(kk + lr) * CD_LD + 16 * tr + lc], -Xb[(kk + lr) * CD_LD + 16 * tc + lc], acc[o], 0, 0, 0);
    };
    if (pending) pend(std::integral_constant<int, EO>{}, 0, 64);
    // strip of pivot block 0
    if (lc < 4) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) sh.Tst[0][lc * 64 + 16 * cd_tr(B + CD_TW * EO) + lr + 4 * reg] = acc[EO][reg];
    }
    __syncthreads();                     // (C) strips, flags, the products of tiles 2, 4, 7, 5 are visible
    // T(k,k) -= X X^T outside the first tile column: the products were formed by the waves that are not T waves (cdp_role) --
    // a T wave's step is what paces the pivot chain, so nothing but the rank-4 updates and the strips stays here
    auto take = [&](auto O) {
        constexpr int o = decltype(O)::value;
        constexpr int slot = cdp_pend_slot(B + CD_TW * o);
        if constexpr (slot >= 0) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) acc[o][reg] -= Pend[(slot * 4 + reg) * 64 + l];
        }
    };
    const bool late_pending = pending && (cdp_pend_slot(B + CD_TW * (NT - 1)) >= 4);      // tile 8 / 9 (T waves 0 / 1): its product arrives later
    bool late_left = late_pending; (void)late_left;
    // The step loop is software-pipelined: the T waves are the slow side of the pivot pipeline (a -DCD_SPINS build: the
    // panel wave takes three to four looks per step before its strips are there, T wave 0 finds the panel flag up nine
    // times out of ten; -DCD_TIMELINE: flag poll, operand fetch, MFMA, strip extraction and post are four dependent LDS /
    // matrix-pipe latencies in a row, ~1000 cycles against ~750 of the panel wave's step).  So the flag and the operands of
    // step s+1 are requested right behind the post of step s -- the panel wave has usually finished step s by then -- and
    // land under this step's remaining MFMAs: a step then starts with its operands in registers.  LDS serves a wave's
    // requests in order: operands read behind a flag that says "posted" are the posted panel.
    bool have = false;                 // a[], b[] hold the operands of the coming step
    double a[NT], b[NT];
    int st = 0, fv = 0;
#define CDP_T_LOADS(vflag, Xsrc) do {                                                                                       \
        vflag = __hip_atomic_load(&sh.flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);                            \
        __atomic_signal_fence(__ATOMIC_SEQ_CST);                                                                           \
        _Pragma("unroll") for (int o = 0; o < NT; ++o) {                                                                     \
            a[o] = (Xsrc)[lr * 64 + 16 * cd_tr(B + CD_TW * o) + lc];                                                         \
            b[o] = (Xsrc)[lr * 64 + 16 * cd_tc(B + CD_TW * o) + lc];                                                         \
        }                                                                                                                  \
        __atomic_signal_fence(__ATOMIC_SEQ_CST);                                                                           \
    } while (0)
#pragma unroll 1
    for (int sb = 0; sb < 4; ++sb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s = 4 * sb + q;
            const int p0 = 16 * sb + 4 * q;
            if (s >= n_piv4) return st;
            CD_STAMP(ts0);
            if (B == 0 && !CD_TL_M_ON) CD_TL(1, blk, s, 0);
            const int bb1 = (p0 + 4) >> 4;           // tile column the strip of block s+1 comes from
            int sp = 0;                              // polls of this step that found the panel flag not yet up
            if (s > 0) {
                if (!have) {
                    // panel s-1 (which also means the panel wave is done with strip buffer (s+1) & 1): flag and operands in one
                    // batch; a flag that is not up is polled alone, then the operands are read again
                    const double* Xp = sh.Xs[(q + 3) & 3];
                    int v;
                    CDP_T_LOADS(v, Xp);
                    if (__builtin_amdgcn_readfirstlane(v) < s) {
                        while (true) {
                            __builtin_amdgcn_s_sleep(1);
                            v = __hip_atomic_load(&sh.flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            ++sp;
                            if (__builtin_amdgcn_readfirstlane(v) >= s) break;
                            if (sp > CD_SPIN_LIMIT) { sh.timeout = 1; break; }
                        }
                        __atomic_signal_fence(__ATOMIC_SEQ_CST);
                        CDP_T_LOADS(v, Xp);
                    }
                }
                if (B == 0 && !CD_TL_M_ON) CD_TL(1, blk, s, 1);
                if (B == 0 && (threadIdx.x & 63) == 0) { CD_SPIN_ADD(2, sp + (have ? 0 : 1)); CD_SPIN_ADD(3, 1); }
                CD_STAMP(ts1);
                CD_ACC_T(3, ts0, ts1, 128);
                if (B == 0 && !CD_TL_M_ON) CD_TL(1, blk, s, 2);
                // (a tile whose 16 columns have all been eliminated -- tile column < sb -- is never read again: no update)
                // The chain waits for the strip of block s+1, which comes out of ONE tile of this wave (tile column bb1): that
                // tile's update goes first and the strip is published behind it; the other tiles' updates -- ~100 cycles of the
                // matrix pipe each -- follow behind the post, off the chain.
#if !defined(ABL_NO_T_MFMA)
#pragma unroll
                for (int o = 0; o < NT; ++o)
                    if (cd_tc(B + CD_TW * o) == bb1) acc[o] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[o], -b[o], acc[o], 0, 0, 0);   // T -= X X^T
#endif
            }
            // strip of block s+1 (updates <= s-1 applied)
            if (p0 + 4 < 64) {
                const int p1 = p0 + 4, bb = p1 >> 4, o1 = p1 & 15;
                double* To = sh.Tst[(q + 1) & 1];
#pragma unroll
                for (int o = 0; o < NT; ++o) {
                    if (cd_tc(B + CD_TW * o) == bb && lc >= o1 && lc < o1 + 4) {
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) To[(lc - o1) * 64 + 16 * cd_tr(B + CD_TW * o) + lr + 4 * reg] = acc[o][reg];
                    }
                }
            }
            cd_post(sh, 2 + B, s + 1);
            if (B == 0 && !CD_TL_M_ON) CD_TL(1, blk, s, 3);
            CD_STAMP(ts2);
            CD_ACC_T(15, ts0, ts2, 128);
            // ---- behind the published strip: work that is not on the chain
            // this step's remaining updates first (they still need a[], b[]) ...
            if (s > 0) {
#pragma unroll
                for (int o = 0; o < NT; ++o)
                    if (cd_tc(B + CD_TW * o) > bb1) {           // (tile columns behind bb1 are never read again: no update)
#if !defined(ABL_NO_T_MFMA)
                        acc[o] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[o], -b[o], acc[o], 0, 0, 0);
#else
                        acc[o][0] += a[o] * b[o];
#endif
                    }
            }
            // ... then the request for the next step's flag and operands (panel s, buffer q & 3): it lands under the MFMAs above
            // and under whatever follows in this step
            int vnext = -1;
            const bool more = (s + 1 < n_piv4) && (s + 1 < 16);
            // (only a wave that did not have to wait in this step asks ahead: one that keeps up with its producer would re-read
            //  the operands at the next step anyway, and the compute unit's LDS pipe is what every round trip queues behind)
            const bool ask = more CD_ASK_IF(sp == 0);
            if (ask) { const double* Xn = sh.Xs[q & 3]; CDP_T_LOADS(vnext, Xn); }
            if (pending && s == 0) {
                // (step 0 has no rank-4 update: the products the other waves left in LDS are taken here)
                static_for<0, NT>([&](auto O) { if (cdp_pend_slot(B + CD_TW * decltype(O)::value) >= 0 && cdp_pend_slot(B + CD_TW * decltype(O)::value) < 4) take(O); });
            }
#if defined(CD_LAZY_ON_T)
            if (late_pending && s >= 1 && s <= 6) {
                // tile 8 (T wave 0; first read at step 7) / tile 9 (T wave 1; step 11): its 16 MFMAs over steps 1..6 (3,3,3,3,2,2)
                const int k0 = s <= 4 ? 12 * (s - 1) : 48 + 8 * (s - 5), k1 = s <= 4 ? k0 + 12 : k0 + 8;
                pend(std::integral_constant<int, NT - 1>{}, k0, k1);
            }
#else
            if (late_left && s == 5) {
                // tile 8 (T wave 0; first read at step 7) / tile 9 (T wave 1; step 11): its product comes from M wave 0 / 1, which
                // forms it behind its first two (empty) steps.  It is taken at ONE fixed step -- where in the sequence of rank-4
                // updates the product is subtracted decides the last bit of the tile, and a take "as soon as the flag is up"
                // made the factor, and with it x and P, differ in the last place from run to run (scripts/repro_bits.py)
                int fvv = __hip_atomic_load(&sh.flags[10 + B], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                int spins = 0;
                while (__builtin_amdgcn_readfirstlane(fvv) == 0) {
                    if (++spins > CD_SPIN_LIMIT) { sh.timeout = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                    fvv = __hip_atomic_load(&sh.flags[10 + B], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
                take(std::integral_constant<int, NT - 1>{});
                late_left = false;
            }
#endif
            if (want_next) {
                // ONE look at the hand-over counter, at step 9, consumed at step 12: by then the strips of the next row block
                // have long handed over (their L^-1 arrives 0.3 us after the previous chain), and from step 12 on a T wave
                // owns at most one live tile -- the fetch (eight LDS-DMA transfers, ten loads) goes out during steps 12..15,
                // where it costs the chain nothing, and the tile's values are stored at the top of the next block.
                if (sb == 2 && q == 1) fv = cdp_count(nx);
                if (sb == 3) {
                    if (q == 0) {
                        asm volatile("" ::: "memory");           // (a real branch: a select on fv would put the wait for the load behind its issue)
                        if (fv >= nx.need) { st = 5; if (stamp && l == 0 && B == 0) stamp[7] = wall_clock64(); }
#if !defined(CDP_ONE_LOOK)
                        else fv = cdp_count(nx);                 // a second look, consumed at the last step
#endif
                    }
                    if (st == 5) { cdp_dma_issue<B>(nx, q, lane_off); cdp_issue_q<B>(q, nx, pf); if (q == 3) st = 4; }
#if !defined(CDP_ONE_LOOK)
                    if (q == 3 && st == 0) {
                        // the hand-over came between steps 9 and 12 (the strips are a little later since the chain got faster):
                        // the whole fetch in the last step -- its transfers land under the end barrier and the next block's top
                        asm volatile("" ::: "memory");
                        if (fv >= nx.need) {
                            if (stamp && l == 0 && B == 0) stamp[7] = wall_clock64();
                            cdp_dma_issue<B>(nx, 0, lane_off); cdp_dma_issue<B>(nx, 1, lane_off); cdp_dma_issue<B>(nx, 2, lane_off); cdp_dma_issue<B>(nx, 3, lane_off);
                            cdp_issue<B, 0>(nx, pf); cdp_issue<B, 1>(nx, pf); cdp_issue<B, 2>(nx, pf); cdp_issue<B, 3>(nx, pf);
                            st = 4;
                        }
                    }
#endif
                }
            }
            have = ask && (__builtin_amdgcn_readfirstlane(vnext) >= s + 1);
            CD_STAMP(ts3);
            CD_ACC_T(4, ts0, ts3, 128);
            CD_ACC_T(14, ts0, ts0 + 1, 128);
        }
    }
#undef CDP_T_LOADS
    if (st == 4) {
        // (round 6) the next block's inputs, as far as this wave fetched them, are LDS-resident BEFORE the end-of-chain barrier: the
        // tile's values go to Tpre (read at the top of a block only: every T wave took its accumulators from it long ago) and
        // the wave's LDS-DMA transfers are waited for here, under the inverse wave's last step, not behind the barrier.  One of
        // three changes that only pay TOGETHER (with the strips' staging in front of their wait for L^-1 and no transfer at
        // block 0): the block period is held by the chain workgroup's own sequence AND by the hand-over loop through the strips,
        // each within 0.3 us of binding -- a cut on one side alone comes back as exposed latency on the other (DESIGN.md
        // section 4, NOTEBOOK.md round 6: each of the three measured alone: nothing; together: C3 -2.2 us, five of five rounds)
        cdp_store<B, 0>(nx, pf); cdp_store<B, 1>(nx, pf); cdp_store<B, 2>(nx, pf); cdp_store<B, 3>(nx, pf);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st = 3;
    }
    return st;
}

// M wave C of the persistent chain: the running inverse (as cd_m_wave), software-pipelined like the T waves: with the T
// waves' step cut to what the chain needs, the M waves' five rank-4 updates per step (~145 cycles of a lone wave's matrix
// pipe each) behind a flag poll and an operand fetch made THEM the slow side (-DCD_SPINS: no failed poll at all).  The flag of
// the inverse wave and the operands of step s+1 are requested behind the post of step s; the tiles of the block row the next
// strip comes from are updated first.
template <int C>
__device__ __forceinline__ void cdp_m_wave(CdShared& sh, int n_piv4, bool pending, const double* Xb, double* Pend, int blk = 0)
{
    const int l = threadIdx.x & 63, lr = l >> 4, lc = l & 15;
    constexpr int NT = (10 - C + CD_MW - 1) / CD_MW;
    d4 acc[NT];
#pragma unroll
    for (int o = 0; o < NT; ++o)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int idx = C + CD_MW * o;
            acc[o][reg] = (16 * cd_tr(idx) + lr + 4 * reg == 16 * cd_tc(idx) + lc) ? 1.0 : 0.0;
        }
    __syncthreads();                                      // (C)
    bool have = false;
    double a[NT], b[NT];
#define CDP_M_LOADS(vflag, Xsrc, Msrc) do {                                                                                 \
        vflag = __hip_atomic_load(&sh.flags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);                            \
        __atomic_signal_fence(__ATOMIC_SEQ_CST);                                                                           \
        _Pragma("unroll") for (int o = 0; o < NT; ++o) {                                                                     \
            a[o] = (Xsrc)[lr * 64 + 16 * cd_tr(C + CD_MW * o) + lc];                                                         \
            b[o] = (Msrc)[lr * 64 + 16 * cd_tc(C + CD_MW * o) + lc];                                                         \
        }                                                                                                                  \
        __atomic_signal_fence(__ATOMIC_SEQ_CST);                                                                           \
    } while (0)
#pragma unroll 1
    for (int sb = 0; sb < 4; ++sb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s = 4 * sb + q;
            const int p0 = 16 * sb + 4 * q;
            if (s >= n_piv4) return;
#if defined(CD_TL_M)
            if (C == 0) CD_TL(1, blk, s, 0);
#endif
            const int b0 = p0 >> 4;                          // block row of the strip of block s
            // needs: Ms of block s-2 (inverse flag >= s-1), which also says the inverse wave is done with strip buffer s & 1
            int sp = 0;
            if (!have) {
                int v;
                const double* Xq = sh.Xs[(q + 2) & 3];       // panel s-2
                const double* Mq = sh.Ms[q & 1];             // Ms of block s-2
                CDP_M_LOADS(v, Xq, Mq);
                if (__builtin_amdgcn_readfirstlane(v) < s - 1) {
                    while (true) {
                        __builtin_amdgcn_s_sleep(1);
                        v = __hip_atomic_load(&sh.flags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        ++sp;
                        if (__builtin_amdgcn_readfirstlane(v) >= s - 1) break;
                        if (sp > CD_SPIN_LIMIT) { sh.timeout = 1; break; }
                    }
                    __atomic_signal_fence(__ATOMIC_SEQ_CST);
                    CDP_M_LOADS(v, Xq, Mq);
                }
            }
            if (C == 0 && l == 0) { CD_SPIN_ADD(4, sp + (have ? 0 : 1)); CD_SPIN_ADD(5, 1); }
#if defined(CD_TL_M)
            if (C == 0) CD_TL(1, blk, s, 1);
#endif
            // (rows of the inverse above the pivot block are final: a tile whose rows all are -- 16 tr + 15 <= 4 (s-2) + 3 --
            //  gets no more updates); the tiles of block row b0 first: the strip comes out of them
            if (s > 1) {
#pragma unroll
                for (int o = 0; o < NT; ++o) {
                    const int idx = C + CD_MW * o;
                    if (cd_tr(idx) == b0 && (sb <= cd_tr(idx) || (sb == cd_tr(idx) + 1 && q == 0)))
                        acc[o] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[o], -b[o], acc[o], 0, 0, 0);          // M -= X (L4^-1 M(p,:))
                }
            }
            // strip of block s (updates <= s-2 applied): rows p0 + lr of a tile in block row b0 = register q
            double* Mo = sh.Mst[q & 1];
#pragma unroll
            for (int o = 0; o < NT; ++o) {
                const int idx = C + CD_MW * o;
                if (cd_tr(idx) == b0) Mo[lr * 64 + 16 * cd_tc(idx) + lc] = acc[o][q];
            }
            cd_post(sh, 2 + CD_TW + C, s + 1);
#if defined(CD_TL_M)
            if (C == 0) CD_TL(1, blk, s, 2);
#endif
            // ---- behind the published strip: the other tiles' updates, then the request for the next step
            if (s > 1) {
#pragma unroll
                for (int o = 0; o < NT; ++o) {
                    const int idx = C + CD_MW * o;
                    if (cd_tr(idx) != b0 && (sb <= cd_tr(idx) || (sb == cd_tr(idx) + 1 && q == 0)))
                        acc[o] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[o], -b[o], acc[o], 0, 0, 0);
                }
            }
#if !defined(CD_LAZY_ON_T)
            if (pending && sb == 0 && q == 1) {
                // The product of trailing tile 8 + C = (3, 2 + C) for T wave C, X(48.., :) X(16 (2 + C).., :)^T over all 64 columns:
                // this wave's first two steps carry no update of the inverse, and its later steps have slack the T waves lack
                d4 pacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
                for (int kk = 0; kk < 64; kk += 4)
                    pacc = __builtin_amdgcn_mfma_f64_16x16x4f64(Xb[(kk + lr) * CD_LD + 48 + lc], Xb[(kk + lr) * CD_LD + 16 * (2 + C) + lc], pacc, 0, 0, 0);
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) Pend[((4 + C) * 4 + reg) * 64 + l] = pacc[reg];
                cd_post(sh, 10 + C, 1);
            }
#endif
            int vnext = -1;
            const bool more = (s + 1 < n_piv4) && (s + 1 < 16);
            const bool ask = more CD_ASK_IF(sp == 0);
            if (ask) { const double* Xn = sh.Xs[(q + 3) & 3]; const double* Mn = sh.Ms[(q + 1) & 1]; CDP_M_LOADS(vnext, Xn, Mn); }
            have = ask && (__builtin_amdgcn_readfirstlane(vnext) >= s);
#if defined(CD_TL_M)
            if (C == 0) CD_TL(1, blk, s, 3);
#endif
        }
    }
#undef CDP_M_LOADS
}

// Inverse wave of the persistent chain: cd_inverse_wave<true> with its rows of L^-1 going to global memory as they become
// final (CdInvOut), software-pipelined like the T and M waves: the flags and the operands of block s+1 are requested behind
// the post of block s (the panel wave and the M waves are usually a step ahead) and land under the global stores.
__device__ __forceinline__ bool cdp_inverse_wave(CdShared& sh, int n_piv4, const CdInvOut io)
{
    const int l = threadIdx.x & 63;
    bool bad = false;
    double mp0 = 0.0, mp1 = 0.0, mp2 = 0.0, mp3 = 0.0;     // own column of Ms of the previous block
    if (n_piv4 < 16) {
        // a last block of fewer than 64 rows: identity beyond the pivots that exist (column l: rows 4 n_piv4 .. 63)
        for (int row = 4 * n_piv4; row < 64; row += 2) st_coh2(io.Lglob + row + 64 * l, row == l ? 1.0 : 0.0, row + 1 == l ? 1.0 : 0.0);
    }
    // (the flag goes out behind the end-of-chain barrier, by the caller: waiting HERE for the last global stores would hold the
    //  whole workgroup at that barrier for a store round trip, ~0.5 us per block, while the X product only needs L^-1 in LDS)
    auto finish = [&]() {};
    // flags this wave needs at block s: panel (lane 0) and both M waves (lanes 2 + CD_TW ..) at s + 1
    const int li = l & 15;
    const bool mine = (li == 0) || (li >= 2 + CD_TW && li < 2 + CD_TW + CD_MW);
    bool have = false;
    double a0, a1, a2, a3, u[16], l10, l20, l30, l21, l31, l32, q0, q1, q2, q3;
#define CDP_I_LOADS(vflag, qq, pp) do {                                                                                     \
        vflag = __hip_atomic_load(&sh.flags[li], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);                           \
        __atomic_signal_fence(__ATOMIC_SEQ_CST);                                                                           \
        { const double* Mc = sh.Mst[(qq) & 1]; const double* Xp = sh.Xs[((qq) + 3) & 3]; const double* Xc = sh.Xs[(qq) & 3];   \
          const double* R = sh.Rs[(qq) & 1];                                                                               \
          a0 = Mc[l]; a1 = Mc[64 + l]; a2 = Mc[128 + l]; a3 = Mc[192 + l];                                                 \
          _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                      \
              _Pragma("unroll") for (int k = 0; k < 4; ++k) u[4 * j + k] = Xp[j * 64 + (pp) + k];                            \
          l10 = Xc[(pp) + 1]; l20 = Xc[(pp) + 2]; l30 = Xc[(pp) + 3];                                                      \
          l21 = Xc[64 + (pp) + 2]; l31 = Xc[64 + (pp) + 3]; l32 = Xc[128 + (pp) + 3];                                      \
          q0 = R[0]; q1 = R[1]; q2 = R[2]; q3 = R[3]; }                                                                    \
        __atomic_signal_fence(__ATOMIC_SEQ_CST);                                                                           \
    } while (0)
#pragma unroll 1
    for (int sb = 0; sb < 4; ++sb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s = 4 * sb + q;                     // block
            const int p0 = 16 * sb + 4 * q;
            if (s >= n_piv4) { finish(); return bad; }
            int sp = 0;
            if (!have) {
                // panel s (L4, reciprocals); strip of block s (M waves, iteration s): flags and operands in one batch
                int v;
                CDP_I_LOADS(v, q, p0);
                if (!__all(!mine || v >= s + 1)) {
                    while (true) {
                        __builtin_amdgcn_s_sleep(1);
                        v = __hip_atomic_load(&sh.flags[li], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        ++sp;
                        if (__all(!mine || v >= s + 1)) break;
                        if (sp > CD_SPIN_LIMIT) { sh.timeout = 1; break; }
                    }
                    __atomic_signal_fence(__ATOMIC_SEQ_CST);
                    CDP_I_LOADS(v, q, p0);
                }
            }
            if (l == 0) { CD_SPIN_ADD(6, sp + (have ? 0 : 1)); CD_SPIN_ADD(7, 1); }
            __builtin_amdgcn_sched_barrier(0);
            bad = bad || !(q0 > 1e-300 && q0 < 1e300 && q1 > 1e-300 && q1 < 1e300 && q2 > 1e-300 && q2 < 1e300 && q3 > 1e-300 && q3 < 1e300);
            {   // strip column l: M(p0+k, l) -= sum_j X_prev(p0+k, j) Ms_prev(j, l)
                const double ms[4] = {mp0, mp1, mp2, mp3};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a0 = fma(-u[4 * j + 0], ms[j], a0); a1 = fma(-u[4 * j + 1], ms[j], a1);
                    a2 = fma(-u[4 * j + 2], ms[j], a2); a3 = fma(-u[4 * j + 3], ms[j], a3);
                }
            }
            // entries right of the pivot columns come out as exact zeros (M is lower triangular)
            mp0 = a0 * q0;
            mp1 = fma(-l10, mp0, a1) * q1;
            mp2 = fma(-l21, mp1, fma(-l20, mp0, a2)) * q2;
            mp3 = fma(-l32, mp2, fma(-l31, mp1, fma(-l30, mp0, a3))) * q3;
            double* Mo = sh.Ms[q & 1];
            Mo[l] = mp0; Mo[64 + l] = mp1; Mo[128 + l] = mp2; Mo[192 + l] = mp3;
            // column l of rows p0..p0+3 of L^-1 (final)
            sh.Mf[l * CD_LD + p0 + 0] = mp0; sh.Mf[l * CD_LD + p0 + 1] = mp1;
            sh.Mf[l * CD_LD + p0 + 2] = mp2; sh.Mf[l * CD_LD + p0 + 3] = mp3;
            cd_post(sh, 1, s + 1);
            // the request for the next block's flags and operands, then this block's rows of L^-1 to global memory
            int vnext = -1;
            const bool more = (s + 1 < n_piv4) && (s + 1 < 16);
            const bool ask = more CD_ASK_IF(sp == 0);
            if (ask) CDP_I_LOADS(vnext, q + 1, p0 + 4);
            st_coh2(io.Lglob + p0 + 64 * l, mp0, mp1); st_coh2(io.Lglob + p0 + 2 + 64 * l, mp2, mp3);
            have = ask && __all(!mine || vnext >= s + 2);
        }
    }
#undef CDP_I_LOADS
    finish();
    return bad;
}

// One wave role of the chain workgroup over all diagonal blocks.  ROLE: 0 panel wave, 1 inverse wave, 2..5 T waves,
// 6..7 M waves.  Every role runs the same barrier sequence; the block loop is per role so that what a role carries
// from block to block (the T waves' fetch registers) does not count against the registers of the others.
template <int ROLE>
__device__ __forceinline__ void cdp_role(double* lds, double* A, long ldA, int nblk, int r_total, double* Linv, double* Ypanel,
                                         SweepFlags* fl, int32_t* status, unsigned long long* dbg, int exp_mask, int n_lower_strips)
{
    CdShared& sh = *reinterpret_cast<CdShared*>(lds);
    double* Aop = lds + CDP_OFF_AOP;
    double* Tpre = Aop + CDP_AOP_DOUBLES;
    double* Xb = sh.Lf;                                   // X = A(k,k-1) L^-T(k-1) lives where L would be collected
    const int t = threadIdx.x;
    const int row = t & 63;
    constexpr int g = ROLE;                               // wave index
    constexpr bool is_t = (ROLE >= 2 && ROLE < 2 + CD_TW);
    constexpr bool is_m = (ROLE >= 2 + CD_TW);
    constexpr int TB = is_t ? ROLE - 2 : 0;
    constexpr int MC = is_m ? ROLE - 2 - CD_TW : 0;
    double* Pend = Tpre + CDP_TPRE_DOUBLES;               // the products of the tiles outside the first tile column (cdp_t_wave)
    int fetch_st = 0;                                     // (T waves) state of the fetch of block k's inputs, see cdp_finish
    double pf[10];
    const unsigned lane_off = td_lane_offset(ldA);        // per-lane part of an LDS-DMA source address
    bool bad = false;
#pragma unroll 1
    for (int k = 0; k < nblk; ++k) {
        const int r_here = min(64, max(0, r_total - 64 * k));
        const int n_piv4 = (r_here + 3) >> 2;
        const bool pending = k > 0;
        unsigned long long* stamp = (dbg && k < SWD_K) ? dbg + k * SWD_SLOT : nullptr;
        if (stamp && t == 0) stamp[0] = wall_clock64();
        // (L^-1(k-1) is in global memory already and its flag is out: the inverse wave wrote its rows as they became final)
        if constexpr (is_t) {
            CdpNext cur;
            cur.a_base = A + (long)k * 64 + (long)(k > 0 ? k - 1 : 0) * 64 * ldA;
            cur.tile = A + (long)k * 64 + (long)k * 64 * ldA;
            cur.ldA = ldA; cur.Aop = Aop; cur.Tpre = Tpre; cur.need_a = k > 0;
#if defined(SW_TILES01_ONE)
            cur.flag = k <= 1 ? &fl->tiles01 : &fl->row_ready[k]; cur.need = k <= 1 ? n_lower_strips : 4; cur.shards = 1;
#else
            cur.flag = k <= 1 ? &fl->tiles01s[0] : &fl->row_ready[k]; cur.need = k <= 1 ? n_lower_strips : 4; cur.shards = k <= 1 ? 4 : 1;
#endif
            cdp_finish<TB>(cur, sh, pf, fetch_st, lane_off);
        }
        __syncthreads();                                  // inputs of block k are in LDS
        if (stamp && t == 0) stamp[1] = wall_clock64();
        if (pending) {
            // X = A(k,k-1) L^-T(k-1): Mf[m * CD_LD + c] = L^-1(c, m) is the B operand image as it stands.  L^-1 is lower
            // triangular: column tile tc of X only needs m < 16 (tc + 1); waves pair the column tiles {0,3} and {1,2}
            const int l = t & 63, lr = l >> 4, lc = l & 15;
            auto x_tiles = [&](auto TR) {
                constexpr int tr = decltype(TR)::value;
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    const int tc = (ROLE & 1) ? 1 + o : 3 * o;
                    d4 acc = {0.0, 0.0, 0.0, 0.0};
                    for (int kk = 0; kk < 16 * (tc + 1); kk += 4)
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Aop[cdp_aop_index(kk + lr, 16 * tr + lc)],
                                                                   sh.Mf[(kk + lr) * CD_LD + 16 * tc + lc], acc, 0, 0, 0);
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) Xb[(16 * tc + lc) * CD_LD + 16 * tr + lr + 4 * reg] = acc[reg];
                }
            };
            // (round 6) The inverse wave's rows of L^-1(k-1) went out as write-through stores during the previous chain; the flag the strips
            // wait for may only follow their acknowledgement.  Waiting for it between the two barriers held the whole workgroup
            // for a store round trip; now the inverse wave waits HERE, beside the X product, and the T wave that shares its SIMD
            // forms its two tiles (the matrix pipe of that SIMD sees the same 40 products)
            if constexpr (ROLE == 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if ((t & 63) == 0) __hip_atomic_store(&fl->linv_ready, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                x_tiles(std::integral_constant<int, (ROLE >> 1)>{});
                if constexpr (ROLE == 5) x_tiles(std::integral_constant<int, 0>{});
            }
        }
        __syncthreads();
        if (stamp && t == 0) stamp[2] = wall_clock64();
        if (pending && !is_t) {
            // X is the panel block L(k,k-1): published for the strips by the four waves that are idle while the T waves
            // apply the first tile column of T -= X X^T (flag xrow_ready after the barrier below)
            constexpr int slot = ROLE < 2 ? ROLE : ROLE - CD_TW;             // 0..3
            double* Xout = Ypanel + (long)k * 64 + (long)(k - 1) * 64 * ldA;
#pragma unroll
            for (int c = slot; c < 64; c += 4) st_coh(Xout + row + (long)c * ldA, Xb[c * CD_LD + row]);
            // ... and each of the four forms ONE product of T(k,k) -= X X^T outside the first tile column -- tiles 2, 4, 7, 5 --
            // for the T wave that owns the tile (cdp_t_wave takes it from LDS behind barrier (C)): 16 MFMAs beside the 16 of
            // the T waves' first tile column, instead of 8 per pivot step on the T waves during the first steps of the chain
            {
                constexpr int pidx = slot == 0 ? 2 : slot == 1 ? 4 : slot == 2 ? 7 : 5;
                constexpr int ptr_ = cd_tr(pidx), ptc = cd_tc(pidx);
                const int l = t & 63, lr = l >> 4, lc = l & 15;
                d4 pacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
                for (int kk = 0; kk < 64; kk += 4)
                    pacc = __builtin_amdgcn_mfma_f64_16x16x4f64(Xb[(kk + lr) * CD_LD + 16 * ptr_ + lc], Xb[(kk + lr) * CD_LD + 16 * ptc + lc], pacc, 0, 0, 0);
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) Pend[(slot * 4 + reg) * 64 + l] = pacc[reg];
            }
            wait_stores();
        }
        if (t < 16) sh.flags[t] = 0;
        if (g < 4) sh.Xs[3][g * 64 + row] = 0.0;                                         // "panel" -1
        else { sh.Mst[0][(g - 4) * 64 + row] = 0.0; sh.Mst[1][(g - 4) * 64 + row] = 0.0; }
        if constexpr (ROLE == 0) {
            __syncthreads();                                         // (C)
            if (pending && t == 0) __hip_atomic_store(&fl->xrow_ready, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (stamp && t == 0) stamp[3] = wall_clock64();
            cd_panel_wave<false>(sh, n_piv4, k);
            if (stamp && t == 0) stamp[4] = wall_clock64();
        } else if constexpr (ROLE == 1) {
            __syncthreads();
            bad = cdp_inverse_wave(sh, n_piv4, CdInvOut{Linv + (long)k * 64 * 64, &fl->linv_ready, k + 1}) || bad;
        } else if constexpr (is_t) {
            const bool want_next = (k + 1 < nblk) && !(exp_mask & 1);
            CdpNext nx;
            nx.a_base = A + (long)(k + 1) * 64 + (long)k * 64 * ldA;
            nx.tile = A + (long)(k + 1) * 64 + (long)(k + 1) * 64 * ldA;
            nx.ldA = ldA; nx.Aop = Aop; nx.Tpre = Tpre; nx.need_a = true;
#if defined(SW_TILES01_ONE)
            nx.flag = k + 1 <= 1 ? &fl->tiles01 : &fl->row_ready[k + 1]; nx.need = k + 1 <= 1 ? n_lower_strips : 4; nx.shards = 1;
#else
            nx.flag = k + 1 <= 1 ? &fl->tiles01s[0] : &fl->row_ready[k + 1]; nx.need = k + 1 <= 1 ? n_lower_strips : 4; nx.shards = k + 1 <= 1 ? 4 : 1;
#endif
            fetch_st = cdp_t_wave<TB>(sh, n_piv4, pending, Xb, Tpre, Pend, nx, want_next, pf, lane_off, stamp, k);
        } else {
            cdp_m_wave<MC>(sh, n_piv4, pending, Xb, Pend, k);
        }
        __syncthreads();                                  // the chain of block k has ended: L^-1(k) is complete in Mf
        if constexpr (ROLE == 1) {
            // the inverse wave's rows of L^-1(k) have reached memory: the strips may fetch them
            if (k + 1 >= nblk)          // (the last block: nothing follows; otherwise beside the next block's X product, see above)
            {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if ((t & 63) == 0) __hip_atomic_store(&fl->linv_ready, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (stamp && t == 0) stamp[5] = wall_clock64();
        if (sh.timeout) { if (t == 0) sw_timed_out(status, sh.timeout == 5 ? -35 : -(36 + 10 * k), k); return; }     // hand-over protocol broke (never expected)
    }
    if (bad) atomicMin(status, -6);                       // RSLAM_ERR_NOT_SPD
}

__device__ __forceinline__ void cd_chain_persistent(double* lds, double* A, long ldA, int nblk, int r_total, double* Linv, double* Ypanel,
                                                    SweepFlags* fl, int32_t* status, unsigned long long* dbg, int exp_mask, int n_lower_strips)
{
    CdShared& sh = *reinterpret_cast<CdShared*>(lds);
    if (threadIdx.x == 0) sh.timeout = 0;
    __syncthreads();
    // Which wave plays which role decides who shares a matrix pipe: waves w and w + 4 run on the same SIMD.  Every rank-4
    // update is one v_mfma_f64_16x16x4 per tile, ~100 cycles of that SIMD's matrix pipe each; T waves 0 / 1 own three tiles,
    // T waves 2 / 3 two, the M waves five each.  With the roles in wave order (T0 + M0 on SIMD 2, T1 + M1 on SIMD 3: eight
    // MFMAs per pivot step there, two on SIMDs 0 and 1) the pivot chain waited ~600 cycles per step for the strips of T
    // waves 0 and 1 (stamped build: the panel wave's own work is ~800 cycles of a ~1500-cycle step).  The M waves therefore
    // share the SIMDs of the two VALU-only waves (panel, inverse) and the T waves pair up: five MFMAs per SIMD and step.
#if !defined(CD_ROLES_REMAPPED)
#define CDP_ROLE_OF_WAVE(w) (w)
#else
#define CDP_ROLE_OF_WAVE(w) ((w) < 4 ? (w) : (w) < 6 ? (w) + 2 : (w) - 2)     // waves 4, 5: M waves (roles 6, 7); waves 6, 7: T waves 2, 3
#endif
#define CDP_CASE(w) case w: cdp_role<CDP_ROLE_OF_WAVE(w)>(lds, A, ldA, nblk, r_total, Linv, Ypanel, fl, status, dbg, exp_mask, n_lower_strips); break
    switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {
    CDP_CASE(0); CDP_CASE(1); CDP_CASE(2); CDP_CASE(3); CDP_CASE(4); CDP_CASE(5); CDP_CASE(6);
    default: cdp_role<CDP_ROLE_OF_WAVE(7)>(lds, A, ldA, nblk, r_total, Linv, Ypanel, fl, status, dbg, exp_mask, n_lower_strips); break;
    }
#undef CDP_CASE
}


// ---- the tile workers of the persistent sweep (fused launches) ---------------------------------------------------------
// K10 + K11 without a launch of their own: P' = 1/2 (P + P^T) - Y Y^T (ExtendKF.cpp:608-609) and the Jnorm congruence
// (:629-634) used to start when the whole sweep had ended, although column block k of Y is final right after step k.  The
// workgroups the sweep does not need -- the launch is as large as the device, one workgroup per CU: the extra blocks behind
// the strips, plus the strips that have no rows of this system -- each own up to WK_TILES lower-triangle tile pairs of P:
// two four-wave MFMA engines (tile_gemm.h, LDS-DMA staging) in lockstep, WK_SLOTS tiles each, accumulators resident in
// registers for the whole sweep.  The accumulators START as the symmetrised P tile and the A operand is negated (neg
// modifier of the MFMA: free), so the epilogue has nothing left to read.  Per column block: poll the flag words of the
// strips that hold the tiles' rows of Y, then two K = 32 chunks per tile, fetched with sc1 transfers (the strips' stores
// are write-through: sweep_persistent_kernel, "Memory protocol").  Only the last block's chunks and the epilogue (tile,
// Jnorm rows / columns, mirrored tile) are left when the pivot chain ends.  Nothing in the sweep waits for a worker.
constexpr int WK_SLOTS = 3;          // (2 slots -- no spills at all -- measured no faster: the spilled values are not in the chunk bodies)
constexpr int WK_TILES = 2 * WK_SLOTS;
static_assert(TD_LDS_DOUBLES >= TS_DOUBLES, "a worker engine's epilogue stages its tile in the operand buffers");

struct WkTile { int bi, bj; bool have; };

__device__ __forceinline__ WkTile wk_tile(const WorkerArgs& wk, int ti, int ntiles)
{
    WkTile r; r.bi = 0; r.bj = 0; r.have = ti < ntiles;
    if (!r.have) return r;
    if (wk.tile_order) { const int e = wk.tile_order[ti]; r.bi = e >> 16; r.bj = e & 0xffff; }
    else {
        int bi = (int)((sqrt(8.0 * (double)ti + 1.0) - 1.0) * 0.5);
        while ((long)bi * (bi + 1) / 2 > ti) --bi;
        while ((long)(bi + 1) * (bi + 2) / 2 <= ti) ++bi;
        r.bi = bi; r.bj = ti - bi * (bi + 1) / 2;
    }
    return r;
}

// Which tiles a worker owns.  A workgroup runs on XCD (blockIdx mod 8) and the eight XCDs have an L2 each: with the tiles dealt
// round-robin over the worker INDEX (the round-3 form: tile widx + j W) every XCD's workers touched every 64-row panel of Y,
// and each Y block was fetched eight times (58 of the launch's 107 MB of reads at C3).  Now (round 6) XCD x's workers own a
// contiguous run of the region-major tile sequence (the 8 x 8 regions of make_rank_update_order: ~16 of the 29 panels at C3),
// sized in proportion to the number of workers the XCD has in THIS launch -- the worker set depends on the inlier count -- and
// dealt round-robin inside the XCD: tile lo + y + j nx.  Which workgroup computes a tile does not change a bit of it.
struct WkMap {
    int widx, W;              // the worker's index and the number of workers (the round-3 form, kept where the run per XCD would not fit)
    int lo, hi, y, nx;        // XCD form: this XCD's run [lo, hi) of the sequence, the worker's rank among the XCD's nx workers
    bool by_xcd;
};

// numbers v in [a, b] with v mod 8 == x (0 for an empty range)
__device__ __forceinline__ int wk_count_res(int a, int b, int x)
{
    if (b < a) return 0;
    return (b - x + 64) / 8 - (a - 1 - x + 64) / 8;        // (+64: floor division of small negatives)
}

// entry s of the region-major sequence, recovered from the XCD-interleaved order array (order[8 idx + x] = seq[start(x) + idx],
// eighths of q or q + 1 entries: make_rank_update_order)
__device__ __forceinline__ int wk_seq_entry(const int32_t* __restrict__ order, int T, int s)
{
    const int q = T / 8, r = T % 8;
    const int x = (s < r * (q + 1)) ? s / (q + 1) : r + (s - r * (q + 1)) / q;
    const int start = x * q + (x < r ? x : r);
    return order[8 * (s - start) + x];
}

__device__ __forceinline__ WkTile wk_tile_j(const WorkerArgs& wk, const WkMap& m, int j, int ntiles)
{
    if (!m.by_xcd) return wk_tile(wk, m.widx + j * m.W, ntiles);
    WkTile r; r.bi = 0; r.bj = 0;
    const int sidx = m.lo + m.y + j * m.nx;
    r.have = sidx < m.hi;
    if (r.have) { const int e = wk_seq_entry(wk.tile_order, ntiles, sidx); r.bi = e >> 16; r.bj = e & 0xffff; }
    return r;
}

// K11 on one tile of the first block column, in its LDS image Cs[col][row] (64 threads, j = row of tile (bi,0) = column of
// its mirror): the Jnorm congruence on rows / columns 3..6 (ExtendKF.cpp:629-634); same arithmetic as a separate pass over
// P would do.  For bi != 0 only the tile's columns 3..6 change (its mirror carries the rows).
__device__ __forceinline__ void wk_jnorm_tile(double* Cs, const double (&T)[16], int bi, int j)
{
    if (bi != 0) {
        double rb[4];
        for (int i = 0; i < 4; ++i) {
            double sacc = 0;
            for (int q = 0; q < 4; ++q) sacc += T[i + 4 * q] * Cs[(3 + q) * TS_LD + j];   // P(3+q, col) by symmetry
            rb[i] = sacc;
        }
        for (int i = 0; i < 4; ++i) Cs[(3 + i) * TS_LD + j] = rb[i];
    } else if (!(j >= 3 && j < 7)) {
        double rb[4];
        for (int i = 0; i < 4; ++i) {
            double sacc = 0;
            for (int q = 0; q < 4; ++q) sacc += T[i + 4 * q] * Cs[j * TS_LD + (3 + q)];
            rb[i] = sacc;
        }
        for (int i = 0; i < 4; ++i) { Cs[j * TS_LD + (3 + i)] = rb[i]; Cs[(3 + i) * TS_LD + j] = rb[i]; }
    } else if (j == 3) {
        double cb[4][4], out[4][4];         // cb = J * P44 ; out = cb * J^T
        for (int i = 0; i < 4; ++i)
            for (int c = 0; c < 4; ++c) {
                double sacc = 0;
                for (int q = 0; q < 4; ++q) sacc += T[i + 4 * q] * Cs[(3 + c) * TS_LD + (3 + q)];
                cb[i][c] = sacc;
            }
        for (int i = 0; i < 4; ++i)
            for (int c = 0; c < 4; ++c) {
                double sacc = 0;
                for (int q = 0; q < 4; ++q) sacc += cb[i][q] * T[c + 4 * q];
                out[i][c] = sacc;
            }
        for (int i = 0; i < 4; ++i)
            for (int c = 0; c < 4; ++c) Cs[(3 + c) * TS_LD + (3 + i)] = out[i][c];
    }
}

// The high-innovation pass found no inliers and the low-innovation covariance was deferred: P_li is written now, by every
// workgroup of the launch (one tile pair at a time per four waves, through LDS): J (sym(P_pred) - Y1 Y1^T) J^T.
__device__ __forceinline__ void wk_materialise_deferred(const WorkerArgs& wk, int NP, double* lds)
{
    const int ntiles = wk.nT * (wk.nT + 1) / 2;
    const int t = threadIdx.x, half = t >> 8, t4 = t & 255, row = t4 & 63, g4 = t4 >> 6;
    for (int i = (int)(blockIdx.x * blockDim.x) + t; i < NP; i += (int)(gridDim.x * blockDim.x)) wk.x_out[i] = wk.x_in[i];
    double T1[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) T1[q] = wk.T_li[q];
    double* Cs = lds + (size_t)half * TD_LDS_DOUBLES;
    double* Ts = Cs + TS_DOUBLES;
    for (int base = 2 * (int)blockIdx.x; base < ntiles; base += 2 * (int)gridDim.x) {      // (uniform trip count: barriers inside)
        const WkTile tl = wk_tile(wk, base + half, ntiles);
        const double* Pij = wk.Ppred + 64L * tl.bi + 64L * tl.bj * wk.ldp;
        const double* Pji = wk.Ppred + 64L * tl.bj + 64L * tl.bi * wk.ldp;
        if (tl.have) {
#pragma unroll 4
            for (int q = 0; q < 16; ++q) {
                const int c = g4 + 4 * q;
                Cs[c * TS_LD + row] = Pij[row + (long)c * wk.ldp];
                Ts[c * TS_LD + row] = Pji[row + (long)c * wk.ldp];
            }
        }
        __syncthreads();
        if (tl.have) {
            double yr[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) yr[c] = wk.Y1[64L * tl.bi + row + (long)c * wk.ldy1];
#pragma unroll 4
            for (int q = 0; q < 16; ++q) {
                const int c = g4 + 4 * q;
                double v = 0.5 * Cs[c * TS_LD + row] + 0.5 * Ts[row * TS_LD + c];
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) v -= yr[cc] * wk.Y1[64L * tl.bj + c + (long)cc * wk.ldy1];
                Cs[c * TS_LD + row] = v;           // (each thread rewrites exactly the entries it read from Cs)
            }
        }
        __syncthreads();
        if (tl.have && tl.bj == 0 && t4 < 64) wk_jnorm_tile(Cs, T1, tl.bi, t4);
        __syncthreads();
        if (tl.have) {
            double* Cij = wk.Pout + 64L * tl.bi + 64L * tl.bj * wk.ldo;
            double* Cji = wk.Pout + 64L * tl.bj + 64L * tl.bi * wk.ldo;
#pragma unroll 4
            for (int q = 0; q < 16; ++q) {
                const int c = g4 + 4 * q;
                Cij[row + (long)c * wk.ldo] = Cs[c * TS_LD + row];
                if (tl.bi != tl.bj) Cji[row + (long)c * wk.ldo] = Cs[row * TS_LD + c];
            }
        }
        __syncthreads();
    }
}

// update() pass-through (no inliers, ExtendKF.cpp:635-638): every workgroup of the launch copies its share
__device__ __forceinline__ void wk_passthrough(const WorkerArgs& wk, int NP)
{
    const int ntiles = wk.nT * (wk.nT + 1) / 2;
    for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < NP; i += (int)(gridDim.x * blockDim.x)) wk.x_out[i] = wk.x_in[i];
    if (wk.Pin == wk.Pout) return;
    for (int ti = (int)blockIdx.x; ti < ntiles; ti += (int)gridDim.x) {
        const WkTile tl = wk_tile(wk, ti, ntiles);
        const int row = threadIdx.x & 63, g = threadIdx.x >> 6;
        for (int c = g; c < 64; c += (int)(blockDim.x >> 6)) {
            wk.Pout[64L * tl.bi + row + (64L * tl.bj + c) * wk.ldo] = wk.Pin[64L * tl.bi + row + (64L * tl.bj + c) * wk.ldp];
            if (tl.bi != tl.bj) wk.Pout[64L * tl.bj + row + (64L * tl.bi + c) * wk.ldo] = wk.Pin[64L * tl.bj + row + (64L * tl.bi + c) * wk.ldp];
        }
    }
}

// one column block (NCH chunks of 32 columns) into the accumulators of this engine's tiles; n_wg slots are in use somewhere
// in the workgroup (uniform), the engine's own tiles are tl[].have
template <int NCH>
__device__ __forceinline__ void wk_block(TgAcc (&acc)[WK_SLOTS], const WkTile (&tl)[WK_SLOTS], int n_wg, const double* __restrict__ Y, long ldy,
                                         int kcol, double* hbase, int wave4)
{
    double* const Abuf[2] = { hbase, hbase + 2 * TD_OPER_DOUBLES };
    double* const Bbuf[2] = { hbase + TD_OPER_DOUBLES, hbase + 3 * TD_OPER_DOUBLES };
    const unsigned lo = td_lane_offset(ldy);
    double a[TG_MI];
    BFrag b[TG_NI];
    if (tl[0].have) td_issue_chunk_w<16>(Y + 64L * tl[0].bi, ldy, Y + 64L * tl[0].bj, ldy, kcol, Abuf[0], Bbuf[0], wave4);
    __syncthreads();                                      // (a barrier waits for this wave's transfers)
    if (tl[0].have) td_read_frags(td_frag_ptr(Abuf[0], Bbuf[0], wave4), 0, a, b);
    static_for<0, WK_SLOTS>([&](auto S) {
        constexpr int slot = decltype(S)::value;
        if (slot < n_wg) {
            static_for<0, NCH>([&](auto C) {
                constexpr int c = decltype(C)::value;
                constexpr int u = slot * NCH + c, cur = u & 1, nxt = cur ^ 1;
                constexpr int nslot = (c + 1 < NCH) ? slot : slot + 1, nc = (c + 1 < NCH) ? c + 1 : 0;
                const bool more = (c + 1 < NCH) || (slot + 1 < n_wg);          // uniform over the workgroup
                const bool nact = more && (nslot < WK_SLOTS) && tl[nslot < WK_SLOTS ? nslot : 0].have;
                // source of the next chunk: the engine's own next tile, or -- when only the sibling engine has one -- anything valid
                const WkTile& nt = tl[(nact && nslot < WK_SLOTS) ? nslot : 0];
                const double* An = Y + 64L * nt.bi;
                const double* Bn = Y + 64L * nt.bj;
                const int kn = kcol + 32 * nc;
                if (tl[slot].have) {
                    if (more) td_compute_chunk_w<true, 16, 1>(Abuf[cur], Bbuf[cur], acc[slot], a, b, An, ldy, lo, Bn, ldy, lo, kn, Abuf[nxt], Bbuf[nxt], wave4);
                    else      td_compute_chunk_w<false, 16, 1>(Abuf[cur], Bbuf[cur], acc[slot], a, b, An, ldy, lo, Bn, ldy, lo, kn, Abuf[nxt], Bbuf[nxt], wave4);
                } else if (more) {
                    if (nact) td_issue_chunk_w<16>(An, ldy, Bn, ldy, kn, Abuf[nxt], Bbuf[nxt], wave4);
                    __syncthreads();
                    if (nact) td_read_frags(td_frag_ptr(Abuf[nxt], Bbuf[nxt], wave4), 0, a, b);
                }
            });
        }
    });
}

__device__ __forceinline__ void sweep_tile_worker(const WkMap& map, int nblk, int r_total, int rp_blocks, long ldA, const double* Ypanel,
                                                  const WorkerArgs& wk, SweepFlags* fl, const int32_t* __restrict__ sel, int slot_k,
                                                  int32_t* status, double* lds, int* abort, unsigned long long* dbg, int exp_mask)
{
    const int widx = map.widx;
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), half = wave >> 2, wave4 = wave & 3;
    const int lane = t & 63;
    const int ntiles = wk.nT * (wk.nT + 1) / 2;
    const int RP = 64 * rp_blocks;
    if (t == 0) *abort = 0;
    WkTile tl[WK_SLOTS];
    int n_wg = 0;
#pragma unroll
    for (int slot = 0; slot < WK_SLOTS; ++slot) {
        tl[slot] = wk_tile_j(wk, map, 2 * slot + half, ntiles);
        if (wk_tile_j(wk, map, 2 * slot, ntiles).have) n_wg = slot + 1;            // (engine 0 has at least as many tiles as engine 1)
    }
    if (n_wg == 0) return;
    const bool timing = dbg && widx == 0;
    if (timing && t == 0) dbg[(5 * SWD_K + 0) * SWD_SLOT + 0] = wall_clock64();
    // ---- accumulators = 1/2 (P + P^T) of the tile, in the MFMA result layout (tg_acc_to_lds).  The HI pass reads what the
    // LI pass wrote -- mirrored pairs, exactly symmetric off the diagonal tiles -- so there the mirror tile need not be read
    // and the tile comes straight from memory (128-byte row segments, under the wait for the first column block).
    // (high-innovation pass behind a deferred low-innovation update: the tiles start from P_pred, Y1 and that update's Jnorm)
    const bool deferred = wk.token == 2 && wk.defer_flag && *wk.defer_flag != 0;
    const double* Pin = deferred ? wk.Ppred : wk.Pin;
    const bool li_wrote = !deferred && wk.li_done_slot >= 0 && sel[wk.li_done_slot] > 0;
    const bool staged = !li_wrote && nblk == 1;
    TgAcc acc[WK_SLOTS];
    double* hbase = lds + (size_t)half * TD_LDS_DOUBLES;
    {
        const int i = lane >> 4, blk = (lane >> 2) & 3, j = lane & 3;
        const int t4 = t & 255, row4 = t4 & 63, g4 = t4 >> 6;
        double* Cs = hbase;                 // tile (bi,bj) as [col][row], tile (bj,bi) as [col][row] behind it
        double* Ts = hbase + TS_DOUBLES;
#pragma unroll
        for (int slot = 0; slot < WK_SLOTS; ++slot) {
            tg_zero(acc[slot]);
            const double* Pij = Pin + 64L * tl[slot].bi + 64L * tl[slot].bj * wk.ldp;
            const double* Pji = Pin + 64L * tl[slot].bj + 64L * tl[slot].bi * wk.ldp;
            if (staged) {
                // both tiles of the pair, each in 512-byte column runs, through LDS (the mirror tile is needed transposed):
                // a system of one diagonal block is done in microseconds, the pass is then a stream over P and this read is
                // on its critical path (elsewhere the tiles arrive under the wait for the first column block)
                if (slot >= n_wg) continue;                          // (uniform: the barriers below are taken by both engines)
                if (tl[slot].have) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {                    // 16 loads of the thread in flight at once, twice
                        double pa[8], pb[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            pa[q] = Pij[row4 + (long)(g4 + 4 * (8 * h + q)) * wk.ldp];
                            pb[q] = Pji[row4 + (long)(g4 + 4 * (8 * h + q)) * wk.ldp];
                        }
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            Cs[(g4 + 4 * (8 * h + q)) * TS_LD + row4] = pa[q];
                            Ts[(g4 + 4 * (8 * h + q)) * TS_LD + row4] = pb[q];
                        }
                    }
                }
                __syncthreads();
                if (tl[slot].have) {
#pragma unroll
                    for (int mi = 0; mi < TG_MI; ++mi)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int row = mi * 16 + 4 * blk + i, col = wave4 * 16 + 4 * ((blk - q) & 3) + j;
                            acc[slot][mi][0][q] = 0.5 * Cs[col * TS_LD + row] + 0.5 * Ts[row * TS_LD + col];
                        }
                }
                __syncthreads();
                continue;
            }
            if (!tl[slot].have) continue;
            const bool mirror_known = li_wrote && tl[slot].bi != tl[slot].bj;
#pragma unroll
            for (int mi = 0; mi < TG_MI; ++mi)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = mi * 16 + 4 * blk + i, col = wave4 * 16 + 4 * ((blk - q) & 3) + j;
                    const double pij = Pij[row + (long)col * wk.ldp];
                    acc[slot][mi][0][q] = mirror_known ? pij : 0.5 * pij + 0.5 * Pji[col + (long)row * wk.ldp];
                }
        }
    }
    if (deferred) {
        // M = sym(P_pred) - Y1 Y1^T entry by entry (rank <= 4: VALU), then the low-innovation update's Jnorm congruence on the
        // tiles of the first block column, through LDS as in the epilogue (uniform barriers: both engines take them)
        const int i = lane >> 4, blk = (lane >> 2) & 3, j = lane & 3;
        double T1[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) T1[q] = wk.T_li[q];
        double* Cs = hbase;
        const int t4 = t & 255;
        static_for<0, WK_SLOTS>([&](auto S) {
            constexpr int slot = decltype(S)::value;
            if (slot < n_wg) {
                const bool mine = tl[slot].have;
                if (mine) {
                    double yc[4][4];                              // Y1 rows of this lane's four columns
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            yc[q][c] = wk.Y1[64L * tl[slot].bj + wave4 * 16 + 4 * ((blk - q) & 3) + j + (long)c * wk.ldy1];
#pragma unroll
                    for (int mi = 0; mi < TG_MI; ++mi) {
                        double yr[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) yr[c] = wk.Y1[64L * tl[slot].bi + mi * 16 + 4 * blk + i + (long)c * wk.ldy1];
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            acc[slot][mi][0][q] -= ((yr[0] * yc[q][0] + yr[1] * yc[q][1]) + yr[2] * yc[q][2]) + yr[3] * yc[q][3];
                    }
                }
                const WkTile t0 = wk_tile_j(wk, map, 2 * slot, ntiles), t1 = wk_tile_j(wk, map, 2 * slot + 1, ntiles);
                if ((t0.have && t0.bj == 0) || (t1.have && t1.bj == 0)) {            // (uniform over the workgroup)
                    const bool fix = mine && tl[slot].bj == 0;
                    __syncthreads();
                    if (fix) tg_acc_to_lds_w(acc[slot], Cs, 1.0, wave4);
                    __syncthreads();
                    if (fix && t4 < 64) wk_jnorm_tile(Cs, T1, tl[slot].bi, t4);
                    __syncthreads();
                    if (fix) {
#pragma unroll
                        for (int mi = 0; mi < TG_MI; ++mi)
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                acc[slot][mi][0][q] = Cs[(wave4 * 16 + 4 * ((blk - q) & 3) + j) * TS_LD + mi * 16 + 4 * blk + i];
                    }
                }
            }
        });
        __syncthreads();
    }
    // ---- the strips whose flag words this workgroup polls: lane = (tile, operand, strip of the 64-row block)
    int my_flag = -1;
    if (t < 8 * WK_TILES) {
        const int s6 = t >> 3, sub = t & 7;                          // tile s6 = 2 slot + engine
        const WkTile tt = wk_tile_j(wk, map, s6, ntiles);
        if (tt.have) my_flag = 4 * rp_blocks + 4 * (sub < 4 ? tt.bi : tt.bj) + (sub & 3);
    }
    const double* Y = Ypanel + RP;
    const bool k11 = wk.T && sel[slot_k] != 0;
    for (int k = 0; k < nblk; ++k) {
        if (timing && t == 0 && k < SWD_K) dbg[(5 * SWD_K + k) * SWD_SLOT + 1] = wall_clock64();
        if (t < 64) {
            SwDeadline dl;
            while (true) {
                const int v = my_flag >= 0 ? ld_flag(&fl->y_flag[my_flag]) : k + 1;
                if (__all(v >= k + 1)) break;
                if (dl.expired()) { if (t == 0) { sw_timed_out(status, -37, k + 1, dl.n, dl.elapsed()); *abort = 1; } break; }
                __builtin_amdgcn_s_sleep(4);
                // (fault injection: a waiter that hardly runs -- ~7 us per poll, fewer than SW_WAIT_MIN_POLLS in 2 ms)
                if (exp_mask & 2048) { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); }
            }
        }
        __syncthreads();
        if (*abort) return;                                          // the host re-runs the update stage: nothing is written
        if (timing && t == 0 && k < SWD_K) dbg[(5 * SWD_K + k) * SWD_SLOT + 2] = wall_clock64();
        const int r_here = r_total - 64 * k;
        if (r_here > 32) wk_block<2>(acc, tl, n_wg, Y, ldA, 64 * k, hbase, wave4);
        else             wk_block<1>(acc, tl, n_wg, Y, ldA, 64 * k, hbase, wave4);
        if (timing && t == 0 && k < SWD_K) dbg[(5 * SWD_K + k) * SWD_SLOT + 3] = wall_clock64();
    }
    // ---- epilogue per tile: through LDS (coalesced stores of the tile and of its mirror); K11 on the first block column
    // (Jnorm congruence on rows / columns 3..6, ExtendKF.cpp:629-634).  The 27.5 MB all workers write at this point are what
    // the epilogue costs (HBM write rate); storing a tile straight from the accumulators behind its last chunk, under the
    // next tile's MFMAs, was tried and is slower: the chunk barrier's vmcnt(0) (it waits for the LDS-DMA) drains those stores too.
    const int t4 = t & 255, row = t4 & 63, g4 = t4 >> 6;
    double* Cs = hbase;
    static_for<0, WK_SLOTS>([&](auto S) {
        constexpr int slot = decltype(S)::value;
        if (slot < n_wg) {
            const bool mine = tl[slot].have;
            const bool fix = mine && k11 && tl[slot].bj == 0;
            const int bi = tl[slot].bi, bj = tl[slot].bj;
            __syncthreads();                                         // the operand buffers / the previous tile's image are free
            if (mine) tg_acc_to_lds_w(acc[slot], Cs, 1.0, wave4);
            __syncthreads();
            if (fix && t4 < 64) {
                // Jnorm comes from the strip of state rows 0..15 of this launch (xacc_finish), long ago
                SwDeadline dl;
                while (ld_flag(wk.xu_flag) < wk.token) {
                    if (dl.expired()) { sw_timed_out(status, -38, 0, dl.n, dl.elapsed()); break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                double T[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) T[q] = ld_coh(wk.T + q);
                wk_jnorm_tile(Cs, T, bi, t4);
            }
            __syncthreads();
            if (mine) {
                double* Cij = wk.Pout + 64L * bi + 64L * bj * wk.ldo;
                double* Cji = wk.Pout + 64L * bj + 64L * bi * wk.ldo;
#pragma unroll 4
                for (int q = 0; q < 16; ++q) {
                    const int c = g4 + 4 * q;
                    Cij[row + (long)c * wk.ldo] = Cs[c * TS_LD + row];
                }
                if (bi != bj) {
#pragma unroll 4
                    for (int q = 0; q < 16; ++q) {
                        const int c = g4 + 4 * q;
                        Cji[row + (long)c * wk.ldo] = Cs[row * TS_LD + c];
                    }
                }
            }
        }
    });
    if (timing && t == 0) dbg[(5 * SWD_K + 0) * SWD_SLOT + 4] = wall_clock64();
}

template <int NJ>
__global__ void __launch_bounds__(CD_THREADS)
sweep_persistent_kernel(double* A, long ldA, const int32_t* __restrict__ sel, int slot_nblk, int slot_k, int rp_blocks, int NP,
                        SysSrc src, double* Linv, double* Ypanel, int32_t* flags, int32_t* flags_other, int32_t* status,
                        unsigned long long* dbg, int exp_mask, WorkerArgs wk)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ int s_wk_abort;
    if (dbg && threadIdx.x == 0 && blockIdx.x < SWD_WG) dbg[SWD_WHO * SWD_K * SWD_SLOT + blockIdx.x] = wall_clock64();
    struct EndStamp {       // (diagnostic: when thread 0 of the workgroup leaves, whatever the path)
        unsigned long long* p;
        __device__ ~EndStamp() { if (p) *p = wall_clock64(); }
    } end_stamp{(dbg && threadIdx.x == 0 && blockIdx.x < SWD_WG) ? dbg + SWD_WHO * SWD_K * SWD_SLOT + SWD_WG + blockIdx.x : nullptr};
    // The hand-over flags are double-buffered between the two sweeps of a frame: this launch uses `flags` (all zero:
    // the previous sweep cleared them) and clears `flags_other` for the next one -- nobody is using that set now.
    if (blockIdx.x == 0 && threadIdx.x < SWEEP_FLAG_INTS) flags_other[threadIdx.x] = 0;
    SweepFlags* fl = reinterpret_cast<SweepFlags*>(flags);
    const bool fused = wk.Pout != nullptr;
    // (the consensus launch has done this low-innovation update itself: li_small_update -- nothing is left for this launch)
    if (fused && wk.token == 1 && wk.defer_flag && *wk.defer_flag == 2) return;
    int nblk = sel[slot_nblk];
    if (nblk > rp_blocks) nblk = rp_blocks;
    if (nblk <= 0) {                                          // no inliers: update() is the identity (ExtendKF.cpp:635-638)
        if (fused) {
            if (wk.token == 2 && wk.defer_flag && *wk.defer_flag != 0) wk_materialise_deferred(wk, NP, lds);   // ... of P_li, which does not exist yet
            else wk_passthrough(wk, NP);
        }
        return;
    }
    const int r_total = 2 * sel[slot_k];
    const bool single = (nblk == 1) && !(exp_mask & 4);       // one diagonal block: every strip factors it itself, no chain workgroup
    const int nstrips = (int)(ldA / 16);
    const int bx = (int)blockIdx.x;
    // A low-innovation update of rank <= 4 (the reference-faithful mode: the consensus set is the hypothesis' own feature) is
    // four microseconds of strip work and then one stream over P for a rank-2 correction.  That stream is DEFERRED: the
    // strips keep Y1 aside, the flag goes up, and until the high-innovation pass writes P every reader of P_li forms it
    // from P_pred, Y1 and this update's Jnorm (DeferArgs; rescue prediction, second P H^T, the HI pass's tile workers).
    const bool li_defer = fused && wk.token == 1 && wk.defer_flag && wk.Y1 && single && r_total <= 4 && !(exp_mask & (8 | 512));
    if (bx == 0) {
        if (li_defer && threadIdx.x == 0) *wk.defer_flag = 1;
        if (single) return;
        // (exp_mask & 16: fault injection for tests/test_gpu_parity.py -- the chain workgroup never shows up, as if it had not
        //  been scheduled: every strip must run into its bounded wait and the host must recover the frame)
        if (!(exp_mask & 16)) cd_chain_persistent(lds, A, ldA, nblk, r_total, Linv, Ypanel, fl, status, dbg, exp_mask, NP / 16 + 1);
        return;
    }
    if (fused) {
        // Workers: the blocks behind the strips, and the strips that hold no rows of this system -- S row blocks 0, 1 (the
        // lower strips assemble those for the chain) and nblk .. (padding), and the three strips below nu^T.
        int widx = -1;
        const int extra = (int)gridDim.x - 1 - nstrips;
        const int lo_blocks = rp_blocks < 2 ? rp_blocks : 2, hi0 = nblk > 2 ? nblk : 2;
        const int n_idle_s = 4 * (lo_blocks + (rp_blocks > hi0 ? rp_blocks - hi0 : 0));
        if (bx > nstrips) widx = bx - 1 - nstrips;
        else {
            const int strip = bx - 1, b = strip >> 2;
            if (b < rp_blocks) { if (b < 2) widx = extra + strip; else if (b >= hi0) widx = extra + 4 * lo_blocks + (strip - 4 * hi0); }
            else if (b == (int)(ldA / 64) - 1 && (strip & 3) != 0) widx = extra + n_idle_s + (strip & 3) - 1;
        }
        // A system of one diagonal block keeps its P H^T / nu strips busy for a few microseconds only: they join the workers
        // afterwards (the covariance pass is then a stream over P shared by every compute unit of the device).
        const int n_late = single ? NP / 16 + 1 : 0;
        const int n_workers = extra + n_idle_s + 3 + n_late;
        // The XCD-aware tile map (WkMap): the workers are the workgroups of up to five runs of block indices -- the strips of
        // S row blocks 0, 1; of the padding row blocks; the three strips below nu^T; the blocks behind the strips; and, for a
        // system of one diagonal block, the P H^T / nu strips that join late -- so a worker's rank inside its XCD and every
        // XCD's worker count are sums of residue counts over those runs.
        auto make_map = [&](int my_bx, int my_widx) {
            WkMap m; m.widx = my_widx; m.W = n_workers; m.lo = m.hi = m.y = 0; m.nx = 1; m.by_xcd = false;
            const int T = wk.nT * (wk.nT + 1) / 2;
            if (!wk.tile_order || T < 64 || (exp_mask & 4096)) return m;       // (bit 12: the round-3 assignment, for A/B measurements)
            const int ra[5] = {1, 4 * hi0 + 1, nstrips - 2, nstrips + 1, 4 * rp_blocks + 1};
            const int rb[5] = {4 * lo_blocks, 4 * rp_blocks, nstrips, (int)gridDim.x - 1, single ? 4 * rp_blocks + NP / 16 + 1 : 0};
            const int x = my_bx & 7;
            int before = 0, nx = 0, y = 0;
            for (int k = 0; k < 5; ++k) {
                for (int xx = 0; xx < x; ++xx) before += wk_count_res(ra[k], rb[k], xx);
                nx += wk_count_res(ra[k], rb[k], x);
                y += wk_count_res(ra[k], my_bx - 1 < rb[k] ? my_bx - 1 : rb[k], x);
            }
            m.lo = (int)((long)T * before / n_workers);
            m.hi = (int)((long)T * (before + nx) / n_workers);
            m.y = y; m.nx = nx;
            // every worker of every XCD must hold its share: the largest share per worker is ceil(T / W) + 1 (rounding of the runs)
            m.by_xcd = nx > 0 && (T + n_workers - 1) / n_workers + 1 <= WK_TILES;
            return m;
        };
        if (widx >= 0) {
            if (!li_defer) sweep_tile_worker(make_map(bx, widx), nblk, r_total, rp_blocks, ldA, Ypanel, wk, fl, sel, slot_k, status, lds, &s_wk_abort, dbg, exp_mask);
            return;
        }
        if (single && bx <= nstrips) {
            // (a second inlined copy of the worker: one call site for both measured 3 us per frame slower -- the allocator then
            //  keeps the strip's registers alive across the worker)
            const int strip = bx - 1;
            sweep_strip<NJ>(A, ldA, rp_blocks, nblk, strip, r_total, NP, src, Linv, Ypanel, fl, status, lds, dbg, single, sel, slot_k,
                            (exp_mask & 8) != 0, wk, exp_mask);
            if (li_defer) return;
            __syncthreads();                                  // the strip's LDS is free
            sweep_tile_worker(make_map(bx, extra + n_idle_s + 3 + (strip - 4 * rp_blocks)), nblk, r_total, rp_blocks, ldA, Ypanel, wk, fl, sel, slot_k,
                              status, lds, &s_wk_abort, nullptr, exp_mask);
            return;
        }
    }
    if (bx <= nstrips)
        sweep_strip<NJ>(A, ldA, rp_blocks, nblk, bx - 1, r_total, NP, src, Linv, Ypanel, fl, status, lds, dbg, single, sel, slot_k,
                        (exp_mask & 8) != 0, wk, exp_mask);
}

// dynamic LDS of the kernels that factor a diagonal block (the fused one also runs tile products in it)
constexpr size_t CD_STAGE_BYTES = sizeof(double) * (64 * CD_LD + 2 * 64 * CD_OPLD);      // Lf + the two staged operands
constexpr size_t cd_max(size_t a, size_t b) { return a > b ? a : b; }
constexpr size_t CD_LDS_BYTES = cd_max(cd_max(sizeof(CdShared), CD_STAGE_BYTES), sizeof(double) * 6 * TG_OPER_DOUBLES);
static_assert(CD_TW == 4 && CD_MW == 2, "the role dispatch of cd_factor_block is written out for 4 T waves and 2 M waves");
static_assert(offsetof(CdShared, Mf) == sizeof(double) * 64 * CD_LD, "the staged operands start where Mf starts");

#if defined(CD_TIMELINE)
int debug_read_cd_log(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cd_log), sizeof(unsigned long long) * 2 * 8 * 16 * 4) == hipSuccess ? 0 : -1; }
#endif
#if defined(CD_STAMPS) || defined(CD_SPINS)
int debug_read_cd_stamps(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cd_stamps), 128) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_cd_stamps), z, 128) != hipSuccess) return -1; }
    return 0;
}
#endif

int init_kernel_attributes()
{
    const int bytes = (int)(sizeof(double) * TG_LDS_DOUBLES);
    hipError_t e = hipSuccess;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(panel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(chol_diag_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CD_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CD_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(trail_stream2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SWP_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
#if !defined(RSLAM_DEV_ONLY_NJ12)      // (development builds of kernel experiments instantiate the C3 shape only: a third of the compile time)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_persistent_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SWP_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_persistent_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SWP_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_persistent_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SWP_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
#endif
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_persistent_kernel<12>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SWP_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    return 0;
}

// The persistent sweep needs every workgroup resident at once: one per CU (its dynamic LDS does not leave room for
// a second), at most 16 column blocks (the accumulators of a strip are a compile-time array).
// diagnostic time stamps of the persistent sweep (the last launch wins); off unless a buffer is installed
// (diagnostic variant of the library only, -DRSLAM_DEBUG: the product build has neither the buffer nor the entry point)
#if defined(RSLAM_DEBUG)
static unsigned long long* g_sweep_dbg = nullptr;
int debug_sweep_stamps(unsigned long long* out /* SWD_TOTAL, nullable = only (un)install */, int enable)
{
    const size_t bytes = sizeof(unsigned long long) * SWD_TOTAL;
    if (enable && !g_sweep_dbg) {
        if (hipMalloc((void**)&g_sweep_dbg, bytes) != hipSuccess) { g_sweep_dbg = nullptr; return -1; }
        (void)hipMemset(g_sweep_dbg, 0, bytes);
    }
    if (out && g_sweep_dbg) { if (hipMemcpy(out, g_sweep_dbg, bytes, hipMemcpyDeviceToHost) != hipSuccess) return -1; (void)hipMemset(g_sweep_dbg, 0, bytes); }
    if (!enable && g_sweep_dbg) { (void)hipFree(g_sweep_dbg); g_sweep_dbg = nullptr; }
    return 0;
}
static unsigned long long* sweep_dbg_buffer() { return g_sweep_dbg; }
#else
static unsigned long long* sweep_dbg_buffer() { return nullptr; }
#endif

// RSLAM_SWEEP_EXP (diagnostic variant of the library only; the product build always runs with mask 0): bit 0 no in-chain fetch of the next block, bit 1 (unused since round 4), bit 2 single-block systems take the
// shared route too, bit 3 no register-only route for systems of <= 4 rows, bit 4 fault injection (the chain
// workgroup does not run), bit 5 fault injection (the strips never announce their Y blocks: tile workers and the x update
// run into their bounded waits), bit 7 the rank update as a launch of its own (not fused into the sweep), bit 8 the time
// stamps of scripts/sweep_stamps.py come from the LI pass instead of the HI pass, bit 9 a low-innovation update of rank <= 4
// streams P at once instead of deferring its covariance to the high-innovation pass, bit 10 fault injection (the P H^T strips
// announce their first Y block 2 ms late), bit 11 fault injection (the tile workers' polls are throttled to ~7 us each), bit 12
// the tile workers' tiles dealt round-robin over the worker index (the round-3 assignment) instead of XCD by XCD (WkMap);
// set_sweep_exp_mask overrides the environment (tests)
#if defined(RSLAM_DEBUG)
static int g_sweep_exp_override = -1;
void set_sweep_exp_mask(int mask) { g_sweep_exp_override = mask; }
int sweep_exp_mask()
{
    static const int env = getenv("RSLAM_SWEEP_EXP") ? atoi(getenv("RSLAM_SWEEP_EXP")) : 0;
    return g_sweep_exp_override >= 0 ? g_sweep_exp_override : env;
}
static bool debug_env(const char* name) { return getenv(name) != nullptr; }      // measurement switches
#else
int sweep_exp_mask() { return 0; }
static bool debug_env(const char*) { return false; }
#endif

static int device_cus()
{
    static int cus = -1;
    if (cus < 0) {
        int dev = 0; hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 0;
    }
    return cus;
}

bool sweep_persistent_eligible(const SystemDims& d)
{
    static const bool off = debug_env("RSLAM_SWEEP_STEPS");       // measurement: the one-launch-per-step sequence
    if (off || d.RP <= 0) return false;
#if defined(RSLAM_DEV_ONLY_NJ12)
    if (d.RP / 64 > 12) return false;
#endif
    return d.RP / 64 <= 16 && 1 + d.ldA / 16 <= device_cus();
}

// The x and covariance update ride inside the persistent sweep when the workgroups it leaves idle -- the launch is as large
// as the device -- can hold every lower-triangle tile pair of P in their accumulators (WK_TILES each), whatever the inlier
// count turns out to be (fewest workers: every S row block in use).
bool sweep_fused_eligible(const SystemDims& d)
{
    static const bool off = debug_env("RSLAM_SWEEP_UNFUSED_K10");  // measurement: rank update as a launch of its own
    if (off || (sweep_exp_mask() & 128) || !sweep_persistent_eligible(d)) return false;
    const int rp_blocks = d.RP / 64, nT = d.NP / 64;
    const int workers = device_cus() - 1 - d.ldA / 16 + 4 * (rp_blocks < 2 ? rp_blocks : 2) + 3;
    return (long)nT * (nT + 1) / 2 <= (long)WK_TILES * workers;
}

// One stream: diag(0), then ONE launch per block step (sweep_step_kernel); for large systems (more than 512 tile
// workgroups at the first step) block steps in pairs: narrow pass, panel, wide pass (trail_stream2_kernel).
// (A two-stream lookahead variant and a three-kernels-per-step sequence existed in rounds 1-2 for measurement: the cross-stream
// event dependencies cost more than the trailing kernels they hide -- C3 frame 0.68 ms against 0.52 ms single-stream.)
// Returns the buffer whose rows [RP, RP + NP] hold Y and u^T afterwards (Ystore on every route).
double* launch_factor_sweep(hipStream_t s, const SystemDims& d,
                            const int32_t* sel, int slot_k, int slot_nblk, int cap_blocks, double* A, double* Ystore, double* Linv,
                            int32_t* status_sel, int32_t* flags, const SysSrc* src, const WorkerArgs* wk)
{
    const int rp_blocks = d.RP / 64;
    const int steps = cap_blocks < rp_blocks ? cap_blocks : rp_blocks;
    const int row_blocks = d.ldA / 64;
    const size_t lds_bytes = sizeof(double) * TG_LDS_DOUBLES;
    if (flags && src && sweep_persistent_eligible(d)) {
        // one launch for the whole sweep, sized for the largest inlier count the frame can have: the launch sequence
        // never depends on the previous frame (cap_blocks is not used)
        WorkerArgs wa{};                                   // Pout == nullptr: the caller launches the rank update itself
        if (wk && wk->Pout && sweep_fused_eligible(d)) wa = *wk;
        // fused: one workgroup per compute unit -- the ones behind the strips are tile workers
        const dim3 grid(wa.Pout ? device_cus() : 1 + d.ldA / 16), block(CD_THREADS);
        const int exp_mask = sweep_exp_mask();             // measurement / fault-injection switches
        const bool stamp_this = ((exp_mask & 256) != 0) == (slot_k == SEL_K_LI);      // time stamps: the HI pass, or (bit 8) the LI pass
        const int set = (slot_k == SEL_K_LI) ? 0 : 1;              // the LI and the HI sweep of a frame alternate between the two flag sets
        int32_t* fl_cur = flags + set * SWEEP_FLAG_INTS;
        int32_t* fl_other = flags + (1 - set) * SWEEP_FLAG_INTS;
#define SWP_LAUNCH(NJ) sweep_persistent_kernel<NJ><<<grid, block, SWP_LDS_BYTES, s>>>(A, d.ldA, sel, slot_nblk, slot_k, rp_blocks, d.NP, *src, Linv, Ystore, fl_cur, fl_other, status_sel, stamp_this ? sweep_dbg_buffer() : nullptr, exp_mask, wa)
#if defined(RSLAM_DEV_ONLY_NJ12)
        SWP_LAUNCH(12);
#else
        if (rp_blocks <= 4) SWP_LAUNCH(4);
        else if (rp_blocks <= 8) SWP_LAUNCH(8);
        else if (rp_blocks <= 12) SWP_LAUNCH(12);
        else SWP_LAUNCH(16);
#endif
#undef SWP_LAUNCH
        return Ystore;
    }
    if ((long)row_blocks * steps > 512) {
        // large system: block steps in pairs -- narrow pass, panel, wide pass (see trail_stream2_kernel)
        // (steps is the update's exact block count here -- the host has read it: enqueue_update -- or an upper bound: the
        //  kernels deal out the tiles of the device-side count, the grids only bound the number of workgroups)
        if (steps <= 0) return Ystore;
        const int cus = device_cus();
        chol_diag_kernel<<<dim3(1), dim3(CD_THREADS), CD_LDS_BYTES, s>>>(A, d.ldA, 0, sel, slot_nblk, slot_k, Linv, status_sel, 0);
        auto wide_pass = [&](const TrailPass& ps, const double* X) {
            const long nS = steps - ps.j_lo, nP = row_blocks - rp_blocks;
            const long total = nS * (nS + 1) / 2 + nP * nS;
            long W = (total + 1) / 2;
            if (W > cus - 1) W = cus - 1;
            if (W < 1) W = 1;
            trail_stream2_kernel<<<dim3(1 + (int)W), dim3(CD_THREADS), SWP_LDS_BYTES, s>>>(
                A, d.ldA, ps, sel, slot_nblk, slot_k, rp_blocks, row_blocks, Linv, status_sel, X);
        };
        // The panel of the first step of a pair has no launch of its own: the narrow pass forms the two panel blocks each of its
        // tiles needs itself (sweep_step_kernel restricted to column k+1: three 64^3 products per tile instead of one, ~120
        // tiles under an 18 us pivot chain), the diagonal workgroup its own (cd_factor_block, pending = 2), and stores the
        // blocks the wide pass will read into the second buffer -- the raw column k of A stays as it is, so nobody races.
        // Every solved panel therefore lives in Ystore (the panel launch of the second step writes there too).
        int step = 0;
        while (step < steps) {
            if (step + 1 >= steps) {                                  // the last column's panel: Y and u^T of that block
                panel_kernel<<<dim3(row_blocks), dim3(256), lds_bytes, s>>>(A, d.ldA, step, sel, slot_nblk, Linv, rp_blocks, Ystore);
                break;
            }
            const bool pair = step + 2 < steps;
            sweep_step_kernel<<<dim3(1 + row_blocks + (pair ? 1 : 0)), dim3(CD_THREADS), CD_LDS_BYTES, s>>>(
                A, d.ldA, step, sel, slot_nblk, slot_k, rp_blocks, row_blocks, Linv, Ystore, status_sel, pair ? 2 : 1);
            if (!pair) { ++step; continue; }
            panel_kernel<<<dim3(row_blocks), dim3(256), lds_bytes, s>>>(A, d.ldA, step + 1, sel, slot_nblk, Linv, rp_blocks, Ystore);
            wide_pass(TrailPass{step, step + 2}, Ystore);                     // panels k, k+1 onto everything from column k+2 on
            step += 2;
        }
        return Ystore;
    }
    if (steps <= 0) return Ystore;
    chol_diag_kernel<<<dim3(1), dim3(CD_THREADS), CD_LDS_BYTES, s>>>(A, d.ldA, 0, sel, slot_nblk, slot_k, Linv, status_sel, 0);
    for (int step = 0; step < steps; ++step)     // column jj = 0 also stores the panel, so the last step still has one
        sweep_step_kernel<<<dim3(1 + row_blocks * (steps - step)), dim3(CD_THREADS), CD_LDS_BYTES, s>>>(
            A, d.ldA, step, sel, slot_nblk, slot_k, rp_blocks, row_blocks, Linv, Ystore, status_sel, 0);
    return Ystore;
}

// The S stage of the staged route (staged_kernels.hip): the same block steps in pairs as the large-system branch above, on the
// r x r innovation covariance ALONE -- the S rows of the stacked system; P H^T and nu^T are not swept, the R stage solves them
// group by group.  After it: Linv holds the inverse of every diagonal block, the S rows of Ystore every panel block L(a, c),
// a > c (the narrow pass also stores tile (k+1, k)).  progress(f) is called behind each launch after which the diagonal
// blocks [0, f) are factored and the panel blocks of the columns [0, f - 1) are stored: what a group that ends at f needs.
// cus: compute units of the stream's CU mask (sizes the wide passes).
void launch_s_stage(hipStream_t s, const SystemDims& d, const int32_t* sel, int slot_k, int slot_nblk, int steps,
                    double* A, double* Ystore, double* Linv, int32_t* status_sel, int cus, const std::function<void(int)>& progress)
{
    const int rp_blocks = d.RP / 64;
    if (steps > rp_blocks) steps = rp_blocks;
    if (steps <= 0) return;
    const int row_blocks = rp_blocks;                              // S rows only
    const size_t lds_bytes = sizeof(double) * TG_LDS_DOUBLES;
    chol_diag_kernel<<<dim3(1), dim3(CD_THREADS), CD_LDS_BYTES, s>>>(A, d.ldA, 0, sel, slot_nblk, slot_k, Linv, status_sel, 0);
    progress(1);
    int step = 0;
    while (step + 1 < steps) {
        const bool pair = step + 2 < steps;
        sweep_step_kernel<<<dim3(1 + row_blocks + (pair ? 1 : 0)), dim3(CD_THREADS), CD_LDS_BYTES, s>>>(
            A, d.ldA, step, sel, slot_nblk, slot_k, rp_blocks, row_blocks, Linv, Ystore, status_sel, pair ? 2 : 1);
        progress(step + 2);
        if (!pair) break;
        panel_kernel<<<dim3(row_blocks), dim3(256), lds_bytes, s>>>(A, d.ldA, step + 1, sel, slot_nblk, Linv, rp_blocks, Ystore);
        const long nS = steps - (step + 2);
        const long total = nS * (nS + 1) / 2;
        long W = (total + 1) / 2;
        if (W > cus - 1) W = cus - 1;
        if (W < 1) W = 1;
        trail_stream2_kernel<<<dim3(1 + (int)W), dim3(CD_THREADS), SWP_LDS_BYTES, s>>>(
            A, d.ldA, TrailPass{step, step + 2}, sel, slot_nblk, slot_k, rp_blocks, row_blocks, Linv, status_sel, Ystore);
        progress(step + 3);
        step += 2;
    }
}

// ---------------------------------------------------------------------------
// K9: x_k_k = x + K (z - h) = x + Y u (ExtendKF.cpp:606), then quaternion
// normalisation and the Jnorm matrix (:613-627; Q6: exponent -3/2 is integer
// division => 1/|q|^2 in compat mode).
// ---------------------------------------------------------------------------
// x_k_k rows of one 16-row group; group 0 also normalises the quaternion, writes Jnorm (T) and, when
// `flag` is given, publishes flag = token for the workgroups of the same launch that apply Jnorm
__device__ __forceinline__ void xupdate_rows(int group, SystemDims d, const int32_t* __restrict__ sel, int slot_nblk, int slot_k,
                                             const double* __restrict__ A, const double* __restrict__ x_in,
                                             double* __restrict__ x_out, double* __restrict__ T, int compat,
                                             int32_t* flag, int token, double (*part)[17],
                                             double* __restrict__ Y1out = nullptr, long ldy1 = 0, int32_t* defer_flag = nullptr)
{
    // 16 rows x 16 K-slices per workgroup; fixed-order LDS reduction (bitwise reproducible)
    const int K = 64 * sel[slot_nblk];
    const int r = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int row = group * 16 + r;
    const double* Y = A + d.RP;
    const double* u = A + d.RP + d.NP;
    double acc = 0;
#pragma unroll 8
    for (int k = sl; k < K; k += 16) acc += Y[row + (long)k * d.ldA] * u[(long)k * d.ldA];     // (eight load pairs in flight)
    part[sl][r] = acc;
    // deferred covariance: the (at most four) columns of Y are kept aside (the padding columns of the system are zero)
    if (Y1out && sl < 4) Y1out[row + (long)sl * ldy1] = Y[row + (long)sl * d.ldA];
    if (defer_flag && group == 0 && threadIdx.x == 0) *defer_flag = 1;
    __syncthreads();
    if (sl == 0) {
        double ssum = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) ssum += part[q][r];
        x_out[row] = x_in[row] + ssum;
    }
    // the quaternion lives in rows 3..6 of group 0: normalisation and Jnorm ride along
    if (group != 0) return;
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x != 0) return;
    if (sel[slot_k] == 0) {
        if (flag) { __threadfence(); __hip_atomic_store(flag, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
        return;
    }
    double* x = x_out;
    const double qr = x[3], qx = x[4], qy = x[5], qz = x[6];
    const double q2 = qr * qr + qx * qx + qy * qy + qz * qz;
    const double nrm = sqrt(q2);
    x[3] = qr / nrm; x[4] = qx / nrm; x[5] = qy / nrm; x[6] = qz / nrm;
    const double scale = compat ? (1.0 / q2) : (1.0 / (q2 * nrm));
    const double rows[16] = {
        qx*qx+qy*qy+qz*qz, -qr*qx,           -qr*qy,           -qr*qz,
        -qx*qr,            qr*qr+qy*qy+qz*qz, -qx*qy,          -qx*qz,
        -qy*qr,            -qy*qx,           qr*qr+qx*qx+qz*qz, -qy*qz,
        -qz*qr,            -qz*qx,           -qz*qy,           qr*qr+qx*qx+qy*qy };
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) T[i + 4 * j] = scale * rows[4 * i + j];
    if (flag) { __threadfence(); __hip_atomic_store(flag, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
}

// ---------------------------------------------------------------------------
// K10: covariance rank-r update.  For each lower-triangle tile pair (bi >= bj):
//   C(bi,bj) = 1/2 (P(bi,bj) + P(bj,bi)^T) - Y_bi Y_bj^T,   C(bj,bi) = C(bi,bj)^T
// = ExtendKF.cpp:608-609 (P - K S K^T, then 1/2 (P + P^T)) with K S K^T = Y Y^T.
// Safe in place: a workgroup owns both tiles of its pair.
// ---------------------------------------------------------------------------
// (k11_lds, the Jnorm congruence on an LDS tile of the first block column: rank_common.h)
// MAT: the pass may have to start from a deferred P_li (MatArgs); an instantiation of its own so that the plain pass keeps its
// register count (two workgroups per compute unit)
template <bool MAT>
__global__ void __launch_bounds__(256)
rank_update_kernel(int nT, const double* Pin, long ldp, const double* __restrict__ Y, long ldy,
                   const int32_t* __restrict__ sel, int slot_nblk, int fixed_k, double* Pout, long ldo,
                   const int32_t* __restrict__ tile_order, const double* __restrict__ Tq /* nullable */, int slot_k, XuArgs xu,
                   MatArgs mat,
                   int rider0 /* first workgroup of the x update: 0 (then the tiles follow) or the number of tiles */,
                   unsigned long long* dbg /* nullable: [workgroup][4] = start, K loop entered, K loop done, end (100 MHz) + hw id */)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    // The first xu.groups workgroups are K9 riding along: x_k_k = x + Y u needs the same finished sweep as this
    // kernel and nothing of its output, so it does not deserve a launch of its own.  Group 0 also produces Jnorm,
    // which the bj == 0 tiles below wait for (they are dispatched after it and need it ~50 us later).
    if ((int)blockIdx.x >= rider0 && (int)blockIdx.x < rider0 + xu.groups) {
        // (xu.inject: fault injection for the tests -- riders dispatched behind the tiles never publish Jnorm, as if rider 0 had
        //  not found a slot: the first block column must run into its bounded wait and the host must re-run with the riders first)
        xupdate_rows((int)blockIdx.x - rider0, xu.d, sel, slot_nblk, slot_k, xu.A, xu.x_in, xu.x_out, xu.T, xu.compat,
                     (xu.inject && rider0 != 0) ? nullptr : xu.flag, xu.token, reinterpret_cast<double (*)[17]>(lds),
                     xu.Y1out, xu.ldy1, xu.defer_flag);
        return;
    }
    if (xu.riders_only) return;
    const int tile_index = rider0 == 0 ? (int)blockIdx.x - xu.groups : (int)blockIdx.x;
    if (dbg && threadIdx.x == 0) {
        dbg[8L * blockIdx.x + 0] = wall_clock64();
        unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        dbg[8L * blockIdx.x + 4] = ((unsigned long long)xcc << 32) | hw;
    }
    int bi, bj;
    if (tile_order) {
        // XCD-aware order (make_rank_update_order): the tiles one XCD's L2 sees form 8 x 8 regions of
        // the triangle, so it fetches ~16 of the Y row panels instead of all of them
        const int e = tile_order[tile_index];
        bi = e >> 16; bj = e & 0xffff;
    } else {
        // linear index -> (bi >= bj), row-major over the lower triangle
        const int t = tile_index;
        bi = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((long)bi * (bi + 1) / 2 > t) --bi;
        while ((long)(bi + 1) * (bi + 2) / 2 <= t) ++bi;
        bj = t - bi * (bi + 1) / 2;
    }
    (void)nT;
    const int K = (fixed_k >= 0) ? fixed_k : 64 * sel[slot_nblk];
    const int row = threadIdx.x & 63, g = threadIdx.x >> 6;
    // P_li deferred (MatArgs): this pass starts from P_pred and forms P_li = J (sym(P_pred) - Y1 Y1^T) J^T on the way
    const bool deferred = MAT && mat.flag && *mat.flag != 0;        // (uniform)
    if (deferred) { Pin = mat.Ppred; ldp = mat.ldp; }
    const double* Pij = Pin + (long)bi * 64 + (long)bj * 64 * ldp;
    const double* Pji = Pin + (long)bj * 64 + (long)bi * 64 * ldp;
    double* Cij = Pout + (long)bi * 64 + (long)bj * 64 * ldo;
    double* Cji = Pout + (long)bj * 64 + (long)bi * 64 * ldo;
    if (K == 0 && !deferred) {                      // update() pass-through, ExtendKF.cpp:635-638
        if (Pin != Pout) {
            for (int q = 0; q < 16; ++q) {
                const int c = g + 4 * q;
                Cij[row + (long)c * ldo] = Pij[row + (long)c * ldp];
                if (bi != bj) Cji[row + (long)c * ldo] = Pji[row + (long)c * ldp];
            }
        }
        return;
    }
    // The HI pass reads what the LI pass wrote: mirrored tile pairs, i.e. exactly symmetric off the diagonal tiles, so
    // 1/2 (P + P^T) = P there bit for bit and the mirror tile need not be read (a quarter of this pass's reads).  Not
    // when the LI pass was a pass-through (no inliers: its output is the prior as uploaded).
    // (xu.mirror_known: a later pass of a staged update -- the first pass of the same update wrote mirrored pairs)
    const bool mirror_known = !deferred && (bi != bj) && (xu.mirror_known || ((xu.token == 2) && (sel[SEL_NBLK_LI] > 0)));
    // Both P tiles are requested before the K loop and land under it: the epilogue is left with the stores only.
    double pij[16], pji[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c = g + 4 * q;
        pij[q] = Pij[row + (long)c * ldp];
        pji[q] = mirror_known ? 0.0 : Pji[row + (long)c * ldp];
    }
    TgAcc acc;
    tg_zero(acc);
    if (dbg && threadIdx.x == 0) dbg[8L * blockIdx.x + 1] = wall_clock64();
    tile_gemm_nt_dma(Y + (long)bi * 64, ldy, Y + (long)bj * 64, ldy, K, lds, acc);
    if (dbg && threadIdx.x == 0) dbg[8L * blockIdx.x + 2] = wall_clock64();
    double* Cs = lds;
    double* Ts = lds + TS_DOUBLES;
    double* Y1s = lds + 2 * TS_DOUBLES;             // 64 x 4 of block bj (deferred)
    double y1i[4] = {0.0, 0.0, 0.0, 0.0}, y1j = 0.0, Tli[16];
    if (MAT && deferred) {                          // (after the K loop: nothing of this lives in registers across it)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) y1i[cc] = mat.Y1[64L * bi + row + cc * mat.ldy1];
        y1j = mat.Y1[64L * bj + row + g * mat.ldy1];                 // thread (row, g): entry (row, g) of the 64 x 4 block of bj
        if (bj == 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) Tli[k] = mat.T_li[k];
        }
    }
    tg_acc_to_lds(acc, Cs, 1.0);
    if (!mirror_known) {
#pragma unroll
        for (int q = 0; q < 16; ++q) Ts[(g + 4 * q) * TS_LD + row] = pji[q];       // tile (bj,bi), element (row, c) -> Ts[c][row]
    }
    if (MAT && deferred) Y1s[row + 64 * g] = y1j;
    __syncthreads();
    double m[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c = g + 4 * q;
        const double pm = mirror_known ? pij[q] : Ts[row * TS_LD + c];           // P(bj,bi)[c, row]
        m[q] = 0.5 * pij[q] + 0.5 * pm;
    }
    if (MAT && deferred) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int c = g + 4 * q;
            m[q] -= (y1i[0] * Y1s[c] + y1i[1] * Y1s[c + 64]) + (y1i[2] * Y1s[c + 128] + y1i[3] * Y1s[c + 192]);
        }
        if (bj == 0) {                              // Jnorm of the low-innovation update on rows / columns 3..6 of M
            __syncthreads();                        // (Ts read by everybody)
#pragma unroll
            for (int q = 0; q < 16; ++q) Ts[(g + 4 * q) * TS_LD + row] = m[q];
            __syncthreads();
            k11_lds(Ts, Tli, bi);
#pragma unroll
            for (int q = 0; q < 16; ++q) m[q] = Ts[(g + 4 * q) * TS_LD + row];
        }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c = g + 4 * q;
        const double o = m[q] - Cs[c * TS_LD + row];
        Cij[row + (long)c * ldo] = o;
        Cs[c * TS_LD + row] = o;
    }
    __syncthreads();
    // K11 rides along: the Jnorm congruence on rows/columns 3..6 (ExtendKF.cpp:629-634) only touches the first
    // block row/column, i.e. the pairs with bj == 0; same arithmetic as a separate pass over P would do
    if (Tq && bj == 0 && sel[slot_k] != 0) {
        if (xu.groups > 0) {                        // Jnorm comes from group 0 of this launch
            SwDeadline dl;
            while (__hip_atomic_load(xu.flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < xu.token) {
                if (dl.expired()) { atomicMin(xu.flag + (SEL_STATUS - SEL_XU_FLAG), -39); break; }   // (xu.flag = sel + SEL_XU_FLAG)
                __builtin_amdgcn_s_sleep(2);
            }
        }
        double T[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) T[k] = Tq[k];
        const int j = threadIdx.x;                  // row of tile (bi,0) = column of its mirror
        if (bi != 0) {
            if (j < 64) {
                double rb[4];
                for (int i = 0; i < 4; ++i) {
                    double sacc = 0;
                    for (int k = 0; k < 4; ++k) sacc += T[i + 4 * k] * Cs[(3 + k) * TS_LD + j];   // P(3+k, col) by symmetry
                    rb[i] = sacc;
                }
                for (int i = 0; i < 4; ++i) { Cs[(3 + i) * TS_LD + j] = rb[i]; Cij[j + (long)(3 + i) * ldo] = rb[i]; }
            }
        } else {
            if (j < 64 && !(j >= 3 && j < 7)) {
                double rb[4];
                for (int i = 0; i < 4; ++i) {
                    double sacc = 0;
                    for (int k = 0; k < 4; ++k) sacc += T[i + 4 * k] * Cs[j * TS_LD + (3 + k)];
                    rb[i] = sacc;
                }
                for (int i = 0; i < 4; ++i) { Cs[j * TS_LD + (3 + i)] = rb[i]; Cs[(3 + i) * TS_LD + j] = rb[i]; }
            } else if (j == 3) {
                double cb[4][4], out[4][4];         // cb = J * P44 ; out = cb * J^T
                for (int i = 0; i < 4; ++i)
                    for (int c = 0; c < 4; ++c) {
                        double sacc = 0;
                        for (int k = 0; k < 4; ++k) sacc += T[i + 4 * k] * Cs[(3 + c) * TS_LD + (3 + k)];
                        cb[i][c] = sacc;
                    }
                for (int i = 0; i < 4; ++i)
                    for (int c = 0; c < 4; ++c) {
                        double sacc = 0;
                        for (int k = 0; k < 4; ++k) sacc += cb[i][k] * T[c + 4 * k];
                        out[i][c] = sacc;
                    }
                for (int i = 0; i < 4; ++i)
                    for (int c = 0; c < 4; ++c) Cs[(3 + c) * TS_LD + (3 + i)] = out[i][c];
            }
            __syncthreads();
#pragma unroll 4
            for (int q = 0; q < 16; ++q) {
                const int c = g + 4 * q;
                Cij[row + (long)c * ldo] = Cs[c * TS_LD + row];
            }
        }
        __syncthreads();
    }
    if (bi != bj) {
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const int c = g + 4 * q;
            Cji[row + (long)c * ldo] = Cs[row * TS_LD + c];
        }
    }
    if (dbg && threadIdx.x == 0) { dbg[8L * blockIdx.x + 3] = wall_clock64(); dbg[8L * blockIdx.x + 5] = ((unsigned long long)bi << 16) | bj; }
}

// Tile order for rank_update_kernel.  Workgroups are dealt round-robin over the 8 XCDs (block b
// and b + 8 share an L2; speed only, never correctness), so entry b holds the b/8-th tile of the
// (b % 8)-th contiguous slice of a region-major enumeration: 8 x 8 regions of the lower triangle.
void make_rank_update_order(int nT, std::vector<int32_t>& order)
{
    std::vector<int32_t> seq;
    for (int sr = 0; sr < nT; sr += 8)
        for (int sc = 0; sc <= sr; sc += 8)
            for (int bi = sr; bi < sr + 8 && bi < nT; ++bi)
                for (int bj = sc; bj < sc + 8 && bj <= bi; ++bj) seq.push_back((bi << 16) | bj);
    const int T = (int)seq.size(), q = T / 8, r = T % 8;
    order.assign(T, 0);
    for (int b = 0; b < T; ++b) {
        const int x = b % 8, idx = b / 8;
        order[b] = seq[x * q + (x < r ? x : r) + idx];
    }
}

// diagnostic time stamps of the rank update (the last launch wins); off unless a buffer is installed
constexpr int K10_DBG_WGS = 8192;
#if defined(RSLAM_DEBUG)
static unsigned long long* g_k10_dbg = nullptr;
int debug_k10_stamps(unsigned long long* out /* K10_DBG_WGS * 8, nullable */, int enable)
{
    const size_t bytes = sizeof(unsigned long long) * K10_DBG_WGS * 8;
    if (enable && !g_k10_dbg) {
        if (hipMalloc((void**)&g_k10_dbg, bytes) != hipSuccess) { g_k10_dbg = nullptr; return -1; }
        (void)hipMemset(g_k10_dbg, 0, bytes);
    }
    if (out && g_k10_dbg) { if (hipMemcpy(out, g_k10_dbg, bytes, hipMemcpyDeviceToHost) != hipSuccess) return -1; (void)hipMemset(g_k10_dbg, 0, bytes); }
    if (!enable && g_k10_dbg) { (void)hipFree(g_k10_dbg); g_k10_dbg = nullptr; }
    return 0;
}
static unsigned long long* k10_dbg_buffer() { return g_k10_dbg; }
#else
static unsigned long long* k10_dbg_buffer() { return nullptr; }
#endif

void launch_rank_update(hipStream_t s, int NP, const double* Pin, long ldp, const double* Y, long ldy,
                        const int32_t* sel, int slot_nblk, int fixed_k, double* Pout, long ldo,
                        const int32_t* tile_order, const double* Tq, int slot_k, const XuArgs* xu, const MatArgs* mat,
                        int n_tiles /* >= 0: only the first n_tiles entries of tile_order (the tiles a macro-tile launch left over) */)
{
    const int nT = NP / 64;
    int tiles = (n_tiles >= 0 && tile_order) ? n_tiles : nT * (nT + 1) / 2;
    if (tiles <= 0) return;
    XuArgs none{}; none.groups = 0;
    const XuArgs& x = xu ? *xu : none;
    MatArgs m{}; if (mat) m = *mat;
    if (x.riders_only) {                            // the x update alone (deferred covariance): no tile workgroups at all
        if (x.groups <= 0) return;
        rank_update_kernel<false><<<dim3(x.groups), dim3(256), sizeof(double) * TG_LDS_DOUBLES, s>>>(nT, Pin, ldp, Y, ldy, sel, slot_nblk,
                                                                                     fixed_k, Pout, ldo, tile_order, Tq, slot_k, x, m, 0, nullptr);
        return;
    }
    // Where the x update rides: in front of the tiles (dispatched first: its Jnorm is out long before the tiles of the first
    // block column ask for it) -- unless every tile finds a slot at once (two workgroups per CU) and slots are left over:
    // then the riders go last, group 0 still starts at once, and no tile waits for a rider to vacate its slot (the late
    // tiles ended the launch ~2 us late at C3).
    // (slots: what the device can hold of this kernel at once, from its real occupancy -- registers and dynamic LDS)
    // (one value per instantiation: the MAT one needs more registers and may hold fewer workgroups per compute unit)
    static int slots_of[2] = {-1, -1};
    int& slots = slots_of[mat ? 1 : 0];
    if (slots < 0) {
        int per_cu = 0;
        const void* fn = mat ? reinterpret_cast<const void*>(rank_update_kernel<true>) : reinterpret_cast<const void*>(rank_update_kernel<false>);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, sizeof(double) * TG_LDS_DOUBLES) != hipSuccess) per_cu = 0;
        slots = per_cu * device_cus();
    }
    static const bool riders_first = debug_env("RSLAM_K10_RIDERS_FIRST");      // measurement
    // behind the tiles only when tiles AND riders all find a slot at once (then rider 0 is resident whatever the dispatch order);
    // x.riders_first: the host saw a timed-out Jnorm wait (somebody else holds compute units) and re-runs the safe order
    const int rider0 = (x.groups > 0 && tiles + x.groups <= slots && !riders_first && !x.riders_first) ? tiles : 0;
    if (mat)
        rank_update_kernel<true><<<dim3(x.groups + tiles), dim3(256), sizeof(double) * TG_LDS_DOUBLES, s>>>(nT, Pin, ldp, Y, ldy, sel, slot_nblk,
                                                                                                   fixed_k, Pout, ldo, tile_order, Tq, slot_k, x, m, rider0,
                                                                                                   (x.groups + tiles <= K10_DBG_WGS) ? k10_dbg_buffer() : nullptr);
    else
        rank_update_kernel<false><<<dim3(x.groups + tiles), dim3(256), sizeof(double) * TG_LDS_DOUBLES, s>>>(nT, Pin, ldp, Y, ldy, sel, slot_nblk,
                                                                                                    fixed_k, Pout, ldo, tile_order, Tq, slot_k, x, m, rider0,
                                                                                                    (x.groups + tiles <= K10_DBG_WGS) ? k10_dbg_buffer() : nullptr);
}

// ---------------------------------------------------------------------------
// EKF prediction, ExtendKF::ekf_prediction, ExtendKF.cpp:333-388, for the
// "constant_velocity" filter the reference instantiates (System.cpp:63).  The 13-state motion
// model (fv :389-400, dfv_by_dxv :444-465, Q = G Pn G' :347-376) is one lane's work; the
// covariance only changes in its first 13 rows/columns (:379-387): a strip kernel.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void v2q_dev(const double v[3], double q[4])
{
    const double theta = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (theta < 2.220446049250313e-16) { q[0] = q[1] = q[2] = q[3] = 0; return; }     // eps, ExtendKF.cpp:24,436
    const double vn[3] = { v[0] / theta, v[1] / theta, v[2] / theta };
    const double nn = sqrt(vn[0] * vn[0] + vn[1] * vn[1] + vn[2] * vn[2]);
    double sh, ch;
    sincos(theta / 2.0, &sh, &ch);
    q[0] = ch;
    for (int a = 0; a < 3; ++a) q[1 + a] = sh * (vn[a] / nn);
}

__global__ void ekf_motion_kernel(const double* __restrict__ x_kk, double dt, double std_a, double std_alpha,
                                  double* __restrict__ x_pred, double* __restrict__ FQ)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double* F = FQ; double* Q = FQ + 169;
    const double* rW = x_kk; const double* q = x_kk + 3; const double* vW = x_kk + 7; const double* wW = x_kk + 10;
    for (int a = 0; a < 3; ++a) x_pred[a] = rW[a] + vW[a] * dt;
    {   // qprod, ExtendKF.cpp:416-427
        const double v[3] = { wW[0] * dt, wW[1] * dt, wW[2] * dt };
        double p[4];
        v2q_dev(v, p);
        const double* qv = q + 1; const double* pu = p + 1;
        x_pred[3] = q[0] * p[0] - (qv[0] * pu[0] + qv[1] * pu[1] + qv[2] * pu[2]);
        x_pred[4] = (q[0] * pu[0] + p[0] * qv[0]) + (qv[1] * pu[2] - qv[2] * pu[1]);
        x_pred[5] = (q[0] * pu[1] + p[0] * qv[1]) + (qv[2] * pu[0] - qv[0] * pu[2]);
        x_pred[6] = (q[0] * pu[2] + p[0] * qv[2]) + (qv[0] * pu[1] - qv[1] * pu[0]);
    }
    for (int a = 0; a < 3; ++a) { x_pred[7 + a] = vW[a]; x_pred[10 + a] = wW[a]; }
    for (int k = 0; k < 169; ++k) { F[k] = 0.0; Q[k] = 0.0; }
    for (int i = 0; i < 13; ++i) F[i + 13 * i] = 1.0;
    {
        const double wt[3] = { wW[0] * dt, wW[1] * dt, wW[2] * dt };
        double qw[4];
        v2q_dev(wt, qw);
        const double blk[16] = { qw[0], -qw[1], -qw[2], -qw[3],  qw[1], qw[0], qw[3], -qw[2],
                                 qw[2], -qw[3], qw[0], qw[1],    qw[3], qw[2], -qw[1], qw[0] };
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) F[(3 + i) + 13 * (3 + j)] = blk[4 * i + j];
    }
    for (int a = 0; a < 3; ++a) F[a + 13 * (7 + a)] = dt;
    // dq3_by_dq1(qOld) * dqomegadt_by_domega(omegaOld, dt), ExtendKF.cpp:463-465,482-529
    double ab[12];
    {
        const double a44[16] = { q[0], -q[1], -q[2], -q[3],  q[1], q[0], -q[3], q[2],
                                 q[2], q[3], q[0], -q[1],    q[3], -q[2], q[1], q[0] };      // row-major
        const double m = sqrt(wW[0] * wW[0] + wW[1] * wW[1] + wW[2] * wW[2]);
        double sh, ch;
        sincos(m * dt / 2.0, &sh, &ch);
        double b43[12];                                                                     // [i][j] row-major 4x3
        for (int j = 0; j < 3; ++j) b43[j] = (-dt / 2.0) * (wW[j] / m) * sh;
        for (int a = 0; a < 3; ++a)
            for (int j = 0; j < 3; ++j)
                b43[3 * (1 + a) + j] = (a == j)
                    ? (dt / 2.0) * wW[a] * wW[a] / (m * m) * ch + (1.0 / m) * (1.0 - wW[a] * wW[a] / (m * m)) * sh
                    : (wW[a] * wW[j] / (m * m)) * ((dt / 2.0) * ch - (1.0 / m) * sh);
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 3; ++j) {
                double sacc = 0;
                for (int k = 0; k < 4; ++k) sacc += a44[4 * i + k] * b43[3 * k + j];
                ab[i + 4 * j] = sacc;
                F[(3 + i) + 13 * (10 + j)] = sacc;
            }
    }
    {   // Q = G Pn G'
        const double la = (std_a * dt) * (std_a * dt), aa = (std_alpha * dt) * (std_alpha * dt);
        double G[13 * 6];
        for (int k = 0; k < 78; ++k) G[k] = 0.0;
        for (int a = 0; a < 3; ++a) { G[(7 + a) + 13 * a] = 1.0; G[(10 + a) + 13 * (3 + a)] = 1.0; G[a + 13 * a] = dt; }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 3; ++j) G[(3 + i) + 13 * (3 + j)] = ab[i + 4 * j];
        for (int i = 0; i < 13; ++i)
            for (int j = 0; j < 13; ++j) {
                double sacc = 0;
                for (int k = 0; k < 6; ++k) sacc += (G[i + 13 * k] * (k < 3 ? la : aa)) * G[j + 13 * k];
                Q[i + 13 * j] = sacc;
            }
    }
}

// Pout must already hold a copy of P (the bottom-right block pk_km5 is unchanged).
__global__ void __launch_bounds__(256)
ekf_cov_kernel(int n, int NP, const double* __restrict__ P, const double* __restrict__ FQ, double* __restrict__ Pout)
{
    __shared__ double F[169], Q[169];
    for (int k = threadIdx.x; k < 169; k += 256) { F[k] = FQ[k]; Q[k] = FQ[169 + k]; }
    __syncthreads();
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    if (j >= 13) {
        double col[13], rowv[13];
        for (int k = 0; k < 13; ++k) { col[k] = P[k + (long)j * NP]; rowv[k] = P[j + (long)k * NP]; }
        for (int i = 0; i < 13; ++i) {
            double s1 = 0, s2 = 0;
            for (int k = 0; k < 13; ++k) { s1 += F[i + 13 * k] * col[k]; s2 += rowv[k] * F[i + 13 * k]; }
            Pout[i + (long)j * NP] = s1;            // pk_km3 = F * P12
            Pout[j + (long)i * NP] = s2;            // pk_km4 = P21 * F'
        }
    } else {                                        // column j of pk_km2 = (F * P11) * F' + Q
        for (int i = 0; i < 13; ++i) {
            double sacc = 0;
            for (int k = 0; k < 13; ++k) {
                double fp = 0;                      // (F * P11)(i, k)
                for (int mm = 0; mm < 13; ++mm) fp += F[i + 13 * mm] * P[mm + (long)k * NP];
                sacc += fp * F[j + 13 * k];
            }
            Pout[i + (long)j * NP] = sacc + Q[i + 13 * j];
        }
    }
}

void launch_ekf_prediction(hipStream_t s, int n, int NP, const double* x_kk, const double* P_kk, double dt,
                           double std_a, double std_alpha, double* x_pred, double* P_pred, double* FQ)
{
    ekf_motion_kernel<<<dim3(1), dim3(64), 0, s>>>(x_kk, dt, std_a, std_alpha, x_pred, FQ);
    ekf_cov_kernel<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(n, NP, P_kk, FQ, P_pred);
}

// ---------------------------------------------------------------------------
// Generic dense NT GEMM on the same MFMA tile engine (dense form of P*H^T).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
gemm_nt_kernel(int K, double alpha, const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb,
               double beta, double* __restrict__ C, long ldc)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int bi = blockIdx.x, bj = blockIdx.y;
    TgAcc acc;
    tg_zero(acc);
    tile_gemm_nt_dma(A + (long)bi * 64, lda, B + (long)bj * 64, ldb, K, lds, acc);
    tg_acc_to_lds(acc, lds, alpha);
    __syncthreads();
    double* Ct = C + (long)bi * 64 + (long)bj * 64 * ldc;
    const int row = threadIdx.x & 63, g = threadIdx.x >> 6;
    for (int q = 0; q < 16; ++q) {
        const int c = g + 4 * q;
        const double prev = (beta != 0.0) ? beta * Ct[row + (long)c * ldc] : 0.0;
        Ct[row + (long)c * ldc] = lds[c * TS_LD + row] + prev;
    }
}

void launch_gemm_nt(hipStream_t s, int M, int N, int K, double alpha, const double* A, long lda,
                    const double* B, long ldb, double beta, double* C, long ldc)
{
    gemm_nt_kernel<<<dim3(M / 64, N / 64), dim3(256), sizeof(double) * TG_LDS_DOUBLES, s>>>(K, alpha, A, lda, B, ldb, beta, C, ldc);
}

int init_kernel_attributes2()
{
    const int bytes = (int)(sizeof(double) * TG_LDS_DOUBLES);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rank_update_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(rank_update_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (debug_env("RSLAM_DEBUG_OCCUPANCY")) {
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(rank_update_kernel<false>), 256, (size_t)bytes);
        fprintf(stderr, "[rslam] rank_update_kernel: %d workgroups per CU at %d B of dynamic LDS\n", nb, bytes);
    }
    return (int)e;
}

// ---------------------------------------------------------------------------
// Drop-in API: the covariance crosses PCIe as ONE linear transfer of the caller's n x n matrix (a pitched 2-D copy of 1813
// columns of 14.5 KB each runs at a fraction of the link rate); the change of leading dimension n <-> NP and the zero
// padding happen on the device, at HBM speed.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
repitch_kernel(const double* __restrict__ src, long lds_, double* __restrict__ dst, long ldd, int rows_src, int rows_dst, int cols_src)
{
    // one column of the destination per blockIdx.y; rows beyond the source's (and columns beyond it) are zero
    const int c = blockIdx.y;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < rows_dst; r += gridDim.x * 256)
        dst[r + (long)c * ldd] = (r < rows_src && c < cols_src) ? src[r + (long)c * lds_] : 0.0;
}

void launch_repitch(hipStream_t s, const double* src, long ld_src, double* dst, long ld_dst, int rows_src, int rows_dst, int cols_src, int cols_dst)
{
    if (rows_dst <= 0 || cols_dst <= 0) return;
    repitch_kernel<<<dim3((rows_dst + 255) / 256 > 8 ? 8 : (rows_dst + 255) / 256, cols_dst), dim3(256), 0, s>>>(src, ld_src, dst, ld_dst, rows_src, rows_dst, cols_src);
}

// ---------------------------------------------------------------------------
// Probes: FP64 MFMA issue rate and streaming-copy bandwidth of this device.
// ---------------------------------------------------------------------------
// mode 0: 4 independent 16x16x4 accumulators per wave; 1: 8 accumulators; 2: v_mfma_f64_4x4x4_4b (4 x 512 flop)
__global__ void __launch_bounds__(256)
mfma_probe_kernel(int iters, int mode, double* out, unsigned long long* stamps)
{
    const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    if (mode == 0) {
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
    } else if (mode == 1) {
        for (int i = 0; i < iters; i += 2) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c4, 0, 0, 0);
            c5 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c5, 0, 0, 0);
            c6 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c6, 0, 0, 0);
            c7 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c7, 0, 0, 0);
        }
    } else {
        for (int i = 0; i < iters; ++i) {
            s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s1, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s2, 0, 0, 0);
            s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s3, 0, 0, 0);
        }
    }
    const d4 sum = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    const double ssum = sum[0] + s0 + s1 + s2 + s3;
    asm volatile("" :: "v"(ssum));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (ssum == -1.0) out[blockIdx.x * 256 + threadIdx.x] = sum[1] + sum[2] + sum[3];
    if (stamps && threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}

void launch_mfma_probe(hipStream_t s, int blocks, int iters, int mode, double* out, unsigned long long* stamps)
{
    mfma_probe_kernel<<<dim3(blocks), dim3(256), 0, s>>>(iters, mode, out, stamps);
}

// one v_mfma_f64_4x4x4_4b_f64 with caller-chosen per-lane operands (layout discovery / unit test)
template <int CBSZ, int ABID>
__global__ void __launch_bounds__(64)
mfma4_raw_kernel(const double* a, const double* b, const double* c, double* d)
{
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], CBSZ, ABID, 0);
}

void launch_mfma4_raw(hipStream_t s, int cbsz, int abid, const double* a, const double* b, const double* c, double* d)
{
    if (cbsz == 0) mfma4_raw_kernel<0, 0><<<1, 64, 0, s>>>(a, b, c, d);
    else if (cbsz == 2 && abid == 0) mfma4_raw_kernel<2, 0><<<1, 64, 0, s>>>(a, b, c, d);
    else if (cbsz == 2 && abid == 1) mfma4_raw_kernel<2, 1><<<1, 64, 0, s>>>(a, b, c, d);
    else if (cbsz == 2 && abid == 2) mfma4_raw_kernel<2, 2><<<1, 64, 0, s>>>(a, b, c, d);
    else if (cbsz == 2 && abid == 3) mfma4_raw_kernel<2, 3><<<1, 64, 0, s>>>(a, b, c, d);
    else if (cbsz == 1 && abid == 0) mfma4_raw_kernel<1, 0><<<1, 64, 0, s>>>(a, b, c, d);
    else if (cbsz == 1 && abid == 1) mfma4_raw_kernel<1, 1><<<1, 64, 0, s>>>(a, b, c, d);
}

__global__ void __launch_bounds__(256)
copy_probe_kernel(const d2* __restrict__ src, d2* __restrict__ dst, long n2)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256) dst[i] = src[i];
}

void launch_copy_probe(hipStream_t s, const double* src, double* dst, long n)
{
    copy_probe_kernel<<<dim3(2048), dim3(256), 0, s>>>(reinterpret_cast<const d2*>(src), reinterpret_cast<d2*>(dst), n / 2);
}

}  // namespace rslam
