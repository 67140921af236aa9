// match_kernels.hip -- the NCC search of Tracking::matching (src/Tracking.cpp:279-351) with the
// correlation of Converter::corrcoef_opencv (src/Converter.cpp:188-209), one workgroup per feature.
//
// The reference builds, per feature, a 169 x (candidates + 1) double matrix of patches and the FULL
// (candidates + 1)^2 correlation matrix, of which it reads one row (Tracking.cpp:338-340).  Here
// the search window of the 8-bit image is staged once in LDS, every thread scores candidates
// against the predicted patch directly, and the first maximum (Eigen's maxCoeff order: column j
// outer, row i inner) is selected by a (value, index) reduction.  Byte traffic: the image window
// (<= 53 x 53 B) and the 169-double patch per feature -- latency-bound, no roofline claim.
#include "kernels.h"

namespace rslam {

constexpr int MT_THREADS = 256;
constexpr int MT_HALF = 6;                         // half_patch_size_when_matching, Map.cpp:294
constexpr int MT_SIDE = 2 * MT_HALF + 1;
constexpr int MT_NPIX = MT_SIDE * MT_SIDE;
constexpr int MT_MAX_HS = 20;                      // ceil(2 sqrt(S_ii)) with the largest eigenvalue of S below 100
constexpr int MT_WIN = 2 * (MT_MAX_HS + MT_HALF) + 1;

__global__ void __launch_bounds__(MT_THREADS)
match_kernel(Cam cam, const uint8_t* __restrict__ image, const double* __restrict__ patches, int L,
             const double* __restrict__ h, const uint8_t* __restrict__ has_h, const double* __restrict__ S,
             double corr_threshold, double chi2, double* __restrict__ z, uint8_t* __restrict__ ic,
             double* __restrict__ corr_out)
{
    __shared__ double P[MT_NPIX];                  // predicted patch (float32 values), then centred
    __shared__ uint8_t W[MT_WIN * MT_WIN];         // image window, row-major, pitch ww
    __shared__ double red[MT_THREADS];
    __shared__ int redi[MT_THREADS];
    __shared__ double stat[2];
    const int f = blockIdx.x, t = threadIdx.x;
    if (f >= L) return;
    bool search = has_h[f] != 0;
    const double S0 = S[4 * f], S1 = S[4 * f + 1], S2 = S[4 * f + 2], S3 = S[4 * f + 3];
    if (search) {
        // SelfAdjointEigenSolver reads the lower triangle (Tracking.cpp:301-303)
        const double lmax = 0.5 * (S0 + S3) + sqrt(0.25 * (S0 - S3) * (S0 - S3) + S1 * S1);
        search = lmax < 100.0;
    }
    if (!search) {                                  // uniform over the workgroup
        if (t == 0) { ic[f] = 0; corr_out[f] = -2.0; }
        return;
    }
    const double h0 = h[2 * f], h1 = h[2 * f + 1];
    const int hsx = (int)ceil(2 * sqrt(S0)), hsy = (int)ceil(2 * sqrt(S3));
    const int x0 = (int)round(h0), y0 = (int)round(h1);
    const int nC = cam.nCols, nR = cam.nRows;
    // window of image pixels any admissible candidate can touch, clipped to the image
    const int wx0 = max(x0 - hsx - MT_HALF, 0), wx1 = min(x0 + hsx + MT_HALF, nC - 1);
    const int wy0 = max(y0 - hsy - MT_HALF, 0), wy1 = min(y0 + hsy + MT_HALF, nR - 1);
    const int ww = wx1 - wx0 + 1, wh = wy1 - wy0 + 1;
    if (ww > 0 && wh > 0)
        for (int e = t; e < ww * wh; e += MT_THREADS) W[e] = image[(long)(wy0 + e / ww) * nC + wx0 + e % ww];
    // predicted patch through float32 (toCvMat_f, Converter.cpp:193), mean and centred energy
    if (t < MT_NPIX) P[t] = (double)(float)patches[(long)f * MT_NPIX + t];
    __syncthreads();
    if (t == 0) {
        double pm = 0;
        for (int k = 0; k < MT_NPIX; ++k) pm += P[k];
        pm /= MT_NPIX;
        double pe = 0;
        for (int k = 0; k < MT_NPIX; ++k) pe += (P[k] - pm) * (P[k] - pm);
        stat[0] = pm; stat[1] = pe;
    }
    __syncthreads();
    const double pm = stat[0], pe = stat[1];
    __syncthreads();
    if (t < MT_NPIX) P[t] -= pm;
    __syncthreads();
    double Si[4] = {S0, S1, S2, S3}, Sinv[4];
    inv2_lu(Si, Sinv);                              // dynamic-size .inverse() = PartialPivLU (Tracking.cpp:321)
    const int ny = 2 * hsy + 1, ncand = (2 * hsx + 1) * ny;
    double best = -2.0; int besti = 0x7fffffff;
    for (int c = t; c < ncand; c += MT_THREADS) {   // c ascending per thread: strict > keeps the first maximum
        const int j = x0 - hsx + c / ny, i = y0 - hsy + c % ny;
        const double n0 = j - h0, n1 = i - h1;
        const double t0 = n0 * Sinv[0] + n1 * Sinv[1], t1 = n0 * Sinv[2] + n1 * Sinv[3];
        const double d2 = t0 * n0 + t1 * n1;
        if (!(d2 < chi2)) continue;
        if (!((j > MT_HALF) && (j < nC - MT_HALF) && (i > MT_HALF) && (i < nR - MT_HALF))) continue;
        const uint8_t* w = W + (i - MT_HALF - wy0) * ww + (j - MT_HALF - wx0);
        int sum = 0;
        for (int cc = 0; cc < MT_SIDE; ++cc)
            for (int r = 0; r < MT_SIDE; ++r) sum += w[r * ww + cc];
        const double cm = (double)sum / MT_NPIX;
        double ce = 0, pc = 0;
        for (int cc = 0; cc < MT_SIDE; ++cc)
            for (int r = 0; r < MT_SIDE; ++r) {
                const double v = w[r * ww + cc] - cm;
                ce += v * v;
                pc += P[r + MT_SIDE * cc] * v;
            }
        const double corr = pc / sqrt(pe * ce);
        if (corr > best) { best = corr; besti = c; }
    }
    red[t] = best; redi[t] = besti;
    __syncthreads();
    for (int sft = MT_THREADS / 2; sft > 0; sft >>= 1) {
        if (t < sft) {
            const double ov = red[t + sft]; const int oi = redi[t + sft];
            if (ov > red[t] || (ov == red[t] && oi < redi[t])) { red[t] = ov; redi[t] = oi; }
        }
        __syncthreads();
    }
    if (t == 0) {
        const int c = redi[0];
        const bool any = c != 0x7fffffff;
        const bool ok = any && red[0] > corr_threshold;
        ic[f] = ok ? 1 : 0;
        corr_out[f] = any ? red[0] : -2.0;
        if (ok) { z[2 * f] = x0 - hsx + c / ny; z[2 * f + 1] = y0 - hsy + c % ny; }
    }
}

void launch_match(hipStream_t s, const Cam& cam, const uint8_t* image, const double* patches, int L, const double* h,
                  const uint8_t* has_h, const double* S, double corr_threshold, double chi2, double* z, uint8_t* ic,
                  double* corr)
{
    if (L <= 0) return;
    match_kernel<<<dim3(L), dim3(MT_THREADS), 0, s>>>(cam, image, patches, L, h, has_h, S, corr_threshold, chi2, z, ic, corr);
}

}  // namespace rslam

// ---------------------------------------------------------------------------
// Tracking::pred_patch_fc (src/Tracking.cpp:164-278): the 13 x 13 patch a feature is expected
// to show in the current image = its 41 x 41 initialisation patch warped by the plane-induced
// homography between the two camera poses, through the distortion model, sampled as
// cv::remap(INTER_LINEAR, BORDER_CONSTANT) samples a CV_32F image (coordinates quantised to
// 1/32 pixel, float weights, float accumulation).  One workgroup per feature: lane 0 sets up
// the homography (a few hundred flops), 169 lanes warp one pixel each.
// ---------------------------------------------------------------------------
namespace rslam {

constexpr int PP_HALF_F = 20;                      // half_patch_size_when_initialized, Map.cpp:293
constexpr int PP_SIDE_F = 2 * PP_HALF_F + 1;
constexpr int PP_NPIX_F = PP_SIDE_F * PP_SIDE_F;
constexpr int PP_REC = 14;                         // uv(2) R(9, col-major) r(3)

__device__ static void pp_inv3(const double M[9], double R[9]) { inv3(M, R); }

__device__ static void pp_inv4(const double m[16], double r[16])
{
#define A(i, j) m[(i) + 4 * (j)]
#define R_(i, j) r[(i) + 4 * (j)]
    const double s0 = A(0,0) * A(1,1) - A(1,0) * A(0,1), s1 = A(0,0) * A(1,2) - A(1,0) * A(0,2);
    const double s2 = A(0,0) * A(1,3) - A(1,0) * A(0,3), s3 = A(0,1) * A(1,2) - A(1,1) * A(0,2);
    const double s4 = A(0,1) * A(1,3) - A(1,1) * A(0,3), s5 = A(0,2) * A(1,3) - A(1,2) * A(0,3);
    const double c5 = A(2,2) * A(3,3) - A(3,2) * A(2,3), c4 = A(2,1) * A(3,3) - A(3,1) * A(2,3);
    const double c3 = A(2,1) * A(3,2) - A(3,1) * A(2,2), c2 = A(2,0) * A(3,3) - A(3,0) * A(2,3);
    const double c1 = A(2,0) * A(3,2) - A(3,0) * A(2,2), c0 = A(2,0) * A(3,1) - A(3,0) * A(2,1);
    const double id = 1.0 / (s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0);
    R_(0,0) = ( A(1,1) * c5 - A(1,2) * c4 + A(1,3) * c3) * id;
    R_(0,1) = (-A(0,1) * c5 + A(0,2) * c4 - A(0,3) * c3) * id;
    R_(0,2) = ( A(3,1) * s5 - A(3,2) * s4 + A(3,3) * s3) * id;
    R_(0,3) = (-A(2,1) * s5 + A(2,2) * s4 - A(2,3) * s3) * id;
    R_(1,0) = (-A(1,0) * c5 + A(1,2) * c2 - A(1,3) * c1) * id;
    R_(1,1) = ( A(0,0) * c5 - A(0,2) * c2 + A(0,3) * c1) * id;
    R_(1,2) = (-A(3,0) * s5 + A(3,2) * s2 - A(3,3) * s1) * id;
    R_(1,3) = ( A(2,0) * s5 - A(2,2) * s2 + A(2,3) * s1) * id;
    R_(2,0) = ( A(1,0) * c4 - A(1,1) * c2 + A(1,3) * c0) * id;
    R_(2,1) = (-A(0,0) * c4 + A(0,1) * c2 - A(0,3) * c0) * id;
    R_(2,2) = ( A(3,0) * s4 - A(3,1) * s2 + A(3,3) * s0) * id;
    R_(2,3) = (-A(2,0) * s4 + A(2,1) * s2 - A(2,3) * s0) * id;
    R_(3,0) = (-A(1,0) * c3 + A(1,1) * c1 - A(1,2) * c0) * id;
    R_(3,1) = ( A(0,0) * c3 - A(0,1) * c1 + A(0,2) * c0) * id;
    R_(3,2) = (-A(3,0) * s3 + A(3,1) * s1 - A(3,2) * s0) * id;
    R_(3,3) = ( A(2,0) * s3 - A(2,1) * s1 + A(2,2) * s0) * id;
#undef A
#undef R_
}

__device__ static void pp_pose(const double R[9], const double r[3], double H[16])
{   // [R 0; 0 1] * [I r; 0 1], Tracking.cpp:189-194
    for (int k = 0; k < 16; ++k) H[k] = 0.0;
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) H[i + 4 * j] = R[i + 3 * j];
    for (int i = 0; i < 3; ++i) H[i + 12] = R[i] * r[0] + R[i + 3] * r[1] + R[i + 6] * r[2];
    H[15] = 1.0;
}

__device__ static void pp_undistort(const Cam& cam, double ud, double vd, double& uu, double& vu)
{   // ExtendKF::undistort_fm, src/ExtendKF.cpp:266-285
    const double xd = (ud - cam.Cx) * cam.dx, yd = (vd - cam.Cy) * cam.dy;
    const double rd2 = xd * xd + yd * yd;
    const double D = 1 + cam.k1 * rd2 + cam.k2 * rd2 * rd2;
    uu = xd * D / cam.dx + cam.Cx;
    vu = yd * D / cam.dy + cam.Cy;
}

__global__ void __launch_bounds__(192)
pred_patch_kernel(Cam cam, int compat, int L, const uint8_t* __restrict__ type, const int32_t* __restrict__ off,
                  const int32_t* __restrict__ xyz_src, const double* __restrict__ x, const double* __restrict__ h,
                  const uint8_t* __restrict__ has_h, const int32_t* __restrict__ slot, const double* __restrict__ rec,
                  const float* __restrict__ rec_patch, double* __restrict__ out, int32_t* __restrict__ status)
{
    __shared__ double Ms[9];
    __shared__ double misc[4];       // xs, ys, u offset, v offset
    __shared__ int st;
    const int f = blockIdx.x, t = threadIdx.x;
    if (f >= L) return;
    double* o = out + (long)f * MT_NPIX;
    const int sl = slot[f];
    const double* rc = rec + (long)sl * PP_REC;
    if (t == 0) {
        int s = 1;
        if (!has_h[f]) s = 2;
        else {
            const double h0 = h[2 * f], h1 = h[2 * f + 1];
            if (!((h0 > MT_HALF) && (h0 < cam.nCols - MT_HALF) && (h1 > MT_HALF) && (h1 < cam.nRows - MT_HALF))) s = 0;
            else {
                const double fk = cam.f / cam.dx;
                const double* uvf = rc; const double* Rf = rc + 2; const double* rf = rc + 11;
                double Rwc[9], Hf[16], Hk[16], Hfi[16], Hr[16];
                q2r(x + 3, Rwc);
                pp_pose(Rf, rf, Hf);
                pp_pose(Rwc, x, Hk);
                pp_inv4(Hf, Hfi);
                for (int j = 0; j < 4; ++j)
                    for (int i = 0; i < 4; ++i) {
                        double a = 0;
                        for (int k = 0; k < 4; ++k) a += Hfi[i + 4 * k] * Hk[k + 4 * j];
                        Hr[i + 4 * j] = a;
                    }
                double n1[3] = { uvf[0] - cam.Cx, uvf[1] - cam.Cy, -fk };
                double nn = sqrt(n1[0] * n1[0] + n1[1] * n1[1] + n1[2] * n1[2]);
                for (int a = 0; a < 3; ++a) n1[a] = n1[a] / nn;
                const double n2in[4] = { h0 - cam.Cx, h1 - cam.Cy, -fk, 1.0 };
                double n2[4];
                for (int i = 0; i < 4; ++i) n2[i] = Hr[i] * n2in[0] + Hr[i + 4] * n2in[1] + Hr[i + 8] * n2in[2] + Hr[i + 12] * n2in[3];
                const double w2 = n2[3];
                for (int i = 0; i < 4; ++i) n2[i] = n2[i] / w2;
                nn = sqrt(n2[0] * n2[0] + n2[1] * n2[1] + n2[2] * n2[2]);
                double n[3];
                for (int a = 0; a < 3; ++a) n[a] = n1[a] + n2[a] / nn;
                nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                for (int a = 0; a < 3; ++a) n[a] = n[a] / nn;
                // world point (search_IC_matches, Tracking.cpp:52-61): refreshed for inverse-depth features only
                double W[3] = {0.0, 0.0, 0.0};
                const int src = xyz_src[f];
                if (src >= 0) {
                    const int os = off[src];
                    if (type[src] == 0) {
                        double st_, ct, sp, cp;
                        sincos(x[os + 3], &st_, &ct);
                        sincos(x[os + 4], &sp, &cp);
                        const double m[3] = { cp * st_, -sp, cp * ct };
                        for (int a = 0; a < 3; ++a) W[a] = x[os + a] + (1.0 / x[os + 5]) * m[a];
                    } else { W[0] = x[os]; W[1] = x[os + 1]; W[2] = x[os + 2]; }
                }
                double X[4];
                for (int i = 0; i < 4; ++i) X[i] = Hfi[i] * W[0] + Hfi[i + 4] * W[1] + Hfi[i + 8] * W[2] + Hfi[i + 12];
                const double w3 = X[3];
                for (int i = 0; i < 4; ++i) X[i] = X[i] / w3;
                const double d = -(n[0] * X[0] + n[1] * X[1] + n[2] * X[2]);
                const double K[9] = { fk, 0, 0,  0, cam.f / cam.dy, 0,  cam.Cx, cam.Cy, 1 };
                double Kinv[9], G[9], T1[9], M[9], Minv[9];
                pp_inv3(K, Kinv);
                for (int j = 0; j < 3; ++j)
                    for (int i = 0; i < 3; ++i) G[i + 3 * j] = Hr[i + 4 * j] - Hr[i + 12] * n[j] / d;
                for (int j = 0; j < 3; ++j)
                    for (int i = 0; i < 3; ++i) T1[i + 3 * j] = K[i] * G[3 * j] + K[i + 3] * G[1 + 3 * j] + K[i + 6] * G[2 + 3 * j];
                for (int j = 0; j < 3; ++j)
                    for (int i = 0; i < 3; ++i) M[i + 3 * j] = T1[i] * Kinv[3 * j] + T1[i + 3] * Kinv[1 + 3 * j] + T1[i + 6] * Kinv[2 + 3 * j];
                pp_inv3(M, Minv);
                double c1u, c1v, c2u, c2v;
                pp_undistort(cam, uvf[0], uvf[1], c1u, c1v);
                double t3[3];
                for (int i = 0; i < 3; ++i) t3[i] = Minv[i] * c1u + Minv[i + 3] * c1v + Minv[i + 6];
                distort_fm(cam, t3[0] / t3[2], t3[1] / t3[2], c2u, c2v);
                const int xs = (int)(c2u - MT_HALF), xe = (int)(c2u + MT_HALF), ys = (int)(c2v - MT_HALF), ye = (int)(c2v + MT_HALF);
                if (xe - xs + 1 != MT_SIDE || ye - ys + 1 != MT_SIDE) s = -1;
                const double offp = compat ? (double)(PP_HALF_F + 1) : (double)PP_HALF_F;      // Tracking.cpp:263-264 (MATLAB indices)
                for (int k = 0; k < 9; ++k) Ms[k] = M[k];
                misc[0] = xs; misc[1] = ys; misc[2] = uvf[0] - offp; misc[3] = uvf[1] - offp;
            }
        }
        st = s;
        status[f] = s;
    }
    __syncthreads();
    if (t >= MT_NPIX) return;
    if (st != 1) { o[t] = 0.0; return; }
    const int j = t / MT_SIDE, i = t % MT_SIDE;          // column-major output: t = i + 13 j, column j = u, row i = v
    double pu, pv, qu, qv;
    pp_undistort(cam, misc[0] + j, misc[1] + i, pu, pv);
    const double q0 = Ms[0] * pu + Ms[3] * pv + Ms[6], q1 = Ms[1] * pu + Ms[4] * pv + Ms[7], q2 = Ms[2] * pu + Ms[5] * pv + Ms[8];
    distort_fm(cam, q0 / q2, q1 / q2, qu, qv);
    const float mapx = (float)(qu - misc[2]), mapy = (float)(qv - misc[3]);
    // cv::remap: 1/32-pixel coordinates (cvRound = round half to even), float table weights, float taps
    const int sx = (int)rint((double)mapx * 32.0), sy = (int)rint((double)mapy * 32.0);
    const int ix = sx >> 5, iy = sy >> 5;
    const float fx = (float)(sx & 31) / 32.f, fy = (float)(sy & 31) / 32.f;
    const float w00 = __fmul_rn(__fsub_rn(1.f, fy), __fsub_rn(1.f, fx)), w01 = __fmul_rn(__fsub_rn(1.f, fy), fx);
    const float w10 = __fmul_rn(fy, __fsub_rn(1.f, fx)), w11 = __fmul_rn(fy, fx);
    const float* sp = rec_patch + (long)sl * PP_NPIX_F;  // row-major 41 x 41
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int xx = ix + (k & 1), yy = iy + (k >> 1);
        v[k] = (xx >= 0 && xx < PP_SIDE_F && yy >= 0 && yy < PP_SIDE_F) ? sp[yy * PP_SIDE_F + xx] : 0.f;
    }
    const float r = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(v[0], w00), __fmul_rn(v[1], w01)), __fmul_rn(v[2], w10)), __fmul_rn(v[3], w11));
    o[t] = (double)r;
}

void launch_pred_patches(hipStream_t s, const Cam& cam, int compat, int L, const uint8_t* type, const int32_t* off,
                         const int32_t* xyz_src, const double* x, const double* h, const uint8_t* has_h,
                         const int32_t* slot, const double* rec, const float* rec_patch, double* out, int32_t* status)
{
    if (L <= 0) return;
    pred_patch_kernel<<<dim3(L), dim3(192), 0, s>>>(cam, compat, L, type, off, xyz_src, x, h, has_h, slot, rec, rec_patch, out, status);
}

}  // namespace rslam
