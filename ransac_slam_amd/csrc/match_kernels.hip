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
