// map_kernels.hip -- state surgery of Map::map_management on the resident posterior
// (x_k_k, p_k_k): delete a feature (src/Map.cpp:69-104), convert one inverse-depth feature to
// Cartesian (src/Map.cpp:105-196) and append a new inverse-depth feature
// (src/Map.cpp:281-292 with add_a_feature_covariance_inverse_depth, :339-400, and
// ExtendKF::hinv, src/ExtendKF.cpp:236-265).
//
// All three are congruences P' = G P G^T (+ R on the new block) in which G is the identity
// except for `special` consecutive rows that hold a small dense block (3 x 6 for the conversion,
// 6 x 13 for the insertion).  One out-of-place pass streams P once and writes P' once with its
// new leading dimension: HBM-bound, 16 n^2 bytes, no n x n temporaries (the reference builds
// J_all as a dense (n-3) x n matrix and multiplies twice, Map.cpp:154-189).
#include "kernels.h"
#include <limits.h>

namespace rslam {

// d_mapcoef layout (doubles)
constexpr int MC_G = 0;        // special x cnt, col-major G[r + special * k]  (<= 6 x 13)
constexpr int MC_R = 96;       // 6 x 6 additive block of the insertion
constexpr int MC_X = 140;      // values of the special state entries
static_assert(MC_X + 6 <= MAP_COEF_DOUBLES, "coefficient buffer too small");

__device__ static void undistort_fm_dev(const Cam& cam, double ud, double vd, double& uu, double& vu)
{   // ExtendKF::undistort_fm, src/ExtendKF.cpp:266-285
    const double xd = (ud - cam.Cx) * cam.dx, yd = (vd - cam.Cy) * cam.dy;
    const double rd2 = xd * xd + yd * yd;
    const double D = 1 + cam.k1 * rd2 + cam.k2 * rd2 * rd2;
    uu = xd * D / cam.dx + cam.Cx;
    vu = yd * D / cam.dy + cam.Cy;
}

__device__ static void convert_coefficients(const double* x, int o, double* coef)
{
    const double theta = x[o + 3], phi = x[o + 4], rho = x[o + 5];
    double st, ct, sp, cp;
    sincos(theta, &st, &ct);
    sincos(phi, &sp, &cp);
    const double mi[3] = { cp * st, -sp, cp * ct };
    const double dmt[3] = { cp * ct, 0.0, -cp * st };
    const double dmp[3] = { -sp * st, -cp, -sp * ct };
    double* G = coef + MC_G;      // 3 x 6
    for (int k = 0; k < 18; ++k) G[k] = 0.0;
    for (int a = 0; a < 3; ++a) {
        G[a + 3 * a] = 1.0;
        G[a + 3 * 3] = (1 / rho) * dmt[a];
        G[a + 3 * 4] = (1 / rho) * dmp[a];
        G[a + 3 * 5] = -mi[a] / (rho * rho);
        coef[MC_X + a] = x[o + a] + (1.0 / rho) * mi[a];           // inversedepth2cartesian, ExtendKF.cpp:137-152
    }
}

__device__ static void insert_coefficients(const Cam& cam, const double* x, double ud, double vd, double rho0,
                                           double std_z, double std_rho, double* coef)
{
    const double fku = cam.f / cam.dx, fkv = cam.f / cam.dy;
    const double* q = x + 3;
    double R[9], uu, vu;
    q2r(q, R);
    undistort_fm_dev(cam, ud, vd, uu, vu);
    const double c[3] = { -(cam.Cx - uu) / fku, -(cam.Cy - vu) / fkv, 1.0 };
    const double Xw = R[0] * c[0] + R[3] * c[1] + R[6] * c[2];
    const double Yw = R[1] * c[0] + R[4] * c[1] + R[7] * c[2];
    const double Zw = R[2] * c[0] + R[5] * c[1] + R[8] * c[2];
    // hinv
    coef[MC_X + 0] = x[0]; coef[MC_X + 1] = x[1]; coef[MC_X + 2] = x[2];
    coef[MC_X + 3] = atan2(Xw, Zw);
    coef[MC_X + 4] = atan2(-Yw, sqrt(Xw * Xw + Zw * Zw));
    coef[MC_X + 5] = rho0;
    // dgw_dqwr = dRq_times_a_by_dq(q, c), ExtendKF.cpp:286-311
    double dg[3][4];
    dg[0][0] = 2 * q[0] * c[0] - 2 * q[3] * c[1] + 2 * q[2] * c[2];
    dg[1][0] = 2 * q[3] * c[0] + 2 * q[0] * c[1] - 2 * q[1] * c[2];
    dg[2][0] = -2 * q[2] * c[0] + 2 * q[1] * c[1] + 2 * q[0] * c[2];
    dg[0][1] = 2 * q[1] * c[0] + 2 * q[2] * c[1] + 2 * q[3] * c[2];
    dg[1][1] = 2 * q[2] * c[0] - 2 * q[1] * c[1] - 2 * q[0] * c[2];
    dg[2][1] = 2 * q[3] * c[0] + 2 * q[0] * c[1] - 2 * q[1] * c[2];
    dg[0][2] = -2 * q[2] * c[0] + 2 * q[1] * c[1] + 2 * q[0] * c[2];
    dg[1][2] = 2 * q[1] * c[0] + 2 * q[2] * c[1] + 2 * q[3] * c[2];
    dg[2][2] = -2 * q[0] * c[0] + 2 * q[3] * c[1] - 2 * q[2] * c[2];
    dg[0][3] = -2 * q[3] * c[0] - 2 * q[0] * c[1] + 2 * q[1] * c[2];
    dg[1][3] = 2 * q[0] * c[0] - 2 * q[3] * c[1] + 2 * q[2] * c[2];
    dg[2][3] = 2 * q[1] * c[0] + 2 * q[2] * c[1] + 2 * q[3] * c[2];
    const double xz = Xw * Xw + Zw * Zw, xyz = Xw * Xw + Yw * Yw + Zw * Zw, sxz = sqrt(xz);
    const double dth[3] = { Zw / xz, 0.0, -Xw / xz };
    const double dph[3] = { (Xw * Yw) / (xyz * sxz), -sxz / xyz, (Zw * Yw) / (xyz * sxz) };
    double* G = coef + MC_G;      // dy_dxv, 6 x 13
    for (int k = 0; k < 78; ++k) G[k] = 0.0;
    for (int a = 0; a < 3; ++a) G[a + 6 * a] = 1.0;
    for (int k = 0; k < 4; ++k) {
        G[3 + 6 * (3 + k)] = dth[0] * dg[0][k] + dth[1] * dg[1][k] + dth[2] * dg[2][k];
        G[4 + 6 * (3 + k)] = dph[0] * dg[0][k] + dph[1] * dg[1][k] + dph[2] * dg[2][k];
    }
    // dy_dhd = [dyprima_dgw * R_wc * dgc_dhu * dhu_dhd, 0; 0 0 1]
    const double du = ud - cam.Cx, dv = vd - cam.Cy;
    const double rd2 = (du * cam.dx) * (du * cam.dx) + (dv * cam.dy) * (dv * cam.dy);
    const double g = 1 + cam.k1 * rd2 + cam.k2 * rd2 * rd2, gk = cam.k1 + 2 * cam.k2 * rd2;
    const double J00 = g + du * gk * (2 * du * cam.dx * cam.dx), J01 = du * gk * (2 * dv * cam.dy * cam.dy);
    const double J10 = dv * gk * (2 * du * cam.dx * cam.dx),     J11 = g + dv * gk * (2 * dv * cam.dy * cam.dy);
    double B[3][2];
    for (int a = 0; a < 3; ++a) {
        const double a0 = R[a] * (1 / fku), a1 = R[a + 3] * (1 / fkv);
        B[a][0] = a0 * J00 + a1 * J10;
        B[a][1] = a0 * J01 + a1 * J11;
    }
    double E[6][3];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 3; ++j) E[i][j] = 0.0;
    for (int j = 0; j < 2; ++j) {
        E[3][j] = dth[0] * B[0][j] + dth[1] * B[1][j] + dth[2] * B[2][j];
        E[4][j] = dph[0] * B[0][j] + dph[1] * B[1][j] + dph[2] * B[2][j];
    }
    E[5][2] = 1.0;
    const double padd[3] = { std_z * std_z, std_z * std_z, std_rho * std_rho };
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += (E[i][k] * padd[k]) * E[j][k];
            coef[MC_R + i + 6 * j] = s;
        }
}

// one block: thread 0 writes the coefficients, then the block moves x into its new layout
__global__ void __launch_bounds__(256)
map_state_kernel(int mode, Cam cam, const double* __restrict__ x_old, int o, double ud, double vd, double rho0,
                 double std_z, double std_rho, double* __restrict__ coef,
                 double* __restrict__ x_new, int n_new, int NP_new, int cut, int special, int shift)
{
    if (threadIdx.x == 0) {
        if (mode == 1) convert_coefficients(x_old, o, coef);
        else if (mode == 2) insert_coefficients(cam, x_old, ud, vd, rho0, std_z, std_rho, coef);
    }
    __threadfence_block();
    __syncthreads();
    for (int i = threadIdx.x; i < NP_new; i += 256) {
        double v = 0.0;
        if (i < cut) v = x_old[i];
        else if (i < cut + special) v = coef[MC_X + (i - cut)];
        else if (i < n_new) v = x_old[i + shift];
        x_new[i] = v;
    }
}

// P' = G P G^T (+ R): column j' per blockIdx.y, 256 rows per block
__global__ void __launch_bounds__(256)
map_cov_kernel(const double* __restrict__ P, int ld_old, double* __restrict__ Pn, int ld_new, int n_new,
               int cut, int special, int shift, int sp_base, int sp_cnt, int add_r, const double* __restrict__ coef)
{
    const int j = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ld_new) return;
    double v = 0.0;
    if (i < n_new && j < n_new) {
        const bool sj = (j >= cut && j < cut + special), si = (i >= cut && i < cut + special);
        const int jo = j < cut ? j : j + shift, io = i < cut ? i : i + shift;
        const double* G = coef + MC_G;
        if (!si && !sj) v = P[io + (long)jo * ld_old];
        else if (si && !sj) {
            for (int k = 0; k < sp_cnt; ++k) v += G[(i - cut) + special * k] * P[(sp_base + k) + (long)jo * ld_old];
        } else if (!si && sj) {
            for (int k = 0; k < sp_cnt; ++k) v += P[io + (long)(sp_base + k) * ld_old] * G[(j - cut) + special * k];
        } else {
            for (int k = 0; k < sp_cnt; ++k) {
                double t = 0.0;
                for (int m = 0; m < sp_cnt; ++m) t += G[(i - cut) + special * m] * P[(sp_base + m) + (long)(sp_base + k) * ld_old];
                v += t * G[(j - cut) + special * k];
            }
            if (add_r) v += coef[MC_R + (i - cut) + 6 * (j - cut)];
        }
    }
    Pn[i + (long)j * ld_new] = v;
}

// linearity index of every inverse-depth feature (Map.cpp:124-149); *first = lowest feature
// index whose value is below the threshold (INT_MAX when none)
__global__ void __launch_bounds__(64)
map_linearity_kernel(const double* __restrict__ x, const double* __restrict__ P, int NP, int L,
                     const uint8_t* __restrict__ type, const int32_t* __restrict__ off, double threshold,
                     double* __restrict__ out, int32_t* __restrict__ first)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= L) return;
    double val = -1.0;
    if (type[i] == 0) {
        const int o = off[i];
        const double std_rho = sqrt(P[(o + 5) + (long)(o + 5) * NP]);
        const double rho = x[o + 5];
        const double std_d = std_rho / (rho * rho);
        double st, ct, sp, cp;
        sincos(x[o + 3], &st, &ct);
        sincos(x[o + 4], &sp, &cp);
        const double mi[3] = { cp * st, -sp, cp * ct };
        double d1[3], d2[3];
        for (int a = 0; a < 3; ++a) {
            const double X = x[o + a] + (1.0 / rho) * mi[a];
            d1[a] = X - x[o + a];
            d2[a] = X - x[a];
        }
        const double d_c2p = sqrt(d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2]);
        const double aa = d1[0] * d2[0] + d1[1] * d2[1] + d1[2] * d2[2];
        const double bb = sqrt(d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2]) * d_c2p;
        val = 4 * std_d * (aa / bb) / d_c2p;
        if (val < threshold) atomicMin(first, i);
    }
    out[i] = val;
}

void launch_map_state(hipStream_t s, int mode, const Cam& cam, const double* x_old, int o, double ud, double vd,
                      double rho0, double std_z, double std_rho, double* coef, double* x_new, int n_new, int NP_new,
                      int cut, int special, int shift)
{
    map_state_kernel<<<dim3(1), dim3(256), 0, s>>>(mode, cam, x_old, o, ud, vd, rho0, std_z, std_rho, coef, x_new, n_new,
                                                   NP_new, cut, special, shift);
}

void launch_map_cov(hipStream_t s, const double* P, int ld_old, double* Pn, int ld_new, int n_new, int cut, int special,
                    int shift, int sp_base, int sp_cnt, int add_r, const double* coef)
{
    map_cov_kernel<<<dim3((ld_new + 255) / 256, ld_new), dim3(256), 0, s>>>(P, ld_old, Pn, ld_new, n_new, cut, special, shift,
                                                                          sp_base, sp_cnt, add_r, coef);
}

void launch_map_linearity(hipStream_t s, const double* x, const double* P, int NP, int L, const uint8_t* type,
                          const int32_t* off, double threshold, double* out, int32_t* first)
{
    if (L <= 0) return;
    map_linearity_kernel<<<dim3((L + 63) / 64), dim3(64), 0, s>>>(x, P, NP, L, type, off, threshold, out, first);
}

}  // namespace rslam
