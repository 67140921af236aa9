// camera_model.h -- device-side camera model and measurement Jacobians (gfx950).
//
// Arithmetic follows the reference's ExtendKF camera helpers and
// Tracking::calculate_Hi_* (citations per function, paths into the reference
// repository); the code is written for one-lane-per-feature execution: plain
// scalars in registers, multiplication chains instead of pow(), structurally
// sparse Jacobians (13 non-zero columns: 7 camera-pose + 6 feature).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rslam {

struct Cam {
    double k1, k2, Cx, Cy, f, dx, dy;
    int nRows, nCols;
    double inv_dx, inv_dy, f_ku;     // 1/dx, 1/dy, f * (1/dx): correctly rounded on the host once (scoring kernel)
    double ru2_fast;                 // squared undistorted radius up to which distort_fm_score's six steps have converged (host)
};

// ExtendKF::q2r (src/ExtendKF.cpp:91-102); q = (r,x,y,z); R column-major.
__device__ __forceinline__ void q2r(const double q[4], double R[9])
{
    const double r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = r * r + x * x - y * y - z * z;  R[3] = 2 * (x * y - r * z);            R[6] = 2 * (z * x + r * y);
    R[1] = 2 * (x * y + r * z);            R[4] = r * r - x * x + y * y - z * z;  R[7] = 2 * (y * z - r * x);
    R[2] = 2 * (z * x - r * y);            R[5] = 2 * (y * z + r * x);            R[8] = r * r - x * x - y * y + z * z;
}

// Matrix3d::inverse() as Eigen's fixed-size path computes it (cofactors / det);
// used for Rrw = q2r(q).inverse() (src/Tracking.cpp:90,136; src/ExtendKF.cpp:83).
__device__ __forceinline__ void inv3(const double M[9], double R[9])
{
    const double c00 = M[4] * M[8] - M[7] * M[5];
    const double c10 = M[6] * M[5] - M[3] * M[8];
    const double c20 = M[3] * M[7] - M[6] * M[4];
    const double det = c00 * M[0] + c10 * M[1] + c20 * M[2];
    const double id = 1.0 / det;
    R[0] = c00 * id;  R[3] = c10 * id;  R[6] = c20 * id;
    R[1] = (M[7] * M[2] - M[1] * M[8]) * id;
    R[4] = (M[0] * M[8] - M[6] * M[2]) * id;
    R[7] = (M[6] * M[1] - M[0] * M[7]) * id;
    R[2] = (M[1] * M[5] - M[4] * M[2]) * id;
    R[5] = (M[3] * M[2] - M[0] * M[5]) * id;
    R[8] = (M[0] * M[4] - M[3] * M[1]) * id;
}

// Dynamic 2x2 MatrixXd::inverse() == PartialPivLU (src/Tracking.cpp:421,591):
// row pivot on |a00| vs |a10|, then solve for the identity.  M, R column-major.
__device__ __forceinline__ void inv2_lu(const double M[4], double R[4])
{
    double a = M[0], c = M[1], b = M[2], d = M[3];   // [a b; c d]
    bool swap = fabs(c) > fabs(a);
    if (swap) { double t = a; a = c; c = t; t = b; b = d; d = t; }
    const double l = c / a;
    const double u11 = d - l * b;
    // columns of the (row-permuted) identity
    double e0a = swap ? 0.0 : 1.0, e0b = swap ? 1.0 : 0.0;   // rhs for column 0: P*e0
    double e1a = swap ? 1.0 : 0.0, e1b = swap ? 0.0 : 1.0;
    double y1 = e0b - l * e0a;  double x1 = y1 / u11;  double x0 = (e0a - b * x1) / a;
    R[0] = x0; R[1] = x1;
    y1 = e1b - l * e1a;  x1 = y1 / u11;  x0 = (e1a - b * x1) / a;
    R[2] = x0; R[3] = x1;
}

// 1/d to full double precision without the IEEE division sequence: v_rcp_f64 (~2^-26) and two Newton steps
__device__ __forceinline__ double rcp_nr2(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

__device__ __forceinline__ void distort_fm(const Cam& cam, double u, double v, double& ud, double& vd);

// distort_fm for the hypothesis scoring (Tracking.cpp:472-476 only compares the result with a threshold; the decisions
// are audited for a 1e-9 margin, the values need not be bit-identical to a divide-based evaluation -- they agree with it to
// ~1e-13 px, tests/test_gpu_scoring.py).  The reference's 10 Newton steps (ExtendKF.cpp:186-200) divide by the slope f'(rd);
// any factor within 2^-26 of 1/f' gives the same fixed point -- the error of step n+1 is eps * (error of step n) +
// O(error^2) -- so the slope reciprocal is the raw v_rcp_f64.  The divisions that enter the result (1/D, 1/dx, 1/dy) are
// full-precision reciprocals (dx, dy: from the host).
// Six steps instead of ten: the starting point ru / (1 + k1 ru^2 + k2 ru^4) is ~25 % low at the image corner (ru = 2.3 mm
// with the reference's camera) and the steps converge quadratically from there: at ru <= cam.ru2_fast^(1/2) -- found on the
// host by running this very iteration, ~1.5 corner radii -- six steps are at the fixed point to < 1e-13 relative.  Beyond
// that radius (projections far outside the image: a hypothesis that puts a feature almost in the image plane) neither six nor
// ten steps need to have converged, the NON-converged ten-step value is what the reference compares -- it can even land back
// inside the image -- so those pairs take the reference's own sequence (distort_fm below: ten steps, IEEE divisions).
__device__ __forceinline__ void distort_fm_score(const Cam& cam, double u, double v, double& ud, double& vd)
{
    const double xu = (u - cam.Cx) * cam.dx;
    const double yu = (v - cam.Cy) * cam.dy;
    const double ru2 = xu * xu + yu * yu;
    if (!(ru2 <= cam.ru2_fast)) { distort_fm(cam, u, v, ud, vd); return; }      // (also NaN / inf)
    const double ru = sqrt(ru2);
    double rd = ru * __builtin_amdgcn_rcp(1 + cam.k1 * ru2 + cam.k2 * (ru2 * ru2));     // starting point only
    const double k1_3 = 3 * cam.k1, k2_5 = 5 * cam.k2;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double rd2 = rd * rd, rd4 = rd2 * rd2;
        const double f = rd + cam.k1 * (rd2 * rd) + cam.k2 * (rd4 * rd) - ru;
        const double fp = 1 + k1_3 * rd2 + k2_5 * rd4;
        rd = rd - f * __builtin_amdgcn_rcp(fp);
    }
    const double rd2 = rd * rd;
    const double rD = rcp_nr2(1 + cam.k1 * rd2 + cam.k2 * (rd2 * rd2));
    ud = (xu * rD) * cam.inv_dx + cam.Cx;
    vd = (yu * rD) * cam.inv_dy + cam.Cy;
}

// sin / cos of (a0 + d) from the tabulated sin a0, cos a0 and a short series in d (|d| <= 1/8: the truncation error of
// both series is < 1e-19): the scoring kernel evaluates every feature's angles once per hypothesis, where they differ from
// the prior's angles only by the hypothesis' state correction.
__device__ __forceinline__ void sincos_delta(double s0, double c0, double d, double& s, double& c)
{
    const double q = d * d;
    const double sd = d * fma(q, fma(q, fma(q, fma(q, fma(q, fma(q, 1.0 / 6227020800.0, -1.0 / 39916800.0), 1.0 / 362880.0), -1.0 / 5040.0),
                                         1.0 / 120.0), -1.0 / 6.0), 1.0);
    const double cd = fma(q, fma(q, fma(q, fma(q, fma(q, fma(q, 1.0 / 479001600.0, -1.0 / 3628800.0), 1.0 / 40320.0), -1.0 / 720.0),
                                    1.0 / 24.0), -0.5), 1.0);
    s = fma(s0, cd, c0 * sd);
    c = fma(c0, cd, -(s0 * sd));
}

// ExtendKF::distort_fm (src/ExtendKF.cpp:175-204): 10 fixed Newton steps.
__device__ __forceinline__ void distort_fm(const Cam& cam, double u, double v, double& ud, double& vd)
{
    const double xu = (u - cam.Cx) * cam.dx;
    const double yu = (v - cam.Cy) * cam.dy;
    const double ru = sqrt(xu * xu + yu * yu);
    const double ru2 = ru * ru;
    double rd = ru / (1 + cam.k1 * ru2 + cam.k2 * (ru2 * ru2));
    const double k1_3 = 3 * cam.k1, k2_5 = 5 * cam.k2;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const double rd2 = rd * rd, rd4 = rd2 * rd2;
        const double f = rd + cam.k1 * (rd2 * rd) + cam.k2 * (rd4 * rd) - ru;
        const double fp = 1 + k1_3 * rd2 + k2_5 * rd4;
        rd = rd - f / fp;
    }
    const double rd2 = rd * rd;
    const double D = 1 + cam.k1 * rd2 + cam.k2 * (rd2 * rd2);
    ud = xu / D / cam.dx + cam.Cx;
    vd = yu / D / cam.dy + cam.Cy;
}

// ExtendKF::hi_cartesian (src/ExtendKF.cpp:103-132) with hu (:153-174):
// +-60 degree FOV gate, pinhole, distortion, image-bounds gate.
__device__ __forceinline__ bool hi_cartesian(const Cam& cam, const double hrl[3], double& ud, double& vd)
{
    const double k = 180.0 / 3.14159265358979323846;
    const double ax = atan2(hrl[0], hrl[2]) * 180 / 3.14159265358979323846;
    const double ay = atan2(hrl[1], hrl[2]) * 180 / 3.14159265358979323846;
    (void)k;
    if (ax < -60 || ax > 60 || ay < -60 || ay > 60) return false;
    const double ku = 1.0 / cam.dx, kv = 1.0 / cam.dy;
    const double uu = cam.Cx + (hrl[0] / hrl[2]) * cam.f * ku;
    const double vu = cam.Cy + (hrl[1] / hrl[2]) * cam.f * kv;
    double u, v;
    distort_fm(cam, uu, vu, u, v);
    if (u > 0 && u < cam.nCols && v > 0 && v < cam.nRows) { ud = u; vd = v; return true; }
    return false;
}

// Camera-frame ray of a feature: hrl = R_cw * ((y - t) * rho + m) for inverse
// depth, R_cw * (y - t) for Cartesian (src/ExtendKF.cpp:73-75,83).
__device__ __forceinline__ void feature_arg(const double* x, int off, bool is_id, double arg[3], double mi[3])
{
    if (is_id) {
        const double th = x[off + 3], ph = x[off + 4], rho = x[off + 5];
        double st, ct, sp, cp;
        sincos(th, &st, &ct);
        sincos(ph, &sp, &cp);
        mi[0] = cp * st; mi[1] = -sp; mi[2] = cp * ct;
#pragma unroll
        for (int a = 0; a < 3; ++a) arg[a] = (x[off + a] - x[a]) * rho + mi[a];
    } else {
        mi[0] = mi[1] = mi[2] = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) arg[a] = x[off + a] - x[a];
    }
}

// Measurement prediction of one feature (src/ExtendKF.cpp:69-88).
__device__ __forceinline__ bool predict_feature(const Cam& cam, const double* x, int off, bool is_id,
                                                double& ud, double& vd)
{
    double R[9], arg[3], mi[3], hrl[3];
    q2r(x + 3, R);
    feature_arg(x, off, is_id, arg, mi);
    if (is_id) {
        // r_wc.transpose() * arg
#pragma unroll
        for (int a = 0; a < 3; ++a) hrl[a] = R[3 * a + 0] * arg[0] + R[3 * a + 1] * arg[1] + R[3 * a + 2] * arg[2];
    } else {
        double Ri[9];
        inv3(R, Ri);
#pragma unroll
        for (int a = 0; a < 3; ++a) hrl[a] = Ri[a] * arg[0] + Ri[a + 3] * arg[1] + Ri[a + 6] * arg[2];
    }
    return hi_cartesian(cam, hrl, ud, vd);
}

// Compact Jacobian of one feature: H13[p*13 + c], p = pixel row (0,1),
// c = 0..6 camera position+quaternion columns, c = 7..12 the feature's own
// columns (Cartesian: 7..9, rest zero).  The velocity columns 7..12 of the
// state are structurally zero (a32, src/Tracking.cpp:107,150) and not stored.
// Follows Tracking::calculate_Hi_inverse_depth (src/Tracking.cpp:113-163),
// calculate_Hi_cartesian (:71-112), jacob_undistor_fm (src/ExtendKF.cpp:312-332)
// and dRq_times_a_by_dq (:286-311).
// The Jacobian in two steps, so that the part that does not depend on the predicted pixel can run beside the prediction
// (predict_kernel gives them to two waves): jacobian_core = a2 * D (2 x 13; a2 = d pinhole / d hc, D = d hc / d state),
// jacobian_finish = a1 * (that), a1 = jacob_undistor_fm(h)^-1 (fixed 2x2 closed form).  Same terms as the reference's
// (a1 a2) D, associated the other way round.
__device__ __forceinline__ void jacobian_core(const Cam& cam, const double* x, int off, bool is_id, double B13[26])
{
    double Rq[9], Rrw[9], arg[3], mi[3], hc[3];
    q2r(x + 3, Rq);
    inv3(Rq, Rrw);
    feature_arg(x, off, is_id, arg, mi);
#pragma unroll
    for (int a = 0; a < 3; ++a) hc[a] = Rrw[a] * arg[0] + Rrw[a + 3] * arg[1] + Rrw[a + 6] * arg[2];
    const double fku = cam.f * (1 / cam.dx), fkv = cam.f * (1 / cam.dy);
    const double a2_00 = fku / hc[2], a2_02 = -hc[0] * fku / (hc[2] * hc[2]);
    const double a2_11 = fkv / hc[2], a2_12 = -hc[1] * fkv / (hc[2] * hc[2]);
    // B = a2 * D first (2 x 13; D = d hc / d state); the 2 x 2 factor a1, the only part that needs h, multiplies it at the end
    const double a12[2][3] = { { a2_00, 0.0, a2_02 }, { 0.0, a2_11, a2_12 } };
    const double rho = is_id ? x[off + 5] : 1.0;

    // columns 0..2: a12 * (-Rrw) [* rho]
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            double s = a12[p][0] * (-Rrw[0 + 3 * c]) + a12[p][1] * (-Rrw[1 + 3 * c]) + a12[p][2] * (-Rrw[2 + 3 * c]);
            B13[p * 13 + c] = is_id ? s * rho : s;
        }
    // columns 3..6: a12 * dRq_times_a_by_dq(qconj, arg) * diag(1,-1,-1,-1)
    {
        const double q0 = x[3], q1 = -x[4], q2 = -x[5], q3 = -x[6];
        double b0[3][4];
        // column 0: [2q0 -2q3 2q2; 2q3 2q0 -2q1; -2q2 2q1 2q0] * arg
        b0[0][0] = 2 * q0 * arg[0] - 2 * q3 * arg[1] + 2 * q2 * arg[2];
        b0[1][0] = 2 * q3 * arg[0] + 2 * q0 * arg[1] - 2 * q1 * arg[2];
        b0[2][0] = -2 * q2 * arg[0] + 2 * q1 * arg[1] + 2 * q0 * arg[2];
        // column 1: [2q1 2q2 2q3; 2q2 -2q1 -2q0; 2q3 2q0 -2q1]
        b0[0][1] = 2 * q1 * arg[0] + 2 * q2 * arg[1] + 2 * q3 * arg[2];
        b0[1][1] = 2 * q2 * arg[0] - 2 * q1 * arg[1] - 2 * q0 * arg[2];
        b0[2][1] = 2 * q3 * arg[0] + 2 * q0 * arg[1] - 2 * q1 * arg[2];
        // column 2: [-2q2 2q1 2q0; 2q1 2q2 2q3; -2q0 2q3 -2q2]
        b0[0][2] = -2 * q2 * arg[0] + 2 * q1 * arg[1] + 2 * q0 * arg[2];
        b0[1][2] = 2 * q1 * arg[0] + 2 * q2 * arg[1] + 2 * q3 * arg[2];
        b0[2][2] = -2 * q0 * arg[0] + 2 * q3 * arg[1] - 2 * q2 * arg[2];
        // column 3: [-2q3 -2q0 2q1; 2q0 -2q3 2q2; 2q1 2q2 2q3]
        b0[0][3] = -2 * q3 * arg[0] - 2 * q0 * arg[1] + 2 * q1 * arg[2];
        b0[1][3] = 2 * q0 * arg[0] - 2 * q3 * arg[1] + 2 * q2 * arg[2];
        b0[2][3] = 2 * q1 * arg[0] + 2 * q2 * arg[1] + 2 * q3 * arg[2];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double sgn = (c == 0) ? 1.0 : -1.0;
#pragma unroll
            for (int p = 0; p < 2; ++p)
                B13[p * 13 + 3 + c] = a12[p][0] * (b0[0][c] * sgn) + a12[p][1] * (b0[1][c] * sgn) + a12[p][2] * (b0[2][c] * sgn);
        }
    }
    // feature columns
    if (is_id) {
        const double th = x[off + 3], ph = x[off + 4];
        double st, ct, sp, cp;
        sincos(th, &st, &ct);
        sincos(ph, &sp, &cp);
        const double c2[3] = { cp * ct, 0.0, -cp * st };
        const double c3[3] = { -sp * st, -cp, -sp * ct };
        const double d[3] = { x[off] - x[0], x[off + 1] - x[1], x[off + 2] - x[2] };
        double c0[3][6];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            c0[a][0] = rho * Rrw[a];  c0[a][1] = rho * Rrw[a + 3];  c0[a][2] = rho * Rrw[a + 6];
            c0[a][3] = Rrw[a] * c2[0] + Rrw[a + 3] * c2[1] + Rrw[a + 6] * c2[2];
            c0[a][4] = Rrw[a] * c3[0] + Rrw[a + 3] * c3[1] + Rrw[a + 6] * c3[2];
            c0[a][5] = Rrw[a] * d[0] + Rrw[a + 3] * d[1] + Rrw[a + 6] * d[2];
        }
#pragma unroll
        for (int c = 0; c < 6; ++c)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                B13[p * 13 + 7 + c] = a12[p][0] * c0[0][c] + a12[p][1] * c0[1][c] + a12[p][2] * c0[2][c];
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                B13[p * 13 + 7 + c] = a12[p][0] * Rrw[0 + 3 * c] + a12[p][1] * Rrw[1 + 3 * c] + a12[p][2] * Rrw[2 + 3 * c];
                B13[p * 13 + 10 + c] = 0.0;
            }
    }
}

__device__ __forceinline__ void jacobian_finish(const Cam& cam, double hu_, double hv_, const double B13[26], double H13[26])
{
    const double du = hu_ - cam.Cx, dv = hv_ - cam.Cy;
    const double xd = du * cam.dx, yd = dv * cam.dy;
    const double rd2 = xd * xd + yd * yd;
    const double g = 1 + cam.k1 * rd2 + cam.k2 * rd2 * rd2;
    const double gk = cam.k1 + 2 * cam.k2 * rd2;
    const double uu_ud = g + du * gk * (2 * du * cam.dx * cam.dx);
    const double vu_vd = g + dv * gk * (2 * dv * cam.dy * cam.dy);
    const double uu_vd = du * gk * (2 * dv * cam.dy * cam.dy);
    const double vu_ud = dv * gk * (2 * du * cam.dx * cam.dx);
    const double idet = 1.0 / (uu_ud * vu_vd - uu_vd * vu_ud);
    const double a1_00 = vu_vd * idet, a1_01 = -uu_vd * idet, a1_10 = -vu_ud * idet, a1_11 = uu_ud * idet;
#pragma unroll
    for (int c = 0; c < 13; ++c) {
        H13[c]      = a1_00 * B13[c] + a1_01 * B13[13 + c];
        H13[13 + c] = a1_10 * B13[c] + a1_11 * B13[13 + c];
    }
}

__device__ __forceinline__ void feature_jacobian(const Cam& cam, const double* x, int off, bool is_id,
                                                 double hu_, double hv_, double H13[26])
{
    double B13[26];
    jacobian_core(cam, x, off, is_id, B13);
    jacobian_finish(cam, hu_, hv_, B13, H13);
}

// state index of compact column c (0..12) of a feature at state offset off
__device__ __forceinline__ int col_index(int off, int c) { return c < 7 ? c : off + (c - 7); }

}  // namespace rslam
