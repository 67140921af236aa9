/*
 * rslam_oracle.c -- CPU restatement of the reference's 1-point-RANSAC EKF
 * update path.  TEST INFRASTRUCTURE ONLY (see rslam_oracle.h): never linked
 * into, imported by or called from the product path.
 *
 * PARITY UNPINNED: the reference has no tests/fixtures for this path and
 * cannot be built in this image (Eigen/OpenCV/ROS absent); this file follows
 * the reference source line by line and is pinned by the KATs in
 * tests/test_oracle_kat.py and a numpy/scipy cross-check only.
 *
 * Citations are file:line into /root/reference.  All matrices column-major
 * FP64 like Eigen's default; products are evaluated left to right as the
 * reference's expressions are; each inner product runs over ascending k.
 * Eigen's own blocked kernels may sum in another order (its version is not
 * even pinned: cmake_modules/FindEigen3.cmake:19-30), so agreement with a real
 * build of the reference is to rounding, not bitwise.
 */
#include "rslam_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

struct orc_ctx {
    rslam_config cfg;
    int structure;
    /* layout */
    int n, L;
    uint8_t* type;
    int32_t* offset;
    /* filter members, ExtendKF.h:154-169 */
    double *x_k_km1, *p_k_km1, *x_k_k, *p_k_k;
    /* features_info[], ExtendKF.h:14-42 (hot fields only) */
    uint8_t* has_h;      /* h.cols() != 0                */
    double*  h;          /* L*2                          */
    double*  H;          /* L blocks of 2 x n, col-major */
    double*  S;          /* L*4                          */
    double*  z;          /* L*2                          */
    uint8_t *ic, *li, *hi;
    /* introspection */
    int      n_eval, words, cap_eval;
    int32_t* supports;
    int32_t* positions;
    uint64_t* masks;
    double   score_margin, rescue_margin;
    /* residual capture for the value-level check of the device's scoring arithmetic (orc_enable_residuals):
     * res[p * res_m + j] = residual (pixels) of matched feature j under the hypothesis of matched feature p */
    int      res_on, res_m;
    double*  res;
    double  *x_li, *p_li;
    int      predicted;
};

/* ------------------------------------------------------------------ */
/* small dense helpers                                                  */
/* ------------------------------------------------------------------ */

/* C(m x n) = A(m x k) * B(k x n) */
/* The two dense products below carry almost all of the oracle's time.  Columns of C are independent, so an
 * OpenMP build (make OMP=1) splits them over threads without changing a single rounding: the "optimised CPU,
 * all cores" figure of bench.py.  The default build is single-threaded like the reference. */
static void gemm_nn(int m, int n, int k, const double* A, int lda,
                    const double* B, int ldb, double* C, int ldc)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if ((double)m * n * k > 1e6)
#endif
    for (int j = 0; j < n; ++j) {
        double* c = C + (size_t)j * ldc;
        for (int i = 0; i < m; ++i) c[i] = 0.0;
        for (int p = 0; p < k; ++p) {
            const double b = B[p + (size_t)j * ldb];
            const double* a = A + (size_t)p * lda;
            for (int i = 0; i < m; ++i) c[i] += a[i] * b;
        }
    }
}
/* C(m x n) = A(m x k) * B(n x k)^T */
static void gemm_nt(int m, int n, int k, const double* A, int lda,
                    const double* B, int ldb, double* C, int ldc)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if ((double)m * n * k > 1e6)
#endif
    for (int j = 0; j < n; ++j) {
        double* c = C + (size_t)j * ldc;
        for (int i = 0; i < m; ++i) c[i] = 0.0;
        for (int p = 0; p < k; ++p) {
            const double b = B[j + (size_t)p * ldb];
            const double* a = A + (size_t)p * lda;
            for (int i = 0; i < m; ++i) c[i] += a[i] * b;
        }
    }
}

/* Dynamic-size MatrixXd::inverse() == PartialPivLU(A).inverse()
 * (used at Tracking.cpp:421,591 and ExtendKF.cpp:603). */
int orc_inverse_lu(int n, const double* A, double* Ainv)
{
    double* lu = (double*)malloc(sizeof(double) * (size_t)n * n);
    int* piv = (int*)malloc(sizeof(int) * (size_t)n);
    if (!lu || !piv) { free(lu); free(piv); return RSLAM_ERR_ARG; }
    memcpy(lu, A, sizeof(double) * (size_t)n * n);
    for (int i = 0; i < n; ++i) piv[i] = i;
    for (int k = 0; k < n; ++k) {
        int p = k; double best = fabs(lu[k + (size_t)k * n]);
        for (int i = k + 1; i < n; ++i) {
            double v = fabs(lu[i + (size_t)k * n]);
            if (v > best) { best = v; p = i; }
        }
        if (p != k) {
            for (int j = 0; j < n; ++j) {
                double t = lu[k + (size_t)j * n];
                lu[k + (size_t)j * n] = lu[p + (size_t)j * n];
                lu[p + (size_t)j * n] = t;
            }
            int t = piv[k]; piv[k] = piv[p]; piv[p] = t;
        }
        const double d = lu[k + (size_t)k * n];
        for (int i = k + 1; i < n; ++i) lu[i + (size_t)k * n] /= d;
        for (int j = k + 1; j < n; ++j) {
            const double u = lu[k + (size_t)j * n];
            for (int i = k + 1; i < n; ++i)
                lu[i + (size_t)j * n] -= lu[i + (size_t)k * n] * u;
        }
    }
    /* solve A X = I : X = U^-1 L^-1 P */
    for (int c = 0; c < n; ++c) {
        double* x = Ainv + (size_t)c * n;
        for (int i = 0; i < n; ++i) x[i] = (piv[i] == c) ? 1.0 : 0.0;
        for (int k = 0; k < n; ++k) {          /* unit lower */
            const double v = x[k];
            if (v != 0.0)
                for (int i = k + 1; i < n; ++i) x[i] -= lu[i + (size_t)k * n] * v;
        }
        for (int k = n - 1; k >= 0; --k) {     /* upper */
            x[k] /= lu[k + (size_t)k * n];
            const double v = x[k];
            for (int i = 0; i < k; ++i) x[i] -= lu[i + (size_t)k * n] * v;
        }
    }
    free(lu); free(piv);
    return RSLAM_OK;
}

/* Fixed-size Matrix2d::inverse() (closed form, Eigen compute_inverse_size2):
 * used for a1 = jacob_undistor_fm(zi).inverse(), Tracking.cpp:89,131. */
static void inv2_fixed(const double M[4], double R[4])
{
    const double det = M[0] * M[3] - M[2] * M[1];
    const double invdet = 1.0 / det;
    R[0] =  M[3] * invdet;
    R[1] = -M[1] * invdet;
    R[2] = -M[2] * invdet;
    R[3] =  M[0] * invdet;
}
/* Fixed-size Matrix3d::inverse() (cofactors, Eigen compute_inverse_size3):
 * Rrw = q2r(q).inverse(), Tracking.cpp:90,136; ExtendKF.cpp:83. */
static void inv3_fixed(const double M[9], double R[9])
{
#define MM(i, j) M[(i) + 3 * (j)]
    const double c00 = MM(1,1) * MM(2,2) - MM(1,2) * MM(2,1);
    const double c10 = MM(0,2) * MM(2,1) - MM(0,1) * MM(2,2);   /* cofactor(1,0) */
    const double c20 = MM(0,1) * MM(1,2) - MM(0,2) * MM(1,1);
    const double det = c00 * MM(0,0) + c10 * MM(1,0) + c20 * MM(2,0);
    const double invdet = 1.0 / det;
    R[0 + 3 * 0] = c00 * invdet;
    R[0 + 3 * 1] = c10 * invdet;
    R[0 + 3 * 2] = c20 * invdet;
    R[1 + 3 * 0] = (MM(1,2) * MM(2,0) - MM(1,0) * MM(2,2)) * invdet;
    R[1 + 3 * 1] = (MM(0,0) * MM(2,2) - MM(0,2) * MM(2,0)) * invdet;
    R[1 + 3 * 2] = (MM(0,2) * MM(1,0) - MM(0,0) * MM(1,2)) * invdet;
    R[2 + 3 * 0] = (MM(1,0) * MM(2,1) - MM(1,1) * MM(2,0)) * invdet;
    R[2 + 3 * 1] = (MM(0,1) * MM(2,0) - MM(0,0) * MM(2,1)) * invdet;
    R[2 + 3 * 2] = (MM(0,0) * MM(1,1) - MM(0,1) * MM(1,0)) * invdet;
#undef MM
}

/* ------------------------------------------------------------------ */
/* camera model                                                         */
/* ------------------------------------------------------------------ */

/* ExtendKF::q2r, ExtendKF.cpp:91-102.  q = (r, x, y, z); R col-major. */
void orc_q2r(const double q[4], double R[9])
{
    const double x = q[1], y = q[2], z = q[3], r = q[0];
    R[0 + 3 * 0] = r * r + x * x - y * y - z * z;
    R[0 + 3 * 1] = 2 * (x * y - r * z);
    R[0 + 3 * 2] = 2 * (z * x + r * y);
    R[1 + 3 * 0] = 2 * (x * y + r * z);
    R[1 + 3 * 1] = r * r - x * x + y * y - z * z;
    R[1 + 3 * 2] = 2 * (y * z - r * x);
    R[2 + 3 * 0] = 2 * (z * x - r * y);
    R[2 + 3 * 1] = 2 * (y * z + r * x);
    R[2 + 3 * 2] = r * r - x * x - y * y + z * z;
}

/* ExtendKF::hu, ExtendKF.cpp:153-174 */
void orc_hu(const rslam_camera* cam, const double y[3], double uv[2])
{
    const double u0 = cam->Cx, v0 = cam->Cy, f = cam->f;
    const double ku = 1.0 / cam->dx, kv = 1.0 / cam->dy;
    uv[0] = u0 + (y[0] / y[2]) * f * ku;
    uv[1] = v0 + (y[1] / y[2]) * f * kv;
}

/* ExtendKF::distort_fm, ExtendKF.cpp:175-204 (one column); Eigen's
 * .array().pow() is std::pow per element. */
void orc_distort_fm(const rslam_camera* cam, const double uv[2], double uvd[2])
{
    const double Cx = cam->Cx, Cy = cam->Cy, k1 = cam->k1, k2 = cam->k2;
    const double dx = cam->dx, dy = cam->dy;
    const double xu = (uv[0] - Cx) * dx;
    const double yu = (uv[1] - Cy) * dy;
    const double ru = sqrt(pow(xu, 2) + pow(yu, 2));
    double rd = ru / (1 + k1 * pow(ru, 2) + k2 * pow(ru, 4));
    for (int k = 0; k < 10; ++k) {
        const double f   = rd + k1 * pow(rd, 3) + k2 * pow(rd, 5) - ru;
        const double f_p = 1 + 3 * k1 * pow(rd, 2) + 5 * k2 * pow(rd, 4);
        rd = rd - f / f_p;
    }
    const double D = 1 + k1 * pow(rd, 2) + k2 * pow(rd, 4);
    uvd[0] = xu / D / dx + Cx;
    uvd[1] = yu / D / dy + Cy;
}

/* ExtendKF::undistort_fm, ExtendKF.cpp:266-285 */
void orc_undistort_fm(const rslam_camera* cam, const double uvd[2], double uvu[2])
{
    const double Cx = cam->Cx, Cy = cam->Cy, k1 = cam->k1, k2 = cam->k2;
    const double dx = cam->dx, dy = cam->dy;
    const double xd = (uvd[0] - Cx) * dx;
    const double yd = (uvd[1] - Cy) * dy;
    const double rd = sqrt(pow(xd, 2) + pow(yd, 2));
    const double D = 1 + k1 * pow(rd, 2) + k2 * pow(rd, 4);
    uvu[0] = xd * D / dx + Cx;
    uvu[1] = yd * D / dy + Cy;
}

/* ExtendKF::jacob_undistor_fm, ExtendKF.cpp:312-332.  J col-major 2x2. */
void orc_jacob_undistor_fm(const rslam_camera* cam, const double uvd[2], double J[4])
{
    const double Cx = cam->Cx, Cy = cam->Cy, k1 = cam->k1, k2 = cam->k2;
    const double dx = cam->dx, dy = cam->dy;
    const double ud = uvd[0], vd = uvd[1];
    const double rd2 = pow((ud - Cx) * dx, 2) + pow((vd - Cy) * dy, 2);
    const double uu_ud = (1 + k1 * rd2 + k2 * rd2 * rd2) + (ud - Cx) * (k1 + 2 * k2 * rd2) * (2 * (ud - Cx) * dx * dx);
    const double vu_vd = (1 + k1 * rd2 + k2 * rd2 * rd2) + (vd - Cy) * (k1 + 2 * k2 * rd2) * (2 * (vd - Cy) * dy * dy);
    const double uu_vd = (ud - Cx) * (k1 + 2 * k2 * rd2) * (2 * (vd - Cy) * dy * dy);
    const double vu_ud = (vd - Cy) * (k1 + 2 * k2 * rd2) * (2 * (ud - Cx) * dx * dx);
    J[0] = uu_ud; J[2] = uu_vd;       /* row 0: uu_ud, uu_vd */
    J[1] = vu_ud; J[3] = vu_vd;       /* row 1: vu_ud, vu_vd */
}

/* ExtendKF::dRq_times_a_by_dq, ExtendKF.cpp:286-311.  out col-major 3x4. */
void orc_dRq_times_a_by_dq(const double q[4], const double a[3], double out[12])
{
    double T[9];   /* row-major as written by operator<< */
#define SETT(a0,a1,a2,b0,b1,b2,c0,c1,c2) do { T[0]=a0;T[1]=a1;T[2]=a2;T[3]=b0;T[4]=b1;T[5]=b2;T[6]=c0;T[7]=c1;T[8]=c2; } while (0)
#define MULCOL(c) do { for (int i_ = 0; i_ < 3; ++i_) out[i_ + 3 * (c)] = T[3*i_+0] * a[0] + T[3*i_+1] * a[1] + T[3*i_+2] * a[2]; } while (0)
    SETT( 2*q[0], -2*q[3],  2*q[2],   2*q[3],  2*q[0], -2*q[1],  -2*q[2],  2*q[1],  2*q[0]); MULCOL(0);
    SETT( 2*q[1],  2*q[2],  2*q[3],   2*q[2], -2*q[1], -2*q[0],   2*q[3],  2*q[0], -2*q[1]); MULCOL(1);
    SETT(-2*q[2],  2*q[1],  2*q[0],   2*q[1],  2*q[2],  2*q[3],  -2*q[0],  2*q[3], -2*q[2]); MULCOL(2);
    SETT(-2*q[3], -2*q[0],  2*q[1],   2*q[0], -2*q[3],  2*q[2],   2*q[1],  2*q[2],  2*q[3]); MULCOL(3);
#undef SETT
#undef MULCOL
}

/* ExtendKF::hi_cartesian, ExtendKF.cpp:103-132.  Returns 1 and uv when the
 * point is in the +-60 degree FOV and inside the image, else 0 ("empty"). */
int orc_hi_cartesian(const rslam_camera* cam, const double hrl[3], double uv[2])
{
    if ((atan2(hrl[0], hrl[2]) * 180 / M_PI < -60) ||
        (atan2(hrl[0], hrl[2]) * 180 / M_PI >  60) ||
        (atan2(hrl[1], hrl[2]) * 180 / M_PI < -60) ||
        (atan2(hrl[1], hrl[2]) * 180 / M_PI >  60))
        return 0;
    double uv_u[2], uv_d[2];
    orc_hu(cam, hrl, uv_u);
    orc_distort_fm(cam, uv_u, uv_d);
    if ((uv_d[0] > 0) && (uv_d[0] < cam->nCols) && (uv_d[1] > 0) && (uv_d[1] < cam->nRows)) {
        uv[0] = uv_d[0]; uv[1] = uv_d[1];
        return 1;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* context                                                              */
/* ------------------------------------------------------------------ */

int orc_create(const rslam_config* cfg, orc_ctx** out)
{
    if (!cfg || !out) return RSLAM_ERR_ARG;
    orc_ctx* c = (orc_ctx*)calloc(1, sizeof(orc_ctx));
    if (!c) return RSLAM_ERR_ARG;
    c->cfg = *cfg;
    c->structure = 0;
    *out = c;
    return RSLAM_OK;
}

static void free_frame(orc_ctx* c)
{
    free(c->type); free(c->offset);
    free(c->x_k_km1); free(c->p_k_km1); free(c->x_k_k); free(c->p_k_k);
    free(c->has_h); free(c->h); free(c->H); free(c->S); free(c->z);
    free(c->ic); free(c->li); free(c->hi);
    free(c->x_li); free(c->p_li);
    c->type = NULL; c->offset = NULL;
    c->x_k_km1 = c->p_k_km1 = c->x_k_k = c->p_k_k = NULL;
    c->has_h = NULL; c->h = c->H = c->S = c->z = NULL;
    c->ic = c->li = c->hi = NULL;
    c->x_li = c->p_li = NULL;
}

int orc_destroy(orc_ctx* c)
{
    if (!c) return RSLAM_ERR_ARG;
    free_frame(c);
    free(c->supports); free(c->positions); free(c->masks); free(c->res);
    free(c);
    return RSLAM_OK;
}

int orc_set_structure(orc_ctx* c, int structure)
{
    if (!c || structure < 0 || structure > 1) return RSLAM_ERR_ARG;
    c->structure = structure;
    return RSLAM_OK;
}

/* ------------------------------------------------------------------ */
/* measurement prediction and Jacobians                                 */
/* ------------------------------------------------------------------ */

/* ExtendKF::predict_camera_measurements, ExtendKF.cpp:56-90.  Features that
 * are not visible keep whatever h they had (":77-78 if(hi.rows()!=0)"). */
static void predict_camera_measurements(orc_ctx* c, const double* xkk, uint8_t* visible_now)
{
    const double* t_wc = xkk;
    double r_wc[9];
    orc_q2r(xkk + 3, r_wc);
    for (int i = 0; i < c->L; ++i) {
        const double* yi = xkk + c->offset[i];
        double hrl[3], uv[2];
        if (c->type[i] == RSLAM_FEAT_INVERSE_DEPTH) {
            double mi[3], v[3];
            mi[0] = cos(yi[4]) * sin(yi[3]);
            mi[1] = -sin(yi[4]);
            mi[2] = cos(yi[4]) * cos(yi[3]);
            for (int a = 0; a < 3; ++a) v[a] = (yi[a] - t_wc[a]) * yi[5] + mi[a];
            for (int a = 0; a < 3; ++a)      /* r_wc.transpose() * v */
                hrl[a] = r_wc[0 + 3 * a] * v[0] + r_wc[1 + 3 * a] * v[1] + r_wc[2 + 3 * a] * v[2];
        } else {
            double rinv[9], v[3];
            inv3_fixed(r_wc, rinv);          /* r_wc.inverse() * (yi - t_wc) */
            for (int a = 0; a < 3; ++a) v[a] = yi[a] - t_wc[a];
            for (int a = 0; a < 3; ++a)
                hrl[a] = rinv[a + 3 * 0] * v[0] + rinv[a + 3 * 1] * v[1] + rinv[a + 3 * 2] * v[2];
        }
        const int vis = orc_hi_cartesian(&c->cfg.cam, hrl, uv);
        if (vis) {
            c->h[2 * i] = uv[0]; c->h[2 * i + 1] = uv[1];
            c->has_h[i] = 1;
        }
        if (visible_now) visible_now[i] = (uint8_t)vis;
    }
}

/* helper: M(2x3) = A(2x2) * B(2x3), all col-major */
static void mul_2x2_2x3(const double A[4], const double B[6], double M[6])
{
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 2; ++i)
            M[i + 2 * j] = A[i + 2 * 0] * B[0 + 2 * j] + A[i + 2 * 1] * B[1 + 2 * j];
}
/* helper: M(2xk) = A(2x3) * B(3xk) */
static void mul_2x3_3xk(const double A[6], const double* B, int k, double* M)
{
    for (int j = 0; j < k; ++j)
        for (int i = 0; i < 2; ++i)
            M[i + 2 * j] = A[i + 2 * 0] * B[0 + 3 * j] + A[i + 2 * 1] * B[1 + 3 * j] + A[i + 2 * 2] * B[2 + 3 * j];
}

/* Tracking::calculate_Hi_inverse_depth, Tracking.cpp:113-163 and
 * Tracking::calculate_Hi_cartesian, Tracking.cpp:71-112.
 * Hi is the dense 2 x n block (col-major), zero outside columns 0..6 and the
 * feature's own columns (a32 = 0 for the velocity columns 7..12). */
static void calculate_Hi(orc_ctx* c, const double* x_v, int order, double* Hi)
{
    const rslam_camera* cam = &c->cfg.cam;
    const int n = c->n;
    const double* yi = x_v + c->offset[order];
    const double* zi = c->h + 2 * order;
    memset(Hi, 0, sizeof(double) * 2 * (size_t)n);

    double J[4], a1[4];
    orc_jacob_undistor_fm(cam, zi, J);
    inv2_fixed(J, a1);
    const double f = cam->f, ku = 1 / cam->dx, kv = 1 / cam->dy;
    double Rq[9], Rrw[9];
    orc_q2r(x_v + 3, Rq);
    inv3_fixed(Rq, Rrw);

    double hc[3], arg[3];
    const int is_id = (c->type[order] == RSLAM_FEAT_INVERSE_DEPTH);
    double mi[3] = {0, 0, 0};
    if (is_id) {
        mi[0] = cos(yi[4]) * sin(yi[3]);
        mi[1] = -sin(yi[4]);
        mi[2] = cos(yi[4]) * cos(yi[3]);
        for (int a = 0; a < 3; ++a) arg[a] = (yi[a] - x_v[a]) * yi[5] + mi[a];
    } else {
        for (int a = 0; a < 3; ++a) arg[a] = yi[a] - x_v[a];
    }
    for (int a = 0; a < 3; ++a)
        hc[a] = Rrw[a + 3 * 0] * arg[0] + Rrw[a + 3 * 1] * arg[1] + Rrw[a + 3 * 2] * arg[2];

    double a2[6];   /* 2x3 col-major */
    a2[0 + 2 * 0] = f * ku / (hc[2]);  a2[0 + 2 * 1] = 0;                 a2[0 + 2 * 2] = -hc[0] * f * ku / (hc[2] * hc[2]);
    a2[1 + 2 * 0] = 0;                 a2[1 + 2 * 1] = f * kv / (hc[2]);  a2[1 + 2 * 2] = -hc[1] * f * kv / (hc[2] * hc[2]);

    double a12[6];
    mul_2x2_2x3(a1, a2, a12);                       /* (a1 * a2) */

    /* a30 = a1 * a2 * (-Rrw) [* yi(5)] */
    double nR[9], a30[6];
    for (int k = 0; k < 9; ++k) nR[k] = -Rrw[k];
    mul_2x3_3xk(a12, nR, 3, a30);
    if (is_id) for (int k = 0; k < 6; ++k) a30[k] = a30[k] * yi[5];

    /* b1 = qconj(q); b0 = dRq_times_a_by_dq(b1, arg) * diag(1,-1,-1,-1) */
    double b1[4] = { x_v[3], -x_v[4], -x_v[5], -x_v[6] };
    double b0[12], a31[8];
    orc_dRq_times_a_by_dq(b1, arg, b0);
    for (int col = 1; col < 4; ++col)
        for (int i = 0; i < 3; ++i) b0[i + 3 * col] = b0[i + 3 * col] * -1.0;
    mul_2x3_3xk(a12, b0, 4, a31);

    for (int k = 0; k < 6; ++k) Hi[k] = a30[k];               /* cols 0..2  */
    for (int k = 0; k < 8; ++k) Hi[2 * 3 + k] = a31[k];       /* cols 3..6  */
    /* cols 7..12: a32 = zeros(2,6) */

    double* Hf = Hi + 2 * (size_t)c->offset[order];
    if (is_id) {
        double c0[18];                                         /* 3x6 */
        for (int k = 0; k < 9; ++k) c0[k] = yi[5] * Rrw[k];    /* c1 */
        const double c2[3] = {  cos(yi[4]) * cos(yi[3]), 0, -cos(yi[4]) * sin(yi[3]) };
        const double c3[3] = { -sin(yi[4]) * sin(yi[3]), -cos(yi[4]), -sin(yi[4]) * cos(yi[3]) };
        double d[3];
        for (int a = 0; a < 3; ++a) d[a] = yi[a] - x_v[a];
        for (int a = 0; a < 3; ++a) {
            c0[a + 3 * 3] = Rrw[a + 3 * 0] * c2[0] + Rrw[a + 3 * 1] * c2[1] + Rrw[a + 3 * 2] * c2[2];
            c0[a + 3 * 4] = Rrw[a + 3 * 0] * c3[0] + Rrw[a + 3 * 1] * c3[1] + Rrw[a + 3 * 2] * c3[2];
            c0[a + 3 * 5] = Rrw[a + 3 * 0] * d[0]  + Rrw[a + 3 * 1] * d[1]  + Rrw[a + 3 * 2] * d[2];
        }
        mul_2x3_3xk(a12, c0, 6, Hf);
    } else {
        mul_2x3_3xk(a12, Rrw, 3, Hf);
    }
}

/* Tracking::calculate_derivatives, Tracking.cpp:540-573 */
static void calculate_derivatives(orc_ctx* c, const double* xk)
{
    for (int i = 0; i < c->L; ++i)
        if (c->has_h[i])
            calculate_Hi(c, xk, i, c->H + (size_t)i * 2 * c->n);
}

/* columns of H_i that can be non-zero: 0..6 and the feature's own */
static int nz_cols(const orc_ctx* c, int i, int cols[13])
{
    int k = 0;
    for (int j = 0; j < 7; ++j) cols[k++] = j;
    const int w = (c->type[i] == RSLAM_FEAT_INVERSE_DEPTH) ? 6 : 3;
    for (int j = 0; j < w; ++j) cols[k++] = c->offset[i] + j;
    return k;
}

/* (Hi * P) * Hi^T for one feature; R_add added to the diagonal when != 0.
 * structure 0: dense (2 x n)(n x n)(n x 2) as Tracking.cpp:42,420,589. */
static void HPHt_2x2(orc_ctx* c, int i, const double* P, double radd, double S[4], double* scratch_2n)
{
    const int n = c->n;
    const double* Hi = c->H + (size_t)i * 2 * n;
    if (c->structure == 0) {
        gemm_nn(2, n, n, Hi, 2, P, n, scratch_2n, 2);
        gemm_nt(2, 2, n, scratch_2n, 2, Hi, 2, S, 2);
    } else {
        int cols[13]; const int nc = nz_cols(c, i, cols);
        double HP[26];   /* 2 x nc : (Hi*P)[:, cols] */
        for (int jj = 0; jj < nc; ++jj) {
            double s0 = 0, s1 = 0;
            for (int kk = 0; kk < nc; ++kk) {
                const double p = P[cols[kk] + (size_t)cols[jj] * n];
                s0 += Hi[0 + 2 * cols[kk]] * p;
                s1 += Hi[1 + 2 * cols[kk]] * p;
            }
            HP[0 + 2 * jj] = s0; HP[1 + 2 * jj] = s1;
        }
        for (int b = 0; b < 2; ++b)
            for (int a = 0; a < 2; ++a) {
                double s = 0;
                for (int jj = 0; jj < nc; ++jj) s += HP[a + 2 * jj] * Hi[b + 2 * cols[jj]];
                S[a + 2 * b] = s;
            }
    }
    S[0] += radd; S[3] += radd;
}

int orc_predict(orc_ctx* c, const rslam_layout* lay, const double* x_pred,
                const double* P_pred, double* h, uint8_t* visible, double* S)
{
    if (!c || !lay || !x_pred || !P_pred || !lay->type || !lay->offset) return RSLAM_ERR_ARG;
    const int n = lay->n, L = lay->L;
    if (n < 13 || L < 0) return RSLAM_ERR_ARG;
    free_frame(c);
    c->n = n; c->L = L;
    const size_t Lz = L > 0 ? (size_t)L : 1;
    c->type = (uint8_t*)malloc(Lz); c->offset = (int32_t*)malloc(sizeof(int32_t) * Lz);
    memcpy(c->type, lay->type, (size_t)L); memcpy(c->offset, lay->offset, sizeof(int32_t) * (size_t)L);
    for (int i = 0; i < L; ++i) {
        const int w = (c->type[i] == RSLAM_FEAT_INVERSE_DEPTH) ? 6 : 3;
        if (c->type[i] > 1 || c->offset[i] < 13 || c->offset[i] + w > n) return RSLAM_ERR_ARG;
    }
    c->x_k_km1 = (double*)malloc(sizeof(double) * n);
    c->p_k_km1 = (double*)malloc(sizeof(double) * (size_t)n * n);
    c->x_k_k   = (double*)malloc(sizeof(double) * n);
    c->p_k_k   = (double*)malloc(sizeof(double) * (size_t)n * n);
    c->x_li    = (double*)malloc(sizeof(double) * n);
    c->p_li    = (double*)malloc(sizeof(double) * (size_t)n * n);
    memcpy(c->x_k_km1, x_pred, sizeof(double) * n);
    memcpy(c->p_k_km1, P_pred, sizeof(double) * (size_t)n * n);
    /* Map::map_management resets these at the start of every frame, Map.cpp:48-52 */
    c->has_h = (uint8_t*)calloc(Lz, 1);
    c->h = (double*)calloc(Lz * 2, sizeof(double));
    c->H = (double*)calloc(Lz * 2 * n, sizeof(double));
    c->S = (double*)calloc(Lz * 4, sizeof(double));
    c->z = (double*)calloc(Lz * 2, sizeof(double));
    c->ic = (uint8_t*)calloc(Lz, 1); c->li = (uint8_t*)calloc(Lz, 1); c->hi = (uint8_t*)calloc(Lz, 1);

    /* Tracking::search_IC_matches, Tracking.cpp:35-44 */
    uint8_t* vis = (uint8_t*)calloc(Lz, 1);
    predict_camera_measurements(c, c->x_k_km1, vis);
    calculate_derivatives(c, c->x_k_km1);
    double* scratch = (double*)malloc(sizeof(double) * 2 * n);
    for (int i = 0; i < L; ++i) {
        if (c->has_h[i]) {
            /* features_info[i].R = Identity(2,2), Map.cpp:310 */
            HPHt_2x2(c, i, c->p_k_km1, 1.0, c->S + 4 * i, scratch);
            if (h) { h[2 * i] = c->h[2 * i]; h[2 * i + 1] = c->h[2 * i + 1]; }
            if (S) memcpy(S + 4 * i, c->S + 4 * i, sizeof(double) * 4);
        }
        if (visible) visible[i] = vis[i];
    }
    free(scratch); free(vis);
    c->predicted = 1;
    return RSLAM_OK;
}

/* ------------------------------------------------------------------ */
/* 1-point RANSAC                                                       */
/* ------------------------------------------------------------------ */

/* Tracking.cpp:531-532 */
int orc_adaptive_n_hyp(double p, int support, int num_ic)
{
    const double epsilon = 1 - ((double)support / (double)num_ic);
    return (int)ceil((log(1 - p)) / (log(1 - (1 - epsilon))));
}

static void ensure_eval_capacity(orc_ctx* c, int need, int words)
{
    if (need <= c->cap_eval && words == c->words) return;
    free(c->supports); free(c->positions); free(c->masks);
    c->cap_eval = need; c->words = words;
    c->supports = (int32_t*)calloc((size_t)need + 1, sizeof(int32_t));
    c->positions = (int32_t*)calloc((size_t)need + 1, sizeof(int32_t));
    c->masks = (uint64_t*)calloc(((size_t)need + 1) * (size_t)(words > 0 ? words : 1), sizeof(uint64_t));
}

/* Tracking::ransac_hypotheses, Tracking.cpp:352-539 */
static int ransac_hypotheses(orc_ctx* c, const double* draws, int n_draws, int max_iters,
                             int* best_hyp, int* best_support, int* hyps_evaluated)
{
    const rslam_camera* cam = &c->cfg.cam;
    const int n = c->n, L = c->L;
    const int compat = c->cfg.compat;
    const double p_at_least_one_spurious_free = c->cfg.p_success;   /* :354 */
    const double threshold = c->cfg.sigma_z;                        /* :356 */
    /* :357; with adaptive == 0 the loop is pinned to exactly n_draws iterations */
    int n_hyp = c->cfg.adaptive ? c->cfg.n_hyp_init : n_draws;
    int max_hypothesis_support = 0;

    /* :361-397 state_vector_pattern / z_id / z_euc, kept as index lists.
     * "matched" <=> z.rows() > 0 <=> individually_compatible (Tracking.cpp:345-346). */
    int* id_list  = (int*)malloc(sizeof(int) * (size_t)(L + 1));   /* matched inverse-depth features */
    int* euc_list = (int*)malloc(sizeof(int) * (size_t)(L + 1));
    int* ic_list  = (int*)malloc(sizeof(int) * (size_t)(L + 1));   /* Converter::find(IC, 0), Converter.cpp:210-228 */
    int* rank_of  = (int*)malloc(sizeof(int) * (size_t)(L + 1));   /* mask bit of a matched feature */
    int m_id = 0, m_euc = 0, num_ic = 0;
    for (int i = 0; i < L; ++i) {
        if (c->ic[i]) {
            rank_of[i] = num_ic;
            ic_list[num_ic++] = i;
            if (c->type[i] == RSLAM_FEAT_INVERSE_DEPTH) id_list[m_id++] = i; else euc_list[m_euc++] = i;
        }
    }
    const int m = m_id + m_euc;
    const int words = (m + 63) / 64;
    int iters_cap = n_draws;
    if (max_iters > 0 && max_iters < iters_cap) iters_cap = max_iters;
    ensure_eval_capacity(c, iters_cap, words);
    if (c->res_on) {
        free(c->res);
        c->res_m = m;
        c->res = (double*)malloc(sizeof(double) * ((size_t)m * m + 1));
        for (size_t q = 0; q < (size_t)m * m; ++q) c->res[q] = NAN;
    }
    c->n_eval = 0;
    c->score_margin = DBL_MAX;
    *best_hyp = -1; *best_support = 0; *hyps_evaluated = 0;
    for (int i = 0; i < L; ++i) c->li[i] = 0;

    int rc = RSLAM_OK;
    if (num_ic == 0) { free(id_list); free(euc_list); free(ic_list); free(rank_of); return RSLAM_OK; }
    if (compat && m_euc > 0 && m_euc != m_id) {
        /* Q2: "nu = z_id - h_distorted" with mismatched column counts is an
         * Eigen assertion failure in the reference build (Tracking.cpp:498). */
        free(id_list); free(euc_list); free(ic_list); free(rank_of);
        return RSLAM_ERR_REF_ASSERT;
    }

    double* xi  = (double*)malloc(sizeof(double) * n);
    double* HP  = (double*)malloc(sizeof(double) * 2 * (size_t)n);
    double* PHt = (double*)malloc(sizeof(double) * 2 * (size_t)n);
    double* K   = (double*)malloc(sizeof(double) * 2 * (size_t)n);
    double* ri_v = (double*)malloc(sizeof(double) * 3 * (size_t)(m_id + 1));
    uint64_t* mask = (uint64_t*)calloc((size_t)(words > 0 ? words : 1), sizeof(uint64_t));
    /* structure 1: cache of already scored positions */
    int32_t* cache_sup = NULL; uint64_t* cache_mask = NULL; uint8_t* cache_ok = NULL;
    if (c->structure == 1) {
        cache_sup = (int32_t*)calloc((size_t)L, sizeof(int32_t));
        cache_mask = (uint64_t*)calloc((size_t)L * (size_t)(words > 0 ? words : 1), sizeof(uint64_t));
        cache_ok = (uint8_t*)calloc((size_t)L, 1);
    }

    const double ku = 1 / (double)(cam->dx);
    const double f = cam->f, u0 = cam->Cx, v0 = cam->Cy;

    for (int i = 0; i < n_hyp; ++i) {                                   /* :403 */
        /* the caller's draw list bounds the loop (include/rslam.h: supply
         * n_draws >= rslam_max_hypotheses() so the reference's rule never truncates) */
        if (i >= iters_cap) break;
        /* :412-417 select a random IC match */
        const double t = draws[i];
        int random_match_position = (int)floor(t * (double)num_ic);
        if (random_match_position >= num_ic) random_match_position = num_ic - 1;   /* Q3 guard */
        if (random_match_position < 0) random_match_position = 0;
        const int position = ic_list[random_match_position];
        const double* zi = c->z + 2 * position;
        int hypothesis_support = 0;

        if (c->structure == 1 && cache_ok[position]) {
            hypothesis_support = cache_sup[position];
            memcpy(mask, cache_mask + (size_t)position * words, sizeof(uint64_t) * words);
        } else {
            /* :419-422  S = Hi*P*Hi' + R ; K = P*Hi'*inv(S) ; xi = x + K*(zi - h') */
            const double* Hi = c->H + (size_t)position * 2 * n;
            double S[4], Sinv[4];
            HPHt_2x2(c, position, c->p_k_km1, 1.0, S, HP);
            if (c->structure == 0) {
                gemm_nt(n, 2, n, c->p_k_km1, n, Hi, 2, PHt, n);
            } else {
                int cols[13]; const int nc = nz_cols(c, position, cols);
                for (int a = 0; a < 2; ++a)
                    for (int r = 0; r < n; ++r) {
                        double s = 0;
                        for (int kk = 0; kk < nc; ++kk)
                            s += c->p_k_km1[r + (size_t)cols[kk] * n] * Hi[a + 2 * cols[kk]];
                        PHt[r + (size_t)a * n] = s;
                    }
            }
            orc_inverse_lu(2, S, Sinv);
            gemm_nn(n, 2, 2, PHt, n, Sinv, 2, K, n);
            const double nu0 = zi[0] - c->h[2 * position], nu1 = zi[1] - c->h[2 * position + 1];
            for (int r = 0; r < n; ++r)
                xi[r] = c->x_k_km1[r] + (K[r] * nu0 + K[r + (size_t)n] * nu1);

            /* :425-503 hypothesis support */
            double Rq[9], rotcw[9];
            orc_q2r(xi + 3, Rq);
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) rotcw[a + 3 * b] = Rq[b + 3 * a];
            memset(mask, 0, sizeof(uint64_t) * (size_t)(words > 0 ? words : 1));

            if (m_id) {
                /* :443-449 select(); Q1: anglesi is mapped from ri_v (not anglesi_v) */
                for (int j = 0; j < m_id; ++j)
                    for (int a = 0; a < 3; ++a) ri_v[3 * j + a] = xi[c->offset[id_list[j]] + a];
                for (int j = 0; j < m_id; ++j) {
                    const int fo = c->offset[id_list[j]];
                    double th, ph;
                    if (compat) { th = ri_v[2 * j]; ph = ri_v[2 * j + 1]; }
                    else        { th = xi[fo + 3]; ph = xi[fo + 4]; }
                    const double rho = xi[fo + 5];
                    double mi[3], v[3], hc[3];
                    mi[0] = cos(ph) * sin(th);
                    mi[1] = -sin(ph);
                    mi[2] = cos(ph) * cos(th);
                    for (int a = 0; a < 3; ++a) v[a] = (xi[fo + a] - xi[a]) * rho + mi[a];
                    for (int a = 0; a < 3; ++a)
                        hc[a] = rotcw[a + 3 * 0] * v[0] + rotcw[a + 3 * 1] * v[1] + rotcw[a + 3 * 2] * v[2];
                    double h_norm[2] = { hc[0] / hc[2], hc[1] / hc[2] };
                    double h_image[2] = { f * ku * h_norm[0] + u0, f * ku * h_norm[1] + v0 };   /* :471 ku on both axes */
                    double h_dist[2];
                    orc_distort_fm(cam, h_image, h_dist);
                    const double n0 = c->z[2 * id_list[j]] - h_dist[0];
                    const double n1 = c->z[2 * id_list[j] + 1] - h_dist[1];
                    const double residual = sqrt(pow(n0, 2) + pow(n1, 2));
                    if (c->res_on) c->res[(size_t)random_match_position * m + rank_of[id_list[j]]] = residual;
                    const double mg = fabs(residual - threshold);
                    if (mg < c->score_margin) c->score_margin = mg;
                    if (residual < threshold) {
                        /* bit index = rank of the feature among matched features in feature order */
                        const int rank = rank_of[id_list[j]];
                        mask[rank >> 6] |= (1ull << (rank & 63));
                        hypothesis_support++;
                    }
                }
            }
            if (m_euc) {
                /* :480-503 */
                for (int j = 0; j < m_euc; ++j) {
                    const int fo = c->offset[euc_list[j]];
                    double v[3], hc[3];
                    for (int a = 0; a < 3; ++a) v[a] = xi[fo + a] - xi[a];
                    for (int a = 0; a < 3; ++a)
                        hc[a] = rotcw[a + 3 * 0] * v[0] + rotcw[a + 3 * 1] * v[1] + rotcw[a + 3 * 2] * v[2];
                    double h_norm[2] = { hc[0] / hc[2], hc[1] / hc[2] };
                    double h_image[2] = { f * ku * h_norm[0] + u0, f * ku * h_norm[1] + v0 };
                    double h_dist[2];
                    orc_distort_fm(cam, h_image, h_dist);
                    /* Q2 (:498): compat subtracts from z_id (column j of the inverse-depth list) */
                    const int zsrc = compat ? id_list[j] : euc_list[j];
                    const double n0 = c->z[2 * zsrc] - h_dist[0];
                    const double n1 = c->z[2 * zsrc + 1] - h_dist[1];
                    const double residual = sqrt(pow(n0, 2) + pow(n1, 2));
                    if (c->res_on) c->res[(size_t)random_match_position * m + rank_of[euc_list[j]]] = residual;
                    const double mg = fabs(residual - threshold);
                    if (mg < c->score_margin) c->score_margin = mg;
                    if (residual < threshold) {
                        const int rank = rank_of[euc_list[j]];
                        mask[rank >> 6] |= (1ull << (rank & 63));
                        hypothesis_support++;
                    }
                }
            }
            if (c->structure == 1) {
                cache_ok[position] = 1; cache_sup[position] = hypothesis_support;
                memcpy(cache_mask + (size_t)position * words, mask, sizeof(uint64_t) * words);
            }
        }

        c->supports[i] = hypothesis_support;
        c->positions[i] = position;
        memcpy(c->masks + (size_t)i * words, mask, sizeof(uint64_t) * words);
        c->n_eval = i + 1;

        /* :507-535 */
        if (hypothesis_support > max_hypothesis_support) {
            max_hypothesis_support = hypothesis_support;
            *best_hyp = i; *best_support = hypothesis_support;
            for (int q = 0; q < num_ic; ++q)
                c->li[ic_list[q]] = (uint8_t)((mask[q >> 6] >> (q & 63)) & 1ull);
            if (c->cfg.adaptive) {
                n_hyp = orc_adaptive_n_hyp(p_at_least_one_spurious_free, hypothesis_support, num_ic);
                if (n_hyp == 0) break;
            }
        }
        if (c->cfg.adaptive) { if (i > n_hyp) break; }                    /* :536 */
    }
    *hyps_evaluated = c->n_eval;

    free(xi); free(HP); free(PHt); free(K); free(ri_v); free(mask);
    free(cache_sup); free(cache_mask); free(cache_ok);
    free(id_list); free(euc_list); free(ic_list); free(rank_of);
    return rc;
}

/* ------------------------------------------------------------------ */
/* EKF update                                                           */
/* ------------------------------------------------------------------ */

/* ExtendKF::update, ExtendKF.cpp:597-639.  H is r x n (ld = r). */
int orc_update(int compat, int n, int r, const double* x_km_k, const double* p_km_k,
               const double* H, const double* z, const double* h,
               double* x_out, double* P_out)
{
    if (r == 0) {                                                 /* :635-638 */
        memcpy(x_out, x_km_k, sizeof(double) * n);
        memcpy(P_out, p_km_k, sizeof(double) * (size_t)n * n);
        return RSLAM_OK;
    }
    const size_t nn = (size_t)n * n;
    double* HP   = (double*)malloc(sizeof(double) * (size_t)r * n);
    double* S    = (double*)malloc(sizeof(double) * (size_t)r * r);
    double* Sinv = (double*)malloc(sizeof(double) * (size_t)r * r);
    double* PHt  = (double*)malloc(sizeof(double) * (size_t)n * r);
    double* K    = (double*)malloc(sizeof(double) * (size_t)n * r);
    double* KS   = (double*)malloc(sizeof(double) * (size_t)n * r);
    double* tmp  = (double*)malloc(sizeof(double) * nn);
    double* pkk  = (double*)malloc(sizeof(double) * nn);
    double* xkk  = (double*)malloc(sizeof(double) * n);
    /* :602  S = H * p * H' + R,  R = Identity(r, r) (:594, :676) */
    gemm_nn(r, n, n, H, r, p_km_k, n, HP, r);
    gemm_nt(r, r, n, HP, r, H, r, S, r);
    for (int i = 0; i < r; ++i) S[i + (size_t)i * r] += 1.0;
    /* :603  K = p * H' * S.inverse() */
    gemm_nt(n, r, n, p_km_k, n, H, r, PHt, n);
    int rc = orc_inverse_lu(r, S, Sinv);
    gemm_nn(n, r, r, PHt, n, Sinv, r, K, n);
    /* :606  xkk = x + K * (z - h) */
    for (int i = 0; i < n; ++i) {
        double s = 0;
        for (int k = 0; k < r; ++k) s += K[i + (size_t)k * n] * (z[k] - h[k]);
        xkk[i] = x_km_k[i] + s;
    }
    /* :608  pkk_temp = p - K * S * K' ;  :609 pkk = 0.5*pkk_temp + 0.5*pkk_temp' */
    gemm_nn(n, r, r, K, n, S, r, KS, n);
    gemm_nt(n, n, r, KS, n, K, n, tmp, n);
    for (size_t k = 0; k < nn; ++k) tmp[k] = p_km_k[k] - tmp[k];
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i)
            pkk[i + (size_t)j * n] = 0.5 * tmp[i + (size_t)j * n] + 0.5 * tmp[j + (size_t)i * n];
    /* :613-620 normalise the quaternion (Jacobian uses the pre-normalisation q) */
    const double qr = xkk[3], qx = xkk[4], qy = xkk[5], qz = xkk[6];
    const double nrm = sqrt(qr * qr + qx * qx + qy * qy + qz * qz);
    for (int k = 3; k < 7; ++k) xkk[k] = xkk[k] / nrm;
    memcpy(x_out, xkk, sizeof(double) * n);
    /* :622-627  Q6: pow(., (-3/2)) with integer division => exponent -1 */
    double T[16];   /* col-major 4x4 of the row-major "temp" */
    const double rows[16] = {
        qx*qx+qy*qy+qz*qz, -qr*qx,            -qr*qy,            -qr*qz,
        -qx*qr,            qr*qr+qy*qy+qz*qz, -qx*qy,            -qx*qz,
        -qy*qr,            -qy*qx,            qr*qr+qx*qx+qz*qz, -qy*qz,
        -qz*qr,            -qz*qx,            -qz*qy,            qr*qr+qx*qx+qy*qy };
    const double q2 = qr*qr + qx*qx + qy*qy + qz*qz;
    const double scale = compat ? pow(q2, (double)(-3 / 2)) : pow(q2, -1.5);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) T[i + 4 * j] = scale * rows[4 * i + j];
    /* :629-634 block congruence on rows/cols 3..6 */
    memcpy(P_out, pkk, sizeof(double) * nn);
    /* rows 3..6, all columns: Jnorm * pkk(3:7, :) */
    double* rowblk = (double*)malloc(sizeof(double) * 4 * (size_t)n);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < 4; ++i) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += T[i + 4 * k] * pkk[(3 + k) + (size_t)j * n];
            rowblk[i + 4 * (size_t)j] = s;
        }
    /* columns 3..6, all rows: pkk(:, 3:7) * Jnorm' ; centre block (J*P44)*J' */
    for (int i = 0; i < n; ++i) {
        if (i >= 3 && i < 7) continue;
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += pkk[i + (size_t)(3 + k) * n] * T[j + 4 * k];
            P_out[i + (size_t)(3 + j) * n] = s;
        }
    }
    for (int j = 0; j < n; ++j) {
        if (j >= 3 && j < 7) continue;
        for (int i = 0; i < 4; ++i) P_out[(3 + i) + (size_t)j * n] = rowblk[i + 4 * (size_t)j];
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += rowblk[i + 4 * (size_t)(3 + k)] * T[j + 4 * k];
            P_out[(3 + i) + (size_t)(3 + j) * n] = s;
        }
    free(rowblk);
    free(HP); free(S); free(Sinv); free(PHt); free(K); free(KS); free(tmp); free(pkk); free(xkk);
    return rc;
}

/* ExtendKF::ekf_update_li_inliers (:559-596) / ekf_update_hi_inliers (:640-678):
 * stack z, h, H of the flagged features in feature order, R = I, update(). */
static int ekf_update_flagged(orc_ctx* c, const uint8_t* flag, const double* x_in, const double* P_in,
                              double* x_out, double* P_out)
{
    const int n = c->n;
    int k = 0;
    for (int i = 0; i < c->L; ++i) if (flag[i]) ++k;
    const int r = 2 * k;
    double* z = (double*)malloc(sizeof(double) * (size_t)(r + 1));
    double* h = (double*)malloc(sizeof(double) * (size_t)(r + 1));
    double* H = (double*)calloc((size_t)(r + 1) * n, sizeof(double));
    int row = 0;
    for (int i = 0; i < c->L; ++i) {
        if (!flag[i]) continue;
        z[row] = c->z[2 * i]; z[row + 1] = c->z[2 * i + 1];
        h[row] = c->h[2 * i]; h[row + 1] = c->h[2 * i + 1];
        const double* Hi = c->H + (size_t)i * 2 * n;
        for (int j = 0; j < n; ++j) {
            H[row + (size_t)j * r] = Hi[0 + 2 * j];
            H[row + 1 + (size_t)j * r] = Hi[1 + 2 * j];
        }
        row += 2;
    }
    int rc = orc_update(c->cfg.compat, n, r, x_in, P_in, H, z, h, x_out, P_out);
    free(z); free(h); free(H);
    return rc;
}

/* Tracking::rescue_hi_inliers, Tracking.cpp:574-597 */
static void rescue_hi_inliers(orc_ctx* c)
{
    const int n = c->n;
    const double chi2inv_2_95 = c->cfg.chi2_gate;
    predict_camera_measurements(c, c->x_k_k, NULL);
    calculate_derivatives(c, c->x_k_k);
    double* scratch = (double*)malloc(sizeof(double) * 2 * n);
    c->rescue_margin = DBL_MAX;
    for (int i = 0; i < c->L; ++i) {
        if (c->ic[i] && !c->li[i]) {
            double Si[4], Sinv[4];
            /* Q7 (:589): no "+ R" in the reference */
            HPHt_2x2(c, i, c->p_k_k, c->cfg.compat ? 0.0 : 1.0, Si, scratch);
            const double nu0 = c->z[2 * i] - c->h[2 * i], nu1 = c->z[2 * i + 1] - c->h[2 * i + 1];
            orc_inverse_lu(2, Si, Sinv);
            /* (nui' * Si^-1) * nui */
            const double t0 = nu0 * Sinv[0] + nu1 * Sinv[1];
            const double t1 = nu0 * Sinv[2] + nu1 * Sinv[3];
            const double d2 = t0 * nu0 + t1 * nu1;
            const double mg = fabs(d2 - chi2inv_2_95);
            if (mg < c->rescue_margin) c->rescue_margin = mg;
            c->hi[i] = (d2 < chi2inv_2_95) ? 1 : 0;
        }
    }
    free(scratch);
}

static int load_measurements(orc_ctx* c, const double* z, const uint8_t* ic)
{
    for (int i = 0; i < c->L; ++i) {
        c->ic[i] = ic[i] ? 1 : 0;
        c->li[i] = 0; c->hi[i] = 0;
        if (c->ic[i]) {
            if (!c->has_h[i]) return RSLAM_ERR_IC_NOT_VISIBLE;   /* matching() only runs where h exists, Tracking.cpp:293 */
            c->z[2 * i] = z[2 * i]; c->z[2 * i + 1] = z[2 * i + 1];
        }
    }
    return RSLAM_OK;
}

int orc_ransac_only(orc_ctx* c, const double* z, const uint8_t* ic, const double* draws,
                    int32_t n_draws, int32_t max_iters, uint8_t* li,
                    int32_t* best_hyp, int32_t* best_support, int32_t* hyps_evaluated)
{
    if (!c || !z || !ic || !draws || n_draws < 0) return RSLAM_ERR_ARG;
    if (!c->predicted) return RSLAM_ERR_STATE;
    int rc = load_measurements(c, z, ic);
    if (rc) return rc;
    int bh, bs, he;
    rc = ransac_hypotheses(c, draws, n_draws, max_iters, &bh, &bs, &he);
    if (li) memcpy(li, c->li, (size_t)c->L);
    if (best_hyp) *best_hyp = bh;
    if (best_support) *best_support = bs;
    if (hyps_evaluated) *hyps_evaluated = he;
    return rc;
}

/* System::TrackRunning lines 123-129 from the flags left by orc_ransac_only
 * (lets the CPU-baseline leg time the RANSAC sample and the updates separately). */
int orc_finish_update(orc_ctx* c, double* x_new, double* P_new, uint8_t* li, uint8_t* hi)
{
    if (!c) return RSLAM_ERR_ARG;
    if (!c->predicted) return RSLAM_ERR_STATE;
    const int n = c->n;
    int rc = ekf_update_flagged(c, c->li, c->x_k_km1, c->p_k_km1, c->x_k_k, c->p_k_k);
    if (rc) return rc;
    memcpy(c->x_li, c->x_k_k, sizeof(double) * n);
    memcpy(c->p_li, c->p_k_k, sizeof(double) * (size_t)n * n);
    rescue_hi_inliers(c);
    double* x2 = (double*)malloc(sizeof(double) * n);
    double* P2 = (double*)malloc(sizeof(double) * (size_t)n * n);
    rc = ekf_update_flagged(c, c->hi, c->x_k_k, c->p_k_k, x2, P2);
    memcpy(c->x_k_k, x2, sizeof(double) * n);
    memcpy(c->p_k_k, P2, sizeof(double) * (size_t)n * n);
    free(x2); free(P2);
    if (x_new) memcpy(x_new, c->x_k_k, sizeof(double) * n);
    if (P_new) memcpy(P_new, c->p_k_k, sizeof(double) * (size_t)n * n);
    if (li) memcpy(li, c->li, (size_t)c->L);
    if (hi) memcpy(hi, c->hi, (size_t)c->L);
    return rc;
}

/* System::TrackRunning lines 120-129 */
int orc_ransac_update(orc_ctx* c, const double* z, const uint8_t* ic, const double* draws,
                      int32_t n_draws, double* x_new, double* P_new, uint8_t* li, uint8_t* hi,
                      int32_t* best_hyp, int32_t* best_support, int32_t* hyps_evaluated)
{
    if (!c || !z || !ic || !draws || n_draws < 0) return RSLAM_ERR_ARG;
    if (!c->predicted) return RSLAM_ERR_STATE;
    const int n = c->n;
    int rc = load_measurements(c, z, ic);
    if (rc) return rc;
    int bh, bs, he;
    rc = ransac_hypotheses(c, draws, n_draws, 0, &bh, &bs, &he);            /* :120 */
    if (rc) return rc;
    rc = ekf_update_flagged(c, c->li, c->x_k_km1, c->p_k_km1, c->x_k_k, c->p_k_k);   /* :123 */
    if (rc) return rc;
    memcpy(c->x_li, c->x_k_k, sizeof(double) * n);
    memcpy(c->p_li, c->p_k_k, sizeof(double) * (size_t)n * n);
    rescue_hi_inliers(c);                                                   /* :126 */
    {                                                                       /* :129 */
        double* x2 = (double*)malloc(sizeof(double) * n);
        double* P2 = (double*)malloc(sizeof(double) * (size_t)n * n);
        rc = ekf_update_flagged(c, c->hi, c->x_k_k, c->p_k_k, x2, P2);
        memcpy(c->x_k_k, x2, sizeof(double) * n);
        memcpy(c->p_k_k, P2, sizeof(double) * (size_t)n * n);
        free(x2); free(P2);
    }
    if (x_new) memcpy(x_new, c->x_k_k, sizeof(double) * n);
    if (P_new) memcpy(P_new, c->p_k_k, sizeof(double) * (size_t)n * n);
    if (li) memcpy(li, c->li, (size_t)c->L);
    if (hi) memcpy(hi, c->hi, (size_t)c->L);
    if (best_hyp) *best_hyp = bh;
    if (best_support) *best_support = bs;
    if (hyps_evaluated) *hyps_evaluated = he;
    return rc;
}

/* ------------------------------------------------------------------ */
/* EKF prediction (the step just before the hot path)  */
/* ------------------------------------------------------------------ */

/* ExtendKF::v2q, ExtendKF.cpp:428-443 (eps = DBL_EPSILON, :24) */
static void v2q(const double v[3], double q[4])
{
    const double theta = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (theta < DBL_EPSILON) { q[0] = q[1] = q[2] = q[3] = 0; return; }
    double vn[3] = { v[0] / theta, v[1] / theta, v[2] / theta };
    const double nn = sqrt(vn[0] * vn[0] + vn[1] * vn[1] + vn[2] * vn[2]);
    q[0] = cos(theta / 2.0);
    for (int a = 0; a < 3; ++a) q[1 + a] = sin(theta / 2.0) * (vn[a] / nn);
}

/* ExtendKF::qprod, ExtendKF.cpp:416-427 */
static void qprod(const double q[4], const double wW[3], double delta_t, double qp[4])
{
    double v[3] = { wW[0] * delta_t, wW[1] * delta_t, wW[2] * delta_t }, p[4];
    v2q(v, p);
    const double* qv = q + 1; const double* pu = p + 1;
    const double cx = qv[1] * pu[2] - qv[2] * pu[1];
    const double cy = qv[2] * pu[0] - qv[0] * pu[2];
    const double cz = qv[0] * pu[1] - qv[1] * pu[0];
    qp[0] = q[0] * p[0] - (qv[0] * pu[0] + qv[1] * pu[1] + qv[2] * pu[2]);
    qp[1] = (q[0] * pu[0] + p[0] * qv[0]) + cx;
    qp[2] = (q[0] * pu[1] + p[0] * qv[1]) + cy;
    qp[3] = (q[0] * pu[2] + p[0] * qv[2]) + cz;
}

/* ExtendKF.cpp:513-529 */
static double dq0_by_domegaA(double omegaA, double omega, double delta_t)
{ return (-delta_t / 2.0) * (omegaA / omega) * sin(omega * delta_t / 2.0); }
static double dqA_by_domegaA(double omegaA, double omega, double delta_t)
{
    return (delta_t / 2.0) * omegaA * omegaA / (omega * omega) * cos(omega * delta_t / 2.0)
         + (1.0 / omega) * (1.0 - omegaA * omegaA / (omega * omega)) * sin(omega * delta_t / 2.0);
}
static double dqA_by_domegaB(double omegaA, double omegaB, double omega, double delta_t)
{
    return (omegaA * omegaB / (omega * omega)) *
           ((delta_t / 2.0) * cos(omega * delta_t / 2.0) - (1.0 / omega) * sin(omega * delta_t / 2.0));
}
/* ExtendKF::dqomegadt_by_domega, ExtendKF.cpp:491-512; out 4x3 col-major */
static void dqomegadt_by_domega(const double w[3], double dt, double out[12])
{
    const double m = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
#define O(i, j) out[(i) + 4 * (j)]
    O(0,0) = dq0_by_domegaA(w[0], m, dt); O(0,1) = dq0_by_domegaA(w[1], m, dt); O(0,2) = dq0_by_domegaA(w[2], m, dt);
    O(1,0) = dqA_by_domegaA(w[0], m, dt); O(1,1) = dqA_by_domegaB(w[0], w[1], m, dt); O(1,2) = dqA_by_domegaB(w[0], w[2], m, dt);
    O(2,0) = dqA_by_domegaB(w[1], w[0], m, dt); O(2,1) = dqA_by_domegaA(w[1], m, dt); O(2,2) = dqA_by_domegaB(w[1], w[2], m, dt);
    O(3,0) = dqA_by_domegaB(w[2], w[0], m, dt); O(3,1) = dqA_by_domegaB(w[2], w[1], m, dt); O(3,2) = dqA_by_domegaA(w[2], m, dt);
#undef O
}
/* ExtendKF::dq3_by_dq1, ExtendKF.cpp:482-490; out 4x4 col-major */
static void dq3_by_dq1(const double q[4], double out[16])
{
    const double r[16] = { q[0], -q[1], -q[2], -q[3],
                           q[1],  q[0], -q[3],  q[2],
                           q[2],  q[3],  q[0], -q[1],
                           q[3], -q[2],  q[1],  q[0] };
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out[i + 4 * j] = r[4 * i + j];
}

/* The 13-state motion model of ExtendKF::ekf_prediction for filter_type
 * "constant_velocity" (the only one the reference instantiates, System.cpp:63):
 * fv (ExtendKF.cpp:389-400), dfv_by_dxv (:444-465), Q = G Pn G' (:347-376).
 * F, Q are 13x13 col-major. */
void orc_motion_model(const double xv[13], double delta_t, double std_a, double std_alpha,
                      double xv_pred[13], double F[169], double Q[169])
{
    const double* rW = xv; const double* qWR = xv + 3; const double* vW = xv + 7; const double* wW = xv + 10;
    for (int a = 0; a < 3; ++a) xv_pred[a] = rW[a] + vW[a] * delta_t;
    qprod(qWR, wW, delta_t, xv_pred + 3);
    for (int a = 0; a < 3; ++a) { xv_pred[7 + a] = vW[a]; xv_pred[10 + a] = wW[a]; }

    memset(F, 0, sizeof(double) * 169);
    for (int i = 0; i < 13; ++i) F[i + 13 * i] = 1.0;
    double wt[3] = { wW[0] * delta_t, wW[1] * delta_t, wW[2] * delta_t }, qwt[4];
    v2q(wt, qwt);
    const double blk[16] = { qwt[0], -qwt[1], -qwt[2], -qwt[3],
                             qwt[1],  qwt[0],  qwt[3], -qwt[2],
                             qwt[2], -qwt[3],  qwt[0],  qwt[1],
                             qwt[3],  qwt[2], -qwt[1],  qwt[0] };
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) F[(3 + i) + 13 * (3 + j)] = blk[4 * i + j];
    for (int a = 0; a < 3; ++a) F[a + 13 * (7 + a)] = delta_t;
    double a44[16], b43[12], ab[12];
    dq3_by_dq1(qWR, a44);
    dqomegadt_by_domega(wW, delta_t, b43);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 3; ++j) {
            double sacc = 0;
            for (int k = 0; k < 4; ++k) sacc += a44[i + 4 * k] * b43[k + 4 * j];
            ab[i + 4 * j] = sacc;
            F[(3 + i) + 13 * (10 + j)] = sacc;
        }
    /* Q = G * Pn * G' */
    const double la = pow(std_a * delta_t, 2), aa = pow(std_alpha * delta_t, 2);
    const double Pn[6] = { la, la, la, aa, aa, aa };
    double G[13 * 6];
    memset(G, 0, sizeof(G));
    for (int a = 0; a < 3; ++a) { G[(7 + a) + 13 * a] = 1.0; G[(10 + a) + 13 * (3 + a)] = 1.0; G[a + 13 * a] = delta_t; }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 3; ++j) G[(3 + i) + 13 * (3 + j)] = ab[i + 4 * j];
    for (int i = 0; i < 13; ++i)
        for (int j = 0; j < 13; ++j) {
            double sacc = 0;
            for (int k = 0; k < 6; ++k) sacc += (G[i + 13 * k] * Pn[k]) * G[j + 13 * k];
            Q[i + 13 * j] = sacc;
        }
}

/* ExtendKF::ekf_prediction, ExtendKF.cpp:333-388.  x_kk (n), P_kk (n x n) -> x_pred, P_pred. */
int orc_ekf_prediction(int n, const double* x_kk, const double* P_kk, double delta_t, double std_a,
                       double std_alpha, double* x_pred, double* P_pred)
{
    if (n < 13 || !x_kk || !P_kk || !x_pred || !P_pred) return RSLAM_ERR_ARG;
    double F[169], Q[169], xv[13];
    orc_motion_model(x_kk, delta_t, std_a, std_alpha, xv, F, Q);
    memcpy(x_pred, x_kk, sizeof(double) * n);
    memcpy(x_pred, xv, sizeof(double) * 13);
    memcpy(P_pred, P_kk, sizeof(double) * (size_t)n * n);                 /* pk_km5 = bottom-right block */
    double FP[169];
    for (int i = 0; i < 13; ++i)                                            /* pk_km2 = F*P11*F' + Q */
        for (int j = 0; j < 13; ++j) {
            double sacc = 0;
            for (int k = 0; k < 13; ++k) sacc += F[i + 13 * k] * P_kk[k + (size_t)j * n];
            FP[i + 13 * j] = sacc;
        }
    for (int i = 0; i < 13; ++i)
        for (int j = 0; j < 13; ++j) {
            double sacc = 0;
            for (int k = 0; k < 13; ++k) sacc += FP[i + 13 * k] * F[j + 13 * k];
            P_pred[i + (size_t)j * n] = sacc + Q[i + 13 * j];
        }
    for (int j = 13; j < n; ++j)                                            /* pk_km3 = F * P12 */
        for (int i = 0; i < 13; ++i) {
            double sacc = 0;
            for (int k = 0; k < 13; ++k) sacc += F[i + 13 * k] * P_kk[k + (size_t)j * n];
            P_pred[i + (size_t)j * n] = sacc;
        }
    for (int i = 13; i < n; ++i)                                            /* pk_km4 = P21 * F' */
        for (int j = 0; j < 13; ++j) {
            double sacc = 0;
            for (int k = 0; k < 13; ++k) sacc += P_kk[i + (size_t)k * n] * F[j + 13 * k];
            P_pred[i + (size_t)j * n] = sacc;
        }
    return RSLAM_OK;
}

/* ------------------------------------------------------------------ */
/* Map state surgery: the x / P edits of Map.cpp       */
/* ------------------------------------------------------------------ */

static int feature_width(uint8_t t) { return t == RSLAM_FEAT_INVERSE_DEPTH ? 6 : 3; }
static int feature_offset(const uint8_t* type, int i)
{
    int o = 13;
    for (int k = 0; k < i; ++k) o += feature_width(type[k]);
    return o;
}

/* Map::delete_a_feature, Map.cpp:69-104 (feature 0-based here).  Outputs sized n - width. */
int orc_map_delete_feature(int n, int L, const uint8_t* type, const double* x, const double* P,
                           int feature, double* x_out, double* P_out)
{
    if (feature < 0 || feature >= L) return RSLAM_ERR_ARG;
    const int w = feature_width(type[feature]), o = feature_offset(type, feature), n2 = n - w;
    for (int i = 0, i2 = 0; i < n; ++i) {
        if (i >= o && i < o + w) continue;
        x_out[i2] = x[i];
        for (int j = 0, j2 = 0; j < n; ++j) {
            if (j >= o && j < o + w) continue;
            P_out[i2 + (size_t)j2 * n2] = P[i + (size_t)j * n];
            ++j2;
        }
        ++i2;
    }
    return RSLAM_OK;
}

/* ExtendKF::inversedepth2cartesian, ExtendKF.cpp:137-152 */
static void inversedepth2cartesian(const double y[6], double c[3])
{
    const double theta = y[3], phi = y[4], rho = y[5];
    const double m[3] = { cos(phi) * sin(theta), -sin(phi), cos(phi) * cos(theta) };
    for (int a = 0; a < 3; ++a) c[a] = y[a] + (1.0 / rho) * m[a];
}

/* linearity index of an inverse-depth feature, Map.cpp:124-149 */
double orc_linearity_index(const double* x, const double* P, int n, int o)
{
    const double std_rho = sqrt(P[(o + 5) + (size_t)(o + 5) * n]);
    const double rho = x[o + 5];
    const double std_d = std_rho / (rho * rho);
    double X_out[3];
    inversedepth2cartesian(x + o, X_out);
    double d1[3], d2[3];
    for (int a = 0; a < 3; ++a) { d1[a] = X_out[a] - x[o + a]; d2[a] = X_out[a] - x[a]; }
    const double d_c2p = sqrt(d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2]);
    const double aa = d1[0] * d2[0] + d1[1] * d2[1] + d1[2] * d2[2];
    const double bb = sqrt(d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2]) * d_c2p;
    const double cos_alpha = aa / bb;
    return 4 * std_d * cos_alpha / d_c2p;
}

/* Map::inversedepth_2_cartesian, Map.cpp:105-196: the first inverse-depth feature whose linearity
 * index is below the threshold is converted (one per call); *converted = its index or -1.
 * Outputs sized for n - 3 (untouched when nothing converts). */
int orc_map_convert(int n, int L, const uint8_t* type, const double* x, const double* P, double threshold,
                    double* x_out, double* P_out, int* converted)
{
    *converted = -1;
    for (int i = 0; i < L; ++i) {
        if (type[i] != RSLAM_FEAT_INVERSE_DEPTH) continue;
        const int o = feature_offset(type, i);
        if (!(orc_linearity_index(x, P, n, o) < threshold)) continue;
        const int n2 = n - 3;
        const double theta = x[o + 3], phi = x[o + 4], rho = x[o + 5];
        const double mi[3] = { cos(phi) * sin(theta), -sin(phi), cos(phi) * cos(theta) };
        const double dmt[3] = { cos(phi) * cos(theta), 0, -cos(phi) * sin(theta) };
        const double dmp[3] = { -sin(phi) * sin(theta), -cos(phi), -sin(phi) * cos(theta) };
        double J[18];   /* 3 x 6 col-major: [I3, dm_dtheta/rho, dm_dphi/rho, -mi/rho^2] */
        memset(J, 0, sizeof(J));
        for (int a = 0; a < 3; ++a) {
            J[a + 3 * a] = 1.0;
            J[a + 3 * 3] = (1 / rho) * dmt[a];
            J[a + 3 * 4] = (1 / rho) * dmp[a];
            J[a + 3 * 5] = -mi[a] / (rho * rho);
        }
        double X_out[3];
        inversedepth2cartesian(x + o, X_out);
        for (int k = 0; k < o; ++k) x_out[k] = x[k];
        for (int a = 0; a < 3; ++a) x_out[o + a] = X_out[a];
        for (int k = o + 6; k < n; ++k) x_out[k - 3] = x[k];
        /* P_expansion = J_all * P * J_all' : tmp = J_all * P (n2 x n), then * J_all' */
        double* tmp = (double*)malloc(sizeof(double) * (size_t)n2 * n);
        for (int j = 0; j < n; ++j)
            for (int i2 = 0; i2 < n2; ++i2) {
                double v;
                if (i2 < o) v = P[i2 + (size_t)j * n];
                else if (i2 < o + 3) {
                    v = 0;
                    for (int k = 0; k < 6; ++k) v += J[(i2 - o) + 3 * k] * P[(o + k) + (size_t)j * n];
                } else v = P[(i2 + 3) + (size_t)j * n];
                tmp[i2 + (size_t)j * n2] = v;
            }
        for (int j2 = 0; j2 < n2; ++j2)
            for (int i2 = 0; i2 < n2; ++i2) {
                double v;
                if (j2 < o) v = tmp[i2 + (size_t)j2 * n2];
                else if (j2 < o + 3) {
                    v = 0;
                    for (int k = 0; k < 6; ++k) v += tmp[i2 + (size_t)(o + k) * n2] * J[(j2 - o) + 3 * k];
                } else v = tmp[i2 + (size_t)(j2 + 3) * n2];
                P_out[i2 + (size_t)j2 * n2] = v;
            }
        free(tmp);
        *converted = i;
        return RSLAM_OK;      /* "only convert one feature per step", Map.cpp:192 */
    }
    return RSLAM_OK;
}

/* ExtendKF::hinv, ExtendKF.cpp:236-265 (cam->K(0,0) = K(1,1) = f/d with d = dx = dy, System.cpp:57) */
void orc_hinv(const rslam_camera* cam, const double uvd[2], const double Xv[13], double initial_rho, double y[6])
{
    const double fku = cam->f / cam->dx, fkv = cam->f / cam->dy, U0 = cam->Cx, V0 = cam->Cy;
    double uv[2];
    orc_undistort_fm(cam, uvd, uv);
    const double hx = -(U0 - uv[0]) / fku, hy = -(V0 - uv[1]) / fkv, hz = 1;
    double R[9];
    orc_q2r(Xv + 3, R);
    const double nx = R[0] * hx + R[3] * hy + R[6] * hz;
    const double ny = R[1] * hx + R[4] * hy + R[7] * hz;
    const double nz = R[2] * hx + R[5] * hy + R[8] * hz;
    y[0] = Xv[0]; y[1] = Xv[1]; y[2] = Xv[2];
    y[3] = atan2(nx, nz);
    y[4] = atan2(-ny, sqrt(nx * nx + nz * nz));
    y[5] = initial_rho;
}

/* dy_dxv (6 x 13) and dy_dhd * Padd * dy_dhd' (6 x 6) of Map::add_a_feature_covariance_inverse_depth,
 * Map.cpp:339-388, both col-major; std_rho = 1 in the reference (Map.cpp:218,384). */
void orc_add_feature_jacobians(const rslam_camera* cam, double std_z, double std_rho, const double uvd[2],
                               const double Xv[13], double D[78], double Rn[36])
{
    const double fku = cam->f / cam->dx, fkv = cam->f / cam->dy, U0 = cam->Cx, V0 = cam->Cy;
    const double* q = Xv + 3;
    double R[9], uvu[2];
    orc_q2r(q, R);
    orc_undistort_fm(cam, uvd, uvu);
    const double c[3] = { -(U0 - uvu[0]) / fku, -(V0 - uvu[1]) / fkv, 1 };
    const double Xw = R[0] * c[0] + R[3] * c[1] + R[6] * c[2];
    const double Yw = R[1] * c[0] + R[4] * c[1] + R[7] * c[2];
    const double Zw = R[2] * c[0] + R[5] * c[1] + R[8] * c[2];
    double dgw_dq[12];
    orc_dRq_times_a_by_dq(q, c, dgw_dq);
    const double xz = Xw * Xw + Zw * Zw, xyz = Xw * Xw + Yw * Yw + Zw * Zw;
    const double dth[3] = { Zw / xz, 0, -Xw / xz };
    const double dph[3] = { (Xw * Yw) / (xyz * sqrt(xz)), -sqrt(xz) / xyz, (Zw * Yw) / (xyz * sqrt(xz)) };
    memset(D, 0, sizeof(double) * 78);
    for (int a = 0; a < 3; ++a) D[a + 6 * a] = 1.0;                 /* dy_drw = [I3; 0] */
    for (int k = 0; k < 4; ++k) {                                    /* dy_dqwr rows 3, 4 */
        double s3 = 0, s4 = 0;
        for (int a = 0; a < 3; ++a) { s3 += dth[a] * dgw_dq[a + 3 * k]; s4 += dph[a] * dgw_dq[a + 3 * k]; }
        D[3 + 6 * (3 + k)] = s3; D[4 + 6 * (3 + k)] = s4;
    }
    /* dyprima_dhd = dyprima_dgw * R_wc * dgc_dhu * dhu_dhd  (5 x 2), rows 3,4 non-zero */
    double Jd[4];
    orc_jacob_undistor_fm(cam, uvd, Jd);
    double A32[6];   /* R_wc * dgc_dhu : 3 x 2 col-major */
    for (int a = 0; a < 3; ++a) { A32[a] = R[a] * (1 / fku); A32[a + 3] = R[a + 3] * (1 / fkv); }
    double B32[6];   /* (R dgc_dhu) * dhu_dhd */
    for (int a = 0; a < 3; ++a)
        for (int j = 0; j < 2; ++j) B32[a + 3 * j] = A32[a] * Jd[0 + 2 * j] + A32[a + 3] * Jd[1 + 2 * j];
    double E[18];    /* dy_dhd : 6 x 3 col-major */
    memset(E, 0, sizeof(E));
    for (int j = 0; j < 2; ++j) {
        double s3 = 0, s4 = 0;
        for (int a = 0; a < 3; ++a) { s3 += dth[a] * B32[a + 3 * j]; s4 += dph[a] * B32[a + 3 * j]; }
        E[3 + 6 * j] = s3; E[4 + 6 * j] = s4;
    }
    E[5 + 6 * 2] = 1.0;
    const double padd[3] = { pow(std_z, 2), pow(std_z, 2), pow(std_rho, 2) };
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double sacc = 0;
            for (int k = 0; k < 3; ++k) sacc += (E[i + 6 * k] * padd[k]) * E[j + 6 * k];
            Rn[i + 6 * j] = sacc;
        }
}

/* Map::initialize_a_features lines 281-292 with Map::add_a_feature_covariance_inverse_depth
 * (Map.cpp:339-400): append one inverse-depth feature observed at the distorted pixel uvd.
 * Outputs sized n + 6. */
int orc_map_add_feature(const rslam_camera* cam, double std_z, int n, const double* x, const double* P,
                        const double uvd[2], double initial_rho, double std_rho, double* x_out, double* P_out)
{
    const int n2 = n + 6;
    double y[6], D[78], Rn[36];
    orc_hinv(cam, uvd, x, initial_rho, y);
    orc_add_feature_jacobians(cam, std_z, std_rho, uvd, x, D, Rn);
    memcpy(x_out, x, sizeof(double) * n);
    memcpy(x_out + n, y, sizeof(double) * 6);
    for (int j = 0; j < n; ++j) {
        for (int i = 0; i < n; ++i) P_out[i + (size_t)j * n2] = P[i + (size_t)j * n];
        for (int a = 0; a < 6; ++a) {                  /* dy_dxv * P(0:13, :) */
            double sacc = 0;
            for (int k = 0; k < 13; ++k) sacc += D[a + 6 * k] * P[k + (size_t)j * n];
            P_out[(n + a) + (size_t)j * n2] = sacc;
        }
    }
    for (int b = 0; b < 6; ++b) {
        for (int i = 0; i < n; ++i) {                  /* P(:, 0:13) * dy_dxv' */
            double sacc = 0;
            for (int k = 0; k < 13; ++k) sacc += P[i + (size_t)k * n] * D[b + 6 * k];
            P_out[i + (size_t)(n + b) * n2] = sacc;
        }
        for (int a = 0; a < 6; ++a) {                  /* (dy_dxv * P_xv) * dy_dxv' + dy_dhd Padd dy_dhd' */
            double sacc = 0;
            for (int k = 0; k < 13; ++k) {
                double dp = 0;
                for (int mm = 0; mm < 13; ++mm) dp += D[a + 6 * mm] * P[mm + (size_t)k * n];
                sacc += dp * D[b + 6 * k];
            }
            P_out[(n + a) + (size_t)(n + b) * n2] = sacc + Rn[a + 6 * b];
        }
    }
    return RSLAM_OK;
}

/* ------------------------------------------------------------------ */
/* NCC search: Tracking::matching, Tracking.cpp:279-351  */
/* with Converter::corrcoef_opencv, Converter.cpp:188-209                  */
/* ------------------------------------------------------------------ */

/* One feature.  image: nRows x nCols uint8, row-major (cv::Mat);  patch: 13 x 13 predicted patch,
 * column-major as Eigen stores patch_when_matching;  h, S: prediction and innovation covariance
 * (col-major 2 x 2).  Returns 1 when a match above the correlation threshold exists and writes
 * z = (column j, row i).  best_corr receives the maximum correlation over the candidates (or -2
 * when there is no candidate), n_cand their number.  margins[0..2] (nullable): smallest distance of
 * the largest eigenvalue of S to 100, of a candidate's Mahalanobis distance to the gate and of the
 * best correlation to the threshold / to the runner-up -- the decisions that must not be close calls
 * for integer outputs to be comparable.
 *
 * corrcoef_opencv: the patches go through toCvMat_f (float32), cv::calcCovarMatrix(..., CV_COVAR_NORMAL |
 * CV_COVAR_ROWS) accumulates in double; the common 1/(cols-1) factor cancels in the normalisation
 * (Converter.cpp:196,204-206).  OpenCV is not in this image: the summation order of its covariance is
 * unpinned; this restatement uses the two-pass definition in double.  Only row 0 of the matrix is read
 * (Tracking.cpp:340), so only that row is formed. */
int orc_match_feature(const rslam_camera* cam, const uint8_t* image, const double* patch, int half,
                      const double h[2], const double S[4], double corr_threshold, double chi2,
                      double z[2], double* best_corr, int* n_cand, double margins[3])
{
    const int side = 2 * half + 1, npix = side * side;
    const int nCols = cam->nCols, nRows = cam->nRows;
    *best_corr = -2.0; *n_cand = 0;
    if (margins) { margins[0] = margins[1] = margins[2] = 1e300; }
    /* SelfAdjointEigenSolver reads the lower triangle: closed form for 2 x 2 */
    const double a = S[0], b = S[1], d = S[3];
    const double lmax = 0.5 * (a + d) + sqrt(0.25 * (a - d) * (a - d) + b * b);
    if (margins) margins[0] = fabs(lmax - 100.0);
    if (!(lmax < 100)) return 0;                                                    /* Tracking.cpp:303 */
    const int hsx = (int)ceil(2 * sqrt(S[0])), hsy = (int)ceil(2 * sqrt(S[3]));   /* :306-307 */
    double Sinv[4];
    orc_inverse_lu(2, S, Sinv);
    /* predicted patch as float32, its mean and centred energy */
    double* p = (double*)malloc(sizeof(double) * (size_t)npix);
    double pm = 0;
    for (int k = 0; k < npix; ++k) { p[k] = (double)(float)patch[k]; pm += p[k]; }
    pm /= npix;
    double pe = 0;
    for (int k = 0; k < npix; ++k) pe += (p[k] - pm) * (p[k] - pm);
    const int x0 = (int)round(h[0]), y0 = (int)round(h[1]);
    double best = -2.0, second = -2.0; int bj = 0, bi = 0, count = 0;
    for (int j = x0 - hsx; j <= x0 + hsx; ++j)
        for (int i = y0 - hsy; i <= y0 + hsy; ++i) {
            const double n0 = j - h[0], n1 = i - h[1];
            /* (nu' * S^-1) * nu, left to right */
            const double t0 = n0 * Sinv[0] + n1 * Sinv[1], t1 = n0 * Sinv[2] + n1 * Sinv[3];
            const double d2 = t0 * n0 + t1 * n1;
            if (margins && fabs(d2 - chi2) < margins[1]) margins[1] = fabs(d2 - chi2);
            if (!(d2 < chi2)) continue;
            if (!((j > half) && (j < nCols - half) && (i > half) && (i < nRows - half))) continue;
            /* candidate patch: element (r, c) = image(i - half + r, j - half + c), reshaped column-major */
            double cm = 0;
            for (int c = 0; c < side; ++c)
                for (int r = 0; r < side; ++r) cm += image[(size_t)(i - half + r) * nCols + (j - half + c)];
            cm /= npix;
            double ce = 0, pc = 0;
            for (int c = 0; c < side; ++c)
                for (int r = 0; r < side; ++r) {
                    const double v = image[(size_t)(i - half + r) * nCols + (j - half + c)] - cm;
                    ce += v * v;
                    pc += (p[r + side * c] - pm) * v;
                }
            const double corr = pc / sqrt(pe * ce);
            ++count;
            if (corr > best) { second = best; best = corr; bj = j; bi = i; }     /* maxCoeff: first maximum */
            else if (corr > second) second = corr;
        }
    free(p);
    *n_cand = count;
    if (count == 0) return 0;               /* (the reference takes maxCoeff of an empty vector here) */
    *best_corr = best;
    if (margins) {
        margins[2] = fabs(best - corr_threshold);
        if (count > 1 && best - second < margins[2] && best > corr_threshold) margins[2] = best - second;
    }
    if (best > corr_threshold) { z[0] = bj; z[1] = bi; return 1; }
    return 0;
}

/* Tracking::matching over all features; has_h (L) = prediction exists.  Outputs z (L*2, written where
 * ic), ic (L), corr (L, -2 where no candidate / not searched), margins[3] minima over the features. */
void orc_matching(const rslam_camera* cam, const uint8_t* image, int L, const double* patches, int half,
                  const double* h, const uint8_t* has_h, const double* S, double* z, uint8_t* ic, double* corr,
                  double margins[3])
{
    const int npix = (2 * half + 1) * (2 * half + 1);
    if (margins) margins[0] = margins[1] = margins[2] = 1e300;
    for (int f = 0; f < L; ++f) {
        ic[f] = 0; corr[f] = -2.0;
        if (!has_h[f]) continue;
        double m[3]; int nc;
        ic[f] = (uint8_t)orc_match_feature(cam, image, patches + (size_t)f * npix, half, h + 2 * f, S + 4 * f, 0.80, 5.9915,
                                           z + 2 * f, corr + f, &nc, m);
        if (margins) for (int k = 0; k < 3; ++k) if (m[k] < margins[k]) margins[k] = m[k];
    }
}

/* ------------------------------------------------------------------ */
/* Patch prediction: Tracking::pred_patch_fc,            */
/* Tracking.cpp:164-278, called from search_IC_matches :46-65             */
/* ------------------------------------------------------------------ */

/* fixed-size Matrix4d::inverse(): cofactors over 2 x 2 sub-determinants (col-major in/out) */
static void inv4_fixed(const double m[16], double r[16])
{
#define A(i, j) m[(i) + 4 * (j)]
    const double s0 = A(0,0) * A(1,1) - A(1,0) * A(0,1), s1 = A(0,0) * A(1,2) - A(1,0) * A(0,2);
    const double s2 = A(0,0) * A(1,3) - A(1,0) * A(0,3), s3 = A(0,1) * A(1,2) - A(1,1) * A(0,2);
    const double s4 = A(0,1) * A(1,3) - A(1,1) * A(0,3), s5 = A(0,2) * A(1,3) - A(1,2) * A(0,3);
    const double c5 = A(2,2) * A(3,3) - A(3,2) * A(2,3), c4 = A(2,1) * A(3,3) - A(3,1) * A(2,3);
    const double c3 = A(2,1) * A(3,2) - A(3,1) * A(2,2), c2 = A(2,0) * A(3,3) - A(3,0) * A(2,3);
    const double c1 = A(2,0) * A(3,2) - A(3,0) * A(2,2), c0 = A(2,0) * A(3,1) - A(3,0) * A(2,1);
    const double id = 1.0 / (s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0);
#define R_(i, j) r[(i) + 4 * (j)]
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

static void mul3(const double a[9], const double b[9], double c[9])     /* col-major 3 x 3 */
{
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) c[i + 3 * j] = a[i] * b[3 * j] + a[i + 3] * b[1 + 3 * j] + a[i + 6] * b[2 + 3 * j];
}

/* "H = [R 0; 0 1] * [I r; 0 1]" of Tracking.cpp:189-194 = [R, R r; 0 1], col-major 4 x 4 */
static void pose_matrix(const double R[9], const double r[3], double H[16])
{
    memset(H, 0, sizeof(double) * 16);
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) H[i + 4 * j] = R[i + 3 * j];
    for (int i = 0; i < 3; ++i) H[i + 12] = R[i] * r[0] + R[i + 3] * r[1] + R[i + 6] * r[2];
    H[15] = 1.0;
}

/* cv::remap(src CV_32F, map1/map2 CV_32FC1, INTER_LINEAR, BORDER_CONSTANT 0) at one point, as OpenCV
 * does it: coordinates quantised to 1/32 pixel (cvRound(x * INTER_TAB_SIZE), round half to even), the
 * four weights from the float table (1 - a)(1 - b) ..., float accumulation in tap order.  src is the
 * sh x sw row-major float image. */
/* distance (pixels) of a double map coordinate from the point where its float32 cast would flip */
static double f32_cast_margin(double v)
{
    const float f = (float)v;
    const double err = fabs(v - (double)f);
    const double up = fabs((double)nextafterf(f, INFINITY) - (double)f), dn = fabs((double)f - (double)nextafterf(f, -INFINITY));
    const double half_ulp = 0.5 * (up < dn ? up : dn);
    return half_ulp - err;
}

static float remap_bilinear_f32(const float* src, int sh, int sw, float mapx, float mapy)
{
    /* after the float cast everything below is exact integer / float arithmetic, identical wherever it runs */
    const double qx = (double)mapx * 32.0, qy = (double)mapy * 32.0;
    const int sx = (int)nearbyint(qx), sy = (int)nearbyint(qy);
    const int ix = sx >> 5, iy = sy >> 5;           /* arithmetic shift = floor */
    const float fx = (float)(sx & 31) / 32.f, fy = (float)(sy & 31) / 32.f;
    const float w00 = (1.f - fy) * (1.f - fx), w01 = (1.f - fy) * fx, w10 = fy * (1.f - fx), w11 = fy * fx;
    float v[4];
    for (int k = 0; k < 4; ++k) {
        const int x = ix + (k & 1), y = iy + (k >> 1);
        v[k] = (x >= 0 && x < sw && y >= 0 && y < sh) ? src[(size_t)y * sw + x] : 0.f;
    }
    return v[0] * w00 + v[1] * w01 + v[2] * w10 + v[3] * w11;
}

/* One feature.  xv: camera state r(3) q(4) of x_k_km1; h: predicted pixel; uv_f, R_f (col-major), r_f,
 * patch_f (side_f x side_f col-major, side_f = 2 * half_f + 1): the feature's initialisation record;
 * XYZ_w: the world point handed in by search_IC_matches.  out: 13 x 13 col-major.
 * compat = 1 keeps the reference's one-pixel offset of the remap coordinates (MATLAB 1-based indices,
 * Tracking.cpp:263-264); compat = 0 removes it.  Returns 1 when a patch was warped, 0 when h is too close
 * to the border (zero patch, :174-175,276), -1 when the meshgrid would not be 13 x 13.  *margin (nullable) is
 * lowered to the smallest distance of a map coordinate from a flip of its float32 cast and of the patch centre
 * from an integer (the cv::Range truncation) -- the two steps at which two correct double-precision evaluations
 * of the geometry can end in different patches. */
int orc_pred_patch(const rslam_camera* cam, int compat, const double xv[7], const double h[2], const double uv_f[2],
                   const double R_f[9], const double r_f[3], const double* patch_f, int half_f, const double XYZ_w[3],
                   int half, double* out, double* margin)
{
    const int side = 2 * half + 1, side_f = 2 * half_f + 1;
    memset(out, 0, sizeof(double) * (size_t)side * side);
    if (!((h[0] > half) && (h[0] < cam->nCols - half) && (h[1] > half) && (h[1] < cam->nRows - half))) return 0;
    const double f = cam->f, dx = cam->dx, cx = cam->Cx, cy = cam->Cy;
    double R_wc[9];
    orc_q2r(xv + 3, R_wc);
    double H_f[16], H_k[16], H_f_inv[16], Hr[16];
    pose_matrix(R_f, r_f, H_f);
    pose_matrix(R_wc, xv, H_k);
    inv4_fixed(H_f, H_f_inv);
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) {
            double a = 0;
            for (int k = 0; k < 4; ++k) a += H_f_inv[i + 4 * k] * H_k[k + 4 * j];
            Hr[i + 4 * j] = a;                                            /* H_kpf_k */
        }
    /* plane normal, :196-209 */
    double n1[3] = { uv_f[0] - cx, uv_f[1] - cy, -f / dx };
    double nn = sqrt(n1[0] * n1[0] + n1[1] * n1[1] + n1[2] * n1[2]);
    for (int a = 0; a < 3; ++a) n1[a] = n1[a] / nn;
    const double n2in[4] = { h[0] - cx, h[1] - cy, -f / dx, 1.0 };
    double n2[4];
    for (int i = 0; i < 4; ++i) n2[i] = Hr[i] * n2in[0] + Hr[i + 4] * n2in[1] + Hr[i + 8] * n2in[2] + Hr[i + 12] * n2in[3];
    for (int i = 0; i < 4; ++i) n2[i] = n2[i] / n2[3];      /* n_temp / n_temp(3): element 3 divided last, it is 1 afterwards */
    nn = sqrt(n2[0] * n2[0] + n2[1] * n2[1] + n2[2] * n2[2]);
    double n[3];
    for (int a = 0; a < 3; ++a) n[a] = n1[a] + n2[a] / nn;
    nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    for (int a = 0; a < 3; ++a) n[a] = n[a] / nn;
    /* plane offset, :211-217 */
    double X[4];
    for (int i = 0; i < 4; ++i) X[i] = H_f_inv[i] * XYZ_w[0] + H_f_inv[i + 4] * XYZ_w[1] + H_f_inv[i + 8] * XYZ_w[2] + H_f_inv[i + 12];
    for (int i = 0; i < 4; ++i) X[i] = X[i] / X[3];
    const double d = -(n[0] * X[0] + n[1] * X[1] + n[2] * X[2]);
    /* homography M = K (R - t n'/d) K^-1 and its inverse, :226,247 */
    const double K[9] = { f / dx, 0, 0,  0, f / cam->dy, 0,  cx, cy, 1 };
    double Kinv[9], G[9], T1[9], M[9], Minv[9];
    inv3_fixed(K, Kinv);
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) G[i + 3 * j] = Hr[i + 4 * j] - Hr[i + 12] * n[j] / d;
    mul3(K, G, T1);
    mul3(T1, Kinv, M);
    inv3_fixed(M, Minv);
    /* centre of the predicted patch in the current image, :221-236 */
    double c1u[2], t3[3], c2u[2], c2[2];
    orc_undistort_fm(cam, uv_f, c1u);
    for (int i = 0; i < 3; ++i) t3[i] = Minv[i] * c1u[0] + Minv[i + 3] * c1u[1] + Minv[i + 6];
    c2u[0] = t3[0] / t3[2]; c2u[1] = t3[1] / t3[2];
    orc_distort_fm(cam, c2u, c2);
    /* cv::Range(double, double): truncation towards zero, :239-240 */
    const int xs = (int)(c2[0] - half), xe = (int)(c2[0] + half), ys = (int)(c2[1] - half), ye = (int)(c2[1] + half);
    if (xe - xs + 1 != side || ye - ys + 1 != side) return -1;
    if (margin) {            /* the truncations above are the other discontinuity: distance of the centre from an integer */
        const double mu_ = fabs(c2[0] - nearbyint(c2[0])), mv_ = fabs(c2[1] - nearbyint(c2[1]));
        if (mu_ < *margin) *margin = mu_;
        if (mv_ < *margin) *margin = mv_;
    }
    float* src = (float*)malloc(sizeof(float) * (size_t)side_f * side_f);
    for (int r = 0; r < side_f; ++r)
        for (int c = 0; c < side_f; ++c) src[(size_t)r * side_f + c] = (float)patch_f[r + (size_t)side_f * c];
    const double off = compat ? (double)(half_f + 1) : (double)half_f;
    for (int j = 0; j < side; ++j)              /* column of the output = u */
        for (int i = 0; i < side; ++i) {        /* row = v */
            const double p[2] = { (double)(xs + j), (double)(ys + i) };
            double pu[2], q3[3], qu[2], qd[2];
            orc_undistort_fm(cam, p, pu);
            for (int a = 0; a < 3; ++a) q3[a] = M[a] * pu[0] + M[a + 3] * pu[1] + M[a + 6];
            qu[0] = q3[0] / q3[2]; qu[1] = q3[1] / q3[2];
            orc_distort_fm(cam, qu, qd);
            const double mu = qd[0] - (uv_f[0] - off), mv = qd[1] - (uv_f[1] - off);
            if (margin) {
                const double m1 = f32_cast_margin(mu), m2 = f32_cast_margin(mv);
                if (m1 < *margin) *margin = m1;
                if (m2 < *margin) *margin = m2;
            }
            out[i + (size_t)side * j] = (double)remap_bilinear_f32(src, side_f, side_f, (float)mu, (float)mv);
        }
    free(src);
    return 1;
}

/* The loop of Tracking::search_IC_matches, :46-65.  The reference refreshes XYZ_w only for inverse-depth
 * features; a Cartesian feature is warped with the point of the last inverse-depth feature before it
 * (compat = 1; zeros when there is none, where the reference reads an uninitialised vector); compat = 0
 * uses the feature's own coordinates.  status (L): result of orc_pred_patch, 2 = no prediction. */
void orc_pred_patches(const rslam_camera* cam, int compat, int L, const uint8_t* type, const int32_t* offset,
                      const double* x, const double* h, const uint8_t* has_h, const double* uv_f, const double* R_f,
                      const double* r_f, const double* patch_f, int half_f, int half, double* out, int32_t* status,
                      double* margin /* L, nullable: per feature */)
{
    const int side = 2 * half + 1, side_f = 2 * half_f + 1;
    double XYZ_w[3] = {0, 0, 0};
    for (int i = 0; i < L; ++i) {
        if (type[i] == RSLAM_FEAT_INVERSE_DEPTH) inversedepth2cartesian(x + offset[i], XYZ_w);
        else if (!compat) { XYZ_w[0] = x[offset[i]]; XYZ_w[1] = x[offset[i] + 1]; XYZ_w[2] = x[offset[i] + 2]; }
        double* o = out + (size_t)i * side * side;
        if (margin) margin[i] = 1e300;
        if (!has_h[i]) { status[i] = 2; memset(o, 0, sizeof(double) * (size_t)side * side); continue; }
        status[i] = orc_pred_patch(cam, compat, x, h + 2 * i, uv_f + 2 * i, R_f + 9 * i, r_f + 3 * i,
                                   patch_f + (size_t)i * side_f * side_f, half_f, XYZ_w, half, o, margin ? margin + i : NULL);
    }
}

/* ------------------------------------------------------------------ */
/* introspection                                                        */
/* ------------------------------------------------------------------ */

int orc_get_supports(orc_ctx* c, int32_t* supports, int32_t* positions, uint64_t* masks, int32_t* words)
{
    if (!c) return RSLAM_ERR_ARG;
    if (supports) memcpy(supports, c->supports, sizeof(int32_t) * (size_t)c->n_eval);
    if (positions) memcpy(positions, c->positions, sizeof(int32_t) * (size_t)c->n_eval);
    if (masks) memcpy(masks, c->masks, sizeof(uint64_t) * (size_t)c->n_eval * (size_t)c->words);
    if (words) *words = c->words;
    return c->n_eval;
}

/* Residual capture (test infrastructure for the scoring kernel's value-level check): when on, the next RANSAC stage
 * records sqrt(n0^2 + n1^2) of Tracking.cpp:472-476,499-503 for every (hypothesised matched feature, matched feature)
 * pair it scores; orc_get_residuals copies the m x m table (row = hypothesised feature's rank, NaN = never scored). */
int orc_enable_residuals(orc_ctx* c, int on)
{
    if (!c) return RSLAM_ERR_ARG;
    c->res_on = on ? 1 : 0;
    return RSLAM_OK;
}

int orc_get_residuals(orc_ctx* c, double* out, int32_t* m)
{
    if (!c) return RSLAM_ERR_ARG;
    if (m) *m = c->res ? c->res_m : 0;
    if (out && c->res) memcpy(out, c->res, sizeof(double) * (size_t)c->res_m * c->res_m);
    return RSLAM_OK;
}

int orc_get_margins(orc_ctx* c, double* score_margin, double* rescue_margin)
{
    if (!c) return RSLAM_ERR_ARG;
    if (score_margin) *score_margin = c->score_margin;
    if (rescue_margin) *rescue_margin = c->rescue_margin;
    return RSLAM_OK;
}

int orc_get_H(orc_ctx* c, double* H)
{
    if (!c || !H || !c->predicted) return RSLAM_ERR_ARG;
    memcpy(H, c->H, sizeof(double) * (size_t)c->L * 2 * c->n);
    return RSLAM_OK;
}

int orc_get_li_state(orc_ctx* c, double* x_li, double* P_li)
{
    if (!c || !c->predicted) return RSLAM_ERR_ARG;
    if (x_li) memcpy(x_li, c->x_li, sizeof(double) * c->n);
    if (P_li) memcpy(P_li, c->p_li, sizeof(double) * (size_t)c->n * c->n);
    return RSLAM_OK;
}
