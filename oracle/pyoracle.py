"""ctypes binding of the CPU oracle (oracle/rslam_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importers allowed: tests/, __graft_entry__.smoke(),
and the cpu_baseline leg of bench.py -- never the product path.
PARITY UNPINNED (see rslam_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from ransac_slam_amd.ctypes_defs import Config, Layout, make_layout

_HERE = os.path.dirname(os.path.abspath(__file__))
# RSLAM_ORACLE_LIB selects another build of the same source (the OpenMP one, `make -C oracle omp`)
_LIB_PATH = os.environ.get("RSLAM_ORACLE_LIB", os.path.join(_HERE, "_build", "librslam_oracle.so"))
_lib = None

_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_u64p = C.POINTER(C.c_uint64)


def build(force=False):
    """Compile the C restatement with gcc (-O2, no FMA contraction, 1 thread)."""
    src = os.path.join(_HERE, "rslam_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src),
                                                   os.path.getmtime(os.path.join(_HERE, "rslam_oracle.h")),
                                                   os.path.getmtime(os.path.join(_HERE, "..", "include", "rslam.h")))):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_set_structure.argtypes = [C.c_void_p, C.c_int]
        L.orc_predict.argtypes = [C.c_void_p, C.POINTER(Layout), _dp, _dp, _dp, _u8p, _dp]
        L.orc_ransac_update.argtypes = [C.c_void_p, _dp, _u8p, _dp, C.c_int32, _dp, _dp, _u8p, _u8p,
                                        _i32p, _i32p, _i32p]
        L.orc_ransac_only.argtypes = [C.c_void_p, _dp, _u8p, _dp, C.c_int32, C.c_int32, _u8p,
                                      _i32p, _i32p, _i32p]
        L.orc_finish_update.argtypes = [C.c_void_p, _dp, _dp, _u8p, _u8p]
        L.orc_get_supports.argtypes = [C.c_void_p, _i32p, _i32p, _u64p, _i32p]
        L.orc_get_margins.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_enable_residuals.argtypes = [C.c_void_p, C.c_int]
        L.orc_get_residuals.argtypes = [C.c_void_p, _dp, _i32p]
        L.orc_get_H.argtypes = [C.c_void_p, _dp]
        L.orc_get_li_state.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_q2r.argtypes = [_dp, _dp]
        L.orc_hu.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_distort_fm.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_undistort_fm.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_jacob_undistor_fm.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_dRq_times_a_by_dq.argtypes = [_dp, _dp, _dp]
        L.orc_hi_cartesian.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_adaptive_n_hyp.argtypes = [C.c_double, C.c_int, C.c_int]
        L.orc_update.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        L.orc_inverse_lu.argtypes = [C.c_int, _dp, _dp]
        L.orc_map_delete_feature.argtypes = [C.c_int, C.c_int, _u8p, _dp, _dp, C.c_int, _dp, _dp]
        L.orc_map_convert.argtypes = [C.c_int, C.c_int, _u8p, _dp, _dp, C.c_double, _dp, _dp, C.POINTER(C.c_int)]
        L.orc_linearity_index.argtypes = [_dp, _dp, C.c_int, C.c_int]
        L.orc_linearity_index.restype = C.c_double
        L.orc_map_add_feature.argtypes = [C.c_void_p, C.c_double, C.c_int, _dp, _dp, _dp, C.c_double, C.c_double, _dp, _dp]
        L.orc_hinv.argtypes = [C.c_void_p, _dp, _dp, C.c_double, _dp]
        L.orc_add_feature_jacobians.argtypes = [C.c_void_p, C.c_double, C.c_double, _dp, _dp, _dp, _dp]
        L.orc_matching.argtypes = [C.c_void_p, _u8p, C.c_int, _dp, C.c_int, _dp, _u8p, _dp, _dp, _u8p, _dp, _dp]
        L.orc_matching.restype = None
        L.orc_pred_patches.argtypes = [C.c_void_p, C.c_int, C.c_int, _u8p, _i32p, _dp, _dp, _u8p, _dp, _dp, _dp, _dp, C.c_int,
                                       C.c_int, _dp, _i32p, _dp]
        L.orc_pred_patches.restype = None
        L.orc_ekf_prediction.argtypes = [C.c_int, _dp, _dp, C.c_double, C.c_double, C.c_double, _dp, _dp]
        L.orc_motion_model.argtypes = [_dp, C.c_double, C.c_double, C.c_double, _dp, _dp, _dp]
        _lib = L
    return _lib


def _p(a, t=_dp):
    return a.ctypes.data_as(t)


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__(f"oracle returned {code}")
        self.code = code


class Oracle:
    """Frame-level wrapper with the same two segments as the C ABI."""

    def __init__(self, cfg: Config, structure=0):
        self.cfg = cfg
        self._h = C.c_void_p()
        rc = lib().orc_create(C.byref(cfg), C.byref(self._h))
        if rc:
            raise OracleError(rc)
        lib().orc_set_structure(self._h, structure)
        self.n = self.L = 0

    def close(self):
        if self._h:
            lib().orc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def predict(self, types, x_pred, P_pred):
        lay, keep = make_layout(types)
        self._keep = keep
        self.n, self.L = lay.n, lay.L
        x = np.ascontiguousarray(x_pred, dtype=np.float64)
        P = np.asfortranarray(P_pred, dtype=np.float64)
        assert x.shape == (self.n,) and P.shape == (self.n, self.n)
        h = np.full((self.L, 2), np.nan)
        vis = np.zeros(self.L, np.uint8)
        S = np.full((self.L, 4), np.nan)
        rc = lib().orc_predict(self._h, C.byref(lay), _p(x), _p(P), _p(h), _p(vis, _u8p), _p(S))
        if rc:
            raise OracleError(rc)
        return h, vis, S

    def ransac_update(self, z, ic, draws):
        z = np.ascontiguousarray(z, dtype=np.float64)
        ic = np.ascontiguousarray(ic, dtype=np.uint8)
        draws = np.ascontiguousarray(draws, dtype=np.float64)
        x_new = np.zeros(self.n)
        P_new = np.zeros((self.n, self.n), order="F")
        li = np.zeros(self.L, np.uint8)
        hi = np.zeros(self.L, np.uint8)
        bh, bs, he = C.c_int32(), C.c_int32(), C.c_int32()
        rc = lib().orc_ransac_update(self._h, _p(z), _p(ic, _u8p), _p(draws), len(draws), _p(x_new), _p(P_new),
                                     _p(li, _u8p), _p(hi, _u8p), C.byref(bh), C.byref(bs), C.byref(he))
        if rc:
            raise OracleError(rc)
        return dict(x_new=x_new, P_new=P_new, li=li, hi=hi, best_hyp=bh.value, best_support=bs.value,
                    hyps_evaluated=he.value)

    def ransac_only(self, z, ic, draws, max_iters=0):
        z = np.ascontiguousarray(z, dtype=np.float64)
        ic = np.ascontiguousarray(ic, dtype=np.uint8)
        draws = np.ascontiguousarray(draws, dtype=np.float64)
        li = np.zeros(self.L, np.uint8)
        bh, bs, he = C.c_int32(), C.c_int32(), C.c_int32()
        rc = lib().orc_ransac_only(self._h, _p(z), _p(ic, _u8p), _p(draws), len(draws), max_iters,
                                   _p(li, _u8p), C.byref(bh), C.byref(bs), C.byref(he))
        if rc:
            raise OracleError(rc)
        return dict(li=li, best_hyp=bh.value, best_support=bs.value, hyps_evaluated=he.value)

    def finish_update(self):
        x_new = np.zeros(self.n)
        P_new = np.zeros((self.n, self.n), order="F")
        li = np.zeros(self.L, np.uint8)
        hi = np.zeros(self.L, np.uint8)
        rc = lib().orc_finish_update(self._h, _p(x_new), _p(P_new), _p(li, _u8p), _p(hi, _u8p))
        if rc:
            raise OracleError(rc)
        return dict(x_new=x_new, P_new=P_new, li=li, hi=hi)

    def supports(self):
        words = C.c_int32()
        n_eval = lib().orc_get_supports(self._h, None, None, None, C.byref(words))
        sup = np.zeros(max(n_eval, 1), np.int32)
        pos = np.zeros(max(n_eval, 1), np.int32)
        masks = np.zeros((max(n_eval, 1), max(words.value, 1)), np.uint64)
        lib().orc_get_supports(self._h, _p(sup, _i32p), _p(pos, _i32p), _p(masks, _u64p), C.byref(words))
        return sup[:n_eval], pos[:n_eval], masks[:n_eval, :words.value]

    def enable_residuals(self, on=True):
        lib().orc_enable_residuals(self._h, 1 if on else 0)

    def residuals(self):
        """(m, m) residuals in pixels: row = rank of the hypothesised matched feature, NaN = never scored"""
        m = C.c_int32()
        lib().orc_get_residuals(self._h, None, C.byref(m))
        out = np.full((max(m.value, 1), max(m.value, 1)), np.nan)
        lib().orc_get_residuals(self._h, _p(out), C.byref(m))
        return out[:m.value, :m.value]

    def margins(self):
        a, b = C.c_double(), C.c_double()
        lib().orc_get_margins(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def H(self):
        H = np.zeros((self.L, self.n, 2))   # block i is 2 x n col-major == (n,2) C-order transposed view
        rc = lib().orc_get_H(self._h, _p(H))
        if rc:
            raise OracleError(rc)
        return np.transpose(H, (0, 2, 1))   # L x 2 x n

    def li_state(self):
        x = np.zeros(self.n)
        P = np.zeros((self.n, self.n), order="F")
        lib().orc_get_li_state(self._h, _p(x), _p(P))
        return x, P


# ---- unit functions -------------------------------------------------------
def q2r(q):
    q = np.ascontiguousarray(q, np.float64)
    R = np.zeros((3, 3), order="F")
    lib().orc_q2r(_p(q), _p(R))
    return R


def _cam_call(fn, cam, vin, nout):
    vin = np.ascontiguousarray(vin, np.float64)
    out = np.zeros(nout)
    fn(C.byref(cam), _p(vin), _p(out))
    return out


def hu(cam, y):
    return _cam_call(lib().orc_hu, cam, y, 2)


def distort_fm(cam, uv):
    return _cam_call(lib().orc_distort_fm, cam, uv, 2)


def undistort_fm(cam, uvd):
    return _cam_call(lib().orc_undistort_fm, cam, uvd, 2)


def jacob_undistor_fm(cam, uvd):
    return _cam_call(lib().orc_jacob_undistor_fm, cam, uvd, 4).reshape(2, 2, order="F")


def dRq_times_a_by_dq(q, a):
    q = np.ascontiguousarray(q, np.float64)
    a = np.ascontiguousarray(a, np.float64)
    out = np.zeros((3, 4), order="F")
    lib().orc_dRq_times_a_by_dq(_p(q), _p(a), _p(out))
    return out


def hi_cartesian(cam, hrl):
    hrl = np.ascontiguousarray(hrl, np.float64)
    uv = np.zeros(2)
    vis = lib().orc_hi_cartesian(C.byref(cam), _p(hrl), _p(uv))
    return bool(vis), uv


def adaptive_n_hyp(p, support, num_ic):
    return lib().orc_adaptive_n_hyp(p, support, num_ic)


def update(compat, x, P, H, z, h):
    n, r = len(x), len(z)
    x = np.ascontiguousarray(x, np.float64)
    P = np.asfortranarray(P, np.float64)
    H = np.asfortranarray(np.reshape(H, (r, n)), np.float64)
    z = np.ascontiguousarray(z, np.float64)
    h = np.ascontiguousarray(h, np.float64)
    xo = np.zeros(n)
    Po = np.zeros((n, n), order="F")
    rc = lib().orc_update(compat, n, r, _p(x), _p(P), _p(H), _p(z), _p(h), _p(xo), _p(Po))
    if rc:
        raise OracleError(rc)
    return xo, Po


def inverse_lu(A):
    A = np.asfortranarray(A, np.float64)
    n = A.shape[0]
    out = np.zeros((n, n), order="F")
    lib().orc_inverse_lu(n, _p(A), _p(out))
    return out


def ekf_prediction(x_kk, P_kk, delta_t=1.0, std_a=0.007, std_alpha=0.007):
    x = np.ascontiguousarray(x_kk, np.float64)
    P = np.asfortranarray(P_kk, np.float64)
    n = len(x)
    xo = np.zeros(n)
    Po = np.zeros((n, n), order="F")
    rc = lib().orc_ekf_prediction(n, _p(x), _p(P), delta_t, std_a, std_alpha, _p(xo), _p(Po))
    if rc:
        raise OracleError(rc)
    return xo, Po


def motion_model(xv, delta_t=1.0, std_a=0.007, std_alpha=0.007):
    xv = np.ascontiguousarray(xv, np.float64)
    xp = np.zeros(13)
    F = np.zeros((13, 13), order="F")
    Q = np.zeros((13, 13), order="F")
    lib().orc_motion_model(_p(xv), delta_t, std_a, std_alpha, _p(xp), _p(F), _p(Q))
    return xp, F, Q


def map_delete_feature(types, x, P, feature):
    types = np.ascontiguousarray(types, np.uint8)
    x = np.ascontiguousarray(x, np.float64); P = np.asfortranarray(P, np.float64)
    n = len(x); w = 6 if types[feature] == 0 else 3
    xo = np.zeros(n - w); Po = np.zeros((n - w, n - w), order="F")
    rc = lib().orc_map_delete_feature(n, len(types), _p(types, _u8p), _p(x), _p(P), feature, _p(xo), _p(Po))
    if rc:
        raise OracleError(rc)
    return xo, Po


def map_convert(types, x, P, threshold=0.1):
    types = np.ascontiguousarray(types, np.uint8)
    x = np.ascontiguousarray(x, np.float64); P = np.asfortranarray(P, np.float64)
    n = len(x)
    xo = np.zeros(n - 3); Po = np.zeros((n - 3, n - 3), order="F")
    conv = C.c_int(-1)
    rc = lib().orc_map_convert(n, len(types), _p(types, _u8p), _p(x), _p(P), threshold, _p(xo), _p(Po), C.byref(conv))
    if rc:
        raise OracleError(rc)
    if conv.value < 0:
        return -1, x, P
    return conv.value, xo, Po


def linearity_index(x, P, offset):
    x = np.ascontiguousarray(x, np.float64); P = np.asfortranarray(P, np.float64)
    return lib().orc_linearity_index(_p(x), _p(P), len(x), offset)


def map_add_feature(cam, std_z, x, P, uvd, initial_rho=1.0, std_rho=1.0):
    x = np.ascontiguousarray(x, np.float64); P = np.asfortranarray(P, np.float64)
    uvd = np.ascontiguousarray(uvd, np.float64)
    n = len(x)
    xo = np.zeros(n + 6); Po = np.zeros((n + 6, n + 6), order="F")
    rc = lib().orc_map_add_feature(C.byref(cam), std_z, n, _p(x), _p(P), _p(uvd), initial_rho, std_rho, _p(xo), _p(Po))
    if rc:
        raise OracleError(rc)
    return xo, Po


def hinv(cam, uvd, Xv, initial_rho=1.0):
    uvd = np.ascontiguousarray(uvd, np.float64); Xv = np.ascontiguousarray(Xv, np.float64)
    y = np.zeros(6)
    lib().orc_hinv(C.byref(cam), _p(uvd), _p(Xv), initial_rho, _p(y))
    return y


def add_feature_jacobians(cam, std_z, std_rho, uvd, Xv):
    uvd = np.ascontiguousarray(uvd, np.float64); Xv = np.ascontiguousarray(Xv, np.float64)
    D = np.zeros((6, 13), order="F"); Rn = np.zeros((6, 6), order="F")
    lib().orc_add_feature_jacobians(C.byref(cam), std_z, std_rho, _p(uvd), _p(Xv), _p(D), _p(Rn))
    return D, Rn


def matching(cam, image, patches, h, has_h, S, half=6):
    """Tracking::matching.  image (nRows, nCols) uint8; patches (L, 13, 13) with patches[f][r, c] the
    predicted patch (stored column-major per feature like Eigen); -> z (L,2), ic (L), corr (L), margins (3)"""
    image = np.ascontiguousarray(image, np.uint8)
    L = len(has_h)
    side = 2 * half + 1
    pt = np.ascontiguousarray(np.transpose(np.asarray(patches, np.float64).reshape(L, side, side), (0, 2, 1)))  # col-major per feature
    h = np.ascontiguousarray(h, np.float64); S = np.ascontiguousarray(S, np.float64)
    has_h = np.ascontiguousarray(has_h, np.uint8)
    z = np.zeros((L, 2)); ic = np.zeros(L, np.uint8); corr = np.zeros(L); m = np.zeros(3)
    lib().orc_matching(C.byref(cam), _p(image, _u8p), L, _p(pt), half, _p(h), _p(has_h, _u8p), _p(S), _p(z), _p(ic, _u8p),
                       _p(corr), _p(m))
    return z, ic, corr, m


def pred_patches(cam, compat, types, offsets, x, h, has_h, uv_f, R_f, r_f, patch_f, half_f=20, half=6):
    """Tracking::pred_patch_fc for every feature.  uv_f (L,2); R_f (L,3,3) rotation matrices; r_f (L,3);
    patch_f (L,41,41) with patch_f[f][row, col]  ->  patches (L,13,13) [row, col], status (L), margins (L)"""
    L = len(types)
    sf, so = 2 * half_f + 1, 2 * half + 1
    types = np.ascontiguousarray(types, np.uint8); offsets = np.ascontiguousarray(offsets, np.int32)
    x = np.ascontiguousarray(x, np.float64); h = np.ascontiguousarray(np.nan_to_num(h), np.float64)
    has_h = np.ascontiguousarray(has_h, np.uint8)
    uv = np.ascontiguousarray(uv_f, np.float64)
    Rf = np.ascontiguousarray(np.transpose(np.asarray(R_f, np.float64).reshape(L, 3, 3), (0, 2, 1)))      # col-major
    rf = np.ascontiguousarray(r_f, np.float64)
    pf = np.ascontiguousarray(np.transpose(np.asarray(patch_f, np.float64).reshape(L, sf, sf), (0, 2, 1)))
    out = np.zeros((L, so, so)); status = np.zeros(L, np.int32); m = np.zeros(max(L, 1))
    lib().orc_pred_patches(C.byref(cam), compat, L, _p(types, _u8p), _p(offsets, _i32p), _p(x), _p(h), _p(has_h, _u8p),
                           _p(uv), _p(Rf), _p(rf), _p(pf), half_f, half, _p(out), _p(status, _i32p), _p(m))
    return np.transpose(out, (0, 2, 1)).copy(), status, m[:L]
