"""ctypes binding of the product library librslam_hip.so (C ABI: include/rslam.h).

Plumbing only: every call goes to the HIP implementation.  There is no CPU
fallback -- loading fails loudly when the library is missing and
``RslamHip()`` raises when no HIP device can be opened.
"""
import ctypes as C
import os

import numpy as np

from .ctypes_defs import Config, Layout, StageTimes, make_layout

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product library, and its diagnostic variant (same sources with -DRSLAM_DEBUG: the rslam_debug_* entry points --
# value-level probes, time stamps, fault injection -- exist only there; build.py).  This module is test / benchmark
# plumbing: RSLAM_HIP_LIB names another build of the product library for A/B measurements (scripts/ab_frame.py).
LIB_PATH = os.environ.get("RSLAM_HIP_LIB", os.path.join(_HERE, "librslam_hip.so"))
LIB_PATH_DEBUG = os.environ.get("RSLAM_HIP_LIB_DEBUG", os.path.join(_HERE, "librslam_hip_dbg.so"))
_lib = None
_lib_debug = None

_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_u64p = C.POINTER(C.c_uint64)

# every symbol include/rslam.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "rslam_create": (C.c_int, [C.POINTER(Config), C.c_int, C.POINTER(C.c_void_p)]),
    "rslam_destroy": (C.c_int, [C.c_void_p]),
    "rslam_error_string": (C.c_char_p, [C.c_int]),
    "rslam_version": (C.c_char_p, []),
    "rslam_predict": (C.c_int, [C.c_void_p, C.POINTER(Layout), _dp, _dp, _dp, _u8p, _dp]),
    "rslam_ransac_update": (C.c_int, [C.c_void_p, _dp, _u8p, _dp, C.c_int32, _dp, _dp, _u8p, _u8p, _i32p, _i32p, _i32p]),
    "rslam_set_posterior": (C.c_int, [C.c_void_p, C.POINTER(Layout), _dp, _dp]),
    "rslam_ekf_prediction": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double]),
    "rslam_fetch_prior": (C.c_int, [C.c_void_p, _dp, _dp]),
    "rslam_match": (C.c_int, [C.c_void_p, _u8p, _dp, _dp, _u8p, _dp]),
    "rslam_set_feature_records": (C.c_int, [C.c_void_p, C.c_int32, _dp, _dp, _dp, _dp]),
    "rslam_append_feature_record": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp]),
    "rslam_predict_patches": (C.c_int, [C.c_void_p, _dp, _i32p]),
    "rslam_map_delete_feature": (C.c_int, [C.c_void_p, C.c_int32]),
    "rslam_map_convert": (C.c_int, [C.c_void_p, C.c_double, _i32p, _dp]),
    "rslam_map_add_feature": (C.c_int, [C.c_void_p, _dp, C.c_double, C.c_double]),
    "rslam_map_predict": (C.c_int, [C.c_void_p, _dp, _u8p]),
    "rslam_get_layout": (C.c_int, [C.c_void_p, _i32p, _i32p, _u8p, _i32p]),
    "rslam_fetch_cov": (C.c_int, [C.c_void_p, _dp]),
    "rslam_fetch_state": (C.c_int, [C.c_void_p, _dp]),
    "rslam_unpin_host_buffers": (C.c_int, [C.c_void_p]),
    "rslam_enable_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "rslam_timings": (C.c_int, [C.c_void_p, C.POINTER(StageTimes)]),
    "rslam_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rslam_load_frame": (C.c_int, [C.c_void_p, C.POINTER(Layout), _dp, _dp, _dp, _u8p, _dp, C.c_int32]),
    "rslam_load_measurements": (C.c_int, [C.c_void_p, _dp, _u8p, _dp, C.c_int32]),
    "rslam_get_counters": (C.c_int, [C.c_void_p, _i32p, _i32p]),
    "rslam_update_mode": (C.c_int, [C.c_void_p]),
    "rslam_last_raw_status": (C.c_int, [C.c_void_p]),
    "rslam_last_wait_detail": (C.c_int, [C.c_void_p]),
    "rslam_last_wait_polls": (C.c_int, [C.c_void_p]),
    "rslam_step_predict": (C.c_int, [C.c_void_p]),
    "rslam_step_score": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "rslam_step_update": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rslam_step_phase": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]),
    "rslam_step_frame": (C.c_int, [C.c_void_p, C.c_int32]),
    "rslam_shard_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "rslam_shard_frame_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "rslam_sync": (C.c_int, [C.c_void_p]),
    "rslam_fetch_prediction": (C.c_int, [C.c_void_p, _dp, _u8p, _dp]),
    "rslam_fetch_results": (C.c_int, [C.c_void_p, _dp, _u8p, _u8p, _i32p, _i32p, _i32p, _i32p, _i32p]),
    "rslam_fetch_supports": (C.c_int, [C.c_void_p, _i32p, _u64p, _i32p]),
    "rslam_k_rank_update": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                      C.c_void_p, C.c_int32]),
    "rslam_k_rank_update_time": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _dp]),
    "rslam_k_gemm_nt": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_int32,
                                  C.c_void_p, C.c_int32, C.c_double, C.c_void_p, C.c_int32]),
    "rslam_k_mfma_f64_peak": (C.c_int, [C.c_void_p, _dp]),
    "rslam_k_mfma_f64_probe": (C.c_int, [C.c_void_p, C.c_int32, _dp, _dp, _dp]),
    "rslam_k_mfma4_raw": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _dp, _dp, _dp, _dp]),
    "rslam_k_hbm_copy_peak": (C.c_int, [C.c_void_p, C.c_int64, _dp]),
}


class RslamError(RuntimeError):
    def __init__(self, code, where=""):
        L = _lib or _lib_debug
        msg = L.rslam_error_string(code).decode() if L is not None else str(code)
        super().__init__(f"{where}: rslam error {code} ({msg})")
        self.code = code


# the diagnostic variant's extra entry points (not part of include/rslam.h)
DEBUG_SYMBOLS = {
    "rslam_debug_score_residuals": (C.c_int, [C.c_void_p, _dp, _i32p]),
    "rslam_debug_distort": (C.c_int, [C.c_void_p, C.c_int32, _dp, _dp, _dp]),
    "rslam_debug_set_sweep_exp": (C.c_int, [C.c_int]),
    "rslam_debug_set_k10_inject": (C.c_int, [C.c_void_p, C.c_int]),
    "rslam_debug_sweep_stamps": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]),
    "rslam_debug_k10_stamps": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]),
}


def _load(path, symbols, lenient):
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the HIP extension is mandatory, there is no CPU fallback)")
    L = C.CDLL(path)
    for name, (res, args) in symbols.items():
        if lenient and not hasattr(L, name):
            continue                          # an older build under A/B measurement may lack the newest entry points
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    return L


def lib(debug=False):
    """Load librslam_hip.so (debug: its diagnostic variant librslam_hip_dbg.so); raises if it has not been built (no fallback)."""
    global _lib, _lib_debug
    if debug:
        if _lib_debug is None:
            _lib_debug = _load(LIB_PATH_DEBUG, {**SYMBOLS, **DEBUG_SYMBOLS}, False)
        return _lib_debug
    if _lib is None:
        _lib = _load(LIB_PATH, SYMBOLS, "RSLAM_HIP_LIB" in os.environ)
    return _lib


_lib_other = {}


def lib_at(path):
    """a diagnostic-variant build at another path (ransac_slam_amd/_dev/*.so); raises if it is missing"""
    path = os.path.abspath(path)
    if path not in _lib_other:
        _lib_other[path] = _load(path, {**SYMBOLS, **DEBUG_SYMBOLS}, False)
    return _lib_other[path]


def _p(a, t=_dp):
    return a.ctypes.data_as(t)


def _chk(rc, where):
    if rc != 0:
        raise RslamError(rc, where)


class RslamHip:
    """One context on one GPU (one process per GPU in multi-GPU runs)."""

    def __init__(self, cfg: Config, device=0, debug=False, lib_path=None):
        """debug: the context lives in the diagnostic variant of the library (needed by the debug_* methods);
        lib_path: another build of the diagnostic variant (build.py build_dev / build_fenced) loaded beside the others"""
        self.cfg = cfg
        self.debug = bool(debug) or lib_path is not None
        self._L = lib_at(lib_path) if lib_path is not None else lib(self.debug)
        self._h = C.c_void_p()
        _chk(self._L.rslam_create(C.byref(cfg), device, C.byref(self._h)), "rslam_create")
        self.n = self.L = self.H = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.rslam_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- drop-in API ----------------------------------------------------
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
        _chk(self._L.rslam_predict(self._h, C.byref(lay), _p(x), _p(P), _p(h), _p(vis, _u8p), _p(S)), "rslam_predict")
        return h, vis, S

    def predict_resident(self):
        """Segment 1 on the prior that ekf_prediction() left in HBM (no upload)."""
        h = np.full((self.L, 2), np.nan)
        vis = np.zeros(self.L, np.uint8)
        S = np.full((self.L, 4), np.nan)
        _chk(self._L.rslam_predict(self._h, None, None, None, _p(h), _p(vis, _u8p), _p(S)), "rslam_predict")
        return h, vis, S

    def set_posterior(self, types, x_kk, P_kk):
        lay, keep = make_layout(types)
        self._keep = keep
        self.n, self.L = lay.n, lay.L
        x = np.ascontiguousarray(x_kk, dtype=np.float64)
        P = np.asfortranarray(P_kk, dtype=np.float64)
        _chk(self._L.rslam_set_posterior(self._h, C.byref(lay), _p(x), _p(P)), "rslam_set_posterior")

    def ekf_prediction(self, delta_t=1.0, std_a=0.007, std_alpha=0.007):
        _chk(self._L.rslam_ekf_prediction(self._h, delta_t, std_a, std_alpha), "rslam_ekf_prediction")

    # ---- Tracking::matching on the resident prediction ---------------------
    def match(self, image, patches=None):
        """image (nRows, nCols) uint8; patches (L, 13, 13) with patches[f][row, col], or None for the ones
        predict_patches() left on the device -> z (L,2), ic (L), corr (L)"""
        image = np.ascontiguousarray(image, np.uint8)
        pt = None
        if patches is not None:
            pt = np.ascontiguousarray(np.transpose(np.asarray(patches, np.float64).reshape(self.L, 13, 13), (0, 2, 1)))
        z = np.zeros((max(self.L, 1), 2)); ic = np.zeros(max(self.L, 1), np.uint8); corr = np.zeros(max(self.L, 1))
        _chk(self._L.rslam_match(self._h, _p(image, _u8p), _p(pt) if pt is not None else None, _p(z), _p(ic, _u8p), _p(corr)),
             "rslam_match")
        return z[:self.L], ic[:self.L], corr[:self.L]

    # ---- Tracking::pred_patch_fc and its feature store ------------------------
    @staticmethod
    def _records(uv_f, R_f, r_f, patch_f):
        uv = np.ascontiguousarray(uv_f, np.float64).reshape(-1, 2)
        n = len(uv)
        Rf = np.ascontiguousarray(np.transpose(np.asarray(R_f, np.float64).reshape(n, 3, 3), (0, 2, 1)))      # column-major
        rf = np.ascontiguousarray(r_f, np.float64).reshape(n, 3)
        pf = np.ascontiguousarray(np.transpose(np.asarray(patch_f, np.float64).reshape(n, 41, 41), (0, 2, 1)))
        return n, uv, Rf, rf, pf

    def set_feature_records(self, uv_f, R_f, r_f, patch_f):
        """uv_f (L,2); R_f (L,3,3); r_f (L,3); patch_f (L,41,41) with patch_f[f][row, col]"""
        n, uv, Rf, rf, pf = self._records(uv_f, R_f, r_f, patch_f)
        _chk(self._L.rslam_set_feature_records(self._h, n, _p(uv), _p(Rf), _p(rf), _p(pf)), "rslam_set_feature_records")

    def append_feature_record(self, uv_f, R_f, r_f, patch_f):
        _, uv, Rf, rf, pf = self._records(uv_f, R_f, r_f, patch_f)
        _chk(self._L.rslam_append_feature_record(self._h, _p(uv), _p(Rf), _p(rf), _p(pf)), "rslam_append_feature_record")

    def predict_patches(self, fetch=True):
        """-> patches (L,13,13) [row, col] (None when fetch is False), status (L)"""
        out = np.zeros((max(self.L, 1), 13, 13)) if fetch else None
        st = np.zeros(max(self.L, 1), np.int32)
        _chk(self._L.rslam_predict_patches(self._h, _p(out) if fetch else None, _p(st, _i32p)), "rslam_predict_patches")
        return (np.transpose(out[:self.L], (0, 2, 1)).copy() if fetch else None), st[:self.L]

    # ---- Map::map_management state surgery on the resident posterior -------
    def get_layout(self):
        n, L = C.c_int32(), C.c_int32()
        _chk(self._L.rslam_get_layout(self._h, C.byref(n), C.byref(L), None, None), "rslam_get_layout")
        types = np.zeros(max(L.value, 1), np.uint8)
        offs = np.zeros(max(L.value, 1), np.int32)
        _chk(self._L.rslam_get_layout(self._h, C.byref(n), C.byref(L), _p(types, _u8p), _p(offs, _i32p)), "rslam_get_layout")
        self.n, self.L = n.value, L.value
        return n.value, types[:L.value].copy(), offs[:L.value].copy()

    def map_delete_feature(self, feature):
        _chk(self._L.rslam_map_delete_feature(self._h, int(feature)), "rslam_map_delete_feature")
        self.get_layout()

    def map_convert(self, threshold=0.1):
        """-> (converted feature index or -1, linearity index of every feature)"""
        conv = C.c_int32(-1)
        lin = np.zeros(max(self.L, 1))
        _chk(self._L.rslam_map_convert(self._h, threshold, C.byref(conv), _p(lin)), "rslam_map_convert")
        L_before = self.L
        self.get_layout()
        return conv.value, lin[:L_before]

    def map_add_feature(self, uvd, initial_rho=1.0, std_rho=1.0):
        uvd = np.ascontiguousarray(uvd, dtype=np.float64)
        _chk(self._L.rslam_map_add_feature(self._h, _p(uvd), initial_rho, std_rho), "rslam_map_add_feature")
        self.get_layout()

    def map_predict(self):
        h = np.full((max(self.L, 1), 2), np.nan)
        vis = np.zeros(max(self.L, 1), np.uint8)
        _chk(self._L.rslam_map_predict(self._h, _p(h), _p(vis, _u8p)), "rslam_map_predict")
        return h[:self.L], vis[:self.L]

    def fetch_posterior(self):
        x = np.zeros(self.n)
        P = np.zeros((self.n, self.n), order="F")
        _chk(self._L.rslam_fetch_state(self._h, _p(x)), "rslam_fetch_state")
        _chk(self._L.rslam_fetch_cov(self._h, _p(P)), "rslam_fetch_cov")
        return x, P

    def sync_stream(self):
        """wait for everything enqueued so far (no frame status involved)"""
        _chk(self._L.rslam_fetch_prior(self._h, None, None), "rslam_fetch_prior")

    def fetch_prior(self):
        x = np.zeros(self.n)
        P = np.zeros((self.n, self.n), order="F")
        _chk(self._L.rslam_fetch_prior(self._h, _p(x), _p(P)), "rslam_fetch_prior")
        return x, P

    def ransac_update(self, z, ic, draws, want_P=True, P_out=None):
        """P_out: an (n, n) Fortran-ordered float64 array to receive p_k_k (reused frame after frame by a caller that asked
        for page-locked covariance buffers: Config.reserved & 1)"""
        z = np.ascontiguousarray(z, dtype=np.float64)
        ic = np.ascontiguousarray(ic, dtype=np.uint8)
        draws = np.ascontiguousarray(draws, dtype=np.float64)
        self.H = len(draws)
        x_new = np.zeros(self.n)
        if want_P and P_out is not None:
            assert P_out.shape == (self.n, self.n) and P_out.flags.f_contiguous and P_out.dtype == np.float64
            P_new = P_out
        else:
            P_new = np.zeros((self.n, self.n), order="F") if want_P else None
        li = np.zeros(self.L, np.uint8)
        hi = np.zeros(self.L, np.uint8)
        bh, bs, he = C.c_int32(), C.c_int32(), C.c_int32()
        _chk(self._L.rslam_ransac_update(self._h, _p(z), _p(ic, _u8p), _p(draws), len(draws), _p(x_new),
                                       _p(P_new) if want_P else None, _p(li, _u8p), _p(hi, _u8p),
                                       C.byref(bh), C.byref(bs), C.byref(he)), "rslam_ransac_update")
        return dict(x_new=x_new, P_new=P_new, li=li, hi=hi, best_hyp=bh.value, best_support=bs.value,
                    hyps_evaluated=he.value)

    # ---- resident API ---------------------------------------------------
    def set_stream(self, stream_handle):
        _chk(self._L.rslam_set_stream(self._h, C.c_void_p(stream_handle)), "rslam_set_stream")

    def load_frame(self, types, x_pred, P_pred, z, ic, draws):
        lay, keep = make_layout(types)
        self._keep = keep
        self.n, self.L, self.H = lay.n, lay.L, len(draws)
        x = np.ascontiguousarray(x_pred, dtype=np.float64)
        P = np.asfortranarray(P_pred, dtype=np.float64)
        z = np.ascontiguousarray(z, dtype=np.float64)
        ic = np.ascontiguousarray(ic, dtype=np.uint8)
        draws = np.ascontiguousarray(draws, dtype=np.float64)
        _chk(self._L.rslam_load_frame(self._h, C.byref(lay), _p(x), _p(P), _p(z), _p(ic, _u8p), _p(draws), len(draws)),
             "rslam_load_frame")

    def load_measurements(self, z, ic, draws):
        z = np.ascontiguousarray(z, dtype=np.float64)
        ic = np.ascontiguousarray(ic, dtype=np.uint8)
        draws = np.ascontiguousarray(draws, dtype=np.float64)
        self.H = len(draws)
        _chk(self._L.rslam_load_measurements(self._h, _p(z), _p(ic, _u8p), _p(draws), len(draws)), "rslam_load_measurements")

    def unpin_host_buffers(self):
        """drop the RSLAM_PIN_HOST_COV registrations (before the caller re-allocates a covariance buffer of the same size)"""
        _chk(self._L.rslam_unpin_host_buffers(self._h), "rslam_unpin_host_buffers")

    def counters(self):
        a, b = C.c_int32(), C.c_int32()
        _chk(self._L.rslam_get_counters(self._h, C.byref(a), C.byref(b)), "rslam_get_counters")
        return dict(graph_captures=a.value, sweep_reruns=b.value)

    def step_predict(self):
        _chk(self._L.rslam_step_predict(self._h), "rslam_step_predict")

    def step_score(self, hyp_begin, hyp_end, d_supports_ptr):
        _chk(self._L.rslam_step_score(self._h, hyp_begin, hyp_end, C.c_void_p(d_supports_ptr)), "rslam_step_score")

    def step_update(self, d_supports_ptr):
        _chk(self._L.rslam_step_update(self._h, C.c_void_p(d_supports_ptr)), "rslam_step_update")

    def step_phase(self, phase, hyp_begin, hyp_end, d_supports_ptr, use_graph=True):
        _chk(self._L.rslam_step_phase(self._h, phase, hyp_begin, hyp_end, C.c_void_p(d_supports_ptr), 1 if use_graph else 0),
             "rslam_step_phase")

    def step_frame(self, use_graph=True):
        _chk(self._L.rslam_step_frame(self._h, 1 if use_graph else 0), "rslam_step_frame")

    def shard_frame(self, nccl_comm, rank, world, use_graph=True):
        """one rank's hypothesis-sharded frame with the RCCL all-gather inside (nccl_comm: ncclComm_t as an int / c_void_p, or None)"""
        _chk(self._L.rslam_shard_frame(self._h, C.c_void_p(nccl_comm) if nccl_comm else None, rank, world, 1 if use_graph else 0),
             "rslam_shard_frame")

    def shard_frame_allreduce(self, nccl_comm, rank, world, use_graph=True):
        """the same frame with ONE 8-byte ncclAllReduce(MAX) of (support << 32 | ~index) instead of the all-gather (adaptive = 0 only)"""
        _chk(self._L.rslam_shard_frame_allreduce(self._h, C.c_void_p(nccl_comm) if nccl_comm else None, rank, world, 1 if use_graph else 0),
             "rslam_shard_frame_allreduce")

    def sync(self):
        _chk(self._L.rslam_sync(self._h), "rslam_sync")

    def enable_timing(self, on=True):
        _chk(self._L.rslam_enable_timing(self._h, 1 if on else 0), "rslam_enable_timing")

    def timings(self):
        t = StageTimes()
        _chk(self._L.rslam_timings(self._h, C.byref(t)), "rslam_timings")
        return {k: getattr(t, k) for k, _ in StageTimes._fields_}

    def fetch_prediction(self):
        h = np.zeros((self.L, 2))
        vis = np.zeros(self.L, np.uint8)
        S = np.zeros((self.L, 4))
        _chk(self._L.rslam_fetch_prediction(self._h, _p(h), _p(vis, _u8p), _p(S)), "rslam_fetch_prediction")
        return h, vis, S

    def fetch_results(self, want_P=True):
        x_new = np.zeros(self.n)
        li = np.zeros(self.L, np.uint8)
        hi = np.zeros(self.L, np.uint8)
        v = [C.c_int32() for _ in range(5)]
        _chk(self._L.rslam_fetch_results(self._h, _p(x_new), _p(li, _u8p), _p(hi, _u8p), *[C.byref(a) for a in v]),
             "rslam_fetch_results")
        out = dict(x_new=x_new, li=li, hi=hi, best_hyp=v[0].value, best_support=v[1].value,
                   hyps_evaluated=v[2].value, n_li=v[3].value, n_hi=v[4].value)
        if want_P:
            P = np.zeros((self.n, self.n), order="F")
            _chk(self._L.rslam_fetch_cov(self._h, _p(P)), "rslam_fetch_cov")
            out["P_new"] = P
        return out

    def fetch_supports(self):
        words = C.c_int32()
        _chk(self._L.rslam_fetch_supports(self._h, None, None, C.byref(words)), "rslam_fetch_supports")
        sup = np.zeros(max(self.H, 1), np.int32)
        masks = np.zeros((max(self.H, 1), max(words.value, 1)), np.uint64)
        _chk(self._L.rslam_fetch_supports(self._h, _p(sup, _i32p), _p(masks, _u64p), C.byref(words)),
             "rslam_fetch_supports")
        return sup[:self.H], masks[:self.H, :words.value]

    # ---- kernel-level entry points (device pointers) ----------------------
    def k_rank_update(self, n, r, dA, lda, dY, ldy, dC, ldc):
        _chk(self._L.rslam_k_rank_update(self._h, n, r, C.c_void_p(dA), lda, C.c_void_p(dY), ldy, C.c_void_p(dC), ldc),
             "rslam_k_rank_update")

    def k_rank_update_time(self, n, r, reps=20):
        """mean duration (us) of one stand-alone rank-update launch of shape (n, r), hipEvents on the context's stream"""
        v = C.c_double()
        _chk(self._L.rslam_k_rank_update_time(self._h, n, r, reps, C.byref(v)), "rslam_k_rank_update_time")
        return v.value

    def k_gemm_nt(self, m, n, k, alpha, dA, lda, dB, ldb, beta, dC, ldc):
        _chk(self._L.rslam_k_gemm_nt(self._h, m, n, k, alpha, C.c_void_p(dA), lda, C.c_void_p(dB), ldb, beta,
                                   C.c_void_p(dC), ldc), "rslam_k_gemm_nt")

    def mfma_f64_peak(self):
        v = C.c_double()
        _chk(self._L.rslam_k_mfma_f64_peak(self._h, C.byref(v)), "rslam_k_mfma_f64_peak")
        return v.value

    def mfma_f64_probe(self, waves_per_simd=1, mode=0):
        t, cy, mhz = C.c_double(), C.c_double(), C.c_double()
        _chk(self._L.rslam_k_mfma_f64_probe(self._h, waves_per_simd + 16 * mode, C.byref(t), C.byref(cy), C.byref(mhz)),
             "rslam_k_mfma_f64_probe")
        return dict(tflops=t.value, cycles_per_mfma=cy.value, clock_mhz=mhz.value)

    def mfma4_raw(self, a, b, c, cbsz=0, abid=0):
        a = np.ascontiguousarray(a, np.float64); b = np.ascontiguousarray(b, np.float64)
        c = np.ascontiguousarray(c, np.float64); d = np.zeros(64)
        _chk(self._L.rslam_k_mfma4_raw(self._h, cbsz, abid, _p(a), _p(b), _p(c), _p(d)), "rslam_k_mfma4_raw")
        return d

    # ---- queries ------------------------------------------------------------------
    def update_mode(self):
        """0 launch-per-step sweep, 1 persistent sweep + stand-alone rank update, 2 update inside the persistent sweep launch"""
        return self._L.rslam_update_mode(self._h)

    debug_update_mode = update_mode        # (name of rounds 1-3)

    def last_raw_status(self):
        """raw device-side code of the last bounded wait that ran out (0: none)"""
        return self._L.rslam_last_raw_status(self._h)

    def last_wait_detail(self):
        """the FIRST bounded wait that ran out: dict(code, workgroup, needed) or None"""
        v = self._L.rslam_last_wait_detail(self._h)
        if v == 0:
            return None
        p = self._L.rslam_last_wait_polls(self._h) & 0xffffffff
        return {"code": v & 0xff, "workgroup": (v >> 8) & 0xfff, "needed": (v >> 20) & 0x7ff,
                "polls": (p >> 16) & 0xffff, "elapsed_us": p & 0xffff}

    # ---- diagnostics: contexts created with debug=True only (librslam_hip_dbg.so) -------------------------
    def _dbg(self):
        if not self.debug:
            raise RuntimeError("diagnostic entry points exist in librslam_hip_dbg.so only: RslamHip(cfg, debug=True)")
        return self._L

    def debug_set_sweep_exp(self, mask):
        """RSLAM_SWEEP_EXP switches (measurement / fault injection; process-wide inside the diagnostic library); -1 = environment"""
        _chk(self._dbg().rslam_debug_set_sweep_exp(int(mask)), "rslam_debug_set_sweep_exp")

    def debug_set_k10_inject(self, on):
        _chk(self._dbg().rslam_debug_set_k10_inject(self._h, 1 if on else 0), "rslam_debug_set_k10_inject")

    def debug_sweep_stamps(self):
        """one eager frame with wall-clock stamps inside the persistent sweep (last sweep launch = HI pass) ->
        int64 array [who][block/step][slot] (100 MHz ticks; scripts/sweep_stamps.py explains the slots)"""
        fn = self._dbg().rslam_debug_sweep_stamps
        _chk(fn(self._h, None, 1), "rslam_debug_sweep_stamps")
        self.step_frame(False); self.sync()
        buf = np.zeros(6 * 16 * 8 + 512, np.uint64)
        _chk(fn(self._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), 0), "rslam_debug_sweep_stamps")
        self.wg_start_end = buf[768:].reshape(2, 256).astype(np.int64)      # per workgroup of the launch
        return buf[:768].reshape(6, 16, 8).astype(np.int64)

    def debug_score_residuals(self):
        """(m, m) squared residuals the scoring kernel compares with sigma_z^2: row = matched rank of the hypothesised
        feature, column = matched rank of the scored one (resident frame, after its scoring stage ran)"""
        fn = self._dbg().rslam_debug_score_residuals
        m = C.c_int32()
        _chk(fn(self._h, None, C.byref(m)), "rslam_debug_score_residuals")
        out = np.zeros((max(m.value, 1), max(m.value, 1)))
        _chk(fn(self._h, _p(out), C.byref(m)), "rslam_debug_score_residuals")
        return out[:m.value, :m.value]

    def debug_distort(self, uv):
        """undistorted pixels (n, 2) -> (distort_fm_score, distort_fm) as the device evaluates them"""
        fn = self._dbg().rslam_debug_distort
        uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 2)
        a = np.zeros_like(uv); b = np.zeros_like(uv)
        _chk(fn(self._h, len(uv), _p(uv), _p(a), _p(b)), "rslam_debug_distort")
        return a, b

    def hbm_copy_peak(self, nbytes=1 << 30):
        v = C.c_double()
        _chk(self._L.rslam_k_hbm_copy_peak(self._h, nbytes, C.byref(v)), "rslam_k_hbm_copy_peak")
        return v.value
