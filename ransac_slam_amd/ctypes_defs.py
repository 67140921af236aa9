"""ctypes mirrors of the plain-data structs in include/rslam.h."""
import ctypes as C


class Camera(C.Structure):
    _fields_ = [("k1", C.c_double), ("k2", C.c_double),
                ("Cx", C.c_double), ("Cy", C.c_double),
                ("f", C.c_double), ("dx", C.c_double), ("dy", C.c_double),
                ("nRows", C.c_int32), ("nCols", C.c_int32)]


class Config(C.Structure):
    _fields_ = [("cam", Camera),
                ("sigma_z", C.c_double), ("p_success", C.c_double),
                ("n_hyp_init", C.c_int32),
                ("chi2_gate", C.c_double),
                ("compat", C.c_int32), ("adaptive", C.c_int32),
                ("dedup", C.c_int32), ("reserved", C.c_int32)]


class Layout(C.Structure):
    _fields_ = [("n", C.c_int32), ("L", C.c_int32),
                ("type", C.POINTER(C.c_uint8)),
                ("offset", C.POINTER(C.c_int32))]


class StageTimes(C.Structure):
    _fields_ = [(k, C.c_double) for k in (
        "predict_us", "pht_us", "score_us", "select_us", "update_li_us",
        "rescue_us", "update_hi_us", "factor_li_us", "rank_update_li_us",
        "factor_hi_us", "rank_update_hi_us", "total_us")]


RSLAM_OK = 0
ERR_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_STATE, ERR_REF_ASSERT, ERR_NOT_SPD, ERR_IC_NOT_VISIBLE = \
    -1, -2, -3, -4, -5, -6, -7
FEAT_INVERSE_DEPTH, FEAT_CARTESIAN = 0, 1


def default_camera() -> Camera:
    """Camera of examples/Monocular/initialize_param.yaml:9-21 as System.cpp:34-58 fills it."""
    d = 0.0112
    return Camera(k1=0.06333, k2=0.01390, Cx=1.7945 / d, Cy=1.4433 / d,
                  f=2.1735, dx=0.0112, dy=0.0112, nRows=240, nCols=320)


def default_config(compat=1, adaptive=1, dedup=0) -> Config:
    """Constants of Tracking.cpp:354-357,576 and Sigma.noise (yaml:42)."""
    return Config(cam=default_camera(), sigma_z=1.0, p_success=0.99, n_hyp_init=1000,
                  chi2_gate=5.9915, compat=compat, adaptive=adaptive, dedup=dedup, reserved=0)


def make_layout(types):
    """Build (Layout, keepalive arrays) for a list of feature types (0 = inverse depth, 1 = cartesian)."""
    import numpy as np
    types = np.ascontiguousarray(types, dtype=np.uint8)
    widths = np.where(types == FEAT_INVERSE_DEPTH, 6, 3).astype(np.int64)
    offset = (13 + np.concatenate([[0], np.cumsum(widths)[:-1]])).astype(np.int32) if len(types) else np.zeros(0, np.int32)
    n = int(13 + widths.sum())
    lay = Layout(n=n, L=len(types),
                 type=types.ctypes.data_as(C.POINTER(C.c_uint8)),
                 offset=offset.ctypes.data_as(C.POINTER(C.c_int32)))
    return lay, (types, offset)
