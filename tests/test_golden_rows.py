"""Committed golden vectors of the widened rows (tests/golden/rows/*.npz, made by
tests/golden/make_golden_rows.py): the oracle must reproduce them (CPU), and the device must match
them through the C ABI (GPU)."""
import glob
import os

import numpy as np
import pytest

from ransac_slam_amd import default_camera, default_config

ROWS = os.path.join(os.path.dirname(__file__), "golden", "rows")
IMAGE_ROWS = sorted(glob.glob(os.path.join(ROWS, "image_rows_*.npz")))


def _offsets(types):
    w = np.where(np.asarray(types) == 0, 6, 3)
    return (13 + np.concatenate([[0], np.cumsum(w)[:-1]])).astype(np.int32)


def test_files_present():
    assert os.path.exists(os.path.join(ROWS, "state_rows_L8.npz")) and len(IMAGE_ROWS) == 2


def test_oracle_state_rows(oracle_lib):
    g = np.load(os.path.join(ROWS, "state_rows_L8.npz"))
    cam = default_camera()
    xp, Pp = oracle_lib.ekf_prediction(g["x"], g["P"], 1.0, 0.007, 0.007)
    assert np.array_equal(xp, g["pred_x"]) and np.array_equal(Pp, g["pred_P"])
    xd, Pd = oracle_lib.map_delete_feature(g["types"], g["x"], g["P"], int(g["del_feature"]))
    assert np.array_equal(xd, g["del_x"]) and np.array_equal(Pd, g["del_P"])
    conv, xc, Pc = oracle_lib.map_convert(g["types"], g["x"], g["conv_P_in"], float(g["conv_threshold"]))
    assert conv == int(g["conv_index"]) and np.array_equal(xc, g["conv_x"]) and np.array_equal(Pc, g["conv_P"])
    xa, Pa = oracle_lib.map_add_feature(cam, 1.0, g["x"], g["P"], g["add_uvd"], 1.0, 1.0)
    assert np.array_equal(xa, g["add_x"]) and np.array_equal(Pa, g["add_P"])


@pytest.mark.parametrize("path", IMAGE_ROWS, ids=[os.path.basename(p)[:-4] for p in IMAGE_ROWS])
def test_oracle_image_rows(oracle_lib, path):
    g = np.load(path)
    cam = default_camera()
    compat = int(g["compat"])
    offs = _offsets(g["types"])
    p, st, pm = oracle_lib.pred_patches(cam, compat, g["types"], offs, g["x_pred"], g["h"], g["visible"], g["uv_f"], g["R_f"],
                                        g["r_f"], g["patch_f"].astype(np.float64))
    assert np.array_equal(st, g["patch_status"]) and np.array_equal(p, g["patches"].astype(np.float64))
    z, ic, corr, mm = oracle_lib.matching(cam, g["image"], g["match_patches"].astype(np.float64), g["h"], g["visible"], g["S"])
    assert np.array_equal(ic, g["ic"]) and np.array_equal(z, g["z"]) and np.array_equal(corr, g["corr"])


@pytest.mark.gpu
def test_device_state_rows():
    from ransac_slam_amd import api
    g = np.load(os.path.join(ROWS, "state_rows_L8.npz"))

    def close(a, b, rel=1e-12):
        return np.max(np.abs(a - b)) <= rel * max(np.abs(b).max(), 1e-300)

    ctx = api.RslamHip(default_config())
    ctx.set_posterior(g["types"], g["x"], g["P"])
    ctx.ekf_prediction(1.0, 0.007, 0.007)
    xp, Pp = ctx.fetch_prior()
    assert np.allclose(xp, g["pred_x"], rtol=1e-13, atol=1e-15) and close(Pp, g["pred_P"])
    ctx.set_posterior(g["types"], g["x"], g["P"])
    ctx.map_delete_feature(int(g["del_feature"]))
    xd, Pd = ctx.fetch_posterior()
    assert np.array_equal(xd, g["del_x"]) and np.array_equal(Pd, g["del_P"])
    ctx.set_posterior(g["types"], g["x"], g["conv_P_in"])
    conv, _ = ctx.map_convert(float(g["conv_threshold"]))
    xc, Pc = ctx.fetch_posterior()
    assert conv == int(g["conv_index"]) and np.allclose(xc, g["conv_x"], rtol=1e-13, atol=1e-15) and close(Pc, g["conv_P"])
    ctx.set_posterior(g["types"], g["x"], g["P"])
    ctx.map_add_feature(g["add_uvd"], 1.0, 1.0)
    xa, Pa = ctx.fetch_posterior()
    assert np.allclose(xa, g["add_x"], rtol=1e-13, atol=1e-15) and close(Pa, g["add_P"])
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("path", IMAGE_ROWS, ids=[os.path.basename(p)[:-4] for p in IMAGE_ROWS])
def test_device_image_rows(path):
    from ransac_slam_amd import api
    g = np.load(path)
    compat = int(g["compat"])
    ctx = api.RslamHip(default_config(compat=compat))
    h, vis, S = ctx.predict(g["types"], g["x_pred"], g["P_pred"])
    assert np.array_equal(vis, g["visible"])
    v = vis.astype(bool)
    assert np.allclose(h[v], g["h"][v], atol=1e-9)
    ctx.set_feature_records(g["uv_f"], g["R_f"], g["r_f"], g["patch_f"].astype(np.float64))
    p, st = ctx.predict_patches()
    assert np.array_equal(st, g["patch_status"])
    safe = g["patch_margins"] > 1e-11           # see tests/test_gpu_patch.py: away from the float32-cast / truncation flips
    assert safe.sum() >= len(safe) - 1 and np.array_equal(p[safe], g["patches"].astype(np.float64)[safe])
    assert min(g["match_margins"]) > 1e-7
    z, ic, corr = ctx.match(g["image"], g["match_patches"].astype(np.float64))
    assert np.array_equal(ic, g["ic"]) and np.array_equal(z[g["ic"] == 1], g["z"][g["ic"] == 1])
    assert np.allclose(corr, g["corr"], rtol=0, atol=1e-9)
    ctx.close()
