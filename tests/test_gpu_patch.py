"""Tracking::pred_patch_fc through the C ABI against the oracle, the device-side
feature store behind it, and the whole tracking front end on the device:
predict -> predict_patches -> match -> ransac_update."""
import numpy as np
import pytest

from ransac_slam_amd import default_camera, default_config
from ransac_slam_amd.synth import make_frame, make_feature_records

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from ransac_slam_amd import api
    api.lib()
    return api


@pytest.mark.parametrize("compat", [1, 0])
@pytest.mark.parametrize("case", [dict(L=10, H=4, seed=1501), dict(L=60, H=4, seed=1502, frac_cartesian=0.3),
                                  dict(L=300, H=4, seed=1503)], ids=["L10", "L60mixed", "L300"])
def test_patches_match_oracle(hip, oracle_lib, case, compat):
    cam = default_camera()
    fr = make_frame(**case)
    cfg = default_config(compat=compat)
    g = hip.RslamHip(cfg)
    h1, v1, _ = g.predict(fr.types, fr.x_pred, fr.P_pred)
    uv_f, R_f, r_f, patch_f = make_feature_records(cam, fr, seed=case["seed"] + 7)
    g.set_feature_records(uv_f, R_f, r_f, patch_f)
    p1, st1 = g.predict_patches()
    # the oracle warps about the device's own h so that only the warp itself is compared
    p0, st0, m = oracle_lib.pred_patches(cam, compat, fr.types, fr.offsets, fr.x_pred, h1, v1, uv_f, R_f, r_f, patch_f)
    assert np.array_equal(st1, st0)
    assert (st0 == 1).sum() >= 0.5 * v1.sum()
    # Two correct double-precision evaluations of the geometry can only end in different patches where a map
    # coordinate sits within their rounding noise (~1e-13 px) of a flip of its float32 cast or of the cv::Range
    # truncation; the oracle reports that distance per feature.  Everywhere else: float32 taps and weights in
    # the same order, bit-identical.
    safe = m > 1e-11
    assert safe.sum() >= 0.97 * fr.L
    assert np.array_equal(p1[safe], p0[safe])
    assert np.abs(p1[~safe] - p0[~safe]).max(initial=0.0) <= 64.0
    g.close()


def test_store_follows_map_edits(hip, oracle_lib):
    cam = default_camera()
    fr = make_frame(L=12, H=4, seed=1511)
    cfg = default_config(compat=0)
    g = hip.RslamHip(cfg)
    uv_f, R_f, r_f, patch_f = make_feature_records(cam, fr, seed=3)
    g.set_posterior(fr.types, fr.x_pred, fr.P_pred)
    g.set_feature_records(uv_f, R_f, r_f, patch_f)
    g.map_delete_feature(4)
    g.map_add_feature([140.0, 100.0])
    xkk, _ = g.fetch_posterior()                       # the record of the new feature holds the pose it was seen from
    g.ekf_prediction(1.0, 0.007, 0.007)
    h1, v1, _ = g.predict_resident()
    with pytest.raises(hip.RslamError) as e:
        g.predict_patches()                            # the new feature has no record yet
    assert e.value.code == -4
    rng = np.random.default_rng(5)
    new_patch = rng.integers(0, 256, (41, 41)).astype(float)
    from ransac_slam_amd import synth
    xk, _ = g.fetch_prior()
    g.append_feature_record([140.0, 100.0], synth.q2r(xkk[3:7])[None], xkk[:3][None], new_patch[None])
    p1, st1 = g.predict_patches()
    keep = np.r_[0:4, 5:12]
    n, types, offs = g.get_layout()
    uv2 = np.vstack([uv_f[keep], [140.0, 100.0]]); R2 = np.concatenate([R_f[keep], synth.q2r(xkk[3:7])[None]])
    r2 = np.vstack([r_f[keep], xkk[:3]]); pf2 = np.concatenate([patch_f[keep], new_patch[None]])
    p0, st0, m = oracle_lib.pred_patches(cam, 0, types, offs, xk, h1, v1, uv2, R2, r2, pf2)
    assert m.min() > 1e-11 and np.array_equal(st1, st0) and np.array_equal(p1, p0)
    assert st0[-1] == 1                                # the freshly inserted feature is warped from its own record
    g.close()


@pytest.mark.parametrize("compat", [1, 0])
def test_front_end_on_device(hip, oracle_lib, compat):
    """predict -> predict_patches -> match (patches never leave the device) -> update, against the oracle chain"""
    cam = default_camera()
    fr = make_frame(L=80, H=300, seed=1521)
    cfg = default_config(compat=compat, adaptive=1)
    o = oracle_lib.Oracle(cfg, structure=1)
    g = hip.RslamHip(cfg)
    h0, v0, S0 = o.predict(fr.types, fr.x_pred, fr.P_pred)
    uv_f, R_f, r_f, patch_f = make_feature_records(cam, fr, seed=9)
    p0, st0, m0 = oracle_lib.pred_patches(cam, compat, fr.types, fr.offsets, fr.x_pred, h0, v0, uv_f, R_f, r_f, patch_f)
    # an image that contains every predicted patch near its prediction
    rng = np.random.default_rng(11)
    image = rng.integers(0, 256, (cam.nRows, cam.nCols)).astype(np.uint8)
    order = np.argsort(-h0[:, 0] * v0)                 # paste in a fixed order; later pastes may overwrite earlier ones
    for i in order:
        if st0[i] != 1:
            continue
        x = int(min(max(round(h0[i, 0]) + 1, 7), cam.nCols - 8)); y = int(min(max(round(h0[i, 1]) - 1, 7), cam.nRows - 8))
        image[y - 6:y + 7, x - 6:x + 7] = np.clip(np.round(p0[i]), 0, 255).astype(np.uint8)
    z0, ic0, c0, m = oracle_lib.matching(cam, image, p0, h0, v0, S0)
    assert m0.min() > 1e-11 and min(m) > 1e-7
    assert ic0.sum() >= 20
    r0 = o.ransac_update(z0, ic0, fr.draws)
    assert min(o.margins()) > 1e-8
    g.predict(fr.types, fr.x_pred, fr.P_pred)
    g.set_feature_records(uv_f, R_f, r_f, patch_f)
    _, st1 = g.predict_patches(fetch=False)
    assert np.array_equal(st1, st0)
    z1, ic1, c1 = g.match(image)
    assert np.array_equal(ic1, ic0) and np.array_equal(z1[ic0 == 1], z0[ic0 == 1])
    r1 = g.ransac_update(z1, ic1, fr.draws)
    for k in ("best_hyp", "best_support", "hyps_evaluated"):
        assert r1[k] == r0[k]
    assert np.array_equal(r1["li"], r0["li"]) and np.array_equal(r1["hi"], r0["hi"])
    assert np.max(np.abs(r1["x_new"] - r0["x_new"])) <= 1e-9 * max(1.0, np.abs(r0["x_new"]).max())
    assert np.max(np.abs(r1["P_new"] - r0["P_new"])) <= 1e-9 * np.abs(r0["P_new"]).max()
    g.close()
