"""Tracking::matching (NCC search) through the C ABI against the oracle, on the
prediction that rslam_predict left on the device, and a whole frame predict -> match -> update."""
import numpy as np
import pytest

from ransac_slam_amd import default_camera, default_config
from ransac_slam_amd.synth import make_frame, make_match_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from ransac_slam_amd import api
    api.lib()
    return api


@pytest.mark.parametrize("case", [dict(L=12, H=4, seed=1301), dict(L=60, H=4, seed=1302, frac_cartesian=0.3),
                                  dict(L=300, H=4, seed=1303)], ids=["L12", "L60mixed", "L300"])
def test_match_matches_oracle(hip, oracle_lib, case):
    cam = default_camera()
    fr = make_frame(**case)
    cfg = default_config()
    o = oracle_lib.Oracle(cfg, structure=1)
    h0, v0, S0 = o.predict(fr.types, fr.x_pred, fr.P_pred)
    image, patches, truth = make_match_inputs(cam, h0, v0, seed=case["seed"])
    g = hip.RslamHip(cfg)
    h1, v1, S1 = g.predict(fr.types, fr.x_pred, fr.P_pred)
    assert np.array_equal(v0, v1)
    # the oracle searches on the device's own h / S so that only the search itself is compared
    z0, ic0, c0, m = oracle_lib.matching(cam, image, patches, np.nan_to_num(h1), v1, np.nan_to_num(S1))
    assert m[0] > 1e-6 and m[1] > 1e-9 and m[2] > 1e-9          # no decision is a close call
    z1, ic1, c1 = g.match(image, patches)
    assert np.array_equal(ic1, ic0)
    assert ic0.sum() >= 0.6 * v0.sum()
    sel = ic0.astype(bool)
    assert np.array_equal(z1[sel], z0[sel])
    assert np.allclose(c1, c0, rtol=0, atol=1e-12)
    planted = sel & (truth[:, 0] >= 0)
    assert np.array_equal(z1[planted], truth[planted])
    g.close()


def test_match_gates_and_ties(hip, oracle_lib):
    """ellipse too large, no candidate inside the image margin, exact ties (first maximum wins)"""
    cam = default_camera()
    fr = make_frame(L=6, H=4, seed=1311)
    cfg = default_config()
    g = hip.RslamHip(cfg)
    # inflate the covariance so that some S_i exceed the 100 px^2 eigenvalue limit
    P = np.asarray(fr.P_pred).copy()
    P[:3, :3] *= 400.0
    h1, v1, S1 = g.predict(fr.types, fr.x_pred, P)
    lmax = np.array([np.linalg.eigvalsh(S1[i].reshape(2, 2)).max() if v1[i] else 0 for i in range(fr.L)])
    rng = np.random.default_rng(8)
    tile = rng.integers(0, 256, (8, 8), dtype=np.uint8)
    image = np.tile(tile, (cam.nRows // 8, cam.nCols // 8))
    patches = np.zeros((fr.L, 13, 13))
    for f in range(fr.L):
        if v1[f]:
            x = int(min(max(round(h1[f, 0]), 7), cam.nCols - 8)); y = int(min(max(round(h1[f, 1]), 7), cam.nRows - 8))
            patches[f] = image[y - 6:y + 7, x - 6:x + 7]
    z0, ic0, c0, m = oracle_lib.matching(cam, image, patches, np.nan_to_num(h1), v1, np.nan_to_num(S1))
    z1, ic1, c1 = g.match(image, patches)
    assert (lmax >= 100).any() and (ic0[lmax >= 100] == 0).all()
    assert np.array_equal(ic1, ic0) and np.array_equal(z1[ic0 == 1], z0[ic0 == 1])
    assert np.allclose(c1, c0, rtol=0, atol=1e-12)
    g.close()


def test_call_order(hip):
    g = hip.RslamHip(default_config())
    with pytest.raises(hip.RslamError) as e:
        g.L = 1
        g.match(np.zeros((240, 320), np.uint8), np.zeros((1, 13, 13)))
    assert e.value.code == -4
    g.close()


@pytest.mark.parametrize("compat", [1, 0])
def test_frame_with_device_matching(hip, oracle_lib, compat):
    """segment 1 -> NCC search -> segment 2, the matches never pass through host logic"""
    cam = default_camera()
    fr = make_frame(L=80, H=300, seed=1321)
    cfg = default_config(compat=compat, adaptive=1)
    o = oracle_lib.Oracle(cfg, structure=1)
    g = hip.RslamHip(cfg)
    h0, v0, S0 = o.predict(fr.types, fr.x_pred, fr.P_pred)
    image, patches, _ = make_match_inputs(cam, h0, v0, seed=77, offset_px=1.5)
    z0, ic0, c0, m = oracle_lib.matching(cam, image, patches, h0, v0, S0)
    assert min(m) > 1e-7
    r0 = o.ransac_update(z0, ic0, fr.draws)
    assert min(o.margins()) > 1e-8
    g.predict(fr.types, fr.x_pred, fr.P_pred)
    z1, ic1, _ = g.match(image, patches)
    assert np.array_equal(ic1, ic0) and np.array_equal(z1[ic0 == 1], z0[ic0 == 1])
    r1 = g.ransac_update(z1, ic1, fr.draws)
    for k in ("best_hyp", "best_support", "hyps_evaluated"):
        assert r1[k] == r0[k]
    assert np.array_equal(r1["li"], r0["li"]) and np.array_equal(r1["hi"], r0["hi"])
    assert np.max(np.abs(r1["x_new"] - r0["x_new"])) <= 1e-9 * max(1.0, np.abs(r0["x_new"]).max())
    assert np.max(np.abs(r1["P_new"] - r0["P_new"])) <= 1e-9 * np.abs(r0["P_new"]).max()
    g.close()
