"""ExtendKF::ekf_prediction on the device against the oracle, and a two-frame
sequence that never uploads the covariance for the second frame."""
import numpy as np
import pytest

from ransac_slam_amd import default_config
from ransac_slam_amd.synth import make_frame

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from ransac_slam_amd import api
    api.lib()
    return api


@pytest.mark.parametrize("case", [dict(L=5, H=4, seed=601), dict(L=40, H=4, seed=602, frac_cartesian=0.3),
                                  dict(L=300, H=4, seed=603)], ids=["L5", "L40mixed", "L300"])
def test_ekf_prediction_matches_oracle(hip, oracle_lib, case):
    fr = make_frame(**case)
    x, P = fr.x_pred, np.asarray(fr.P_pred)
    x = x.copy(); x[10:13] += [0.02, -0.01, 0.03]; x[7:10] += [0.1, 0.0, -0.2]
    xp0, Pp0 = oracle_lib.ekf_prediction(x, P, 1.0, 0.007, 0.007)
    g = hip.RslamHip(default_config())
    g.set_posterior(fr.types, x, P)
    g.ekf_prediction(1.0, 0.007, 0.007)
    xp1, Pp1 = g.fetch_prior()
    assert np.array_equal(xp1[13:], x[13:]) and np.array_equal(Pp1[13:, 13:], P[13:, 13:])
    assert np.allclose(xp1, xp0, rtol=1e-13, atol=1e-15)
    assert np.max(np.abs(Pp1 - Pp0)) <= 1e-13 * np.abs(Pp0).max()
    g.close()


@pytest.mark.parametrize("compat", [1, 0])
def test_two_frames_resident(hip, oracle_lib, compat):
    """frame k: upload + update; frame k+1: ekf_prediction on the resident posterior,
    predict with no upload, update -- against the oracle running the same sequence."""
    fr = make_frame(L=80, H=300, seed=611)
    cfg = default_config(compat=compat, adaptive=1)
    o = oracle_lib.Oracle(cfg, structure=1)
    g = hip.RslamHip(cfg)
    h0, v0, _ = o.predict(fr.types, fr.x_pred, fr.P_pred)
    ic = (fr.ic & v0).astype(np.uint8)
    r0 = o.ransac_update(fr.z, ic, fr.draws)
    g.predict(fr.types, fr.x_pred, fr.P_pred)
    r1 = g.ransac_update(fr.z, ic, fr.draws, want_P=False)
    assert np.array_equal(r1["li"], r0["li"]) and np.array_equal(r1["hi"], r0["hi"])
    # next frame
    xp0, Pp0 = oracle_lib.ekf_prediction(r0["x_new"], r0["P_new"], 1.0, 0.007, 0.007)
    h0b, v0b, S0b = o.predict(fr.types, xp0, Pp0)
    g.ekf_prediction(1.0, 0.007, 0.007)
    h1b, v1b, S1b = g.predict_resident()
    assert np.array_equal(v0b, v1b)
    vb = v0b.astype(bool)
    assert np.allclose(h1b[vb], h0b[vb], atol=1e-8) and np.allclose(S1b[vb], S0b[vb], rtol=1e-8)
    rng = np.random.default_rng(5)
    z2 = h0b + rng.normal(0, 0.4, h0b.shape)
    z2[~vb] = 0.0
    ic2 = (ic & v0b).astype(np.uint8)
    draws2 = rng.random(300)
    r0b = o.ransac_update(z2, ic2, draws2)
    r1b = g.ransac_update(z2, ic2, draws2)
    assert o.margins()[0] > 1e-8 and o.margins()[1] > 1e-8
    for k in ("best_hyp", "best_support", "hyps_evaluated"):
        assert r1b[k] == r0b[k]
    assert np.array_equal(r1b["li"], r0b["li"]) and np.array_equal(r1b["hi"], r0b["hi"])
    assert np.max(np.abs(r1b["x_new"] - r0b["x_new"])) <= 1e-9 * max(1.0, np.abs(r0b["x_new"]).max())
    assert np.max(np.abs(r1b["P_new"] - r0b["P_new"])) <= 1e-9 * np.abs(r0b["P_new"]).max()
    g.close()


def test_prediction_call_order(hip):
    g = hip.RslamHip(default_config())
    with pytest.raises(hip.RslamError) as e:
        g.ekf_prediction()
    assert e.value.code == -4            # no posterior resident yet
    g.close()
