"""Several consecutive frames with every widened row in the loop, the filter state never leaving the
device: predict -> pred_patch_fc -> matching -> RANSAC + updates -> map management (delete / convert /
insert) -> ekf_prediction -> next frame; the oracle runs the same sequence on host arrays."""
import numpy as np
import pytest

from ransac_slam_amd import default_camera, default_config, synth
from ransac_slam_amd.synth import make_frame, make_feature_records

pytestmark = pytest.mark.gpu


def _offsets(types):
    w = np.where(np.asarray(types) == 0, 6, 3)
    return (13 + np.concatenate([[0], np.cumsum(w)[:-1]])).astype(np.int32)


def _image_with(cam, rng, patches, status, h, shift):
    image = rng.integers(0, 256, (cam.nRows, cam.nCols)).astype(np.uint8)
    for i in np.argsort(-h[:, 0]):
        if status[i] != 1:
            continue
        x = int(min(max(round(h[i, 0]) + shift[0], 7), cam.nCols - 8)); y = int(min(max(round(h[i, 1]) + shift[1], 7), cam.nRows - 8))
        image[y - 6:y + 7, x - 6:x + 7] = np.clip(np.round(patches[i]), 0, 255).astype(np.uint8)
    return image


@pytest.mark.parametrize("compat", [0, 1])
def test_four_frames_resident(oracle_lib, compat):
    from ransac_slam_amd import api
    cam = default_camera()
    fr = make_frame(L=50, H=200, seed=1601)
    cfg = default_config(compat=compat, adaptive=1)
    o = oracle_lib.Oracle(cfg, structure=1)
    g = api.RslamHip(cfg)
    rng = np.random.default_rng(21)
    # host-side (oracle) copies of everything
    types = fr.types.copy()
    xp, Pp = fr.x_pred.copy(), np.asarray(fr.P_pred).copy()
    uv_f, R_f, r_f, patch_f = make_feature_records(cam, fr, seed=4)
    g.set_feature_records(uv_f, R_f, r_f, patch_f)
    first = True
    for k in range(4):
        offs = _offsets(types)
        # ---- segment 1
        h0, v0, S0 = o.predict(types, xp, Pp)
        if first:
            h1, v1, S1 = g.predict(types, xp, Pp)          # the only covariance upload of the sequence
            first = False
        else:
            h1, v1, S1 = g.predict_resident()
        assert np.array_equal(v0, v1)
        vb = v0.astype(bool)
        assert np.allclose(h1[vb], h0[vb], atol=1e-7) and np.allclose(S1[vb], S0[vb], rtol=1e-6)
        # ---- patch prediction and NCC search (device patches never come back)
        p0, st0, m0 = oracle_lib.pred_patches(cam, compat, types, offs, xp, h0, v0, uv_f, R_f, r_f, patch_f)
        image = _image_with(cam, rng, p0, st0, h0, (1, -1) if k % 2 == 0 else (-2, 1))
        z0, ic0, c0, mm = oracle_lib.matching(cam, image, p0, h0, v0, S0)
        assert min(mm) > 1e-6          # (a flipped remap tap would move a correlation by ~1e-3, far from deciding a match)
        _, st1 = g.predict_patches(fetch=False)
        assert np.array_equal(st1, st0)
        z1, ic1, c1 = g.match(image)
        assert np.array_equal(ic1, ic0) and np.array_equal(z1[ic0 == 1], z0[ic0 == 1])
        assert ic0.sum() >= 10
        # ---- segment 2
        draws = rng.random(200)
        r0 = o.ransac_update(z0, ic0, draws)
        assert min(o.margins()) > 1e-8
        r1 = g.ransac_update(z1, ic1, draws, want_P=False)
        for key in ("best_hyp", "best_support", "hyps_evaluated"):
            assert r1[key] == r0[key]
        assert np.array_equal(r1["li"], r0["li"]) and np.array_equal(r1["hi"], r0["hi"])
        assert np.max(np.abs(r1["x_new"] - r0["x_new"])) <= 1e-8 * max(1.0, np.abs(r0["x_new"]).max())
        x0, P0 = r0["x_new"], r0["P_new"]
        # ---- map management: delete one feature, convert when something is linear enough, insert one
        victim = (7 * k + 3) % len(types)
        x0, P0 = oracle_lib.map_delete_feature(types, x0, P0, victim)
        types = np.delete(types, victim)
        uv_f, R_f, r_f, patch_f = (np.delete(a, victim, axis=0) for a in (uv_f, R_f, r_f, patch_f))
        g.map_delete_feature(victim)
        if compat == 0:          # compat keeps Q2: a frame with unequal Cartesian / inverse-depth match counts asserts
            lin = np.array([oracle_lib.linearity_index(x0, P0, int(of)) if t == 0 else 9e9
                            for t, of in zip(types, _offsets(types))])
            srt = np.sort(lin)
            thr = float(0.5 * (srt[0] + srt[1]))
            conv0, x0, P0 = oracle_lib.map_convert(types, x0, P0, thr)
            conv1, _ = g.map_convert(thr)
            assert conv1 == conv0 and conv0 >= 0
            types[conv0] = 1
        uvd = np.array([40.0 + 60.0 * k, 60.0 + 30.0 * k])
        x0, P0 = oracle_lib.map_add_feature(cam, cfg.sigma_z, x0, P0, uvd, 1.0, 1.0)
        new_patch = rng.integers(0, 256, (41, 41)).astype(float)
        Rn, rn = synth.q2r(x0[3:7]), x0[:3].copy()
        types = np.append(types, 0).astype(np.uint8)
        uv_f = np.vstack([uv_f, uvd]); R_f = np.concatenate([R_f, Rn[None]]); r_f = np.vstack([r_f, rn])
        patch_f = np.concatenate([patch_f, new_patch[None]])
        g.map_add_feature(uvd, 1.0, 1.0)
        xg = g.fetch_posterior()[0]                          # the host reads x_k_k (n doubles) for the new record
        g.append_feature_record(uvd, synth.q2r(xg[3:7])[None], xg[:3][None], new_patch[None])
        R_f[-1] = synth.q2r(xg[3:7]); r_f[-1] = xg[:3]      # same record on both sides
        n1, t1, _ = g.get_layout()
        assert n1 == len(x0) and np.array_equal(t1, types)
        # ---- prediction
        xp, Pp = oracle_lib.ekf_prediction(x0, P0, 1.0, 0.007, 0.007)
        g.ekf_prediction(1.0, 0.007, 0.007)
    xq, Pq = g.fetch_prior()
    assert np.max(np.abs(xq - xp)) <= 1e-7 * max(1.0, np.abs(xp).max())
    assert np.max(np.abs(Pq - Pp)) <= 1e-7 * np.abs(Pp).max()
    g.close()
