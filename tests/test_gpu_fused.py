"""The x / covariance update inside the persistent factor sweep (tile workers, kernels.hip), its bounded waits, and
the host's handling of frames nobody has looked at yet.  Replaces ExtendKF::update's tail (ExtendKF.cpp:606-634)
on a different schedule, so every route is held against the oracle and against the stand-alone rank update.
"""
import time

import numpy as np
import pytest

from ransac_slam_amd import default_config
from ransac_slam_amd.synth import make_frame, remeasure

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (the product path has no CPU fallback)")
    from ransac_slam_amd import api
    return api


def close_x(a, b, tol=1e-9):
    return np.max(np.abs(a - b)) <= tol * max(1.0, float(np.max(np.abs(b))))


def close_P(a, b, tol=1e-9):
    """norm-wise and scale-aware: |dP_ij| <= tol * sqrt(P_ii P_jj) (the diagonal of p_k_k spans 1e-6 .. 0.25 rho^2)"""
    d = np.sqrt(np.abs(np.diag(b)))
    return (np.max(np.abs(a - b)) <= tol * float(np.max(np.abs(b)))
            and np.all(np.abs(a - b) <= tol * np.outer(d, d) + 1e-300))


def oracle_frame(oracle_lib, fr, cfg, z=None, draws=None):
    o = oracle_lib.Oracle(cfg, structure=1)
    _, v0, _ = o.predict(fr.types, fr.x_pred, fr.P_pred)
    ic = (fr.ic & v0).astype(np.uint8)
    r0 = o.ransac_update(fr.z if z is None else z, ic, fr.draws if draws is None else draws)
    assert min(o.margins()) > 1e-9
    return ic, r0


@pytest.mark.parametrize("compat,L,H,seed", [(1, 90, 120, 21), (0, 24, 60, 12), (0, 40, 80, 13), (0, 90, 120, 13), (1, 300, 200, 2),
                                             (0, 300, 200, 2)])
def test_fused_update_equals_standalone_rank_update(hip_dbg, oracle_lib, compat, L, H, seed):
    """Same frame with the update inside the sweep launch (default) and with the rank update as a launch of its own
    (RSLAM_SWEEP_EXP bit 7): both against the oracle, and against each other to rounding.  The frames cover the
    register-only route (r <= 4), the in-LDS single-block route, and systems of several diagonal blocks."""
    fr = make_frame(L=L, H=H, seed=seed)
    cfg = default_config(compat=compat, adaptive=0)
    ic, r0 = oracle_frame(oracle_lib, fr, cfg)
    out = {}
    for mask in (0, 128):
        hip_dbg.set_sweep_exp(mask)
        try:
            g = hip_dbg.RslamHip(cfg)
            g.predict(fr.types, fr.x_pred, fr.P_pred)
            out[mask] = g.ransac_update(fr.z, ic, fr.draws)
            g.close()
        finally:
            hip_dbg.set_sweep_exp(-1)
        r1 = out[mask]
        assert np.array_equal(r1["li"], r0["li"]) and np.array_equal(r1["hi"], r0["hi"])
        assert close_x(r1["x_new"], r0["x_new"]) and close_P(r1["P_new"], r0["P_new"]), mask
        Dsym = r1["P_new"] - r1["P_new"].T          # exactly symmetric, except the 4 x 4 block (J P44) J^T, symmetric to
        assert np.abs(Dsym[3:7, 3:7]).max() <= 1e-14 * np.abs(r1["P_new"][3:7, 3:7]).max()     # rounding (an ulp of the block BEFORE the projection; seen: 4e-15)
        Dsym[3:7, 3:7] = 0                                                                   # (ExtendKF.cpp:632)
        assert not Dsym.any()
        assert abs(np.linalg.norm(r1["x_new"][3:7]) - 1.0) < 1e-14
    assert close_x(out[0]["x_new"], out[128]["x_new"], 1e-12) and close_P(out[0]["P_new"], out[128]["P_new"], 1e-11)


@pytest.mark.parametrize("compat,L,H,seed", [(1, 300, 200, 2), (0, 300, 200, 2), (0, 90, 120, 13), (1, 150, 100, 5)])
def test_xcd_tile_map_equals_round_robin_assignment_bitwise(hip_dbg, compat, L, H, seed):
    """Which tile worker computes a tile pair of P - Y Y^T does not change a bit of it: the XCD-aware assignment of round 6
    (kernels.hip WkMap: every XCD's workers own a contiguous run of the region-major tile sequence, so that an XCD's L2 fetches
    ~16 of the Y row panels instead of all of them) against the round-3 assignment (tile widx + j W; RSLAM_SWEEP_EXP bit 12)
    -- multi-block sweeps (C3 size, both arithmetic modes), a mid-size map, and a system of one diagonal block whose strips
    join the workers late."""
    fr = make_frame(L=L, H=H, seed=seed)
    cfg = default_config(compat=compat, adaptive=0)
    out = {}
    for mask in (0, 4096):
        hip_dbg.set_sweep_exp(mask)
        try:
            g = hip_dbg.RslamHip(cfg)
            _, v0, _ = g.predict(fr.types, fr.x_pred, fr.P_pred)
            ic = (fr.ic & v0).astype(np.uint8)
            g.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, ic, fr.draws)
            for use_graph in (False, True, True):
                g.step_frame(use_graph)
            g.sync()
            assert g.update_mode() == 2 and g.counters()["sweep_reruns"] == 0
            out[mask] = g.fetch_results()
            g.close()
        finally:
            hip_dbg.set_sweep_exp(-1)
    a, b = out[0], out[4096]
    assert int(a["li"].sum()) + int(a["hi"].sum()) > 0
    assert np.array_equal(a["li"], b["li"]) and np.array_equal(a["hi"], b["hi"])
    assert np.array_equal(a["x_new"], b["x_new"]) and np.array_equal(a["P_new"], b["P_new"])


def test_alternating_frames_on_one_context(hip, oracle_lib):
    """The tile workers read Y while the strips of the same launch are still writing later blocks of it, through
    write-through stores and sc1 transfers.  A stale line would carry the PREVIOUS frame's Y: two different
    measurement sets alternate on one context (same buffers, hipGraph replay), each posterior against the oracle."""
    fr = make_frame(L=300, H=200, seed=2)
    cfg = default_config(compat=0, adaptive=0)
    sets = [(fr.z, fr.draws)] + [remeasure(fr, 500 + k, frac_outlier=0.1 + 0.3 * k, H=200)[::2] for k in range(2)]
    g = hip.RslamHip(cfg)
    ic, _ = oracle_frame(oracle_lib, fr, cfg)
    ref = [oracle_frame(oracle_lib, fr, cfg, z=z, draws=d)[1] for (z, d) in sets]
    assert not np.allclose(ref[0]["P_new"], ref[1]["P_new"], rtol=1e-6, atol=0)
    g.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, ic, fr.draws)
    for rep in range(3):
        for k, (z, d) in enumerate(sets):
            g.load_measurements(z, ic, d)
            g.step_frame(True)
            g.step_frame(True)                       # back to back: the second replay starts while nothing was synchronised
            g.sync()
            r1 = g.fetch_results()
            assert np.array_equal(r1["li"], ref[k]["li"]) and np.array_equal(r1["hi"], ref[k]["hi"]), (rep, k)
            assert close_x(r1["x_new"], ref[k]["x_new"]) and close_P(r1["P_new"], ref[k]["P_new"]), (rep, k)
    assert g.counters()["sweep_reruns"] == 0
    g.close()


def test_tile_workers_timeout_falls_back(hip_dbg, oracle_lib):
    """Fault injection (RSLAM_SWEEP_EXP bit 5): the strips never announce their Y blocks.  Every tile worker and every
    x-update strip must leave through its bounded wait WITHOUT writing the posterior, the host must notice, re-run the
    update stage on the launch-per-step sweep + stand-alone rank update, and return the right answer."""
    fr = make_frame(L=90, H=120, seed=321)
    cfg = default_config(compat=0, adaptive=1)
    ic, r0 = oracle_frame(oracle_lib, fr, cfg)
    hip_dbg.set_sweep_exp(32)
    try:
        g = hip_dbg.RslamHip(cfg)
        g.predict(fr.types, fr.x_pred, fr.P_pred)
        r1 = g.ransac_update(fr.z, ic, fr.draws)
    finally:
        hip_dbg.set_sweep_exp(-1)
    assert g.last_raw_status() in (-37, -38)
    assert g.counters()["sweep_reruns"] >= 1
    assert np.array_equal(r1["li"], r0["li"]) and np.array_equal(r1["hi"], r0["hi"])
    assert close_x(r1["x_new"], r0["x_new"]) and close_P(r1["P_new"], r0["P_new"])
    # ... and the diagnosis names the withheld wait (rslam_last_wait_detail / _polls): a tile worker waiting for the FIRST Y
    # block (code 37, needed value 1) or an x-update strip waiting for u^T (38), in a workgroup behind the chain's, and it ran
    # out the way a withheld hand-over does -- both bounds passed: >= 768 polls of its own AND >= 1 ms of wall clock
    d = g.last_wait_detail()
    assert d is not None and d["code"] in (37, 38), d
    assert d["workgroup"] >= 1
    if d["code"] == 37:
        assert d["needed"] == 1, d
    assert d["polls"] >= 768 and d["elapsed_us"] >= 1000, d
    g.close()
    # What a timeout costs: the waits are bounded in TIME (1 ms of the device's wall clock, kernels.hip SW_WAIT_TICKS), so a
    # frame whose workgroups are not all resident is back -- timed out, noticed by the host, re-run on the launch-per-step
    # route -- within ~3 ms of a healthy frame (until round 3 the bound was a spin count worth ~30 ms: 140 frames).
    def frame_ms(mask):
        hip_dbg.set_sweep_exp(mask)
        try:
            gg = hip_dbg.RslamHip(cfg)
            gg.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, ic, fr.draws)
            gg.step_frame(False); gg.sync()                # (first launch of the context: module load, buffers)
            if mask:
                assert gg.counters()["sweep_reruns"] >= 1
            ts = []
            for _ in range(3):
                gg2 = hip_dbg.RslamHip(cfg)                # a fresh context: no fallback state carried over
                gg2.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, ic, fr.draws)
                t0 = time.perf_counter()
                gg2.step_frame(False); gg2.sync()
                ts.append((time.perf_counter() - t0) * 1e3)
                if mask:
                    assert gg2.counters()["sweep_reruns"] == 1 and gg2.last_raw_status() in (-37, -38)
                gg2.close()
            gg.close()
            return min(ts)
        finally:
            hip_dbg.set_sweep_exp(-1)
    t_ok, t_fault = frame_ms(0), frame_ms(32)
    assert t_fault - t_ok <= 3.0, (t_ok, t_fault)


def test_late_hand_over_survives_when_the_waiter_hardly_ran(hip_dbg, oracle_lib):
    """The other side of the doubly bounded waits (kernels.hip SwDeadline: 1 ms of wall clock AND 768 polls of the waiting wave):
    a hand-over that is merely LATE.  RSLAM_SWEEP_EXP bit 10 makes the P H^T strips announce their first Y block 2 ms after they
    stored it -- twice the wall-clock bound.  With the tile workers' polls throttled to ~7 us each (bit 11: fewer than 768 in
    those 2 ms -- a waiter that was hardly running, as when the process's queues are off the device) the wait must NOT expire: the
    frame comes out right, late, with no re-run.  Without the throttle the same 2 ms are > 768 polls of a running wave: the wait
    expires (that is what the bound is for), the stage is re-run on the fallback route, and the answer is still right."""
    fr = make_frame(L=90, H=120, seed=321)
    cfg = default_config(compat=0, adaptive=1)
    ic, r0 = oracle_frame(oracle_lib, fr, cfg)

    def run(mask):
        hip_dbg.set_sweep_exp(mask)
        try:
            g = hip_dbg.RslamHip(cfg)
            g.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, ic, fr.draws)
            g.step_frame(False); g.sync()                  # (first frame of the context: module load, buffers)
            c0 = g.counters()["sweep_reruns"]
            g2 = hip_dbg.RslamHip(cfg)
            g2.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, ic, fr.draws)
            t0 = time.perf_counter()
            g2.step_frame(False); g2.sync()
            ms = (time.perf_counter() - t0) * 1e3
            r = g2.fetch_results()
            out = (r, g2.counters()["sweep_reruns"], g2.last_raw_status(), g2.last_wait_detail(), ms, g2.update_mode())
            g.close(); g2.close()
            return out, c0
        finally:
            hip_dbg.set_sweep_exp(-1)

    (r1, reruns, raw, detail, ms, mode), _ = run(1024 | 2048)
    assert mode == 2                                        # (the route with tile workers)
    assert ms >= 2.0, ms                                    # the hold was in effect: a multi-block sweep waited its 2 ms out
    assert reruns == 0 and raw == 0 and detail is None, (reruns, raw, detail)
    assert np.array_equal(r1["li"], r0["li"]) and np.array_equal(r1["hi"], r0["hi"])
    assert close_x(r1["x_new"], r0["x_new"]) and close_P(r1["P_new"], r0["P_new"])
    (r2, reruns2, raw2, detail2, _, _), _ = run(1024)
    assert reruns2 >= 1 and raw2 == -37, (reruns2, raw2)
    assert detail2 is not None and detail2["code"] == 37 and detail2["polls"] >= 768 and detail2["elapsed_us"] >= 1000, detail2
    assert np.array_equal(r2["li"], r0["li"]) and np.array_equal(r2["hi"], r0["hi"])
    assert close_x(r2["x_new"], r0["x_new"]) and close_P(r2["P_new"], r0["P_new"])


def test_unchecked_timeout_is_settled_before_ekf_prediction(hip_dbg, oracle_lib):
    """A persistent sweep that timed out in a frame nobody synchronised on (rslam_step_frame, no rslam_sync) must be
    noticed and re-run when rslam_ekf_prediction turns its posterior into the next prior -- while the frame's inputs are
    still in place -- and rslam_load_measurements for the next frame must then simply work."""
    fr = make_frame(L=90, H=120, seed=322)
    cfg = default_config(compat=0, adaptive=1)
    ic, r0 = oracle_frame(oracle_lib, fr, cfg)
    xp0, Pp0 = oracle_lib.ekf_prediction(r0["x_new"], r0["P_new"], 1.0, 0.007, 0.007)
    hip_dbg.set_sweep_exp(16)              # the chain workgroup never shows up
    try:
        g = hip_dbg.RslamHip(cfg)
        g.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, ic, fr.draws)
        g.step_frame(True)                               # enqueued, not looked at
        g.ekf_prediction(1.0, 0.007, 0.007)              # must settle the frame first (re-run inside)
    finally:
        hip_dbg.set_sweep_exp(-1)
    assert g.last_raw_status() <= -30 and g.counters()["sweep_reruns"] >= 1
    xp1, Pp1 = g.fetch_prior()
    assert close_x(xp1, xp0) and close_P(Pp1, Pp0)
    z2, _, d2 = remeasure(fr, 77, frac_outlier=0.2, H=120)
    g.predict_resident()
    g.load_measurements(z2, ic, d2)                      # no stale status, no state error
    g.step_frame(True); g.sync()
    g.close()


def test_rank_update_rider_timeout_reruns_with_riders_first(hip_dbg, oracle_lib):
    """The stand-alone rank update (systems the sweep cannot host: here forced by RSLAM_SWEEP_EXP bit 7) places the x-update
    riders behind the tiles when everything is resident at once.  Fault injection: such riders never publish Jnorm -- the
    first block column runs into its bounded wait (-39), the host re-runs the update stage with the riders in front and
    keeps that order."""
    fr = make_frame(L=90, H=120, seed=323)
    cfg = default_config(compat=0, adaptive=1)
    ic, r0 = oracle_frame(oracle_lib, fr, cfg)
    hip_dbg.set_sweep_exp(128)
    try:
        g = hip_dbg.RslamHip(cfg)
        g.debug_set_k10_inject(1)
        g.predict(fr.types, fr.x_pred, fr.P_pred)
        r1 = g.ransac_update(fr.z, ic, fr.draws)
        assert g.last_raw_status() == -39 and g.counters()["sweep_reruns"] >= 1
        assert np.array_equal(r1["li"], r0["li"]) and np.array_equal(r1["hi"], r0["hi"])
        assert close_x(r1["x_new"], r0["x_new"]) and close_P(r1["P_new"], r0["P_new"])
        n_before = g.counters()["sweep_reruns"]
        g.predict(fr.types, fr.x_pred, fr.P_pred)       # riders first from now on: the injection no longer applies
        r2 = g.ransac_update(fr.z, ic, fr.draws)
        assert g.counters()["sweep_reruns"] == n_before and close_P(r2["P_new"], r0["P_new"])
        g.close()
    finally:
        hip_dbg.set_sweep_exp(-1)


def test_status_of_an_unsynced_frame_is_not_lost(hip):
    """Frames enqueued back to back without rslam_sync: the next frame's reset used to erase the status word of the one
    before.  A frame that fails (covariance not positive definite) followed by a good one: the failure is reported by
    the next sync -- once."""
    fr = make_frame(L=20, H=16, seed=403)
    cfg = default_config(compat=0)
    g = hip.RslamHip(cfg)
    g.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, fr.ic, fr.draws)
    g.step_predict(); g.sync()
    ic = (fr.ic & g.fetch_prediction()[1]).astype(np.uint8)
    g.load_frame(fr.types, fr.x_pred, -10.0 * np.eye(fr.n), fr.z, ic, fr.draws)
    g.step_frame(False)                                   # fails on the device; nobody looks
    g.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, ic, fr.draws)     # (uploads wait for the stream, they read no status)
    g.step_frame(False)
    with pytest.raises(hip.RslamError) as e:
        g.sync()
    assert e.value.code == -6
    g.step_frame(False)
    g.sync()                                              # reported once: the context is clean again
    g.close()


@pytest.mark.parametrize("L,H,seed", [(90, 120, 21), (300, 200, 2), (40, 100, 7)])
def test_deferred_li_covariance_equals_immediate(hip_dbg, oracle_lib, L, H, seed):
    """A low-innovation update of rank <= 4 (compat = 1: the consensus set is the hypothesis' own feature) does not stream P:
    Y1 and its Jnorm are kept aside and the rescue prediction, the second P H^T and the HI pass's tile workers form
    P_li = J (sym(P_pred) - Y1 Y1^T) J^T themselves (SEL_LI_DEFER).  Same frame with the deferral (default), with the
    immediate stream (RSLAM_SWEEP_EXP bit 9) and with the stand-alone rank update (bit 7): all against the oracle, and
    the decisions of the rescue gate (which sees the deferred S) identical."""
    fr = make_frame(L=L, H=H, seed=seed)
    cfg = default_config(compat=1, adaptive=0)
    ic, r0 = oracle_frame(oracle_lib, fr, cfg)
    assert int(r0["li"].sum()) in (1, 2)
    out = {}
    for mask in (0, 512, 128):
        hip_dbg.set_sweep_exp(mask)
        try:
            g = hip_dbg.RslamHip(cfg)
            g.predict(fr.types, fr.x_pred, fr.P_pred)
            out[mask] = g.ransac_update(fr.z, ic, fr.draws)
            g.close()
        finally:
            hip_dbg.set_sweep_exp(-1)
        r1 = out[mask]
        assert np.array_equal(r1["li"], r0["li"]) and np.array_equal(r1["hi"], r0["hi"]), mask
        assert close_x(r1["x_new"], r0["x_new"]) and close_P(r1["P_new"], r0["P_new"]), mask
    assert close_P(out[0]["P_new"], out[512]["P_new"], 1e-11) and close_x(out[0]["x_new"], out[512]["x_new"], 1e-12)


def test_deferred_li_covariance_with_asymmetric_prior(hip_dbg, oracle_lib):
    """The invariant the deferred route leans on: a prior is symmetric to rounding (an uploaded p_k_km1 exactly, one left by
    rslam_ekf_prediction to ~1e-16 relative).  Its readers do not all symmetrise the same way -- the second P H^T reads raw
    columns of P_pred, the tile workers sym(P_pred) -- so an asymmetry of relative size e moves the deferred result by O(e)
    against the immediate stream: with e = 1e-12 both must still be within the parity tolerance of the oracle run on the
    symmetrised prior, and of each other to ~1e-10 (nothing amplifies the asymmetry)."""
    fr = make_frame(L=90, H=120, seed=21)
    cfg = default_config(compat=1, adaptive=0)
    P = np.asarray(fr.P_pred).copy()
    rng = np.random.default_rng(17)
    P_as = P * (1.0 + 1e-12 * rng.standard_normal(P.shape))              # entry-wise relative asymmetry ~1e-12
    assert np.max(np.abs(P_as - P_as.T)) > 0
    fr_sym = make_frame(L=90, H=120, seed=21)
    fr_sym.P_pred = 0.5 * (P_as + P_as.T)
    ic, r0 = oracle_frame(oracle_lib, fr_sym, cfg)
    assert int(r0["li"].sum()) in (1, 2)
    out = {}
    for mask in (0, 512):
        hip_dbg.set_sweep_exp(mask)
        try:
            g = hip_dbg.RslamHip(cfg)
            g.predict(fr.types, fr.x_pred, P_as)
            out[mask] = g.ransac_update(fr.z, ic, fr.draws)
            g.close()
        finally:
            hip_dbg.set_sweep_exp(-1)
        r1 = out[mask]
        assert np.array_equal(r1["li"], r0["li"]) and np.array_equal(r1["hi"], r0["hi"]), mask
        assert close_x(r1["x_new"], r0["x_new"]) and close_P(r1["P_new"], r0["P_new"]), mask
    assert close_P(out[0]["P_new"], out[512]["P_new"], 1e-10) and close_x(out[0]["x_new"], out[512]["x_new"], 1e-10)


def test_zero_li_inliers_with_asymmetric_prior(hip, oracle_lib):
    """compat = 1 and NO low-innovation inlier (the winner's own feature falls outside the threshold): update() is the
    identity (ExtendKF.cpp:635-638) and the consensus launch writes it as the DEFERRED identity -- Y1 = 0, Jnorm = I -- so that
    every later reader forms P_li = sym(P_pred).  The reference keeps p_km_k as it is; the two agree exactly for a symmetric
    prior (an uploaded one) and to the prior's own asymmetry otherwise (one left by rslam_ekf_prediction: ~1e-16 relative).
    Held here with an asymmetry of 1e-12: (a) the frame against the oracle run on the asymmetric prior itself and on the
    symmetrised one; (b) with the rescue gate shut (no high-innovation inlier either: wk_materialise_deferred writes the
    posterior) p_k_k is EXACTLY sym(P_pred) -- the documented form -- which is the reference's unchanged p_km_k to 1e-12, and
    x_k_k is x_km_k bit for bit."""
    fr = make_frame(L=40, H=60, seed=3)
    P = np.asarray(fr.P_pred).copy()
    rng = np.random.default_rng(5)
    P_as = P * (1.0 + 1e-12 * rng.standard_normal(P.shape))
    P_sym = 0.5 * (P_as + P_as.T)
    assert np.max(np.abs(P_as - P_as.T)) > 0
    for chi2 in (None, 1e-12):
        cfg = default_config(compat=1, adaptive=0)
        if chi2 is not None:
            cfg.chi2_gate = chi2
        refs = []
        for Pref in (P_as, P_sym):
            fr_ref = make_frame(L=40, H=60, seed=3)
            fr_ref.P_pred = Pref
            ic, r0 = oracle_frame(oracle_lib, fr_ref, cfg)
            refs.append(r0)
        assert int(refs[0]["li"].sum()) == 0
        assert (int(refs[0]["hi"].sum()) > 10) if chi2 is None else (int(refs[0]["hi"].sum()) == 0)
        g = hip.RslamHip(cfg)
        g.predict(fr.types, fr.x_pred, P_as)
        r1 = g.ransac_update(fr.z, ic, fr.draws)
        assert g.counters()["sweep_reruns"] == 0           # (zero inliers is not the guard's case: it is the deferred identity)
        g.close()
        for r0 in refs:
            assert np.array_equal(r1["li"], r0["li"]) and np.array_equal(r1["hi"], r0["hi"])
            assert close_x(r1["x_new"], r0["x_new"]) and close_P(r1["P_new"], r0["P_new"])
        if chi2 is not None:
            n = len(fr.x_pred)
            assert np.array_equal(r1["x_new"], np.asarray(fr.x_pred))
            assert np.array_equal(r1["P_new"], P_sym)
            assert np.max(np.abs(r1["P_new"] - P_as)) <= 2e-12 * np.max(np.abs(P_as))


def test_deferred_li_covariance_without_hi_inliers(hip, oracle_lib):
    """... and when the rescue gate lets nobody through (chi-square gate at ~0) the high-innovation pass is a pass-through of
    a covariance that does not exist yet: it has to be written then (wk_materialise_deferred)."""
    fr = make_frame(L=90, H=120, seed=21)
    cfg = default_config(compat=1, adaptive=0)
    cfg.chi2_gate = 1e-12
    ic, r0 = oracle_frame(oracle_lib, fr, cfg)
    assert int(r0["li"].sum()) >= 1 and int(r0["hi"].sum()) == 0
    g = hip.RslamHip(cfg)
    g.predict(fr.types, fr.x_pred, fr.P_pred)
    r1 = g.ransac_update(fr.z, ic, fr.draws)
    g.close()
    assert np.array_equal(r1["li"], r0["li"]) and not r1["hi"].any()
    assert close_x(r1["x_new"], r0["x_new"]) and close_P(r1["P_new"], r0["P_new"])
