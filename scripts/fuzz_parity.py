"""Randomised parity sweep on the GPU (not part of the test suite: run through gpurun when a route changes):
N random frames -- size, hypotheses, outlier fraction, compatible fraction, arithmetic mode, adaptive or not -- HIP against the
oracle with the tests' own tolerances; frames whose margins are below the audit threshold are skipped (the integer outputs are
then not well defined).  Prints the distribution of the low-innovation counts it met and every failure.
    python scripts/fuzz_parity.py [N] [seed0] [Lmax]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_slam_amd import api, default_config
from ransac_slam_amd.synth import make_frame
from oracle import pyoracle
pyoracle.build()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
Lmax = int(sys.argv[3]) if len(sys.argv) > 3 else 120
X_TOL = P_TOL = 1e-9
rng = np.random.default_rng(seed0)
hist, bad, skipped, modes = {}, [], 0, {}
for i in range(N):
    L = int(rng.integers(1, Lmax + 1)); H = int(rng.integers(2, 200))
    compat = int(rng.integers(0, 2)); adaptive = int(rng.integers(0, 2))
    case = dict(L=L, H=H, seed=seed0 + i, frac_outlier=float(rng.choice([0.0, 0.1, 0.3, 0.6, 1.0])), frac_ic=float(rng.choice([1.0, 0.9, 0.5])))
    fr = make_frame(**case)
    cfg = default_config(compat=compat, adaptive=adaptive)
    o = pyoracle.Oracle(cfg, structure=1)
    h0, v0, S0 = o.predict(fr.types, fr.x_pred, fr.P_pred)
    g = api.RslamHip(cfg)
    h1, v1, S1 = g.predict(fr.types, fr.x_pred, fr.P_pred)
    ic = (fr.ic & v0).astype(np.uint8)
    try:
        r0 = o.ransac_update(fr.z, ic, fr.draws)
    except Exception as e:
        skipped += 1; g.close(); continue
    r1 = g.ransac_update(fr.z, ic, fr.draws)
    sm, rm = o.margins()
    if sm <= 1e-9 or rm <= 1e-9:
        skipped += 1; g.close(); continue
    k = int(r0["li"].sum())
    hist[(compat, min(k, 3))] = hist.get((compat, min(k, 3)), 0) + 1
    modes[g.update_mode()] = modes.get(g.update_mode(), 0) + 1
    d = np.sqrt(np.abs(np.diag(r0["P_new"])))
    ok = (np.array_equal(v0, v1) and r1["best_hyp"] == r0["best_hyp"] and r1["best_support"] == r0["best_support"]
          and r1["hyps_evaluated"] == r0["hyps_evaluated"]
          and np.array_equal(r1["li"], r0["li"]) and np.array_equal(r1["hi"], r0["hi"])
          and np.max(np.abs(r1["x_new"] - r0["x_new"])) <= X_TOL * max(1.0, float(np.max(np.abs(r0["x_new"]))))
          and np.max(np.abs(r1["P_new"] - r0["P_new"])) <= P_TOL * float(np.max(np.abs(r0["P_new"])))
          and bool(np.all(np.abs(r1["P_new"] - r0["P_new"]) <= P_TOL * np.outer(d, d) + 1e-300))
          and g.counters()["sweep_reruns"] == 0)
    if not ok:
        bad.append((case, compat, adaptive, k, int(r0["hi"].sum()), g.counters(), g.last_raw_status()))
    g.close()
    if (i + 1) % 20 == 0:               # (a sign of life: a run that writes nothing for minutes is taken to be hung)
        print("... %d frames, %d failures so far" % (i + 1, len(bad)), flush=True)
print("frames", N, "skipped (reference assertion / margin audit)", skipped, "failures", len(bad))
print("low-innovation counts met (compat, min(k, 3)) -> frames:", dict(sorted(hist.items())), " update modes:", modes)
for b in bad:
    print("FAIL", b)
sys.exit(1 if bad else 0)
