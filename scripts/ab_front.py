"""A/B of the low-innovation update of the reference-faithful mode inside ONE GPU call (diagnostic library, switches from the
environment): (a) the persistent sweep's own launch (strips, register route), (b) inside the consensus launch with the sweep
launch still in the sequence (it returns at once), (c) inside the consensus launch, no sweep launch (the product).
ms per frame of hipGraph replays, interleaved rounds.   python scripts/ab_front.py [L] [H]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_slam_amd import api, default_config
from ransac_slam_amd.synth import make_frame
L = int(sys.argv[1]) if len(sys.argv) > 1 else 300
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
fr = make_frame(L=L, H=H, seed=2)
variants = [("a: sweep launch (strips)          ", dict(RSLAM_NO_LI_SMALL="1", RSLAM_LI_SKIP="0")),
            ("b: consensus launch + empty sweep ", dict(RSLAM_LI_SKIP="0")),
            ("c: consensus launch, no sweep     ", dict(RSLAM_LI_SKIP="1"))]
if len(sys.argv) > 3:                                            # e.g. "cba": the order the contexts are created (and timed) in --
    variants = [variants["abc".index(ch)] for ch in sys.argv[3]]  # contexts of one process differ by ~0.5 % at C5 whatever they run
ctxs = []
for name, env in variants:
    for k in ("RSLAM_NO_LI_SMALL", "RSLAM_LI_SKIP"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ctx = api.RslamHip(default_config(compat=1, adaptive=0), debug=True)
    ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, fr.ic, fr.draws)
    ctx.step_predict(); ctx.sync()
    ic = fr.ic & ctx.fetch_prediction()[1]
    ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, ic, fr.draws)
    for _ in range(50):
        ctx.step_frame(True)          # (the environment is read when the sequence is enqueued: captured here)
    ctx.sync()
    ctxs.append((name, ctx, []))
for k in ("RSLAM_NO_LI_SMALL", "RSLAM_LI_SKIP"):
    os.environ.pop(k, None)
reps = 2000 if L <= 400 else 100
for rnd in range(7):
    for name, ctx, ts in ctxs:
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.step_frame(True)
        ctx.sync()
        ts.append((time.perf_counter() - t0) / reps * 1e3)
ref = None
for name, ctx, ts in ctxs:
    r = ctx.fetch_results(want_P=True)
    if ref is None:
        ref = r
    print(name, "ms/frame", " ".join("%.4f" % t for t in ts), " median %.4f" % np.median(ts), "counters", ctx.counters(),
          "max|dx| %.1e max|dP|/max|P| %.1e" % (np.abs(r["x_new"] - ref["x_new"]).max(), np.abs(r["P_new"] - ref["P_new"]).max() / np.abs(ref["P_new"]).max()))
