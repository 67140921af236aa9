"""A/B of RSLAM_SWEEP_EXP switches inside ONE GPU call (diagnostic library): hipGraph replays of the C3 frame, interleaved rounds.
    python scripts/ab_sweep_exp.py <compat> <mask> [<mask> ...]      e.g. 1 0 1024   (bit 10: tile (0,0) from the strips, the round-4 form)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_slam_amd import api, default_config
from ransac_slam_amd.synth import make_frame
compat = int(sys.argv[1]) if len(sys.argv) > 1 else 1
masks = [int(a) for a in sys.argv[2:]] or [0, 1024]
fr = make_frame(L=300, H=1000, seed=2)
lib = api.lib(debug=True)
ctxs = []
for m in masks:
    assert lib.rslam_debug_set_sweep_exp(m) == 0
    ctx = api.RslamHip(default_config(compat=compat, adaptive=0), debug=True)
    ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, fr.ic, fr.draws)
    ctx.step_predict(); ctx.sync()
    ic = fr.ic & ctx.fetch_prediction()[1]
    ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, ic, fr.draws)
    for _ in range(50):
        ctx.step_frame(True)          # (the switches are read when the sequence is enqueued: captured here)
    ctx.sync()
    ctxs.append((m, ctx, []))
lib.rslam_debug_set_sweep_exp(-1)
for rnd in range(7):
    for m, ctx, ts in ctxs:
        t0 = time.perf_counter()
        for _ in range(2000):
            ctx.step_frame(True)
        ctx.sync()
        ts.append((time.perf_counter() - t0) / 2000 * 1e3)
ref = None
for m, ctx, ts in ctxs:
    r = ctx.fetch_results(want_P=True)
    if ref is None:
        ref = r
    print("exp %5d  compat %d  ms/frame" % (m, compat), " ".join("%.4f" % t for t in ts), " median %.4f" % np.median(ts), ctx.counters(),
          "bitwise equal to the first:", bool(np.array_equal(r["x_new"], ref["x_new"]) and np.array_equal(r["P_new"], ref["P_new"])))
