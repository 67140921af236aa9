"""soak: many back-to-back resident frames (stream-ordered launches; --graph: hipGraph replay), a sync every 100; reports re-runs /
re-captures and the spread"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_slam_amd import api, default_config
from ransac_slam_amd.synth import make_frame
GRAPH = "--graph" in sys.argv
sys.argv = [a for a in sys.argv if a != "--graph"]
compat = int(sys.argv[1]) if len(sys.argv) > 1 else 1
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
L, H, seed = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (300, 1000, 2)
fr = make_frame(L=L, H=H, seed=seed)
ctx = api.RslamHip(default_config(compat=compat, adaptive=0))
ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, fr.ic, fr.draws)
ctx.step_predict(); ctx.sync()
ic = fr.ic & ctx.fetch_prediction()[1]
ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, ic, fr.draws)
for _ in range(30):
    ctx.step_frame(GRAPH)
ctx.sync()
r0 = ctx.fetch_results(want_P=True)
ts = []
for b in range(frames // 100):
    t0 = time.perf_counter()
    for _ in range(100):
        ctx.step_frame(GRAPH)
    ctx.sync()
    ts.append((time.perf_counter() - t0) / 100 * 1e3)
r1 = ctx.fetch_results(want_P=True)
ts = np.array(ts)
same = np.array_equal(r0["x_new"], r1["x_new"]) and np.array_equal(r0["P_new"], r1["P_new"])
print("hipGraph replay" if GRAPH else "stream-ordered", "L", L, "compat", compat, "frames", frames, "ms/frame median %.4f p99 %.4f max %.4f" % (np.median(ts), np.percentile(ts, 99), ts.max()),
      "counters", ctx.counters(), "bitwise stable", same, "last raw status", ctx.last_raw_status(), "first wait", ctx.last_wait_detail(),
      "slow batches (of 100 frames)", [int(i) for i in np.nonzero(ts > 1.2 * np.median(ts))[0][:4]], "of", len(ts))
