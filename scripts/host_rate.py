"""Is the frame loop host-bound?  Times the enqueue of N frames against their completion, for hipGraph replay and (--eager) for
stream-ordered launches."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ransac_slam_amd import default_config, api
from ransac_slam_amd.synth import make_frame
GRAPH = "--eager" not in sys.argv
fr = make_frame(L=300, H=1000, seed=2)
ctx = api.RslamHip(default_config(compat=1, adaptive=0))
ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, fr.ic, fr.draws)
ctx.step_predict(); ctx.sync()
_, vis, _ = ctx.fetch_prediction()
ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, fr.ic & vis, fr.draws)
for _ in range(12):
    ctx.step_frame(GRAPH); ctx.sync()
print("hipGraph replay" if GRAPH else "stream-ordered launches")
for N in (50, 200, 800):
    t0 = time.perf_counter()
    for _ in range(N):
        ctx.step_frame(GRAPH)
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    print(f"N={N}: enqueue {1e3 * (t1 - t0) / N:.4f} ms/frame, complete {1e3 * (t2 - t0) / N:.4f} ms/frame")
# one frame at a time with a sync: pure latency
ts = []
for _ in range(50):
    t0 = time.perf_counter(); ctx.step_frame(GRAPH); ctx.sync(); ts.append(time.perf_counter() - t0)
print(f"single frame + sync: {1e3 * np.median(ts):.4f} ms")
