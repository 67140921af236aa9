"""ms/frame of the resident C3 frame (hipGraph replay) for the library RSLAM_HIP_LIB points at: A/B timing of kernel variants
inside ONE gpurun call (boxes differ by a few per cent).   RSLAM_HIP_LIB=/path/lib.so python scripts/ab_frame.py [compat]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_slam_amd import api, default_config
from ransac_slam_amd.synth import make_frame
compat = int(sys.argv[1]) if len(sys.argv) > 1 else 1
L, H, seed = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (300, 1000, 2)
fr = make_frame(L=L, H=H, seed=seed)
ctx = api.RslamHip(default_config(compat=compat, adaptive=0))
ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, fr.ic, fr.draws)
ctx.step_predict(); ctx.sync()
ic = fr.ic & ctx.fetch_prediction()[1]
ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, ic, fr.draws)
for _ in range(30):
    ctx.step_frame(True)
ctx.sync()
out = []
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(200):
        ctx.step_frame(True)
    ctx.sync()
    out.append((time.perf_counter() - t0) / 200 * 1e3)
r = ctx.fetch_results(want_P=False)
print(os.path.basename(api.LIB_PATH), "compat", compat, "ms/frame", " ".join("%.4f" % v for v in out), "median %.4f" % np.median(out), "n_li", r["n_li"], "n_hi", r["n_hi"])
