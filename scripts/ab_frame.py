"""ms/frame of the resident C3 frame (hipGraph replay) for the library RSLAM_HIP_LIB points at: A/B timing of kernel variants
inside ONE gpurun call (boxes differ by a few per cent).   RSLAM_HIP_LIB=/path/lib.so python scripts/ab_frame.py [compat]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_slam_amd import api, default_config
from ransac_slam_amd.synth import make_frame
DEBUG = "--debug" in sys.argv         # time the diagnostic variant (RSLAM_HIP_LIB_DEBUG=<a build of ransac_slam_amd.build.build_dev>)
GRAPH = "--eager" not in sys.argv     # --eager: the frames of the timed loops as stream-ordered launches, no hipGraph replay
sys.argv = [a for a in sys.argv if a not in ("--debug", "--eager")]
compat = int(sys.argv[1]) if len(sys.argv) > 1 else 1
L, H, seed = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (300, 1000, 2)
fr = make_frame(L=L, H=H, seed=seed)
ctx = api.RslamHip(default_config(compat=compat, adaptive=0), debug=DEBUG)
ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, fr.ic, fr.draws)
ctx.step_predict(); ctx.sync()
ic = fr.ic & ctx.fetch_prediction()[1]
ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, ic, fr.draws)
for _ in range(30):
    ctx.step_frame(GRAPH)
ctx.sync()
out = []
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(200):
        ctx.step_frame(GRAPH)
    ctx.sync()
    out.append((time.perf_counter() - t0) / 200 * 1e3)
r = ctx.fetch_results(want_P=False)
# the dominant launch on its own: hipEvents around the HI sweep launch, mean over eager frames
ctx.enable_timing(True)
fh, tot = [], []
for i in range(45):
    ctx.step_frame(False); ctx.sync()
    if i >= 5:
        tt = ctx.timings(); fh.append(tt["factor_hi_us"]); tot.append(tt["total_us"])
ctx.enable_timing(False)
if DEBUG:
    # the experiment's posterior against the product library's (the in-tree librslam_hip.so) on the same frame
    ref = api.RslamHip(default_config(compat=compat, adaptive=0))
    ref.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, ic, fr.draws)
    ref.step_frame(False); ref.sync()
    r0 = ref.fetch_results(want_P=True); r1 = ctx.fetch_results(want_P=True)
    dP = np.max(np.abs(r1["P_new"] - r0["P_new"])) / np.max(np.abs(r0["P_new"]))
    dx = np.max(np.abs(r1["x_new"] - r0["x_new"]))
    same = np.array_equal(r1["li"], r0["li"]) and np.array_equal(r1["hi"], r0["hi"])
    print("   vs product library: sets %s, max|dx| %.2e, max|dP|/max|P| %.2e %s" % ("equal" if same else "DIFFER", dx, dP, "OK" if (same and dx < 1e-9 and dP < 1e-9) else "** MISMATCH **"))
    ref.close()
print("   eager: factor_hi_us mean %.2f median %.2f  total_us median %.1f" % (np.mean(fh), np.median(fh), np.median(tot)))
print("   counters", ctx.counters(), "last raw status", ctx.last_raw_status())
print(os.path.basename(api.LIB_PATH_DEBUG if DEBUG else api.LIB_PATH), "compat", compat, "ms/frame", " ".join("%.4f" % v for v in out), "median %.4f" % np.median(out), "n_li", r["n_li"], "n_hi", r["n_hi"])
