"""A/B of the large-map route inside ONE GPU call (diagnostic library: the switches are read from the environment):
64 x 64 rank update / macro tiles; one-stream sweep / staged route (group solves beside the S stage).  ms per frame (eager frames, the large-map route is
never graph-captured), the stage times, and every variant's posterior against the first one's.
    python scripts/ab_c5.py [compat] [L]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_slam_amd import api, default_config
from ransac_slam_amd.synth import make_frame
compat = int(sys.argv[1]) if len(sys.argv) > 1 else 1
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
fr = make_frame(L=L, H=1000, seed=4)
variants = [("64x64 rank update, one-stream sweep", dict(RSLAM_NO_MACRO="1")),
            ("macro tiles,       one-stream sweep", dict()),
            ("macro tiles,       staged (group solves)", dict(RSLAM_STAGED_MIN_BLOCKS="12")),
            ("macro tiles, x update as a launch of its own", dict(RSLAM_RIDERS_IN_SMALL="0"))]
SWITCHES = ("RSLAM_NO_MACRO", "RSLAM_STAGED_MIN_BLOCKS", "RSLAM_RIDERS_IN_SMALL")
if os.environ.get("AB_SKIP_STAGED"):
    variants = [v for v in variants if "staged" not in v[0]]
extra = [a for a in sys.argv[3:] if "=" in a]                  # e.g. RSLAM_STAGED_CUS_S=64, applied to every variant
ref = None
if os.environ.get("AB_ONLY"):                                   # one variant only (under a profiler)
    variants = [variants[int(os.environ["AB_ONLY"])]]
for name, env in variants:
    for k in SWITCHES:
        os.environ.pop(k, None)
    os.environ.update(env)
    for a in extra:
        k, v = a.split("=", 1); os.environ[k] = v
    ctx = api.RslamHip(default_config(compat=compat, adaptive=0), debug=True)
    ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, fr.ic, fr.draws)
    ctx.step_predict(); ctx.sync()
    ic = fr.ic & ctx.fetch_prediction()[1]
    ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, ic, fr.draws)
    for _ in range(5):
        ctx.step_frame(False)
    ctx.sync()
    out = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            ctx.step_frame(False)
        ctx.sync()
        out.append((time.perf_counter() - t0) / 10 * 1e3)
    ctx.enable_timing(True)
    st = []
    for i in range(8):
        ctx.step_frame(False); ctx.sync()
        if i >= 2:
            st.append(ctx.timings())
    ctx.enable_timing(False)
    r = ctx.fetch_results(want_P=True)
    msg = ""
    if ref is None:
        ref = r
    else:
        same = np.array_equal(r["li"], ref["li"]) and np.array_equal(r["hi"], ref["hi"])
        dP = np.max(np.abs(r["P_new"] - ref["P_new"])) / np.max(np.abs(ref["P_new"]))
        dx = np.max(np.abs(r["x_new"] - ref["x_new"]))
        msg = "  vs first: sets %s, max|dx| %.1e, max|dP|/max|P| %.1e" % ("equal" if same else "DIFFER", dx, dP)
    med = {k: float(np.median([t[k] for t in st])) for k in st[0]}
    print("%-38s compat %d  ms/frame %s  median %.4f  mode %d%s" % (name, compat, " ".join("%.3f" % v for v in out), np.median(out), ctx.update_mode(), msg))
    print("      li: factor %.0f rank %.0f   hi: factor %.0f rank %.0f   total %.0f us   n_li %d n_hi %d" % (
        med["factor_li_us"], med["rank_update_li_us"], med["factor_hi_us"], med["rank_update_hi_us"], med["total_us"], r["n_li"], r["n_hi"]))
    ctx.close()
