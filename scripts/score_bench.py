"""the scoring launch (K4) of the C3 frame (--L 1000: of the C5 frame) on its own: hipEvent time of the stage for H = 1000 (one frame), 4000 (C4 on one GPU) and
16 000 hypotheses (the x16 grid of bench.py), both arithmetic modes; supports and masks against the product library.
   [RSLAM_HIP_LIB_DEBUG=ransac_slam_amd/_dev/<name>.so] python scripts/score_bench.py [--debug]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_slam_amd import api, default_config
from ransac_slam_amd.synth import make_frame
DEBUG = "--debug" in sys.argv
L = int(sys.argv[sys.argv.index("--L") + 1]) if "--L" in sys.argv else 300          # (--L 1000: the C5 map)
MULTS = (1, 4, 16) if L <= 300 else (1, 4)
fr = make_frame(L=L, H=1000, seed=2 if L == 300 else 4)
for compat in (1, 0):
    cfg = default_config(compat=compat, adaptive=0)
    probe = api.RslamHip(cfg)
    probe.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, fr.ic, fr.draws)
    probe.step_predict(); probe.sync()
    ic = fr.ic & probe.fetch_prediction()[1]
    probe.close()
    m = int(ic.sum())
    line = []
    for mult in MULTS:
        draws = np.random.default_rng(99).random(mult * 1000)
        res = {}
        for dbg in ((False, True) if DEBUG else (False,)):
            c = api.RslamHip(cfg, debug=dbg)
            c.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, ic, draws)
            c.enable_timing(True)
            ts = []
            for _ in range(12):
                c.step_frame(False); c.sync()
                ts.append(c.timings()["score_us"])
            r = c.fetch_results(want_P=False)
            res[dbg] = (float(np.median(ts[2:])), r["best_hyp"], r["best_support"], r["li"].copy(), r["hi"].copy())
            c.close()
        us = res[DEBUG][0]
        same = (not DEBUG) or (res[True][1:3] == res[False][1:3] and np.array_equal(res[True][3], res[False][3]) and np.array_equal(res[True][4], res[False][4]))
        line.append("H %5d: %7.2f us = %5.0f GB/s of the nominal 96 B per pair%s" % (mult * 1000, us, mult * 1000 * m * 96.0 / us * 1e-3, "" if same else "  ** DIFFERS from the product library **"))
    print("compat %d (%d matched):  " % (compat, m) + ";  ".join(line))
