"""Timing of the HI sweep launch for ablation builds (kernel experiments whose RESULTS are wrong on purpose: a status error of
the frame is ignored, only the hipEvent time of the launch is read).  RSLAM_HIP_LIB_DEBUG=<lib> python scripts/ab_abl.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_slam_amd import api, default_config
from ransac_slam_amd.synth import make_frame
fr = make_frame(L=300, H=1000, seed=2)
ctx = api.RslamHip(default_config(compat=1, adaptive=0), debug=True)
ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, fr.ic, fr.draws)
ctx.step_predict(); ctx.sync()
ic = fr.ic & ctx.fetch_prediction()[1]
ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, ic, fr.draws)
ctx.enable_timing(True)
fh, errs = [], 0
for i in range(45):
    ctx.step_frame(False)
    try:
        ctx.sync()
    except api.RslamError:
        errs += 1
    if i >= 5:
        fh.append(ctx.timings()["factor_hi_us"])
print(os.path.basename(api.LIB_PATH_DEBUG), "factor_hi_us mean %.2f median %.2f min %.2f (frames with a status error: %d)" % (np.mean(fh), np.median(fh), np.min(fh), errs))
