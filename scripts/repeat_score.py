import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch
from ransac_slam_amd import api, default_config
from ransac_slam_amd.synth import make_frame
fr = make_frame(L=300, H=1000, seed=2)
ctx = api.RslamHip(default_config(compat=1, adaptive=0))
ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, fr.ic, fr.draws)
ctx.step_predict(); ctx.sync()
ic = fr.ic & ctx.fetch_prediction()[1]
ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, ic, fr.draws)
sup = torch.zeros(1000, dtype=torch.int32, device="cuda:0")
other = torch.zeros(64, dtype=torch.int32, device="cuda:0")
torch.cuda.synchronize()
for it in range(300):
    ctx.step_frame(False)                 # a whole frame (caches as in the sequence)
    for r in range(3):
        ctx.step_predict()                # the same launch three times in a row: #2 and #3 find code, translations and data warm
    for r in range(3):
        ctx.step_score(0, 1000, sup.data_ptr())
    # ... once more behind an idle device (#4), and behind an idle device + ANOTHER tiny kernel (#5): is it the code?
    ctx.sync()
    ctx.step_score(0, 1000, sup.data_ptr())
    ctx.sync()
    other.zero_()
    torch.cuda.synchronize()
    ctx.step_score(0, 1000, sup.data_ptr())
ctx.sync()
