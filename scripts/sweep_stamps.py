"""Where the time of the persistent factor sweep goes: wall-clock stamps (100 MHz) written by the chain
workgroup and four strip workgroups of the LAST sweep launch of a frame (the HI pass).
    python scripts/sweep_stamps.py [--compat 0]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_slam_amd import api, default_config           # noqa: E402
from ransac_slam_amd.synth import make_frame              # noqa: E402

WHO, K, SLOT = 6, 16, 8
ap = argparse.ArgumentParser()
ap.add_argument("--compat", type=int, default=1)
ap.add_argument("--L", type=int, default=300)
ap.add_argument("--H", type=int, default=1000)
ap.add_argument("--seed", type=int, default=2)
ap.add_argument("--li", action="store_true", help="stamps of the LI pass (RSLAM_SWEEP_EXP bit 8) instead of the HI pass")
a = ap.parse_args()
if a.li:
    api.lib(debug=True).rslam_debug_set_sweep_exp(256)
fr = make_frame(L=a.L, H=a.H, seed=a.seed)
ctx = api.RslamHip(default_config(compat=a.compat, adaptive=0), debug=True)     # stamps: diagnostic variant of the library
ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, fr.ic, fr.draws)
ctx.step_predict(); ctx.sync()
ic = fr.ic & ctx.fetch_prediction()[1]
ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, ic, fr.draws)
fn = api.lib(debug=True).rslam_debug_sweep_stamps
for _ in range(3):
    ctx.step_frame(False); ctx.sync()
assert fn(ctx._h, None, 1) == 0
ctx.step_frame(False); ctx.sync()
buf = np.zeros(WHO * K * SLOT + 512, np.uint64)
assert fn(ctx._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), 0) == 0
wg = buf[WHO * K * SLOT:].reshape(2, 256).astype(np.int64)
buf = buf[:WHO * K * SLOT]
r = ctx.fetch_results(want_P=False)
print("n_li", r["n_li"], "n_hi", r["n_hi"])
st = buf.reshape(WHO, K, SLOT).astype(np.int64)
t0 = st[0, 0, 0]
us = lambda v: (v - t0) / 100.0 if v else float("nan")
print("chain: per block [loop top, inputs in LDS, X product done + L^-1 flag out, chain start, chain end, end barrier, next inputs stored (T wave 0), next inputs issued]")
for k in range(K):
    if st[0, k, 0] == 0:
        break
    print(f"  block {k:2d}: " + " ".join(f"{us(v):8.2f}" for v in st[0, k]))
names = {1: "S strip 8 (row block 2)", 2: "first strip of the last S row block", 3: "first P H^T strip", 4: "nu strip"}
for who in (1, 2, 3, 4):
    print(names[who] + ": per step [before linv wait, after, X stored/posted, after panel wait, updates done, handed over]")
    for k in range(K):
        if st[who, k, 0] == 0:
            break
        print(f"  step {k:2d}: " + " ".join(f"{us(v):8.2f}" for v in st[who, k, :6]))
print("tile worker 0 (fused launches): start %.2f, epilogue done %.2f; per block [wait begins, Y flags seen, chunks done]" % (us(st[5, 0, 0]), us(st[5, 0, 4])))
for k in range(K):
    if st[5, k, 1] == 0:
        break
    print(f"  block {k:2d}: " + " ".join(f"{us(v):8.2f}" for v in st[5, k, 1:4]))
live = wg[0] > 0
if live.any():
    s0 = wg[0][live].min()
    print("workgroups: first start %.2f, last start %.2f, first end %.2f, LAST END %.2f (workgroup %d); chain reference t0 = %.2f after the first start"
          % ((s0 - t0) / 100.0, (wg[0][live].max() - t0) / 100.0, (wg[1][live].min() - t0) / 100.0, (wg[1][live].max() - t0) / 100.0,
             int(np.argmax(np.where(live, wg[1], 0))), (t0 - s0) / 100.0))
    order = np.argsort(-np.where(live, wg[1], 0))[:8]
    print("  latest workgroups (block index: end):", ", ".join("%d: %.2f" % (int(i), (wg[1][i] - t0) / 100.0) for i in order))
