"""Per-workgroup time line of the HI rank update of one C3 frame: start, K loop entered, K loop done, end (100 MHz wall
clock), XCC / CU of each workgroup.   python scripts/k10_stamps.py [--compat 0]"""
import argparse
import ctypes as C
import os
import sys
from collections import Counter

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_slam_amd import api, default_config           # noqa: E402
from ransac_slam_amd.synth import make_frame              # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--compat", type=int, default=1)
a = ap.parse_args()
fr = make_frame(L=300, H=1000, seed=2)
ctx = api.RslamHip(default_config(compat=a.compat, adaptive=0), debug=True)     # stamps: diagnostic variant of the library
ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, fr.ic, fr.draws)
ctx.step_predict(); ctx.sync()
ic = fr.ic & ctx.fetch_prediction()[1]
ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, ic, fr.draws)
fn = api.lib(debug=True).rslam_debug_k10_stamps
for _ in range(3):
    ctx.step_frame(False); ctx.sync()
assert fn(ctx._h, None, 1) == 0
ctx.step_frame(False); ctx.sync()
buf = np.zeros(8192 * 8, np.uint64)
assert fn(ctx._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), 0) == 0
st = buf.reshape(8192, 8).astype(np.int64)
st = st[st[:, 3] > 0]                       # tile workgroups of the last launch (HI pass)
t0 = st[:, 0].min()
us = (st[:, :4] - t0) / 100.0
hw = st[:, 4]
xcc = hw >> 32
cu = (hw & 0xffffffff) >> 8 & 0xf
se = (hw & 0xffffffff) >> 13 & 0x7
key = xcc * 1000 + se * 16 + cu
per_cu = Counter(key.tolist())
print("tile workgroups", len(st), " distinct CUs", len(per_cu), " CUs with 1/2/3+ tiles:",
      sum(1 for v in per_cu.values() if v == 1), sum(1 for v in per_cu.values() if v == 2), sum(1 for v in per_cu.values() if v >= 3))
print("start   : min %.2f  p50 %.2f  p95 %.2f  max %.2f" % (us[:, 0].min(), np.percentile(us[:, 0], 50), np.percentile(us[:, 0], 95), us[:, 0].max()))
print("loop in : p50 %.2f  max %.2f" % (np.percentile(us[:, 1], 50), us[:, 1].max()))
print("loop out: min %.2f  p50 %.2f  p95 %.2f  max %.2f" % (us[:, 2].min(), np.percentile(us[:, 2], 50), np.percentile(us[:, 2], 95), us[:, 2].max()))
print("end     : min %.2f  p50 %.2f  p95 %.2f  max %.2f" % (us[:, 3].min(), np.percentile(us[:, 3], 50), np.percentile(us[:, 3], 95), us[:, 3].max()))
loop = us[:, 2] - us[:, 1]
epi = us[:, 3] - us[:, 2]
n_on_cu = np.array([per_cu[k] for k in key.tolist()])
for n in sorted(set(n_on_cu.tolist())):
    m = n_on_cu == n
    print(f"workgroups on CUs with {n} tile(s): {m.sum():4d}  K loop p50 {np.percentile(loop[m], 50):6.2f} us  epilogue p50 {np.percentile(epi[m], 50):5.2f} us  end p50 {np.percentile(us[m, 3], 50):6.2f}")
print("per XCC: tiles, K loop p50 / max of the workgroups that share a CU, end max")
for x in sorted(set(xcc.tolist())):
    m = (xcc == x) & (n_on_cu == 2)
    print(f"  xcc {x}: {int((xcc == x).sum()):3d} tiles  loop p50 {np.percentile(loop[m], 50):6.2f}  max {loop[m].max():6.2f}   end max {us[xcc == x, 3].max():6.2f}")
bi = st[:, 5] >> 16
bj = st[:, 5] & 0xffff
slow = np.argsort(-loop)[:24]
print("slowest K loops (bi, bj, xcc, us):", [(int(bi[i]), int(bj[i]), int(xcc[i]), round(float(loop[i]), 1)) for i in slow])
fast2 = [i for i in np.argsort(loop) if n_on_cu[i] == 2][:12]
print("fastest shared-CU K loops:", [(int(bi[i]), int(bj[i]), int(xcc[i]), round(float(loop[i]), 1)) for i in fast2])
pairs = {}
for i, k in enumerate(key.tolist()):
    pairs.setdefault(k, []).append(i)
rows = []
for k, idx in pairs.items():
    if len(idx) == 2:
        a, b = idx
        rows.append((max(us[a, 2], us[b, 2]), k, round(float(loop[a]), 1), round(float(loop[b]), 1), round(float(us[a, 1]), 1), round(float(us[b, 1]), 1)))
rows.sort()
print("shared CUs, by the time their later K loop ends: (xcc*1000 + se*16 + cu, loop A, loop B, loop-in A, loop-in B)")
for r in rows[:6] + rows[-10:]:
    print("  end %.1f  cu %d  loops %.1f %.1f  in %.1f %.1f" % r)
d = np.array([abs(r[2] - r[3]) for r in rows])
print("loop-time difference inside a CU: p50 %.2f  max %.2f;  later loop end over CUs: p50 %.2f max %.2f" % (np.percentile(d, 50), d.max(), np.percentile([r[0] for r in rows], 50), max(r[0] for r in rows)))
by_se = {}
for r in rows:
    by_se.setdefault((r[1] // 1000, (r[1] % 1000) // 16), []).append(r[0])
print("later loop end by (xcc, se):", {k: round(float(np.mean(v)), 1) for k, v in sorted(by_se.items())})
