"""Shader-clock time line of the pivot pipeline (chain workgroup of the persistent sweep), one diagonal block: -DCD_TIMELINE build.
    python -m ransac_slam_amd.build dev tl -DCD_TIMELINE;  python scripts/cd_timeline.py ransac_slam_amd/_dev/tl.so [block]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ransac_slam_amd import default_config, api
api.LIB_PATH_DEBUG = sys.argv[1]
blk = int(sys.argv[2]) if len(sys.argv) > 2 else 2
from ransac_slam_amd.synth import make_frame
fr = make_frame(L=300, H=1000, seed=2)
ctx = api.RslamHip(default_config(compat=1, adaptive=0), debug=True)
ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, fr.ic, fr.draws)
ctx.step_predict(); ctx.sync()
ic = fr.ic & ctx.fetch_prediction()[1]
ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, ic, fr.draws)
L = api.lib(debug=True)
L.rslam_debug_cd_log.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
for _ in range(4):
    ctx.step_frame(False); ctx.sync()
out = (C.c_ulonglong * (2 * 8 * 16 * 4))()
assert L.rslam_debug_cd_log(ctx._h, out) == 0
v = np.array(list(out), dtype=np.int64).reshape(2, 8, 16, 4)
p, t = v[0, blk], v[1, blk]
t0 = p[0, 0]
print("block", blk, ": shader cycles since the panel wave's first look of step 0")
print("step | panel: look issued, inputs ready, posted | T wave 0: step begins, panel flag seen, operands landed, strip posted")
for s in range(16):
    print("%4d | %7d %7d %7d   (wait %5d, work %5d) | %7d %7d %7d %7d" % (s, p[s, 0] - t0, p[s, 1] - t0, p[s, 2] - t0, p[s, 1] - p[s, 0], p[s, 2] - p[s, 1],
          t[s, 0] - t0, t[s, 1] - t0, t[s, 2] - t0, t[s, 3] - t0))
