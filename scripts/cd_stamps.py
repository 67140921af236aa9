import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ransac_slam_amd import default_config, api
api.LIB_PATH_DEBUG = sys.argv[1]          # a -DRSLAM_DEBUG -DCD_STAMPS build (ransac_slam_amd/build.py: extra_flags)
from ransac_slam_amd.synth import make_frame
fr = make_frame(L=300, H=1000, seed=2)
ctx = api.RslamHip(default_config(compat=1, adaptive=0), debug=True)
ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, fr.ic, fr.draws)
L = api.lib(debug=True)
L.rslam_debug_cd_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 16)()
for _ in range(3):
    ctx.step_frame(False); ctx.sync()
L.rslam_debug_cd_stamps(ctx._h, out, 1)
N = 5
for _ in range(N):
    ctx.step_frame(False); ctx.sync()
L.rslam_debug_cd_stamps(ctx._h, out, 1)
v = np.array(list(out), dtype=float)
steps = N * (1 + 7 * 16 + 13)      # LI: 1 pivot step; HI: 7 full blocks + 13 steps
names = ["chain wait", "chain lds read+strip", "chain post", "T wave wait", "T wave total", "chain chol", "chain total", "chain lds write"]
for n, x in zip(names, v[:8]):
    print(f"{n:22s} {x / steps:9.1f} cycles per 4-pivot step  ({100 * x / max(v[6], 1):5.1f} %)")
calls = v[13] if v[13] > 0 else 1          # one tick per cd_factor_block call of thread 0
for n, k in (("stage operands", 8), ("X product", 9), ("pending + init (A..C)", 10), ("chain", 11), ("whole block", 12)):
    print(f"{n:22s} {v[k] / calls:9.1f} cycles per diagonal block ({int(calls)} blocks; pending paths only where used)")
