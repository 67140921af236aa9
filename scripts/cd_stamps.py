"""Where the cycles of a pivot step go (chain workgroup of the persistent sweep): accumulated s_memtime differences of a
-DCD_STAMPS build.   python -m ransac_slam_amd.build dev stamps -DCD_STAMPS;  python scripts/cd_stamps.py ransac_slam_amd/_dev/stamps.so"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ransac_slam_amd import default_config, api
api.LIB_PATH_DEBUG = sys.argv[1]          # a -DRSLAM_DEBUG -DCD_STAMPS build (ransac_slam_amd/build.py build_dev)
compat = int(sys.argv[2]) if len(sys.argv) > 2 else 1
from ransac_slam_amd.synth import make_frame
fr = make_frame(L=300, H=1000, seed=2)
ctx = api.RslamHip(default_config(compat=compat, adaptive=0), debug=True)
ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, fr.ic, fr.draws)
ctx.step_predict(); ctx.sync()
ic = fr.ic & ctx.fetch_prediction()[1]
ctx.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, ic, fr.draws)
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
r = ctx.fetch_results(want_P=False)
steps_p = v[6] and None
names = {0: "panel: flags + loads", 1: "panel: strip update", 5: "panel: factor", 7: "panel: lds write", 2: "panel: post", 6: "panel: whole step",
         3: "T wave 0: wait for panel", 15: "T wave 0: wait + mfma + strip + post", 4: "T wave 0: whole step (incl. lazy updates, fetch)"}
tsteps = v[14] if v[14] > 0 else 1
print("n_li", r["n_li"], "n_hi", r["n_hi"], " T-wave steps counted:", int(tsteps), "in", N, "frames")
for k in (0, 1, 5, 7, 2, 6, 3, 15, 4):
    print(f"{names[k]:50s} {v[k] / tsteps:9.1f} shader cycles per 4-pivot step")
