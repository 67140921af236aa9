"""Who waits for whom in the pivot pipeline of the chain workgroup: polls that found their flag not yet up (-DCD_SPINS build).
    python -m ransac_slam_amd.build dev spins -DCD_SPINS;  python scripts/cd_spins.py ransac_slam_amd/_dev/spins.so [compat]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ransac_slam_amd import default_config, api
api.LIB_PATH_DEBUG = sys.argv[1]
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
N = 10
ctx.enable_timing(True)
fh = []
for _ in range(N):
    ctx.step_frame(False); ctx.sync(); fh.append(ctx.timings()["factor_hi_us"])
L.rslam_debug_cd_stamps(ctx._h, out, 1)
v = np.array(list(out), dtype=float)
print("factor_hi_us mean %.2f" % np.mean(fh))
for name, a, b in (("panel wave: extra looks per step", 0, 1), ("T wave 0: failed polls per step (~170 cycles each)", 2, 3),
                   ("M wave 0: failed polls per step", 4, 5), ("inverse wave: failed polls per step", 6, 7)):
    print(f"{name:55s} {v[a] / max(v[b], 1):7.2f}   ({int(v[b])} steps)")
