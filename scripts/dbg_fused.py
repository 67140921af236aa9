"""Diagnostic: which tiles of the posterior differ from the oracle (fused update)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_slam_amd import api, default_config
from ransac_slam_amd.synth import make_frame
from oracle import pyoracle as po
L, H, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
fr = make_frame(L=L, H=H, seed=seed)
for compat in (1, 0):
    cfg = default_config(compat=compat, adaptive=0)
    o = po.Oracle(cfg, structure=1)
    _, v0, _ = o.predict(fr.types, fr.x_pred, fr.P_pred)
    ic = (fr.ic & v0).astype(np.uint8)
    r0 = o.ransac_update(fr.z, ic, fr.draws)
    for mask in (0, 64, 128):
        api.lib(debug=True).rslam_debug_set_sweep_exp(mask)
        g = api.RslamHip(cfg, debug=True)
        g.predict(fr.types, fr.x_pred, fr.P_pred)
        try:
            r1 = g.ransac_update(fr.z, ic, fr.draws)
        except Exception as e:
            print("compat", compat, "mask", mask, "ERR", e); g.close(); continue
        d = np.abs(r1["P_new"] - r0["P_new"]); tol = 1e-9 * np.abs(r0["P_new"]).max()
        n = fr.n; nT = (n + 63) // 64
        bad = [(bi, bj) for bi in range(nT) for bj in range(nT) if d[64*bi:64*bi+64, 64*bj:64*bj+64].max() > tol]
        print("compat", compat, "mask", mask, "n_li", int(r1["li"].sum()), "n_hi", int(r1["hi"].sum()), "bad tiles", len(bad), bad[:12],
              "dx", np.abs(r1["x_new"] - r0["x_new"]).max())
        g.close()
api.lib(debug=True).rslam_debug_set_sweep_exp(-1)
