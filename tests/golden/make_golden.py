"""Regenerates tests/golden/*.npz: seeded synthetic frames and the outputs of the
CPU oracle (oracle/rslam_oracle.c, reference-structure mode) on them.

The reference itself cannot be built in this image (no Eigen/OpenCV/ROS) and ships
no fixtures, so these vectors come from our restatement of its algorithm
("parity unpinned", see oracle/rslam_oracle.h); they pin the oracle against
regressions and give the GPU tests stored inputs *and* expected outputs that do
not depend on the numpy version of the generator.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from ransac_slam_amd import default_config            # noqa: E402
from ransac_slam_amd.synth import make_frame            # noqa: E402
from oracle import pyoracle as po                       # noqa: E402

CASES = [  # name, make_frame kwargs
    ("L4_H16", dict(L=4, H=16, seed=101)),
    ("L16_H200", dict(L=16, H=200, seed=103)),
    ("L30_H200_partial_ic", dict(L=30, H=200, seed=105, frac_ic=0.7)),
    ("L24_H64_mixed", dict(L=24, H=64, seed=106, frac_cartesian=0.35)),
]
MODES = [(1, 1), (0, 1), (0, 0)]   # (compat, adaptive)


def main():
    po.build()
    for name, kw in CASES:
        fr = make_frame(**kw)
        out = dict(types=fr.types, x_pred=fr.x_pred, P_pred=np.asarray(fr.P_pred), z=fr.z, draws=fr.draws)
        ic = None
        for compat, adaptive in MODES:
            cfg = default_config(compat=compat, adaptive=adaptive)
            o = po.Oracle(cfg, structure=0)
            h, vis, S = o.predict(fr.types, fr.x_pred, fr.P_pred)
            if ic is None:
                ic = (fr.ic & vis).astype(np.uint8)
                out.update(ic=ic, h=h, visible=vis, S=S)
            tag = f"c{compat}a{adaptive}"
            try:
                r = o.ransac_update(fr.z, ic, fr.draws)
            except po.OracleError as e:
                out[f"{tag}_error"] = np.int32(e.code)
                continue
            sup, pos, masks = o.supports()
            sm, rm = o.margins()
            out[f"{tag}_error"] = np.int32(0)
            out[f"{tag}_supports"] = sup
            out[f"{tag}_positions"] = pos
            out[f"{tag}_masks"] = masks
            out[f"{tag}_margins"] = np.array([sm, rm])
            out[f"{tag}_scalars"] = np.array([r["best_hyp"], r["best_support"], r["hyps_evaluated"]], np.int32)
            out[f"{tag}_li"] = r["li"]
            out[f"{tag}_hi"] = r["hi"]
            out[f"{tag}_x_new"] = r["x_new"]
            if adaptive:      # the non-adaptive mode pins supports/masks; keep the fixtures small
                out[f"{tag}_P_new"] = np.asarray(r["P_new"])
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, "n =", fr.n, "->", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
