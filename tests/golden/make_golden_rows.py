"""Regenerates tests/golden/rows/*.npz: stored inputs and oracle outputs of the widened rows
(SURVEY 8f): ekf_prediction, map state surgery, NCC search, patch prediction.  Same status as
tests/golden/make_golden.py: the reference cannot be built here and ships no fixtures, so the vectors
come from our restatement ("parity unpinned"); they pin it against regressions and give the GPU
tests stored inputs and expected outputs.

    python tests/golden/make_golden_rows.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from ransac_slam_amd import default_camera, default_config                      # noqa: E402
from ransac_slam_amd.synth import make_frame, make_match_inputs, make_feature_records   # noqa: E402
from oracle import pyoracle as po                                               # noqa: E402


def main():
    po.build()
    cam = default_camera()
    out_dir = os.path.join(HERE, "rows")
    os.makedirs(out_dir, exist_ok=True)

    # --- ekf_prediction + map surgery on a small mixed map
    fr = make_frame(L=8, H=2, seed=2101, frac_cartesian=0.25)
    x, P = fr.x_pred.copy(), np.asarray(fr.P_pred).copy()
    x[7:13] += [0.1, 0.0, -0.2, 0.02, -0.01, 0.03]
    xp, Pp = po.ekf_prediction(x, P, 1.0, 0.007, 0.007)
    xd, Pd = po.map_delete_feature(fr.types, x, P, 3)
    ids = np.flatnonzero(fr.types == 0)
    Pc = P.copy(); o = int(fr.offsets[ids[1]]); Pc[o + 5, :] *= 1e-3; Pc[:, o + 5] *= 1e-3
    conv, xc, Pcv = po.map_convert(fr.types, x, Pc, 1e-3)
    xa, Pa = po.map_add_feature(cam, 1.0, x, P, np.array([123.0, 77.0]), 1.0, 1.0)
    np.savez_compressed(os.path.join(out_dir, "state_rows_L8.npz"), types=fr.types, x=x, P=P,
                        pred_x=xp, pred_P=Pp, del_feature=np.int32(3), del_x=xd, del_P=Pd,
                        conv_P_in=Pc, conv_threshold=1e-3, conv_index=np.int32(conv), conv_x=xc, conv_P=Pcv,
                        add_uvd=np.array([123.0, 77.0]), add_x=xa, add_P=Pa)

    # --- patch prediction + NCC search on a 12-feature frame
    fr = make_frame(L=12, H=2, seed=2102, frac_cartesian=0.25)
    for compat in (1, 0):
        o = po.Oracle(default_config(compat=compat), structure=1)
        h, vis, S = o.predict(fr.types, fr.x_pred, fr.P_pred)
        uv_f, R_f, r_f, patch_f = make_feature_records(cam, fr, seed=31)
        patches, status, pm = po.pred_patches(cam, compat, fr.types, fr.offsets, fr.x_pred, h, vis, uv_f, R_f, r_f, patch_f)
        image, mpatches, truth = make_match_inputs(cam, h, vis, seed=32)
        z, ic, corr, mm = po.matching(cam, image, mpatches, h, vis, S)
        np.savez_compressed(os.path.join(out_dir, f"image_rows_L12_c{compat}.npz"), compat=np.int32(compat), types=fr.types,
                            x_pred=fr.x_pred, P_pred=np.asarray(fr.P_pred), h=h, visible=vis, S=S,
                            uv_f=uv_f, R_f=R_f, r_f=r_f, patch_f=patch_f.astype(np.uint8),
                            patches=patches.astype(np.float32), patch_status=status, patch_margins=pm,
                            image=image, match_patches=mpatches.astype(np.float32), z=z, ic=ic, corr=corr, match_margins=mm)
    for f in sorted(os.listdir(out_dir)):
        print(f, os.path.getsize(os.path.join(out_dir, f)), "bytes")


if __name__ == "__main__":
    main()
