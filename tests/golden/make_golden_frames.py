"""Regenerates tests/golden/rows/real_frames_c{0,1}.npz: the widened rows (pred_patch_fc + matching,
Tracking.cpp:46-69,164-351) on REAL frames of the reference's own sequence (BASELINE config 0's only GPU-relevant
content).  Runs in the build container, where /root/reference/data/images_sequences exists; the fixture holds
DATA only -- three 320 x 240 PGM frames as uint8 arrays, the feature records cropped from the first one, the
filter state, and what the oracle computes from them -- no reference source travels.

What it replays, with the oracle's restatements of the reference functions:
  frame A   filter initialised as ExtendKF::initialize_x_and_p does (ExtendKF.cpp:32-54, initialize_param.yaml:40-47);
            24 features initialised from salient pixels of the frame as Map::initialize_a_features does
            (hinv + add_a_feature_covariance_inverse_depth, Map.cpp:271-312,339-400; record = pixel, pose, 41 x 41 crop);
  frames B, C  (the next two frames of the sequence)  ekf_prediction -> predict_camera_measurements / S_i ->
            pred_patch_fc for every feature -> matching against the real image -> 1-point RANSAC + both updates.
The salient pixels are the maxima of a box-filtered gradient energy on a grid: a stand-in for the FAST corners
of Map.cpp:232-250 (OpenCV is not in the image); which pixels are chosen is an input of the fixture, not a result.

    python tests/golden/make_golden_frames.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from ransac_slam_amd import default_camera, default_config      # noqa: E402
from oracle import pyoracle as po                               # noqa: E402

SEQ = "/root/reference/data/images_sequences"
EPS = np.finfo(float).eps


def read_pgm(path):
    b = open(path, "rb").read()
    assert b[:2] == b"P5"
    vals, i = [], 2
    while len(vals) < 3:
        while b[i:i + 1].isspace():
            i += 1
        if b[i:i + 1] == b"#":
            while b[i:i + 1] != b"\n":
                i += 1
            continue
        j = i
        while not b[j:j + 1].isspace():
            j += 1
        vals.append(int(b[i:j])); i = j
    w, h, mx = vals
    assert mx == 255
    return np.frombuffer(b, np.uint8, w * h, i + 1).reshape(h, w).copy()


def salient_pixels(img, count, margin=30, cell=40):
    """one pixel per grid cell: the maximum of the 5 x 5 box-filtered squared gradient; strongest `count` cells"""
    f = img.astype(np.float64)
    gx = np.zeros_like(f); gy = np.zeros_like(f)
    gx[:, 1:-1] = f[:, 2:] - f[:, :-2]; gy[1:-1, :] = f[2:, :] - f[:-2, :]
    e = gx * gx + gy * gy
    c = np.cumsum(np.cumsum(np.pad(e, ((3, 2), (3, 2))), axis=0), axis=1)
    box = c[5:, 5:] - c[:-5, 5:] - c[5:, :-5] + c[:-5, :-5]
    # corner-ness: both gradient directions present
    cxx = np.cumsum(np.cumsum(np.pad(gx * gx, ((3, 2), (3, 2))), axis=0), axis=1)
    cyy = np.cumsum(np.cumsum(np.pad(gy * gy, ((3, 2), (3, 2))), axis=0), axis=1)
    bxx = cxx[5:, 5:] - cxx[:-5, 5:] - cxx[5:, :-5] + cxx[:-5, :-5]
    byy = cyy[5:, 5:] - cyy[:-5, 5:] - cyy[5:, :-5] + cyy[:-5, :-5]
    score = np.minimum(bxx, byy)
    H, W = img.shape
    cand = []
    for y0 in range(margin, H - margin, cell):
        for x0 in range(margin, W - margin, cell):
            blk = score[y0:min(y0 + cell, H - margin), x0:min(x0 + cell, W - margin)]
            iy, ix = np.unravel_index(np.argmax(blk), blk.shape)
            cand.append((float(blk[iy, ix]), x0 + ix, y0 + iy))
    cand.sort(reverse=True)
    return [(x, y) for _, x, y in cand[:count]]


def q2r(q):
    r, x, y, z = q
    return np.array([[r*r + x*x - y*y - z*z, 2*(x*y - r*z), 2*(z*x + r*y)],
                     [2*(x*y + r*z), r*r - x*x + y*y - z*z, 2*(y*z - r*x)],
                     [2*(z*x - r*y), 2*(y*z + r*x), r*r - x*x - y*y + z*z]])


def main():
    po.build()
    cam = default_camera()
    names = sorted(os.listdir(SEQ))[:3]
    frames = [read_pgm(os.path.join(SEQ, n)) for n in names]
    assert all(f.shape == (cam.nRows, cam.nCols) for f in frames)
    out_dir = os.path.join(HERE, "rows")
    L = 24
    pix = salient_pixels(frames[0], L)
    assert len(pix) == L
    for compat in (1, 0):
        cfg = default_config(compat=compat, adaptive=1)
        # ExtendKF::initialize_x_and_p (p_k_k(5,5) is left at zero there: kept)
        x = np.zeros(13); x[3] = 1.0; x[10:13] = 1e-11
        P = np.zeros((13, 13))
        for i in (0, 1, 2, 3, 4, 6):
            P[i, i] = EPS
        for i in range(7, 13):
            P[i, i] = 0.025 ** 2
        types = np.zeros(0, np.uint8)
        uv_f, R_f, r_f, patch_f = [], [], [], []
        padded = np.pad(frames[0], 20, mode="edge").astype(np.float64)
        for (u, v) in pix:
            uvd = np.array([float(u), float(v)])
            x, P = po.map_add_feature(cam, cfg.sigma_z, x, P, uvd, 1.0, 1.0)
            types = np.append(types, 0).astype(np.uint8)
            uv_f.append(uvd); R_f.append(q2r(x[3:7])); r_f.append(x[:3].copy())
            patch_f.append(padded[v:v + 41, u:u + 41])
        uv_f, R_f, r_f, patch_f = map(np.array, (uv_f, R_f, r_f, patch_f))
        offs = (13 + 6 * np.arange(L)).astype(np.int32)
        rng = np.random.Generator(np.random.PCG64(0x5EED0000 + 77))
        store = dict(compat=np.int32(compat), names=np.array(names), types=types, x0=x, P0=P, pixels=np.array(pix, np.int32),
                     uv_f=uv_f, R_f=R_f, r_f=r_f, patch_f=patch_f.astype(np.uint8), image0=frames[0])
        for k, img in enumerate(frames[1:], start=1):
            xp, Pp = po.ekf_prediction(x, P, 1.0, 0.007, 0.007)
            o = po.Oracle(cfg, structure=1)
            h, vis, S = o.predict(types, xp, Pp)
            patches, status, pm = po.pred_patches(cam, compat, types, offs, xp, h, vis, uv_f, R_f, r_f, patch_f)
            z, ic, corr, mm = po.matching(cam, img, patches, h, vis, S)
            draws = rng.random(64)
            r = o.ransac_update(z, ic, draws)
            sm, rm = o.margins()
            store.update({f"image{k}": img, f"x_pred{k}": xp, f"P_pred{k}": Pp, f"h{k}": h, f"visible{k}": vis, f"S{k}": S,
                          f"patches{k}": patches.astype(np.float32), f"patch_status{k}": status, f"patch_margins{k}": pm,
                          f"z{k}": z, f"ic{k}": ic, f"corr{k}": corr, f"match_margins{k}": mm, f"draws{k}": draws,
                          f"li{k}": r["li"], f"hi{k}": r["hi"], f"x_new{k}": r["x_new"], f"P_new{k}": r["P_new"],
                          f"scalars{k}": np.array([r["best_hyp"], r["best_support"], r["hyps_evaluated"]], np.int32),
                          f"update_margins{k}": np.array([sm, rm])})
            print(f"compat {compat} frame {names[k]}: visible {int(vis.sum())}, warped {int((status == 1).sum())}, "
                  f"matched {int(ic.sum())}, li {int(r['li'].sum())}, hi {int(r['hi'].sum())}, "
                  f"min margins patch {pm[status == 1].min() if (status == 1).any() else np.nan:.2e} match {mm.min():.2e}")
            x, P = r["x_new"], r["P_new"]          # Map management would follow here; the map is left as it is
        path = os.path.join(out_dir, f"real_frames_c{compat}.npz")
        np.savez_compressed(path, **store)
        print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
