"""Synthetic EKF-SLAM frames for the 1-point-RANSAC update benchmark and tests.

No dataset can be fetched here and BASELINE config 0 (the PGM sequence) cannot
run, so frames are generated: a camera near the origin, L inverse-depth (or a mix
with Cartesian) landmarks seen in the 320x240 image of
examples/Monocular/initialize_param.yaml, a dense SPD prior covariance whose
dominant uncertainty is the camera pose (as in a converged EKF-SLAM map), and
measurements z = h(x_true) + noise with a fraction of gross outliers.

The camera functions below restate ExtendKF.cpp:91-102 (q2r), :153-174 (hu),
:175-204 (distort_fm), :266-285 (undistort_fm) in numpy; they are generator
utilities, independent of both the HIP path and the C oracle.
"""
from dataclasses import dataclass

import numpy as np

from .ctypes_defs import FEAT_CARTESIAN, FEAT_INVERSE_DEPTH, default_camera

SEED_BASE = 0x5EED0000

# BASELINE.json configs (index = config id): (L, H)
CONFIGS = {
    1: dict(name="C2: 100 landmarks x 200 hypotheses", L=100, H=200),
    2: dict(name="C3: 300 landmarks x 1000 hypotheses", L=300, H=1000),
    3: dict(name="C4: 300 landmarks x 4000 hypotheses (sharded)", L=300, H=4000),
    4: dict(name="C5: 1000 landmarks, r forced large", L=1000, H=1000),
}


def q2r(q):
    r, x, y, z = q
    return np.array([
        [r*r + x*x - y*y - z*z, 2*(x*y - r*z),         2*(z*x + r*y)],
        [2*(x*y + r*z),         r*r - x*x + y*y - z*z, 2*(y*z - r*x)],
        [2*(z*x - r*y),         2*(y*z + r*x),         r*r - x*x - y*y + z*z]])


def undistort(cam, uvd):
    xd = (uvd[..., 0] - cam.Cx) * cam.dx
    yd = (uvd[..., 1] - cam.Cy) * cam.dy
    rd2 = xd * xd + yd * yd
    D = 1 + cam.k1 * rd2 + cam.k2 * rd2 * rd2
    return np.stack([xd * D / cam.dx + cam.Cx, yd * D / cam.dy + cam.Cy], axis=-1)


def distort(cam, uv):
    xu = (uv[..., 0] - cam.Cx) * cam.dx
    yu = (uv[..., 1] - cam.Cy) * cam.dy
    ru = np.sqrt(xu * xu + yu * yu)
    rd = ru / (1 + cam.k1 * ru**2 + cam.k2 * ru**4)
    for _ in range(10):
        f = rd + cam.k1 * rd**3 + cam.k2 * rd**5 - ru
        fp = 1 + 3 * cam.k1 * rd**2 + 5 * cam.k2 * rd**4
        rd = rd - f / fp
    D = 1 + cam.k1 * rd**2 + cam.k2 * rd**4
    return np.stack([xu / D / cam.dx + cam.Cx, yu / D / cam.dy + cam.Cy], axis=-1)


def project(cam, x, types, offsets):
    """h_i(x) for every feature (no visibility gate), L x 2."""
    t = x[0:3]
    R = q2r(x[3:7])
    out = np.zeros((len(types), 2))
    for i, (ty, o) in enumerate(zip(types, offsets)):
        if ty == FEAT_INVERSE_DEPTH:
            y = x[o:o + 6]
            m = np.array([np.cos(y[4]) * np.sin(y[3]), -np.sin(y[4]), np.cos(y[4]) * np.cos(y[3])])
            hrl = R.T @ ((y[0:3] - t) * y[5] + m)
        else:
            hrl = np.linalg.solve(R, x[o:o + 3] - t)
        uv = np.array([cam.Cx + (hrl[0] / hrl[2]) * cam.f / cam.dx,
                       cam.Cy + (hrl[1] / hrl[2]) * cam.f / cam.dy])
        out[i] = distort(cam, uv)
    return out


@dataclass
class Frame:
    types: np.ndarray      # uint8 L
    offsets: np.ndarray    # int32 L
    n: int
    x_pred: np.ndarray     # n
    P_pred: np.ndarray     # n x n, Fortran order (column-major)
    z: np.ndarray          # L x 2 (row i = feature i)
    ic: np.ndarray         # uint8 L
    draws: np.ndarray      # H
    x_true: np.ndarray
    outlier: np.ndarray    # bool L (ground truth)
    diagD: np.ndarray = None   # P_pred = diag(diagD) + U U^T (kept so that further measurements can be drawn)
    U: np.ndarray = None

    @property
    def L(self):
        return len(self.types)


def make_frame(L=16, H=32, seed=0, frac_cartesian=0.0, frac_outlier=0.2, frac_ic=1.0,
               meas_sigma=0.5, cam=None, pose_sigma_scale=1.0) -> Frame:
    """One synthetic frame.  Deterministic in (arguments, numpy version)."""
    cam = cam or default_camera()
    rng = np.random.Generator(np.random.PCG64(SEED_BASE + seed))
    types = np.where(rng.random(L) < frac_cartesian, FEAT_CARTESIAN, FEAT_INVERSE_DEPTH).astype(np.uint8)
    widths = np.where(types == FEAT_INVERSE_DEPTH, 6, 3)
    offsets = (13 + np.concatenate([[0], np.cumsum(widths)[:-1]])).astype(np.int32)
    n = int(13 + widths.sum())

    x = np.zeros(n)
    x[0:3] = rng.normal(0, 0.01, 3)
    q = np.array([1.0, 0, 0, 0]) + rng.normal(0, 0.01, 4)
    x[3:7] = q / np.linalg.norm(q)
    x[7:13] = rng.normal(0, 0.01, 6)
    Rwc = q2r(x[3:7])

    # landmarks: pixel uniform in the image interior, depth uniform in [1, 10] m
    pix = np.stack([rng.uniform(31, 289, L), rng.uniform(31, 209, L)], axis=-1)
    und = undistort(cam, pix)
    depth = rng.uniform(1.0, 10.0, L)
    diagD = np.zeros(n)
    diagD[0:3] = (0.01 * pose_sigma_scale) ** 2        # camera position, 1 cm
    diagD[3:7] = (0.004 * pose_sigma_scale) ** 2       # quaternion, ~0.5 deg
    diagD[7:13] = 6.25e-4
    for i in range(L):
        # ray in the camera frame through the undistorted pixel, then to the world
        ray_c = np.array([(und[i, 0] - cam.Cx) * cam.dx / cam.f, (und[i, 1] - cam.Cy) * cam.dy / cam.f, 1.0])
        o = offsets[i]
        if types[i] == FEAT_INVERSE_DEPTH:
            anchor = rng.normal(0, 0.05, 3)
            # point = cam + depth*ray (in the world); direction from the anchor
            pw = x[0:3] + Rwc @ (ray_c * depth[i])
            d = pw - anchor
            dist = np.linalg.norm(d)
            m = d / dist
            theta = np.arctan2(m[0], m[2])
            phi = np.arctan2(-m[1], np.hypot(m[0], m[2]))
            x[o:o + 6] = [*anchor, theta, phi, 1.0 / dist]
            diagD[o:o + 3] = 1e-6
            diagD[o + 3:o + 5] = 1e-7
            diagD[o + 5] = (0.02 / dist) ** 2
        else:
            x[o:o + 3] = x[0:3] + Rwc @ (ray_c * depth[i])
            diagD[o:o + 3] = 1e-6 * depth[i] ** 2

    # dense low-rank coupling (map <-> camera correlations of a converged filter)
    U = rng.normal(0, 1.0, (n, 13)) * 3e-4
    U[0:13, :] *= 3.0
    P = (U @ U.T)
    P[np.diag_indices(n)] += diagD
    P = np.asfortranarray(0.5 * (P + P.T))

    # truth ~ N(x, P) sampled through the factors
    x_true = x + np.sqrt(diagD) * rng.normal(0, 1, n) + U @ rng.normal(0, 1, 13)
    h_true = project(cam, x_true, types, offsets)
    outlier = rng.random(L) < frac_outlier
    z = h_true + rng.normal(0, meas_sigma, (L, 2))
    z[outlier] += rng.uniform(-10, 10, (int(outlier.sum()), 2))
    ic = (rng.random(L) < frac_ic).astype(np.uint8)
    draws = rng.random(H)
    return Frame(types=types, offsets=offsets, n=n, x_pred=x, P_pred=P, z=np.ascontiguousarray(z),
                 ic=ic, draws=draws, x_true=x_true, outlier=outlier, diagD=diagD, U=U)


def remeasure(fr: Frame, seed, frac_outlier=0.2, meas_sigma=0.5, H=None, cam=None):
    """Another frame's worth of measurements for the SAME prior (a sequence whose covariance stays
    resident): a new truth ~ N(x_pred, P_pred), z = h(truth) + noise, a fraction of gross outliers,
    new draws.  Returns (z, outlier, draws)."""
    cam = cam or default_camera()
    rng = np.random.Generator(np.random.PCG64(SEED_BASE + 0x0F00000 + seed))
    x_true = fr.x_pred + np.sqrt(fr.diagD) * rng.normal(0, 1, fr.n) + fr.U @ rng.normal(0, 1, fr.U.shape[1])
    h_true = project(cam, x_true, fr.types, fr.offsets)
    outlier = rng.random(fr.L) < frac_outlier
    z = h_true + rng.normal(0, meas_sigma, (fr.L, 2))
    z[outlier] += rng.uniform(-10, 10, (int(outlier.sum()), 2))
    return np.ascontiguousarray(z), outlier, rng.random(H if H is not None else len(fr.draws))


def make_config_frame(config_id: int) -> Frame:
    """The frames of BASELINE.json configs 1..4 (all inverse-depth landmarks)."""
    c = CONFIGS[config_id]
    return make_frame(L=c["L"], H=c["H"], seed=config_id)


def make_match_inputs(cam, h, visible, seed=0, offset_px=2.0, noise=3.0, frac_unmatched=0.15):
    """Synthetic inputs of the NCC search (Tracking::matching): a random 8-bit image of the camera's
    size and, per feature, the 13 x 13 patch the reference would have predicted (pred_patch_fc):
    the image around a point a few pixels from h plus noise, or an unrelated patch for
    `frac_unmatched` of the features.  Returns image (nRows, nCols) uint8, patches (L, 13, 13)
    float64 with patches[f][row, col], truth (L, 2) = (column, row) of the planted match or -1."""
    rng = np.random.default_rng(SEED_BASE + 0x3A7C0000 + seed)
    nR, nC = int(cam.nRows), int(cam.nCols)
    # low-pass filtered noise: correlation peaks are unique but neighbours are not independent
    base = rng.normal(0.0, 1.0, (nR + 8, nC + 8))
    k = np.array([1.0, 4.0, 6.0, 4.0, 1.0]); k /= k.sum()
    for ax in (0, 1):
        base = sum(np.roll(base, s - 2, axis=ax) * k[s] for s in range(5))
    base = base[4:-4, 4:-4]
    image = np.clip(128.0 + 60.0 * base / base.std(), 0, 255).astype(np.uint8)
    L = len(visible)
    patches = np.zeros((L, 13, 13))
    truth = -np.ones((L, 2))
    for f in range(L):
        if not visible[f]:
            continue
        if rng.random() < frac_unmatched:
            patches[f] = rng.uniform(0, 255, (13, 13))
            continue
        x = int(round(h[f, 0] + rng.uniform(-offset_px, offset_px)))
        y = int(round(h[f, 1] + rng.uniform(-offset_px, offset_px)))
        x = min(max(x, 7), nC - 8); y = min(max(y, 7), nR - 8)
        patches[f] = image[y - 6:y + 7, x - 6:x + 7] + rng.normal(0.0, noise, (13, 13))
        truth[f] = (x, y)
    return image, patches, truth


def make_feature_records(cam, fr, seed=0, pose_jitter=0.02):
    """Initialisation records of every feature as Map::initialize_a_features stores them
    (Map.cpp:286-292): uv_when_initialized, R_wc / r_wc_when_initialized of a camera pose near the
    current one (so that the homography of pred_patch_fc stays close to the identity) and a smooth
    random 41 x 41 patch_when_initialized.  Returns uv_f (L,2), R_f (L,3,3), r_f (L,3), patch_f (L,41,41)."""
    rng = np.random.default_rng(SEED_BASE + 0x51AB0000 + seed)
    L = fr.L
    uv_f = np.zeros((L, 2)); R_f = np.zeros((L, 3, 3)); r_f = np.zeros((L, 3)); patch_f = np.zeros((L, 41, 41))
    k = np.array([1.0, 4.0, 6.0, 4.0, 1.0]); k /= k.sum()
    for i in range(L):
        x0 = fr.x_pred.copy()
        x0[0:3] += rng.normal(0, pose_jitter, 3)
        q = x0[3:7] + rng.normal(0, pose_jitter * 0.5, 4)
        x0[3:7] = q / np.linalg.norm(q)
        uv = project(cam, x0, fr.types[i:i + 1], fr.offsets[i:i + 1])[0]
        uv_f[i] = np.round(np.clip(uv, [25, 25], [cam.nCols - 26, cam.nRows - 26]))     # FAST corners are integer pixels
        R_f[i] = q2r(x0[3:7]); r_f[i] = x0[0:3]
        base = rng.normal(0.0, 1.0, (49, 49))
        for ax in (0, 1):
            base = sum(np.roll(base, s - 2, axis=ax) * k[s] for s in range(5))
        base = base[4:-4, 4:-4]
        patch_f[i] = np.clip(128.0 + 60.0 * base / base.std(), 0, 255).astype(np.uint8)   # toMatrixd_atuc of an image crop
    return uv_f, R_f, r_f, patch_f


def make_scene_records(cam, fr, h, visible, seed=0):
    """A static scene consistent with a prediction: a smooth random image and, per feature, the record
    Map::initialize_a_features would have stored had the feature been initialised from THIS image at the
    predicted pixel with the prior's pose (Map.cpp:286-292): uv = round(h), R_wc / r_wc of x_pred, the 41 x 41
    crop around uv.  pred_patch_fc then warps with a homography close to the identity and the NCC search
    finds every visible feature.  Returns image (nRows, nCols) uint8, uv_f, R_f, r_f, patch_f."""
    rng = np.random.default_rng(SEED_BASE + 0x7C3E0000 + seed)
    nR, nC = int(cam.nRows), int(cam.nCols)
    base = rng.normal(0.0, 1.0, (nR + 8, nC + 8))
    k = np.array([1.0, 4.0, 6.0, 4.0, 1.0]); k /= k.sum()
    for ax in (0, 1):
        base = sum(np.roll(base, s - 2, axis=ax) * k[s] for s in range(5))
    base = base[4:-4, 4:-4]
    image = np.clip(128.0 + 60.0 * base / base.std(), 0, 255).astype(np.uint8)
    L = fr.L
    uv_f = np.zeros((L, 2)); R_f = np.zeros((L, 3, 3)); r_f = np.zeros((L, 3)); patch_f = np.zeros((L, 41, 41))
    Rwc = q2r(fr.x_pred[3:7])
    padded = np.pad(image, 20, mode="edge").astype(np.float64)
    for i in range(L):
        u, v = (h[i] if visible[i] else (nC / 2, nR / 2))
        u = int(min(max(round(float(u)), 0), nC - 1)); v = int(min(max(round(float(v)), 0), nR - 1))
        uv_f[i] = (u, v); R_f[i] = Rwc; r_f[i] = fr.x_pred[:3]
        patch_f[i] = padded[v:v + 41, u:u + 41]                  # rows = image rows around v, columns around u
    return image, uv_f, R_f, r_f, patch_f
