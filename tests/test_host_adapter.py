"""The C++ host-side mirror of the reference interface (ransac_slam_amd/host/ransac_slam_hip.hpp):
the five hot calls of System::TrackRunning driven from C++ through the C ABI, checked against
the oracle.  Needs an MI355X."""
import os
import struct
import subprocess

import numpy as np
import pytest

from ransac_slam_amd import default_camera, default_config
from ransac_slam_amd.synth import make_frame

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("compat", [1, 0])
def test_cpp_adapter_frame(oracle_lib, tmp_path, compat):
    from ransac_slam_amd import build
    build.build()
    exe = build.build_host_example()
    fr = make_frame(L=60, H=1100, seed=501, frac_ic=0.9)
    cfg = default_config(compat=compat, adaptive=1)
    o = oracle_lib.Oracle(cfg, structure=1)
    h0, v0, S0 = o.predict(fr.types, fr.x_pred, fr.P_pred)
    ic = (fr.ic & v0).astype(np.uint8)
    r0 = o.ransac_update(fr.z, ic, fr.draws)
    fin, fout = tmp_path / "frame.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("4i", fr.n, fr.L, len(fr.draws), compat))
        f.write(fr.types.tobytes()); f.write(fr.ic.astype(np.uint8).tobytes())
        f.write(fr.x_pred.tobytes()); f.write(np.asfortranarray(fr.P_pred).tobytes(order="F"))
        f.write(np.ascontiguousarray(fr.z).tobytes()); f.write(fr.draws.tobytes())
    fmap = tmp_path / "map.bin"
    subprocess.check_call([exe, str(fin), str(fout), str(fmap)], timeout=120)
    raw = open(fout, "rb").read()
    n, L = fr.n, fr.L
    sc = np.frombuffer(raw, np.int32, 3); p = 12
    li = np.frombuffer(raw, np.uint8, L, p); p += L
    hi = np.frombuffer(raw, np.uint8, L, p); p += L
    has_h = np.frombuffer(raw, np.uint8, L, p); p += L
    h = np.frombuffer(raw, np.float64, 2 * L, p).reshape(L, 2); p += 16 * L
    S = np.frombuffer(raw, np.float64, 4 * L, p).reshape(L, 4); p += 32 * L
    x = np.frombuffer(raw, np.float64, n, p); p += 8 * n
    P = np.frombuffer(raw, np.float64, n * n, p).reshape(n, n, order="F")
    assert np.array_equal(has_h, v0)
    v = v0.astype(bool)
    assert np.allclose(h[v], h0[v], atol=1e-9) and np.allclose(S[v], S0[v], rtol=1e-10)
    assert list(sc) == [r0["best_hyp"], r0["best_support"], r0["hyps_evaluated"]]
    assert np.array_equal(li, r0["li"]) and np.array_equal(hi, r0["hi"])
    assert np.max(np.abs(x - r0["x_new"])) <= 1e-9 * max(1.0, np.abs(r0["x_new"]).max())
    assert np.max(np.abs(P - r0["P_new"])) <= 1e-9 * np.abs(r0["P_new"]).max()

    # the Map mirror: delete feature 3 (1-based), no conversion, one insertion, prediction -- all on the device
    x0, P0 = oracle_lib.map_delete_feature(fr.types, r0["x_new"], r0["P_new"], 2)
    t0 = np.delete(fr.types, 2)
    x0, P0 = oracle_lib.map_add_feature(default_camera(), cfg.sigma_z, x0, P0, np.array([140.0, 100.0]), 1.0, 1.0)
    t0 = np.append(t0, 0).astype(np.uint8)
    xp0, Pp0 = oracle_lib.ekf_prediction(x0, P0, 1.0, 0.007, 0.007)
    raw = open(fmap, "rb").read()
    n2, L2, converted, visible = np.frombuffer(raw, np.int32, 4); p = 16
    assert n2 == len(xp0) and L2 == len(t0) and converted == -1 and 0 < visible <= L2 - 1
    assert np.array_equal(np.frombuffer(raw, np.uint8, L2, p), t0); p += L2
    xp = np.frombuffer(raw, np.float64, n2, p); p += 8 * n2
    Pp = np.frombuffer(raw, np.float64, n2 * n2, p).reshape(n2, n2, order="F")
    assert np.max(np.abs(xp - xp0)) <= 1e-9 * max(1.0, np.abs(xp0).max())
    assert np.max(np.abs(Pp - Pp0)) <= 1e-9 * np.abs(Pp0).max()
