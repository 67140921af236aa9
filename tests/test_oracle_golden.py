"""The oracle against the committed golden vectors (tests/golden/*.npz, produced by
tests/golden/make_golden.py): integer outputs bit-exact, floating point to 1e-12."""
import glob
import os

import numpy as np
import pytest

from ransac_slam_amd import default_config

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
MODES = [(1, 1), (0, 1), (0, 0)]


def test_golden_files_present():
    assert len(GOLDEN) >= 4


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
@pytest.mark.parametrize("structure", [0, 1])
def test_oracle_reproduces_golden(oracle_lib, path, structure):
    g = np.load(path)
    for compat, adaptive in MODES:
        tag = f"c{compat}a{adaptive}"
        o = oracle_lib.Oracle(default_config(compat=compat, adaptive=adaptive), structure=structure)
        h, vis, S = o.predict(g["types"], g["x_pred"], g["P_pred"])
        assert np.array_equal(vis, g["visible"])
        v = vis.astype(bool)
        assert np.array_equal(h[v], g["h"][v])
        assert np.allclose(S[v], g["S"][v], rtol=1e-12, atol=0)
        err = int(g[f"{tag}_error"])
        if err:
            with pytest.raises(oracle_lib.OracleError) as e:
                o.ransac_update(g["z"], g["ic"], g["draws"])
            assert e.value.code == err
            continue
        r = o.ransac_update(g["z"], g["ic"], g["draws"])
        sup, pos, masks = o.supports()
        assert np.array_equal(sup, g[f"{tag}_supports"])
        assert np.array_equal(pos, g[f"{tag}_positions"])
        assert np.array_equal(masks, g[f"{tag}_masks"])
        assert [r["best_hyp"], r["best_support"], r["hyps_evaluated"]] == list(g[f"{tag}_scalars"])
        assert np.array_equal(r["li"], g[f"{tag}_li"]) and np.array_equal(r["hi"], g[f"{tag}_hi"])
        assert np.allclose(r["x_new"], g[f"{tag}_x_new"], rtol=1e-12, atol=1e-14)
        if f"{tag}_P_new" in g:
            assert np.allclose(r["P_new"], g[f"{tag}_P_new"], rtol=1e-9, atol=1e-16)
        # margin audit: no scored pair sits within 1e-9 px of the threshold, no rescue
        # candidate within 1e-9 of the chi-square gate => inlier sets are well defined
        # independently of summation order / libm differences
        sm, rm = g[f"{tag}_margins"]
        assert sm > 1e-9 and rm > 1e-9
