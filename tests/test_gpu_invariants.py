"""The two hand-placed relaxations the pivot pipeline of the persistent sweep rests on (kernels.hip):

  * the LDS hand-over of the chain workgroup (cd_post / cd_wait) uses compiler barriers instead of workgroup-scope
    release / acquire fences -- it relies on the LDS executing one wave's operations in order;
  * the in-chain fetch of the next diagonal block's inputs is an LDS-DMA written out in inline asm, invisible to the
    compiler's wait-count pass, and waited for by hand (cdp_finish: s_waitcnt vmcnt(0) in front of the barrier).

ransac_slam_amd/_dev/fenced.so is the same code with the hardware fences back (-DCD_HW_FENCES) and the transfer as the
compiler's builtin (-DCDP_DMA_BUILTIN), i.e. with every wait the toolchain derives by itself (build.py build_fenced;
__graft_entry__.build() makes it).  If the product's results did depend on either relaxation -- a hand-over read too
early, a tile consumed before its transfer landed -- they would differ from the conservative twin's; the pipeline is
deterministic (test_resident_frame_is_bitwise_reproducible), so the comparison is BITWISE: same x_k_k, same p_k_k, same
decisions, for the headline frame in both arithmetic modes and for a small multi-block frame."""
import os

import numpy as np
import pytest

from ransac_slam_amd import default_config
from ransac_slam_amd.synth import make_frame

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def twin():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (the product path has no CPU fallback)")
    from ransac_slam_amd import api, build
    if not os.path.exists(build.FENCED):
        pytest.fail("%s is missing: __graft_entry__.build() (build.build_fenced) makes it" % build.FENCED)
    api.lib()
    api.lib_at(build.FENCED)
    return api, build.FENCED


def _frame_results(api, fr, cfg, lib_path, replays):
    c = api.RslamHip(cfg, lib_path=lib_path) if lib_path else api.RslamHip(cfg)
    _, vis, _ = c.predict(fr.types, fr.x_pred, fr.P_pred)
    ic = (fr.ic & vis).astype(np.uint8)
    c.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, ic, fr.draws)
    out = []
    for i in range(replays):
        c.step_frame(i % 2 == 0)                 # hipGraph replay and launch by launch
        c.sync()
        out.append(c.fetch_results())
    mode = c.update_mode()
    c.close()
    return out, mode


@pytest.mark.parametrize("case", [("C3", 300, 1000, 2, 1), ("C3", 300, 1000, 2, 0), ("small multi-block", 70, 300, 41, 0)],
                         ids=lambda c: "%s_compat%d" % (c[0].replace(" ", "_"), c[4]))
def test_product_equals_fenced_twin_bitwise(twin, case):
    api, fenced = twin
    _, L, H, seed, compat = case
    fr = make_frame(L=L, H=H, seed=seed)
    cfg = default_config(compat=compat, adaptive=1)
    prod, mode_p = _frame_results(api, fr, cfg, None, 4)
    cons, mode_c = _frame_results(api, fr, cfg, fenced, 4)
    assert mode_p == mode_c == 2                 # both took the fused persistent sweep (the code under test)
    ref = prod[0]
    n_in = int(ref["li"].sum()) + int(ref["hi"].sum())
    assert 2 * n_in > 64                         # more than one diagonal block: the chain hands over between blocks
    for r in prod[1:] + cons:
        for k in ("best_hyp", "best_support", "hyps_evaluated", "n_li", "n_hi"):
            assert r[k] == ref[k], k
        assert np.array_equal(r["li"], ref["li"]) and np.array_equal(r["hi"], ref["hi"])
        assert np.array_equal(r["x_new"], ref["x_new"])
        assert np.array_equal(r["P_new"], ref["P_new"])
