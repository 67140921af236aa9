"""Regenerates tests/golden/c5/c5_compat{0,1}.npz: BASELINE config C5 (1000 landmarks, n = 6013, 1000 hypotheses)
through the CPU oracle, so that the device's C5 frame is checked against the oracle instead of properties only.

The posterior covariance is 289 MB and is not committable; the fixture keeps what pins it: every integer output
(supports, 64-bit masks, consensus scalars, LI / HI flags), x_k_k, the whole diagonal of p_k_k, 8192 seeded sample
entries (half of them in the rows of the quaternion block, whose magnitudes are six orders below the largest
entry), the Frobenius norm and the trace.  The inputs are `make_frame(L=1000, H=1000, seed=4)` (ransac_slam_amd/synth.py,
the generator of every other C5 figure) and are stored in c5_inputs.npz -- the prior covariance in the factored form the
generator builds it from, P = sym(U U^T) + diag(D) with U n x 13 (the generator's BLAS products round differently from
machine to machine, so regenerating the frame elsewhere would not give bit-identical inputs).

Runs the structured oracle (identical arithmetic, structural zeros of H skipped, repeated hypotheses cached) in its
OpenMP build on all host cores: a few minutes.

    make -C oracle omp && python tests/golden/make_golden_c5.py
"""
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
OMP_LIB = os.path.join(ROOT, "oracle", "_build", "librslam_oracle_omp.so")
if "RSLAM_ORACLE_LIB" not in os.environ:      # the oracle binding reads this at import time
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "omp"])
    os.environ["RSLAM_ORACLE_LIB"] = OMP_LIB
    os.environ.setdefault("OMP_NUM_THREADS", str(os.cpu_count() or 1))

import numpy as np                                        # noqa: E402

from ransac_slam_amd import default_config               # noqa: E402
from ransac_slam_amd.synth import make_frame              # noqa: E402
from oracle import pyoracle as po                         # noqa: E402

sys.path.insert(0, HERE)
from c5_samples import sample_indices                     # noqa: E402  (shared with tests/test_gpu_parity.py)


def main():
    os.makedirs(os.path.join(HERE, "c5"), exist_ok=True)
    fr = make_frame(L=1000, H=1000, seed=4)
    ic_saved = None
    for compat in (1, 0):
        t0 = time.time()
        cfg = default_config(compat=compat, adaptive=0)
        o = po.Oracle(cfg, structure=1)
        h, vis, S = o.predict(fr.types, fr.x_pred, fr.P_pred)
        ic = (fr.ic & vis).astype(np.uint8)
        r = o.ransac_update(fr.z, ic, fr.draws)
        sup, pos, masks = o.supports()
        sm, rm = o.margins()
        P = np.asarray(r["P_new"])
        rows, cols = sample_indices(fr.n)
        if ic_saved is None:
            ic_saved = ic
            np.savez_compressed(os.path.join(HERE, "c5", "c5_inputs.npz"), types=fr.types, x_pred=fr.x_pred, z=fr.z, ic=ic,
                                draws=fr.draws, diagD=fr.diagD, U=fr.U)
        assert np.array_equal(ic, ic_saved)
        out = dict(compat=np.int32(compat), visible=vis, ic=ic, h=h, S=S,
                   supports=sup, positions=pos, masks=masks, margins=np.array([sm, rm]),
                   scalars=np.array([r["best_hyp"], r["best_support"], r["hyps_evaluated"]], np.int32),
                   li=r["li"], hi=r["hi"], x_new=r["x_new"], P_diag=np.diag(P).copy(), P_samples=P[rows, cols].copy(),
                   P_fro=np.float64(np.linalg.norm(P)), P_trace=np.float64(np.trace(P)),
                   P_asym=np.float64(np.max(np.abs(P - P.T))))
        path = os.path.join(HERE, "c5", f"c5_compat{compat}.npz")
        np.savez_compressed(path, **out)
        print(f"compat={compat}: n_li={int(r['li'].sum())} n_hi={int(r['hi'].sum())} best_support={r['best_support']} "
              f"margins=({sm:.2e}, {rm:.2e}) -> {os.path.getsize(path) // 1024} KiB in {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
