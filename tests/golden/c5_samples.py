"""Sample positions of the C5 fixtures (tests/golden/c5/): shared by make_golden_c5.py and tests/test_gpu_parity.py."""
import numpy as np

N_SAMPLES = 8192


def sample_indices(n, seed=20240):
    """(rows, cols): half of the samples in rows 3..6 (the quaternion block, six orders below the largest entry of
    p_k_k), the rest anywhere; seeded"""
    rng = np.random.default_rng(seed)
    rq = rng.integers(3, 7, N_SAMPLES // 2); cq = rng.integers(0, n, N_SAMPLES // 2)
    ra = rng.integers(0, n, N_SAMPLES // 2); ca = rng.integers(0, n, N_SAMPLES // 2)
    return np.concatenate([rq, ra]).astype(np.int64), np.concatenate([cq, ca]).astype(np.int64)
