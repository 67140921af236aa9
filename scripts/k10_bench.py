"""K10 alone: P - Y Y^T (rslam_k_rank_update) at a given size, back-to-back launches, TFLOP/s against n(n+1)r.
    python scripts/k10_bench.py [n r]..."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_slam_amd import api, default_config   # noqa: E402

cases = [(1813, 498), (1813, 64), (1813, 256), (6013, 1600), (613, 160)]
if len(sys.argv) > 2:
    cases = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
ctx = api.RslamHip(default_config())
dev = torch.device("cuda:0")
for n, r in cases:
    NP, KP = -(-n // 64) * 64, max(32, -(-r // 32) * 32)
    g = torch.Generator(device=dev).manual_seed(1)
    P = torch.randn(NP, NP, dtype=torch.float64, device=dev, generator=g)
    Y = torch.zeros(KP, NP, dtype=torch.float64, device=dev)
    Y[:r, :n] = torch.randn(r, n, dtype=torch.float64, device=dev, generator=g) * 1e-2
    C = torch.empty_like(P)
    torch.cuda.synchronize()
    for _ in range(5):
        ctx.k_rank_update(n, r, P.data_ptr(), NP, Y.data_ptr(), NP, C.data_ptr(), NP)
    ctx.sync_stream() if hasattr(ctx, "sync_stream") and ctx.n else torch.cuda.synchronize()
    reps = 50
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.k_rank_update(n, r, P.data_ptr(), NP, Y.data_ptr(), NP, C.data_ptr(), NP)
    ctx.sync()
    us = (time.perf_counter() - t0) / reps * 1e6
    fl = float(n) * (n + 1) * r
    print(f"n={n} r={r}: {us:8.2f} us  {fl / us * 1e-6:6.2f} TFLOP/s ({fl / us * 1e-6 / 78.6:.3f} of 78.6)")
