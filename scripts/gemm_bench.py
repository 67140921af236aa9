import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ransac_slam_amd import default_config, api
if len(sys.argv) > 1:
    api.LIB_PATH = sys.argv[1]
ctx = api.RslamHip(default_config())
dev = torch.device("cuda:0")
for (n, K) in [(1856, 512), (1856, 64), (6016, 1600)]:
    P = torch.randn(n, n, dtype=torch.float64, device=dev)
    Y = torch.randn(K, n, dtype=torch.float64, device=dev)
    Cm = torch.zeros(n, n, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    for kind in ("gemm", "rank"):
        def run():
            if kind == "gemm":
                ctx.k_gemm_nt(n, n, K, 1.0, Y.data_ptr(), n, Y.data_ptr(), n, 0.0, Cm.data_ptr(), n)
            else:
                ctx.k_rank_update(n, K, P.data_ptr(), n, Y.data_ptr(), n, Cm.data_ptr(), n)
        for _ in range(3): run()
        ctx.sync()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps): run()
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
        fl = (2.0 * n * n * K) if kind == "gemm" else (1.0 * n * (n + 64) * K)
        print(f"{os.path.basename(api.LIB_PATH)} {kind} n={n} K={K}: {dt*1e6:8.1f} us  {fl/dt*1e-12:6.2f} TF/s")
