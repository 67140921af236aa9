"""First-contact GPU check: every stage against the oracle, verbose."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ransac_slam_amd import default_config
from ransac_slam_amd.api import RslamHip
from ransac_slam_amd.synth import make_frame
from oracle import pyoracle as po

def rel(a, b):
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))

def kernels(ctx):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for (M, N, K) in [(64, 64, 32), (128, 192, 64), (256, 128, 480 // 32 * 32)]:
        A = torch.randn(K, M, dtype=torch.float64, device=dev)   # col-major M x K == row-major K x M
        B = torch.randn(K, N, dtype=torch.float64, device=dev)
        Cm = torch.zeros(N, M, dtype=torch.float64, device=dev)  # col-major M x N
        torch.cuda.synchronize()
        ctx.k_gemm_nt(M, N, K, 1.0, A.data_ptr(), M, B.data_ptr(), N, 0.0, Cm.data_ptr(), M)
        ctx.sync()
        ref = (A.T @ B).T.contiguous()      # (M x N) stored col-major == (N x M) row-major
        ref = (B.T @ A)                      # N x M row-major: element [j,i] = sum_k B[k,j] A[k,i] = C[i,j]
        print(f"gemm_nt {M}x{N}x{K}: max err {float((Cm - ref).abs().max()):.3e}")
    n, r = 200, 70
    NP, KP = 256, 96
    P = torch.randn(NP, NP, dtype=torch.float64, device=dev)
    P[n:, :] = 0; P[:, n:] = 0
    Y = torch.zeros(KP, NP, dtype=torch.float64, device=dev)    # col-major NP x KP
    Y[:r, :n] = torch.randn(r, n, dtype=torch.float64, device=dev)
    C = torch.zeros(NP, NP, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    ctx.k_rank_update(n, r, P.data_ptr(), NP, Y.data_ptr(), NP, C.data_ptr(), NP)
    ctx.sync()
    ref = 0.5 * (P + P.T) - Y.T @ Y
    print(f"rank_update: max err {float((C - ref).abs().max()):.3e}  sym {float((C - C.T).abs().max()):.3e}")
    Pi = P.clone()
    ctx.k_rank_update(n, r, Pi.data_ptr(), NP, Y.data_ptr(), NP, Pi.data_ptr(), NP)
    ctx.sync()
    print(f"rank_update in place: max err {float((Pi - ref).abs().max()):.3e}")

def frame_case(L, H, seed, compat, adaptive, dedup=0, frac_cart=0.0, frac_ic=1.0, structure=0):
    fr = make_frame(L=L, H=H, seed=seed, frac_cartesian=frac_cart, frac_ic=frac_ic)
    cfg = default_config(compat=compat, adaptive=adaptive, dedup=dedup)
    o = po.Oracle(cfg, structure=structure)
    h0, v0, S0 = o.predict(fr.types, fr.x_pred, fr.P_pred)
    ic = fr.ic & v0
    try:
        r0 = o.ransac_update(fr.z, ic, fr.draws)
    except po.OracleError as e:
        print("oracle error", e.code); r0 = None
    sup0, pos0, masks0 = o.supports()
    g = RslamHip(cfg)
    h1, v1, S1 = g.predict(fr.types, fr.x_pred, fr.P_pred)
    okv = np.array_equal(v0, v1)
    vis = v0.astype(bool)
    print(f"[L={L} H={H} seed={seed} compat={compat} adaptive={adaptive} dedup={dedup} cart={frac_cart}] vis eq {okv}; h err {np.max(np.abs(h0[vis]-h1[vis])):.2e}; S rel {rel(S1[vis], S0[vis]):.2e}")
    try:
        r1 = g.ransac_update(fr.z, ic, fr.draws)
    except Exception as e:
        print("   gpu error:", e); return
    if r0 is None: return
    sup1, masks1 = g.fetch_supports()
    ne = len(sup0)
    print(f"   supports eq (first {ne}): {np.array_equal(sup0, sup1[:ne])}  masks eq: {np.array_equal(masks0, masks1[:ne])}  margins {o.margins()}")
    if not np.array_equal(sup0, sup1[:ne]):
        bad = np.flatnonzero(sup0 != sup1[:ne])[:10]
        print("   mismatch at", bad, sup0[bad], sup1[bad])
    for k in ("best_hyp", "best_support", "hyps_evaluated"):
        print(f"   {k}: oracle {r0[k]} gpu {r1[k]}")
    print(f"   li eq {np.array_equal(r0['li'], r1['li'])} ({r0['li'].sum()})  hi eq {np.array_equal(r0['hi'], r1['hi'])} ({r0['hi'].sum()})")
    print(f"   x rel {rel(r1['x_new'], r0['x_new']):.2e}  P rel {rel(r1['P_new'], r0['P_new']):.2e}  P sym {np.max(np.abs(r1['P_new']-r1['P_new'].T)):.1e}")
    g.close()

if __name__ == "__main__":
    cfg = default_config()
    ctx = RslamHip(cfg)
    print("version", ctx and po and "ok")
    kernels(ctx)
    print("mfma f64 peak TF/s:", ctx.mfma_f64_peak())
    print("hbm copy GB/s:", ctx.hbm_copy_peak(1 << 30))
    ctx.close()
    frame_case(16, 32, 0, 0, 0)
    frame_case(16, 32, 0, 1, 0)
    frame_case(50, 200, 1, 0, 1)
    frame_case(50, 200, 1, 1, 1)
    frame_case(40, 64, 2, 0, 0, frac_cart=0.4, frac_ic=0.8)
    frame_case(100, 200, 3, 0, 1, dedup=1)
    frame_case(100, 200, 3, 1, 0, dedup=1)
    t = time.time()
    frame_case(300, 1000, 2, 0, 1, structure=1)
    frame_case(300, 1000, 2, 1, 1, structure=1)
    print("C3 cases took", time.time() - t)
