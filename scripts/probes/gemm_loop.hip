// Where does the inner loop of tile_gemm_nt_dma lose its rate?  Ablations of the K loop of the 64 x 64 tile engine
// (ransac_slam_amd/csrc/tile_gemm.h) on a full chip: no epilogue, long K, all workgroups resident at once.
// build: hipcc --offload-arch=gfx950 -O3 -Iransac_slam_amd/csrc scripts/probes/gemm_loop.hip -o scripts/probes/gemm_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "tile_gemm.h"
using namespace rslam;


// td_compute_chunk with (ILV) one DMA pair of the next chunk issued after each of the first k-steps instead of all up
// front, and (SGB) the LDS reads of k-step s+1 pinned between the MFMAs of k-step s
template <bool ILV, bool SGB, int RD = 0>
__device__ __forceinline__ void compute_chunk_x(const double* As, const double* Bs, TgAcc& acc,
                                                const double* __restrict__ A, const double* __restrict__ B, long ld, int k0, double* An, double* Bn, bool more)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / TG_WN, wn = wave % TG_WN;
    const int kq = lane >> 4, ij = lane & 15;
    const int koff = (kq & 1) * TD_SEG + (kq >> 1) * 64;
    const double* ap = As + koff + wm * 16 * TG_MI + ij;
    const double* bb = Bs + koff + wn * 16 * TG_NI;
    const double* bp0 = bb + ij;
    const double* bp1 = bb + ((ij - 4) & 15);
    const double* bp2 = bb + ((ij - 8) & 15);
    const double* bp3 = bb + ((ij - 12) & 15);
    const int r = 2 * (lane & 31), up = lane >> 5;
    double a[TG_MI];
    BFrag b[TG_NI];
#pragma unroll
    for (int mi = 0; mi < TG_MI; ++mi) a[mi] = ap[16 * mi];
#pragma unroll
    for (int ni = 0; ni < TG_NI; ++ni) b[ni] = BFrag{ bp0[16 * ni], bp1[16 * ni], bp2[16 * ni], bp3[16 * ni] };
#pragma unroll
    for (int kk = 0; kk < TG_KC; kk += 4) {
        double na[TG_MI];
        BFrag nb[TG_NI];
        if (ILV && more && kk / 4 < TG_KC / 8) {
            const int q = kk / 4;
            const int sg = wave + 4 * q;
            const long k = k0 + 4 * (sg >> 1) + (sg & 1) + 2 * up;
            __builtin_amdgcn_global_load_lds((tg_glb_void*)(A + r + k * ld), (tg_lds_void*)(An + sg * TD_SEG), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((tg_glb_void*)(B + r + k * ld), (tg_lds_void*)(Bn + sg * TD_SEG), 16, 0, 0);
        }
        if (kk + 4 < TG_KC) {
            const int o = (kk / 4 + 1) * TD_STEP;
#pragma unroll
            for (int mi = 0; mi < TG_MI; ++mi) na[mi] = ap[o + 16 * mi];
#pragma unroll
            for (int ni = 0; ni < TG_NI; ++ni) {
                if (RD == 1) { const double v = bp0[o + 16 * ni]; nb[ni] = BFrag{ v, v, v, v }; }
                else nb[ni] = BFrag{ bp0[o + 16 * ni], bp1[o + 16 * ni], bp2[o + 16 * ni], bp3[o + 16 * ni] };
            }
        }
        if (RD == 2) {
#pragma unroll
            for (int mi = 0; mi < TG_MI; ++mi) asm volatile("" :: "v"(a[mi]));
            asm volatile("" :: "v"(b[0].r0), "v"(b[0].r1), "v"(b[0].r2), "v"(b[0].r3));
        } else {
#pragma unroll
        for (int mi = 0; mi < TG_MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < TG_NI; ++ni) tg_mma_16x16x4(a[mi], b[ni], acc[mi][ni]);
        }
        if (SGB && RD == 1) {
            if (kk + 4 < TG_KC) {
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
        } else if (SGB && RD == 0) {
            if (ILV && kk / 4 < TG_KC / 8) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   // the two DMAs first
            if (kk + 4 < TG_KC) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);    // 2 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // 1 DS read
                }
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            }
        }
        if (kk + 4 < TG_KC) {
#pragma unroll
            for (int mi = 0; mi < TG_MI; ++mi) a[mi] = na[mi];
#pragma unroll
            for (int ni = 0; ni < TG_NI; ++ni) b[ni] = nb[ni];
        }
    }
}

template <bool ILV, bool SGB>
__device__ __forceinline__ void gemm_x(const double* __restrict__ A, const double* __restrict__ B, long ld, int K, double* lds, TgAcc& acc)
{
    double* As0 = lds;
    double* Bs0 = lds + TD_OPER_DOUBLES;
    double* As1 = lds + 2 * TD_OPER_DOUBLES;
    double* Bs1 = lds + 3 * TD_OPER_DOUBLES;
    const int nchunks = K / TG_KC;
    td_issue_chunk(A, ld, B, ld, 0, As0, Bs0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const bool more = c + 1 < nchunks;
        if (!ILV && more) { if (c & 1) td_issue_chunk(A, ld, B, ld, (c + 1) * TG_KC, As0, Bs0); else td_issue_chunk(A, ld, B, ld, (c + 1) * TG_KC, As1, Bs1); }
        if (c & 1) compute_chunk_x<ILV, SGB>(As1, Bs1, acc, A, B, ld, (c + 1) * TG_KC, As0, Bs0, more);
        else       compute_chunk_x<ILV, SGB>(As0, Bs0, acc, A, B, ld, (c + 1) * TG_KC, As1, Bs1, more);
        __syncthreads();
    }
}

// MODE 0: the engine as shipped (DMA + LDS reads + MFMA + barrier)
//      1: no DMA in the loop (same LDS image every chunk), barrier kept
//      2: no DMA, no barrier
//      3: MFMAs only (fragments held in registers), no barrier
//      4: DMA + barrier, MFMAs on register fragments (no LDS reads)
//      6: as 4 but the barrier does not wait for the DMA (is it latency or bandwidth?)
//      8: shipped with the DMA issue spread over the k-steps; 9: shipped with pinned LDS-read/MFMA interleave; 10: both
//      7: as 4 but the DMA re-reads the same 8 chunks (L2 hits)
template <int MODE>
__global__ void __launch_bounds__(256) loop_kernel(const double* __restrict__ Y, long ldy, int K, int ntile, double* out)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int bi = blockIdx.x % ntile, bj = (blockIdx.x / ntile) % ntile;
    const double* A = Y + (long)bi * 64;
    const double* B = Y + (long)bj * 64;
    TgAcc acc;
    tg_zero(acc);
    double* As0 = lds;
    double* Bs0 = lds + TD_OPER_DOUBLES;
    double* As1 = lds + 2 * TD_OPER_DOUBLES;
    double* Bs1 = lds + 3 * TD_OPER_DOUBLES;
    const int nchunks = K / TG_KC;
    if (MODE == 0 || MODE == 5) {
        tile_gemm_nt_dma(A, ldy, B, ldy, K, lds, acc);
    } else if (MODE == 11 || MODE == 12 || MODE == 13) {
        td_issue_chunk(A, ldy, B, ldy, 0, As0, Bs0);
        td_issue_chunk(A, ldy, B, ldy, TG_KC, As1, Bs1);
        __syncthreads();
        for (int c = 0; c < nchunks; ++c) {
            double* Ac = (c & 1) ? As1 : As0; double* Bc = (c & 1) ? Bs1 : Bs0;
            if (MODE == 11) compute_chunk_x<false, true, 0>(Ac, Bc, acc, A, B, ldy, 0, Ac, Bc, false);
            if (MODE == 12) compute_chunk_x<false, true, 1>(Ac, Bc, acc, A, B, ldy, 0, Ac, Bc, false);
            if (MODE == 13) compute_chunk_x<false, false, 2>(Ac, Bc, acc, A, B, ldy, 0, Ac, Bc, false);
        }
    } else if (MODE == 8) { gemm_x<true, false>(A, B, ldy, K, lds, acc);
    } else if (MODE == 9) { gemm_x<false, true>(A, B, ldy, K, lds, acc);
    } else if (MODE == 10) { gemm_x<true, true>(A, B, ldy, K, lds, acc);
    } else if (MODE == 1 || MODE == 2) {
        td_issue_chunk(A, ldy, B, ldy, 0, As0, Bs0);
        td_issue_chunk(A, ldy, B, ldy, TG_KC, As1, Bs1);
        __syncthreads();
        for (int c = 0; c < nchunks; ++c) {
            if (c & 1) compute_chunk_x<false, false, 0>(As1, Bs1, acc, A, B, ldy, 0, As1, Bs1, false); else compute_chunk_x<false, false, 0>(As0, Bs0, acc, A, B, ldy, 0, As0, Bs0, false);
            if (MODE == 1) __syncthreads();
        }
    } else if (MODE == 3 || MODE == 4 || MODE == 6 || MODE == 7) {
        td_issue_chunk(A, ldy, B, ldy, 0, As0, Bs0);
        __syncthreads();
        const int lane = threadIdx.x & 63;
        double a[4], b[4];
        for (int q = 0; q < 4; ++q) { a[q] = As0[lane + 64 * q]; b[q] = Bs0[lane + 64 * q]; }
        for (int c = 0; c < nchunks; ++c) {
            if ((MODE == 4 || MODE == 6) && c + 1 < nchunks) { if (c & 1) td_issue_chunk(A, ldy, B, ldy, (c + 1) * TG_KC, As0, Bs0); else td_issue_chunk(A, ldy, B, ldy, (c + 1) * TG_KC, As1, Bs1); }
            if (MODE == 7 && c + 1 < nchunks) { if (c & 1) td_issue_chunk(A, ldy, B, ldy, ((c + 1) & 7) * TG_KC, As0, Bs0); else td_issue_chunk(A, ldy, B, ldy, ((c + 1) & 7) * TG_KC, As1, Bs1); }
#pragma unroll
            for (int kk = 0; kk < TG_KC; kk += 4)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[mi][0][t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[mi], b[t], acc[mi][0][t], 0, 0, 0);
            if (MODE == 4 || MODE == 7) __syncthreads();
            if (MODE == 6) __builtin_amdgcn_s_barrier();      // no vmcnt wait: the DMA is never waited for (timing only)
        }
    }
    double s = 0;
    for (int mi = 0; mi < TG_MI; ++mi) for (int ni = 0; ni < TG_NI; ++ni) for (int t = 0; t < 4; ++t) s += acc[mi][ni][t];
    if (s == -1.2345) out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static double run(const double* Y, long ldy, int K, int ntile, int grid, double* out, int reps)
{
    hipFuncSetAttribute(reinterpret_cast<const void*>(loop_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * TD_LDS_DOUBLES));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) loop_kernel<MODE><<<grid, 256, sizeof(double) * TD_LDS_DOUBLES>>>(Y, ldy, K, ntile, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) loop_kernel<MODE><<<grid, 256, sizeof(double) * TD_LDS_DOUBLES>>>(Y, ldy, K, ntile, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 / reps;
}

int main(int argc, char** argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 4096;
    const int n = 1856, ntile = n / 64;
    double* Y; double* out;
    hipMalloc(&Y, sizeof(double) * (size_t)n * K); hipMalloc(&out, sizeof(double) * 1024 * 256);
    std::vector<double> h((size_t)n * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1e-3 * (double)((i * 2654435761u) % 1000) - 0.5;
    hipMemcpy(Y, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
    const char* names[] = { "engine of tile_gemm.h", "no DMA", "no DMA, no barrier", "MFMA only", "DMA + barrier, no LDS reads", "DMA never waited for, no LDS reads", "DMA of L2-resident chunks, no LDS reads", "DMA spread", "reads pinned", "DMA spread + reads pinned", "no DMA, no barrier, reads pinned", "no DMA, no barrier, 5 of 8 reads, pinned", "LDS reads only" };
    run<3>(Y, n, 4096, ntile, 512, out, 200);          // ~50 ms of MFMA load first: clocks and power state settle
    for (int grid : { 512 }) {
        double t[13];
        t[0] = run<0>(Y, n, K, ntile, grid, out, 5);
        t[1] = run<1>(Y, n, K, ntile, grid, out, 5);
        t[2] = run<2>(Y, n, K, ntile, grid, out, 5);
        t[3] = run<3>(Y, n, K, ntile, grid, out, 5);
        t[4] = run<4>(Y, n, K, ntile, grid, out, 5);
        t[5] = run<6>(Y, n, K, ntile, grid, out, 5);
        t[6] = run<7>(Y, n, K, ntile, grid, out, 5);
        t[7] = run<8>(Y, n, K, ntile, grid, out, 5);
        t[8] = run<9>(Y, n, K, ntile, grid, out, 5);
        t[9] = run<10>(Y, n, K, ntile, grid, out, 5);
        t[10] = run<11>(Y, n, K, ntile, grid, out, 5);
        t[11] = run<12>(Y, n, K, ntile, grid, out, 5);
        t[12] = run<13>(Y, n, K, ntile, grid, out, 5);
        const double again = run<0>(Y, n, K, ntile, grid, out, 5);
        printf("grid %4d K %d  %-40s %8.1f us  (first kernel of the process measured again at the end)\n", grid, K, names[0], again * 1e6);
        for (int m = 0; m < 13; ++m) {
            const double fl = 2.0 * 64 * 64 * (double)K * grid;
            printf("grid %4d K %d  %-40s %8.1f us  %6.2f TFLOP/s\n", grid, K, names[m], t[m] * 1e6, fl / t[m] * 1e-12);
        }
    }
    return 0;
}
