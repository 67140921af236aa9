// staged_kernels.hip -- the update of LARGE systems (more 16-row strips than compute units: 1000 landmarks, C5) as two
// stages on disjoint sets of compute units (ExtendKF.cpp:602-609: S^-1, K = P H^T S^-1, P - K S K^T).
//
//   S stage   the blocked Cholesky sweep of the r x r innovation covariance ALONE (kernels.hip, launch_s_stage): the serial
//             pivot chain of r dependent pivots with a small trailing matrix -- latency-bound, a few compute units.
//   R stage   everything that has n = 6013 rows and is MFMA work, per GROUP g of column blocks [b0, b1) of the factor, as
//             soon as the S stage has finished that group:
//               M_g   = L_gg^-1                               group_inverse_kernel   (the group's diagonal block of L, G x G)
//               Y_g   = W_g M_g^T                              staged_gemm_kernel     (W = [P H^T; nu^T]: 6077 rows; triangular)
//               W_k  -= Y_g L(k,g)^T  for every later block k  staged_gemm_kernel     (right-looking: K = 64 x group size)
//             i.e. the triangular solve Y L^T = W of the sweep as a blocked forward substitution whose block is a GROUP of four
//             diagonal blocks, beside the S stage; then ONE pass P -= Y Y^T over the whole width on every compute unit
//             (rank_macro.hip).  (Round 5 also ran one rank-update pass per group beside the S stage: each pass moves P once
//             more, and at K = 256 .. 512 a pass runs at 29 .. 35 TFLOP/s -- slower in sum than the single pass, NOTEBOOK.md.)
//
// Until round 4 the sweep carried the 6077 rows of W through all 25 block steps (launch per step: 13 narrow passes that
// leave half the chip idle, 12 wide passes at 38 % MFMA-busy, 0.79 ms) and the rank update (1.05 ms) started when it
// ended.  A second stream could not run them side by side: a sweep workgroup (104 KiB of LDS) only fits a compute unit
// that holds NO rank-update workgroup (72 KiB), and the rank update refills every slot the moment it frees
// (profiles/r04_c5_ksplit_side_stream_trace.txt).  Streams with disjoint CU masks (hipExtStreamCreateWithCUMask) remove that
// coupling (scripts/probes/cu_mask.hip: a chain of 120 KiB launches keeps its 16 us period beside a saturating bulk stream).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"
#include "tile_gemm.h"

namespace rslam {

// ---------------------------------------------------------------------------------------------------------------
// M_g = L_gg^-1 of one group of nb diagonal blocks, from the panel blocks L(a,c) (a > c, in the S rows of Ystore) and the
// inverses of the 64 x 64 diagonal blocks (Linv) the S stage leaves behind.  Column-oriented block recursion,
//     M(a,b) = -Linv(a) * sum_{c=b}^{a-1} L(a,c) M(c,b),      a > b,    M(b,b) = Linv(b),
// one workgroup per block column b (nothing crosses workgroups), on the transposed factor so that every product is the
// tile engine's A B^T:   Mt(b,a) := M(a,b)^T = -[ sum_c Mt(b,c) L(a,c)^T ] Linv(a)^T.
// Both orientations are written: Mt feeds the recursion, M (row block k = 64 x K, rows contiguous) is the B operand of
// Y_g = T_g M_g^T.  Off the critical path of either stage: it runs on a stream of its own while the S stage factors the
// next group and the R stage is busy with the previous one.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
group_inverse_kernel(int b0, int nb, const double* __restrict__ Lp, long ldl, const double* __restrict__ Linv,
                     double* M, double* Mt, long ldm)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int b = b0 + (int)blockIdx.x;
    const int row = threadIdx.x & 63, g = threadIdx.x >> 6;
    double* Mbb = M + 64L * b + 64L * b * ldm;
    double* Mtbb = Mt + 64L * b + 64L * b * ldm;
    {   // diagonal block: M(b,b) = Linv(b) (lower triangular, zero above), Mt(b,b) its transpose
        const double* Lb = Linv + 64L * 64 * b;
        double* Cs = lds;
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const int c = g + 4 * q;
            const double v = Lb[row + 64 * c];
            Mbb[row + (long)c * ldm] = v;
            Cs[c * TS_LD + row] = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const int c = g + 4 * q;
            Mtbb[row + (long)c * ldm] = Cs[row * TS_LD + c];
        }
        __threadfence();
        __syncthreads();
    }
    for (int a = b + 1; a < b0 + nb; ++a) {
        // D = sum_{c=b}^{a-1} Mt(b,c) L(a,c)^T    (K = 64 (a - b): the row block of Mt this workgroup has written so far)
        TgAcc acc;
        tg_zero(acc);
        tile_gemm_nt_dma(Mt + 64L * b + 64L * b * ldm, ldm, Lp + 64L * a + 64L * b * ldl, ldl, 64 * (a - b), lds, acc);
        // Mt(b,a) = -D Linv(a)^T : D becomes an LDS-resident operand (a 64 x 64 result written with the operand leading
        // dimension IS two operand buffers), Linv(a) the other
        double* bufA = lds;
        double* bufB = lds + 2 * TG_OPER_DOUBLES;
        tg_fill64(Linv + 64L * 64 * a, 64, bufB);
        tg_acc_to_lds<TG_LD>(acc, bufA, -1.0);
        __syncthreads();
        TgAcc out;
        tg_zero(out);
        tg_gemm64_lds(bufA, bufB, out);
        __syncthreads();
        double* Cs = lds;
        tg_acc_to_lds(out, Cs, 1.0);
        __syncthreads();
        double* Mtba = Mt + 64L * b + 64L * a * ldm;
        double* Mab = M + 64L * a + 64L * b * ldm;
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const int c = g + 4 * q;
            Mtba[row + (long)c * ldm] = Cs[c * TS_LD + row];
            Mab[row + (long)c * ldm] = Cs[row * TS_LD + c];
        }
        __threadfence();        // the next step's LDS-DMA reads Mt(b,a) back (lines this workgroup has never read before)
        __syncthreads();
    }
}

void launch_group_inverse(hipStream_t s, int b0, int nb, const double* Lp, long ldl, const double* Linv, double* M, double* Mt, long ldm)
{
    if (nb <= 0) return;
    group_inverse_kernel<<<dim3(nb), dim3(256), sizeof(double) * TG_LDS_DOUBLES, s>>>(b0, nb, Lp, ldl, Linv, M, Mt, ldm);
}

// ---------------------------------------------------------------------------------------------------------------
// C(i, k) = [Cin(i, k)] + alpha * A(i, :) B(k, :)^T over 64 x 64 tiles, i < row_tiles, k < nb; K = k_fixed, or 64 (k + 1)
// (tri: B is block lower triangular -- the group inverse).  The tile engine of the rank update (LDS-DMA staging).
// Workgroup -> tile: workgroups are dealt round-robin over the 8 XCDs, so XCD x takes the row tiles i = x (mod 8) with all
// their k (the A panel of a row tile is fetched into ONE L2 and reused nb times; the B panels, a few MB, live in all of
// them); inside an XCD the long-K tiles of a triangular product go first.
// ---------------------------------------------------------------------------------------------------------------
struct StagedGemm {
    const double* A; long lda;          // row tile i: A + 64 i
    const double* B; long ldb;          // column tile k: B + 64 k
    const double* Cin; long ldi;        // nullable
    double* Cout; long ldo;
    int row_tiles, nb, k_fixed, tri;
    double alpha;
};

__global__ void __launch_bounds__(256)
staged_gemm_kernel(StagedGemm p)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int w = (int)blockIdx.x, x = w & 7, slot = w >> 3;
    const int rows_x = (p.row_tiles - x + 7) / 8;              // row tiles of this XCD: x, x + 8, ...
    if (slot >= rows_x * p.nb) return;
    // (tri) k descending inside a round of row tiles: the K = 64 nb tiles are dispatched first
    const int kslot = slot / rows_x, i = x + 8 * (slot % rows_x);
    const int k = p.tri ? p.nb - 1 - kslot : kslot;
    const int K = p.tri ? 64 * (k + 1) : p.k_fixed;
    const int row = threadIdx.x & 63, g = threadIdx.x >> 6;
    double cin[16];
    if (p.Cin) {
        const double* Ci = p.Cin + 64L * i + 64L * k * p.ldi;
#pragma unroll
        for (int q = 0; q < 16; ++q) cin[q] = Ci[row + (long)(g + 4 * q) * p.ldi];
    }
    TgAcc acc;
    tg_zero(acc);
    tile_gemm_nt_dma(p.A + 64L * i, p.lda, p.B + 64L * k, p.ldb, K, lds, acc);
    tg_acc_to_lds(acc, lds, p.alpha);
    __syncthreads();
    double* Co = p.Cout + 64L * i + 64L * k * p.ldo;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c = g + 4 * q;
        Co[row + (long)c * p.ldo] = (p.Cin ? cin[q] : 0.0) + lds[c * TS_LD + row];
    }
}

static void launch_staged_gemm(hipStream_t s, const StagedGemm& p)
{
    if (p.row_tiles <= 0 || p.nb <= 0) return;
    const int per_x = (p.row_tiles + 7) / 8;
    staged_gemm_kernel<<<dim3(8 * per_x * p.nb), dim3(256), sizeof(double) * TG_LDS_DOUBLES, s>>>(p);
}

// Right-looking step of the blocked substitution: the solved group g = [b0, b0 + nb) goes onto every later column block,
// W(:, k) -= Y_g L(k, g)^T for k = b1 .. nblk - 1, in place in the system matrix A (rows RP ..: P H^T and nu^T).  K = 64 nb:
// short products, but they run beside the S stage on compute units it does not use, and the LAST group then only waits
// for its own inverse and one product (left-looking, its T would be a K = 1472 product behind the end of the S stage).
void launch_staged_update(hipStream_t s, const SystemDims& d, int b0, int nb, int nblk, double* A, const double* Ystore)
{
    const int b1 = b0 + nb;
    if (b1 >= nblk) return;
    StagedGemm p{};
    p.A = Ystore + d.RP + 64L * b0 * d.ldA; p.lda = d.ldA;
    p.B = Ystore + 64L * b1 + 64L * b0 * d.ldA; p.ldb = d.ldA;
    p.Cin = A + d.RP + 64L * b1 * d.ldA; p.ldi = d.ldA;
    p.Cout = A + d.RP + 64L * b1 * d.ldA; p.ldo = d.ldA;
    p.row_tiles = (d.NP + 64) / 64; p.nb = nblk - b1; p.k_fixed = 64 * nb; p.tri = 0; p.alpha = -1.0;
    launch_staged_gemm(s, p);
}

// Y_g = T_g M_g^T  (T_g in A, Y_g into Ystore: rows RP .. of the group's columns)
void launch_staged_Y(hipStream_t s, const SystemDims& d, int b0, int nb, const double* A, double* Ystore, const double* M, long ldm)
{
    StagedGemm p{};
    p.A = A + d.RP + 64L * b0 * d.ldA; p.lda = d.ldA;
    p.B = M + 64L * b0 + 64L * b0 * ldm; p.ldb = ldm;
    p.Cin = nullptr; p.ldi = 0;
    p.Cout = Ystore + d.RP + 64L * b0 * d.ldA; p.ldo = d.ldA;
    p.row_tiles = (d.NP + 64) / 64; p.nb = nb; p.k_fixed = 0; p.tri = 1; p.alpha = 1.0;
    launch_staged_gemm(s, p);
}

int init_staged_kernel_attributes()
{
    const int bytes = (int)(sizeof(double) * TG_LDS_DOUBLES);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(group_inverse_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(staged_gemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    return (int)e;
}

}  // namespace rslam
