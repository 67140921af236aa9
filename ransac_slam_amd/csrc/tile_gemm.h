// tile_gemm.h -- FP64 MFMA tile engine for gfx950 (CDNA4).
//
// One workgroup (256 threads = 4 waves, one per SIMD) owns a 64 x 64 output
// tile C = A * B^T where A is 64 x K and B is 64 x K, both column-major with the
// row index contiguous ("NT" form).  Every dense contraction of the EKF update
// has this shape (covariance rank-r update Y*Y^T, panel solve X*Linv^T,
// trailing update L_ik*L_jk^T), so there is exactly one MFMA inner loop.
//
// v_mfma_f64_16x16x4_f64: lane l supplies A[i = l&15][k = l>>4] and
// B[k = l>>4][j = l&15]; the 4 results of lane l are C[(l>>4) + 4*reg][l&15].
// Each wave computes a 32 x 32 sub-tile as 2 x 2 MFMA tiles (4 independent
// accumulators keep the 64-cycle matrix pipe back to back).
//
// LDS image per operand and buffer: [KC][LDS_LD] doubles, row index contiguous,
// so global->LDS is a straight 16-byte copy and every ds_read_b64 of a fragment
// touches 16 consecutive doubles per k.  LDS_LD = 80 doubles (= 160 dwords = 32
// mod 64 banks) puts the two k-rows a 32-lane half reads on disjoint banks.
#pragma once
#include <hip/hip_runtime.h>

namespace rslam {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int TG_TILE = 64;     // output tile edge per workgroup
constexpr int TG_KC = 32;       // K chunk staged per barrier
constexpr int TG_LD = 80;       // LDS leading dimension (doubles)
constexpr int TG_THREADS = 256;
constexpr int TG_OPER_DOUBLES = TG_KC * TG_LD;              // one operand, one buffer
constexpr int TG_LDS_DOUBLES = 4 * TG_OPER_DOUBLES;         // A,B x 2 buffers = 80 KiB

struct TileRegs { d2 a[4]; d2 b[4]; };

// Each thread moves rows (2*rp, 2*rp+1) of columns cq + 8*q, q = 0..3.
__device__ __forceinline__ void tg_load_chunk(const double* __restrict__ A, long lda,
                                              const double* __restrict__ B, long ldb,
                                              int k0, TileRegs& r)
{
    const int t = threadIdx.x;
    const int rp = t & 31, cq = t >> 5;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long c = k0 + cq + 8 * q;
        r.a[q] = *reinterpret_cast<const d2*>(A + 2 * rp + c * lda);
        r.b[q] = *reinterpret_cast<const d2*>(B + 2 * rp + c * ldb);
    }
}

__device__ __forceinline__ void tg_store_chunk(double* As, double* Bs, const TileRegs& r)
{
    const int t = threadIdx.x;
    const int rp = t & 31, cq = t >> 5;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = cq + 8 * q;
        *reinterpret_cast<d2*>(As + c * TG_LD + 2 * rp) = r.a[q];
        *reinterpret_cast<d2*>(Bs + c * TG_LD + 2 * rp) = r.b[q];
    }
}

__device__ __forceinline__ void tg_compute_chunk(const double* As, const double* Bs, d4 (&acc)[2][2])
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int kq = lane >> 4, ij = lane & 15;
    const double* ap = As + kq * TG_LD + wm * 32 + ij;
    const double* bp = Bs + kq * TG_LD + wn * 32 + ij;
#pragma unroll
    for (int kk = 0; kk < TG_KC; kk += 4) {
        const double a0 = ap[kk * TG_LD], a1 = ap[kk * TG_LD + 16];
        const double b0 = bp[kk * TG_LD], b1 = bp[kk * TG_LD + 16];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
}

// acc += A(64 x K) * B(64 x K)^T.  A, B point at row 0 / column 0 of the
// operand panels; K must be a multiple of TG_KC; lds holds TG_LDS_DOUBLES.
// Ends with a barrier: lds may be reused by the caller immediately.
__device__ __forceinline__ void tile_gemm_nt(const double* __restrict__ A, long lda,
                                             const double* __restrict__ B, long ldb,
                                             int K, double* lds, d4 (&acc)[2][2])
{
    double* As0 = lds;
    double* Bs0 = lds + TG_OPER_DOUBLES;
    double* As1 = lds + 2 * TG_OPER_DOUBLES;
    double* Bs1 = lds + 3 * TG_OPER_DOUBLES;
    const int nchunks = K / TG_KC;
    if (nchunks <= 0) return;
    TileRegs r;
    tg_load_chunk(A, lda, B, ldb, 0, r);
    tg_store_chunk(As0, Bs0, r);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const bool more = (c + 1 < nchunks);
        if (more) tg_load_chunk(A, lda, B, ldb, (c + 1) * TG_KC, r);   // in flight under the MFMAs
        if (c & 1) tg_compute_chunk(As1, Bs1, acc); else tg_compute_chunk(As0, Bs0, acc);
        if (more) { if (c & 1) tg_store_chunk(As0, Bs0, r); else tg_store_chunk(As1, Bs1, r); }
        __syncthreads();
    }
}

__device__ __forceinline__ void tg_zero(d4 (&acc)[2][2])
{
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
}

// Spill the accumulators into an LDS tile Cs[col][row] (ld = TS_LD) so that the
// epilogue can read columns (coalesced global stores) or rows (transposes).
constexpr int TS_LD = 65;
constexpr int TS_DOUBLES = 64 * TS_LD;

__device__ __forceinline__ void tg_acc_to_lds(const d4 (&acc)[2][2], double* Cs, double scale)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = wm * 32 + mi * 16 + (lane >> 4) + 4 * reg;
                const int col = wn * 32 + ni * 16 + (lane & 15);
                Cs[col * TS_LD + row] = acc[mi][ni][reg] * scale;
            }
}

}  // namespace rslam
