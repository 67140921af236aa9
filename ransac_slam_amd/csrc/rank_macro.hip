// rank_macro.hip -- K10, the covariance rank update P' = 1/2 (P + P^T) - Y Y^T (ExtendKF.cpp:608-609), on 128 x 128 macro
// tiles (tile_gemm128.h) for large maps: lower-triangle macro-tile pairs (I >= J), four 64 x 64 quadrants each, every
// quadrant with exactly the epilogue of the 64 x 64 form (kernels.hip, rank_update_kernel): symmetrisation with the mirror
// tile, the deferred low-innovation covariance (MatArgs), mirrored write, K11 (Jnorm congruence on rows / columns 3..6,
// ExtendKF.cpp:629-634) for the first block column.
//
// Why a second form: at C5 (n = 6013, r = 1568) the 64 x 64 form runs at 54 TFLOP/s (0.69 of the FP64 MFMA peak), re-reads Y
// 4.4 times over (2.9 GB per launch for 0.65 GB of operands) and its loop is bound by what a wave issues around its MFMAs.
// A macro tile halves the operand bytes and the fragment reads per flop; its K loop alone reaches 68 TFLOP/s
// (scripts/probes/gemm128_loop.hip; a vendor DGEMM on the same box: 70.6, scripts/probes/rocblas_yardstick.cpp).
// One workgroup per compute unit (144 KiB of LDS) and 52 MFLOP per tile at K = 1600 make the LAST round expensive (1128
// macro tiles on 256 compute units: the fifth round would run at 40 % occupancy), so the launch covers whole rounds only and
// the host hands the remaining tiles to the 64 x 64 form (launch_rank_update_tiles), two workgroups per compute unit.
// K9 (x_k_k = x + Y u, quaternion, Jnorm) does not ride here: it rides in the 64 x 64 launch of the partial round, which the
// host enqueues FIRST (enqueue_rank_pass), so Jnorm is simply there when the first block column's quadrants reach their epilogue.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "kernels.h"
#include "tile_gemm128.h"
#include "rank_common.h"

namespace rslam {

struct MacroArgs {
    const double* Pin; long ldp; double* Pout; long ldo;
    const double* Y; long ldy; int K;                 // Y: row 0 of the P H^T rows, first column of this pass; K columns (multiple of 64)
    const int32_t* order; int n_tiles;                // (I << 16 | J) per workgroup
    const int32_t* sel; int slot_k;
    const double* Tq;                                 // nullable: Jnorm of THIS update (4 x 4), applied when sel[slot_k] != 0
    int mirror_flag;                                  // != 0: Pin holds exactly mirrored pairs (a later pass of the same update)
    int token;                                        // 2: high-innovation pass (the LI pass may have left mirrored pairs)
    MatArgs mat;                                      // mat.flag nullable: P_li may be deferred (first pass of the HI update)
};

// LDS-only workgroup barrier: the epilogue's barriers order LDS traffic; __syncthreads() would also wait for every store of
// the previous quadrant to be acknowledged (s_waitcnt vmcnt(0)) -- four exposed memory round trips per quadrant.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// What a quadrant's epilogue reads from global memory, requested one quadrant ahead (and, for the first one, in front of the K loop)
struct QuadIn { double pij[16], pji[16], y1i[4], y1j; bool live, mirror_known; };

// quadrant (QI, qj) of the macro tile (I, J): block (bi, bj) = (2 I + QI, 2 J + qj) of P
template <bool MAT>
__device__ __forceinline__ void quad_load(const MacroArgs& a, int I, int J, int QI, int qj, bool deferred, QuadIn& in)
{
    const int bi = 2 * I + QI, bj = 2 * J + qj;
    in.live = bj <= bi;                                // (uniform) the upper quadrant of a diagonal macro tile is written as a mirror
    if (!in.live) return;
    const int row = threadIdx.x & 63, g = threadIdx.x >> 6;
    const double* Pin = deferred ? a.mat.Ppred : a.Pin;
    const long ldp = deferred ? a.mat.ldp : a.ldp;
    const double* Pij = Pin + 64L * bi + 64L * bj * ldp;
    const double* Pji = Pin + 64L * bj + 64L * bi * ldp;
    in.mirror_known = !deferred && (bi != bj) && (a.mirror_flag || ((a.token == 2) && (a.sel[SEL_NBLK_LI] > 0)));
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c = g + 4 * q;
        in.pij[q] = Pij[row + (long)c * ldp];
        in.pji[q] = in.mirror_known ? 0.0 : Pji[row + (long)c * ldp];
    }
    if (MAT && deferred) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) in.y1i[cc] = a.mat.Y1[64L * bi + row + cc * a.mat.ldy1];
        in.y1j = a.mat.Y1[64L * bj + row + g * a.mat.ldy1];
    }
}

// The epilogue in two halves.  quad_compute: one quadrant's result -- sym(P) [deferred: the materialised P_li] minus the
// product, K11 applied -- into ITS OWN LDS tile Cs_q[col][row]; nothing is stored to global memory.  quad_store: at the very
// end, all four tiles out (direct and mirrored).  Why: loads and stores share one counter (vmcnt) and may complete out of
// order with each other, so a wave that has stores in flight can only wait for a load with vmcnt(0) -- with the stores of
// quadrant q issued before the loads of quadrant q + 1 were consumed, every quadrant paid a full store acknowledgement plus
// a load round trip (5.7 us per quadrant, 23 us per tile; a start stagger of the workgroups did not change it: it is not
// bandwidth).  Now every load of a tile is consumed before its first store is issued.
template <int QI, bool MAT>
__device__ __forceinline__ void quad_compute(const MacroArgs& a, const T8Acc& acc, int I, int J, int qj, bool deferred, const QuadIn& in, double* Cs,
                                             double* Y1s)
{
    if (!in.live) return;
    const int bi = 2 * I + QI, bj = 2 * J + qj;
    const int row = threadIdx.x & 63, g = threadIdx.x >> 6;
    const bool mirror_known = in.mirror_known;
    double pm[16];
    if (!mirror_known) {
        // the mirror tile (bj, bi) transposed through this quadrant's own LDS tile (it holds nothing yet)
#pragma unroll
        for (int q = 0; q < 16; ++q) Cs[(g + 4 * q) * TS_LD + row] = in.pji[q];
        lds_barrier();
#pragma unroll
        for (int q = 0; q < 16; ++q) pm[q] = Cs[row * TS_LD + (g + 4 * q)];
        lds_barrier();
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) pm[q] = in.pij[q];
    }
    t8_quadrant_to_lds<QI>(acc, qj, Cs, 1.0);
    if (MAT && deferred) Y1s[row + 64 * g] = in.y1j;
    lds_barrier();
    double m[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) m[q] = 0.5 * in.pij[q] + 0.5 * pm[q];
    if (MAT && deferred) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int c = g + 4 * q;
            m[q] -= (in.y1i[0] * Y1s[c] + in.y1i[1] * Y1s[c + 64]) + (in.y1i[2] * Y1s[c + 128] + in.y1i[3] * Y1s[c + 192]);
        }
    }
    double prod[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) prod[q] = Cs[(g + 4 * q) * TS_LD + row];
    if (MAT && deferred && bj == 0) {                  // Jnorm of the low-innovation update on rows / columns 3..6 of M
        double Tli[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) Tli[k] = a.mat.T_li[k];
        lds_barrier();                                 // (the product has been read by everybody)
#pragma unroll
        for (int q = 0; q < 16; ++q) Cs[(g + 4 * q) * TS_LD + row] = m[q];
        lds_barrier();
        k11_lds(Cs, Tli, bi);
#pragma unroll
        for (int q = 0; q < 16; ++q) m[q] = Cs[(g + 4 * q) * TS_LD + row];
    }
    lds_barrier();                                     // (the product / M has been read by everybody: the tile takes the result)
#pragma unroll
    for (int q = 0; q < 16; ++q) Cs[(g + 4 * q) * TS_LD + row] = m[q] - prod[q];
    lds_barrier();
    // K11: the Jnorm congruence of this update on rows / columns 3..6 -- the first block column only (same arithmetic as the
    // 64 x 64 form; Jnorm was written by an earlier launch on this stream)
    if (a.Tq && bj == 0 && a.sel[a.slot_k] != 0) {
        double T[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) T[k] = a.Tq[k];
        const int j = threadIdx.x;
        if (bi != 0) {
            if (j < 64) {
                double rb[4];
                for (int i = 0; i < 4; ++i) {
                    double sacc = 0;
                    for (int k = 0; k < 4; ++k) sacc += T[i + 4 * k] * Cs[(3 + k) * TS_LD + j];
                    rb[i] = sacc;
                }
                for (int i = 0; i < 4; ++i) Cs[(3 + i) * TS_LD + j] = rb[i];
            }
        } else {
            if (j < 64 && !(j >= 3 && j < 7)) {
                double rb[4];
                for (int i = 0; i < 4; ++i) {
                    double sacc = 0;
                    for (int k = 0; k < 4; ++k) sacc += T[i + 4 * k] * Cs[j * TS_LD + (3 + k)];
                    rb[i] = sacc;
                }
                for (int i = 0; i < 4; ++i) { Cs[j * TS_LD + (3 + i)] = rb[i]; Cs[(3 + i) * TS_LD + j] = rb[i]; }
            } else if (j == 3) {
                double cb[4][4], out[4][4];
                for (int i = 0; i < 4; ++i)
                    for (int c = 0; c < 4; ++c) {
                        double sacc = 0;
                        for (int k = 0; k < 4; ++k) sacc += T[i + 4 * k] * Cs[(3 + c) * TS_LD + (3 + k)];
                        cb[i][c] = sacc;
                    }
                for (int i = 0; i < 4; ++i)
                    for (int c = 0; c < 4; ++c) {
                        double sacc = 0;
                        for (int k = 0; k < 4; ++k) sacc += cb[i][k] * T[c + 4 * k];
                        out[i][c] = sacc;
                    }
                for (int i = 0; i < 4; ++i)
                    for (int c = 0; c < 4; ++c) Cs[(3 + c) * TS_LD + (3 + i)] = out[i][c];
            }
        }
        lds_barrier();
    }
}

__device__ __forceinline__ void quad_store(const MacroArgs& a, int I, int J, int QI, int qj, const double* Cs)
{
    const int bi = 2 * I + QI, bj = 2 * J + qj;
    if (bj > bi) return;
    const int row = threadIdx.x & 63, g = threadIdx.x >> 6;
    double* Cij = a.Pout + 64L * bi + 64L * bj * a.ldo;
#pragma unroll 4
    for (int q = 0; q < 16; ++q) {
        const int c = g + 4 * q;
        Cij[row + (long)c * a.ldo] = Cs[c * TS_LD + row];
    }
    if (bi != bj) {
        double* Cji = a.Pout + 64L * bj + 64L * bi * a.ldo;
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const int c = g + 4 * q;
            Cji[row + (long)c * a.ldo] = Cs[row * TS_LD + c];
        }
    }
}

static_assert(T8_LDS_DOUBLES >= 4 * TS_DOUBLES + 256, "the epilogue keeps the four quadrants' results in LDS until all loads are in");

template <bool MAT>
__global__ void __launch_bounds__(T8_THREADS)
rank_update_macro_kernel(MacroArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if ((int)blockIdx.x >= a.n_tiles) return;
    const int e = a.order[blockIdx.x];
    const int I = e >> 16, J = e & 0xffff;
    const bool deferred = MAT && a.mat.flag && *a.mat.flag != 0;       // (uniform)
    const double* Ya = a.Y + 128L * I;
    const double* Yb = a.Y + 128L * J;
    T8Src s{{Ya, Ya + 64, Yb, Yb + 64}, a.ldy};
    T8Acc acc;
    t8_zero(acc);
    // quadrant order (0,0), (1,0), (0,1), (1,1); each one's inputs are requested while the one before is computed
    QuadIn in0, in1;
    quad_load<MAT>(a, I, J, 0, 0, deferred, in0);      // (lands under the K loop)
    tile_gemm128_nt<0, 0>(s, a.K, lds, acc);
#if defined(MACRO_EXP_NO_EPILOGUE)      // timing experiment only (wrong results): what the K loops alone cost inside the frame
    { double sum = 0; for (int mi = 0; mi < T8_MI; ++mi) for (int ni = 0; ni < T8_NI; ++ni) for (int q = 0; q < 4; ++q) sum += acc[mi][ni][q];
      if (sum == -1.2345) a.Pout[threadIdx.x] = sum + in0.pij[0]; }
    return;
#endif
    double* Y1s = lds + 4 * TS_DOUBLES;
    quad_load<MAT>(a, I, J, 1, 0, deferred, in1);
    quad_compute<0, MAT>(a, acc, I, J, 0, deferred, in0, lds, Y1s);
    quad_load<MAT>(a, I, J, 0, 1, deferred, in0);
    quad_compute<1, MAT>(a, acc, I, J, 0, deferred, in1, lds + TS_DOUBLES, Y1s);
    quad_load<MAT>(a, I, J, 1, 1, deferred, in1);
    quad_compute<0, MAT>(a, acc, I, J, 1, deferred, in0, lds + 2 * TS_DOUBLES, Y1s);
    quad_compute<1, MAT>(a, acc, I, J, 1, deferred, in1, lds + 3 * TS_DOUBLES, Y1s);
    // every load of this tile has been consumed: the stores
    quad_store(a, I, J, 0, 0, lds);
    quad_store(a, I, J, 1, 0, lds + TS_DOUBLES);
    quad_store(a, I, J, 0, 1, lds + 2 * TS_DOUBLES);
    quad_store(a, I, J, 1, 1, lds + 3 * TS_DOUBLES);
}

// Tile lists for a map of nT block rows on a device of `cus` compute units.
//   macro: (I << 16 | J), whole rounds of `cus` macro tiles, dealt to the XCDs in contiguous slices of a region-major order
//          (the 32 macro tiles one XCD holds at a time share a dozen 128-row panels of Y);
//   small: (bi << 16 | bj) of every 64 x 64 tile pair the macro launch does not cover (the last, partial round; the last
//          block row of a map with an odd number of block rows).
void make_macro_order(int nT, int cus, std::vector<int32_t>& macro, std::vector<int32_t>& small)
{
    macro.clear(); small.clear();
    const int nM = nT / 2, R = 6;
    std::vector<int32_t> seq;
    for (int sr = 0; sr < nM; sr += R)
        for (int sc = 0; sc <= sr; sc += R)
            for (int I = sr; I < sr + R && I < nM; ++I)
                for (int J = sc; J < sc + R && J <= I; ++J) seq.push_back((I << 16) | J);
    const int total = (int)seq.size();
    const int n_macro = cus > 0 ? (total / cus) * cus : 0;
    // the tiles of the partial round are taken from the END of the region-major order (the bottom rows of the triangle)
    if (n_macro > 0) {
        const int q = n_macro / 8, r = n_macro % 8;
        macro.assign(n_macro, 0);
        for (int b = 0; b < n_macro; ++b) {
            const int x = b % 8, idx = b / 8;
            macro[b] = seq[x * q + (x < r ? x : r) + idx];
        }
    }
    for (int t = n_macro; t < total; ++t) {
        const int I = seq[t] >> 16, J = seq[t] & 0xffff;
        for (int qi = 0; qi < 2; ++qi)
            for (int qj = 0; qj < 2; ++qj) {
                const int bi = 2 * I + qi, bj = 2 * J + qj;
                if (bj <= bi) small.push_back((bi << 16) | bj);
            }
    }
    if (nT & 1) for (int bj = 0; bj < nT; ++bj) small.push_back(((nT - 1) << 16) | bj);
}

void launch_rank_update_macro(hipStream_t s, const double* Pin, long ldp, const double* Y, long ldy, int K, double* Pout, long ldo,
                              const int32_t* macro_order, int n_macro, const int32_t* sel, int slot_k, const double* Tq,
                              int mirror_flag, int token, const MatArgs* mat)
{
    if (n_macro <= 0 || K <= 0) return;
    MacroArgs a{};
    a.Pin = Pin; a.ldp = ldp; a.Pout = Pout; a.ldo = ldo; a.Y = Y; a.ldy = ldy; a.K = K;
    a.order = macro_order; a.n_tiles = n_macro; a.sel = sel; a.slot_k = slot_k; a.Tq = Tq;
    a.mirror_flag = mirror_flag; a.token = token;
    if (mat) a.mat = *mat;
    const size_t bytes = sizeof(double) * T8_LDS_DOUBLES;
    if (mat) rank_update_macro_kernel<true><<<dim3(n_macro), dim3(T8_THREADS), bytes, s>>>(a);
    else     rank_update_macro_kernel<false><<<dim3(n_macro), dim3(T8_THREADS), bytes, s>>>(a);
}

int init_macro_kernel_attributes()
{
    const int bytes = (int)(sizeof(double) * T8_LDS_DOUBLES);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rank_update_macro_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(rank_update_macro_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    return (int)e;
}

}  // namespace rslam
