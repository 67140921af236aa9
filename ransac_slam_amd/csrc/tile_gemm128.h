// tile_gemm128.h -- the FP64 MFMA tile engine of tile_gemm.h with a 128 x 128 macro tile per workgroup.
//
// Why: a 64 x 64 tile fetches 64 + 64 operand rows for 64 x 64 x K products -- at C5 (n = 6013, r = 1568) the rank update
// moved 2.9 GB for 0.65 GB of operands and spent half of the LDS read port's time on fragments (8 fragment reads per 16
// MFMAs, two workgroups per compute unit).  A 128 x 128 tile halves the operand bytes per flop, and with the wave tile
// below a fragment read feeds four MFMAs instead of two.
//
// One workgroup = 4 waves (one per SIMD), one workgroup per compute unit (144 KiB of LDS).  Wave w owns ALL 128 rows and
// the columns [32 w, 32 w + 32): 8 x 2 tiles of 16 x 16 (16 accumulators of 4 doubles per lane = 128 registers).  The B
// fragment of a 16-column tile is needed in four block rotations (v_mfma_f64_4x4x4_4b, tile_gemm.h), the A fragment once:
// a tall wave tile reads 8 + 2 x 4 = 16 fragments for 64 MFMAs.
// Operands are staged as four 64-row halves (A rows 0..63 / 64..127, B likewise), each in the LDS-DMA image of tile_gemm.h
// (TD_SEG / TD_STEP: conflict-free fragment reads), in a ring of FOUR buffers of 16 columns (4 x 36 KiB) whose transfers run
// TWO chunks ahead: with one wave per SIMD nobody covers a wave that waits at the chunk barrier for memory, so the barrier
// only ever waits for transfers issued a whole chunk earlier (counted s_waitcnt).  Per chunk a wave issues 8 transfers
// (two per k-step) and 256 MFMAs; one workgroup barrier per chunk, in front of the last k-step's MFMAs.
// (Measured on the way, scripts/probes/gemm128_loop.hip, K = 1600, 1024 tiles, no epilogue: double-buffered chunks of 32 with
//  the transfers spread over the whole chunk 59.4 TFLOP/s, issued in its first half 63.5.)
#pragma once
#include "tile_gemm.h"

namespace rslam {

constexpr int T8_MI = 8, T8_NI = 2;
typedef d4 T8Acc[T8_MI][T8_NI];
constexpr int T8_KC = 16;                                      // K chunk: 4 k-steps
constexpr int T8_NBUF = 4;                                     // ring of chunk buffers; transfers run two chunks ahead
constexpr int T8_HALF_DOUBLES = (T8_KC / 2) * TD_SEG;          // one 64-row operand half of one chunk: 8 segments = 9 KiB
constexpr int T8_BUF_DOUBLES = 4 * T8_HALF_DOUBLES;            // A0, A1, B0, B1
constexpr int T8_LDS_DOUBLES = T8_NBUF * T8_BUF_DOUBLES;       // 144 KiB
constexpr int T8_THREADS = 256;
constexpr int T8_XFERS = 4 * (T8_KC / 2) / 4;                  // transfers per wave and chunk: 8

struct T8Src {                 // the four 64-row operand halves (row 0, column 0 of each) and their common leading dimension
    const double* h[4];        // A0, A1, B0, B1
    long ld;
};

__device__ __forceinline__ void t8_zero(T8Acc& acc)
{
#pragma unroll
    for (int i = 0; i < T8_MI; ++i)
#pragma unroll
        for (int j = 0; j < T8_NI; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
}

// transfer idx (0..7) of this wave for the chunk that starts at column k0: operand half idx & 3, segment wave + 4 (idx >> 2)
template <int AUX = 0>
__device__ __forceinline__ void t8_issue(const T8Src& s, unsigned lane_off, int k0, int idx, double* buf, int wave)
{
    const int hf = idx & 3, sg = wave + 4 * (idx >> 2);
    const long ku = k0 + 4 * (sg >> 1) + (sg & 1);            // wave-uniform
    const char* g = reinterpret_cast<const char*>(s.h[hf] + ku * s.ld) + lane_off;
    __builtin_amdgcn_global_load_lds((tg_glb_void*)g, (tg_lds_void*)(buf + hf * T8_HALF_DOUBLES + sg * TD_SEG), 16, 0, AUX);
}

template <int AUX = 0>
__device__ __forceinline__ void t8_issue_chunk(const T8Src& s, unsigned lane_off, int k0, double* buf, int wave)
{
#pragma unroll
    for (int idx = 0; idx < T8_XFERS; ++idx) t8_issue<AUX>(s, lane_off, k0, idx, buf, wave);
}

struct T8Frag { double a[T8_MI]; BFrag b[T8_NI]; };

// fragment read pointers of a wave inside one chunk buffer (k-step 0)
struct T8Ptr { const double* a0; const double* a1; const double* b[4]; };
__device__ __forceinline__ T8Ptr t8_ptr(const double* buf, int wave)
{
    const int lane = threadIdx.x & 63;
    const int kq = lane >> 4, ij = lane & 15;
    const int koff = (kq & 1) * TD_SEG + (kq >> 1) * 64;
    const double* bb = buf + (2 + (wave >> 1)) * T8_HALF_DOUBLES + koff + 32 * (wave & 1);
    T8Ptr p;
    p.a0 = buf + koff + ij;
    p.a1 = buf + T8_HALF_DOUBLES + koff + ij;
    p.b[0] = bb + ij; p.b[1] = bb + ((ij - 4) & 15); p.b[2] = bb + ((ij - 8) & 15); p.b[3] = bb + ((ij - 12) & 15);
    return p;
}
__device__ __forceinline__ void t8_read(const T8Ptr& p, int o, T8Frag& f)
{
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) f.a[mi] = p.a0[o + 16 * mi];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) f.a[4 + mi] = p.a1[o + 16 * mi];
#pragma unroll
    for (int ni = 0; ni < T8_NI; ++ni) f.b[ni] = BFrag{ p.b[0][o + 16 * ni], p.b[1][o + 16 * ni], p.b[2][o + 16 * ni], p.b[3][o + 16 * ni] };
}

// wait until at most N of this wave's transfers are outstanding (they complete in order), then the workgroup barrier
template <int N>
__device__ __forceinline__ void t8_wait_barrier()
{
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(N) : "memory");
}

// One staged chunk (4 k-steps of 64 MFMAs); on entry f holds the fragments of its k-step 0.
//   AHEAD2: the chunk two ahead (column k0 + 2 T8_KC) flies into `far`, two transfers per k-step;
//   AHEAD1: there is a next chunk (`nxt`): in front of the last k-step's MFMAs this wave waits for ITS transfers of that chunk
//           (issued a whole chunk ago: vmcnt(8) leaves the ones just issued in flight), the workgroup barrier follows, and the
//           first fragments of the next chunk are read under the last k-step's MFMAs.
// A wave at the barrier therefore never waits for memory that was requested less than one chunk (4096 MFMA cycles) ago.
// NEGA: acc -= A B^T.
template <bool AHEAD2, bool AHEAD1, int AUX = 0, int NEGA = 0>
__device__ __forceinline__ void t8_compute_chunk(const double* cur, T8Acc& acc, T8Frag& f, const T8Src& s, unsigned lane_off,
                                                 int k0, double* nxt, double* far, int wave)
{
    const T8Ptr pc = t8_ptr(cur, wave);
#pragma unroll
    for (int ks = 0; ks < T8_KC / 4; ++ks) {
        T8Frag nf;
        const bool last = (ks == T8_KC / 4 - 1);
        if (AHEAD2) { t8_issue<AUX>(s, lane_off, k0 + 2 * T8_KC, 2 * ks, far, wave); t8_issue<AUX>(s, lane_off, k0 + 2 * T8_KC, 2 * ks + 1, far, wave); }
        if (!last) {
            t8_read(pc, (ks + 1) * TD_STEP, nf);
        } else if (AHEAD1) {
            if (AHEAD2) t8_wait_barrier<T8_XFERS>(); else t8_wait_barrier<0>();
            __builtin_amdgcn_sched_barrier(0);
            t8_read(t8_ptr(nxt, wave), 0, nf);
        }
#pragma unroll
        for (int mi = 0; mi < T8_MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < T8_NI; ++ni) tg_mma_16x16x4<NEGA>(f.a[mi], f.b[ni], acc[mi][ni]);
        if (AHEAD2) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);         // the two transfers
        if (!last || AHEAD1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);             // 4 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);             // 1 LDS read
            }
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * T8_MI * T8_NI, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!last || AHEAD1) f = nf;
    }
}

// acc (+/-)= A(128 x K) B(128 x K)^T; K a multiple of 64 (four chunks: the ring's buffers are compile-time constants);
// lds holds T8_LDS_DOUBLES; ends with a barrier.  The waits count this wave's transfers: loads the caller issued BEFORE the
// call are fine (older: they complete first), stores must not be outstanding.
template <int AUX = 0, int NEGA = 0>
__device__ __forceinline__ void tile_gemm128_nt(const T8Src& s, int K, double* lds, T8Acc& acc)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lane_off = td_lane_offset(s.ld);
    double* b0 = lds;
    double* b1 = lds + T8_BUF_DOUBLES;
    double* b2 = lds + 2 * T8_BUF_DOUBLES;
    double* b3 = lds + 3 * T8_BUF_DOUBLES;
    const int nq = K / (4 * T8_KC);
    if (nq <= 0) return;
    // (loads the caller has in flight are older than every transfer: the counted waits below cover them -- they only must
    //  not be STORES, which may be acknowledged out of order with loads)
    t8_issue_chunk<AUX>(s, lane_off, 0, b0, wave);
    t8_issue_chunk<AUX>(s, lane_off, T8_KC, b1, wave);
    t8_wait_barrier<T8_XFERS>();                            // chunk 0 is in
    T8Frag f;
    t8_read(t8_ptr(b0, wave), 0, f);
    for (int q = 0; q + 1 < nq; ++q) {
        const int k0 = 4 * T8_KC * q;
        t8_compute_chunk<true, true, AUX, NEGA>(b0, acc, f, s, lane_off, k0, b1, b2, wave);
        t8_compute_chunk<true, true, AUX, NEGA>(b1, acc, f, s, lane_off, k0 + T8_KC, b2, b3, wave);
        t8_compute_chunk<true, true, AUX, NEGA>(b2, acc, f, s, lane_off, k0 + 2 * T8_KC, b3, b0, wave);
        t8_compute_chunk<true, true, AUX, NEGA>(b3, acc, f, s, lane_off, k0 + 3 * T8_KC, b0, b1, wave);
    }
    {
        const int k0 = 4 * T8_KC * (nq - 1);
        t8_compute_chunk<true, true, AUX, NEGA>(b0, acc, f, s, lane_off, k0, b1, b2, wave);
        t8_compute_chunk<true, true, AUX, NEGA>(b1, acc, f, s, lane_off, k0 + T8_KC, b2, b3, wave);
        t8_compute_chunk<false, true, AUX, NEGA>(b2, acc, f, s, lane_off, k0 + 2 * T8_KC, b3, b0, wave);
        t8_compute_chunk<false, false, AUX, NEGA>(b3, acc, f, s, lane_off, k0 + 3 * T8_KC, b0, b1, wave);
    }
    __syncthreads();
}

// The 64 x 64 quadrant (QI, qj) of the macro tile into an LDS tile Cs[col][row] (ld = TS_LD): written by the two waves that
// own the quadrant's columns (wave >> 1 == qj), rows 64 qi .. of their accumulators.  Callers put barriers around it.
template <int QI>
__device__ __forceinline__ void t8_quadrant_to_lds(const T8Acc& acc, int qj, double* Cs, double scale)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if ((wave >> 1) != qj) return;
    const int lane = threadIdx.x & 63;
    const int i = lane >> 4, blk = (lane >> 2) & 3, j = lane & 3;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < T8_NI; ++ni)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = mi * 16 + 4 * blk + i;
                const int col = 32 * (wave & 1) + ni * 16 + 4 * ((blk - t) & 3) + j;
                Cs[col * TS_LD + row] = acc[4 * QI + mi][ni][t] * scale;      // (QI is a template argument: accumulators are never indexed dynamically)
            }
}

}  // namespace rslam
