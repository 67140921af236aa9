// tile_gemm.h -- FP64 MFMA tile engine for gfx950 (CDNA4).
//
// One workgroup (256 threads = 4 waves, one per SIMD) owns a 64 x 64 output
// tile C = A * B^T where A is 64 x K and B is 64 x K, both column-major with the
// row index contiguous ("NT" form).  Every dense contraction of the EKF update
// has this shape (covariance rank-r update Y*Y^T, panel solve X*Linv^T,
// trailing update L_ik*L_jk^T), so there is exactly one MFMA inner loop.
//
// Instruction choice (measured on MI355X, scripts/probe.py): v_mfma_f64_16x16x4_f64
// issues at ~100 cycles per 2048 flop however many waves are resident (49 TF/s
// chip-wide), v_mfma_f64_4x4x4_4b_f64 at 16-20 cycles per 512 flop (60-75 TF/s,
// the 78.6 TF/s datasheet rate), so the engine is built on the 4x4x4 form.
// Its lane maps (verified by tests/test_gpu_parity.py::test_mfma4_lane_map):
//   A: lane l supplies A_blk[i][k],  B: B_blk[k][j],  with k = l>>4, blk = (l>>2)&3, i|j = l&3
//   D: lane l receives D_blk[i][j]  with i = l>>4, blk = (l>>2)&3, j = l&3
// i.e. four independent 4x4x4 products per instruction (the CBSZ/ABID operand
// broadcast is ignored for f64).  A 16 x 16 x 4 product needs all 16 (row-block,
// column-block) pairs: 4 instructions on the same A fragment and the B fragment
// rotated by 0/4/8/12 lanes inside each 16-lane row (DPP row_ror: VALU work that
// overlaps the matrix pipe), instead of extra LDS reads.  The per-lane operand
// fetch is the same as for 16x16x4 (row = l & 15, k = l >> 4).
//
// Each wave computes a 32 x 32 sub-tile as 2 x 2 tiles of 16 x 16 (16 independent
// accumulators).  LDS image per operand and buffer: [KC][LDS_LD] doubles, row index
// contiguous, so global->LDS is a straight 16-byte copy and every ds_read_b64 of
// a fragment touches 16 consecutive doubles per k.  LDS_LD = 80 doubles (= 160
// dwords = 32 mod 64 banks) puts the two k-rows a 32-lane half reads on disjoint banks.
#pragma once
#include <hip/hip_runtime.h>

namespace rslam {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int TG_TILE = 64;     // output tile edge per workgroup
#ifndef TG_KC_VALUE
#define TG_KC_VALUE 32
#endif
constexpr int TG_KC = TG_KC_VALUE;       // K chunk staged per barrier
constexpr int TG_LD = 80;       // LDS leading dimension (doubles)
constexpr int TG_THREADS = 256;
constexpr int TG_OPER_DOUBLES = TG_KC * TG_LD;              // one operand, one buffer
constexpr int TG_LDS_DOUBLES = 4 * TG_OPER_DOUBLES;         // A,B x 2 buffers = 80 KiB

// Sub-tile of a wave: TG_MI x TG_NI tiles of 16 x 16.  The B fragment of a 16-column tile has to be read four
// times (block rotations), the A fragment of a 16-row tile once, so a tall, narrow wave tile reads less LDS per
// MFMA: 2 x 2 (waves 2 x 2) = 10 reads per 16 MFMAs, 4 x 1 (waves 1 x 4) = 8.
#if defined(TG_LAYOUT_2X2)
constexpr int TG_MI = 2, TG_NI = 2;
#else
constexpr int TG_MI = 4, TG_NI = 1;
#endif
constexpr int TG_WN = 4 / TG_NI;   // waves along N
typedef d4 TgAcc[TG_MI][TG_NI];

constexpr int TG_NQ = TG_KC / 8;   // 16-byte loads per thread, operand and chunk
struct TileRegs { d2 a[TG_NQ]; d2 b[TG_NQ]; };

// Each thread moves rows (2*rp, 2*rp+1) of columns cq + 8*q, q = 0..3.
__device__ __forceinline__ void tg_load_chunk(const double* __restrict__ A, long lda,
                                              const double* __restrict__ B, long ldb,
                                              int k0, TileRegs& r)
{
    const int t = threadIdx.x;
    const int rp = t & 31, cq = t >> 5;
#pragma unroll
    for (int q = 0; q < TG_NQ; ++q) {
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
    for (int q = 0; q < TG_NQ; ++q) {
        const int c = cq + 8 * q;
        *reinterpret_cast<d2*>(As + c * TG_LD + 2 * rp) = r.a[q];
        *reinterpret_cast<d2*>(Bs + c * TG_LD + 2 * rp) = r.b[q];
    }
}

// rotate a double by N lanes inside every 16-lane row: lane i receives lane (i - N) mod 16
template <int N>
__device__ __forceinline__ double row_ror_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x120 + N, 0xF, 0xF, true);   // every lane has a source: no "old" operand
    hi = __builtin_amdgcn_mov_dpp(hi, 0x120 + N, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// B fragment of one 16-column tile with its three block rotations
struct BFrag { double r0, r1, r2, r3; };
__device__ __forceinline__ BFrag tg_rotations(double b)
{
    BFrag f;
#if defined(TG_FAKE_ROT)        // timing experiment only (wrong results): no DPP work
    f.r0 = b; f.r1 = b; f.r2 = b; f.r3 = b;
#else
    f.r0 = b; f.r1 = row_ror_f64<4>(b); f.r2 = row_ror_f64<8>(b); f.r3 = row_ror_f64<12>(b);
#endif
    return f;
}

// acc[t] accumulates, for lane (i = l>>4, blk = (l>>2)&3, j = l&3), the element
//   row 4*blk + i,  column 4*((blk - t) & 3) + j   of a 16 x 16 tile.
// NEGA: acc -= a b (for f64 MFMAs the BLGP field of the instruction is the neg:[a,b,c] modifier: free)
template <int NEGA = 0>
__device__ __forceinline__ void tg_mma_16x16x4(double a, const BFrag& b, d4& acc)
{
    acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b.r0, acc[0], 0, 0, NEGA);
    acc[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b.r1, acc[1], 0, 0, NEGA);
    acc[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b.r2, acc[2], 0, 0, NEGA);
    acc[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b.r3, acc[3], 0, 0, NEGA);
}

// One staged K chunk.  Software pipelined by hand: the fragments of k-step s+1 are read
// from LDS while the 16 MFMAs of k-step s occupy the matrix pipe.  The three rotated copies
// of each B fragment are read straight from LDS with rotated per-lane addresses (an LDS read
// is asynchronous; the DPP alternative costs 12 VALU issues per k-step on the same SIMD the
// MFMAs issue from -- measured 15 % slower).
__device__ __forceinline__ void tg_compute_chunk(const double* As, const double* Bs, TgAcc& acc)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave / TG_WN, wn = wave % TG_WN;
    const int kq = lane >> 4, ij = lane & 15;
    const double* ap = As + kq * TG_LD + wm * 16 * TG_MI + ij;
    const double* bb = Bs + kq * TG_LD + wn * 16 * TG_NI;
    const double* bp0 = bb + ij;
    const double* bp1 = bb + ((ij - 4) & 15);
    const double* bp2 = bb + ((ij - 8) & 15);
    const double* bp3 = bb + ((ij - 12) & 15);
    double a[TG_MI];
    BFrag b[TG_NI];
#pragma unroll
    for (int mi = 0; mi < TG_MI; ++mi) a[mi] = ap[16 * mi];
#if defined(TG_ROT_DPP)
#pragma unroll
    for (int ni = 0; ni < TG_NI; ++ni) b[ni] = tg_rotations(bp0[16 * ni]);
#else
#pragma unroll
    for (int ni = 0; ni < TG_NI; ++ni) b[ni] = BFrag{ bp0[16 * ni], bp1[16 * ni], bp2[16 * ni], bp3[16 * ni] };
#endif
#pragma unroll
    for (int kk = 0; kk < TG_KC; kk += 4) {
        double na[TG_MI];
        BFrag nb[TG_NI];
        if (kk + 4 < TG_KC) {
#if defined(TG_EXP_NO_LDS_READ)
            const int o = 0;
#else
            const int o = (kk + 4) * TG_LD;
#endif
#pragma unroll
            for (int mi = 0; mi < TG_MI; ++mi) na[mi] = ap[o + 16 * mi];
#if defined(TG_ROT_DPP)
#pragma unroll
            for (int ni = 0; ni < TG_NI; ++ni) nb[ni] = tg_rotations(bp0[o + 16 * ni]);
#else
#pragma unroll
            for (int ni = 0; ni < TG_NI; ++ni) nb[ni] = BFrag{ bp0[o + 16 * ni], bp1[o + 16 * ni], bp2[o + 16 * ni], bp3[o + 16 * ni] };
#endif
        }
#pragma unroll
        for (int mi = 0; mi < TG_MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < TG_NI; ++ni) tg_mma_16x16x4(a[mi], b[ni], acc[mi][ni]);
        if (kk + 4 < TG_KC) {
#pragma unroll
            for (int mi = 0; mi < TG_MI; ++mi) a[mi] = na[mi];
#pragma unroll
            for (int ni = 0; ni < TG_NI; ++ni) b[ni] = nb[ni];
        }
    }
}

// acc += A(64 x K) * B(64 x K)^T.  A, B point at row 0 / column 0 of the
// operand panels; K must be a multiple of TG_KC; lds holds TG_LDS_DOUBLES.
// Ends with a barrier: lds may be reused by the caller immediately.
__device__ __forceinline__ void tile_gemm_nt(const double* __restrict__ A, long lda,
                                             const double* __restrict__ B, long ldb,
                                             int K, double* lds, TgAcc& acc)
{
    double* As0 = lds;
    double* Bs0 = lds + TG_OPER_DOUBLES;
    double* As1 = lds + 2 * TG_OPER_DOUBLES;
    double* Bs1 = lds + 3 * TG_OPER_DOUBLES;
    const int nchunks = K / TG_KC;
    if (nchunks <= 0) return;
    if (nchunks == 2) {
        // K = 64 (panel solve / trailing update of the factor sweep): both chunks fit the two
        // buffers, so all global loads are issued at once and there is a single fill latency
        TileRegs r0, r1;
        tg_load_chunk(A, lda, B, ldb, 0, r0);
        tg_load_chunk(A, lda, B, ldb, TG_KC, r1);
        tg_store_chunk(As0, Bs0, r0);
        tg_store_chunk(As1, Bs1, r1);
        __syncthreads();
        tg_compute_chunk(As0, Bs0, acc);
        tg_compute_chunk(As1, Bs1, acc);
        __syncthreads();
        return;
    }
    TileRegs r;
    tg_load_chunk(A, lda, B, ldb, 0, r);
    tg_store_chunk(As0, Bs0, r);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const bool more = (c + 1 < nchunks);
#if !defined(TG_EXP_NO_GLOBAL)
        if (more) tg_load_chunk(A, lda, B, ldb, (c + 1) * TG_KC, r);   // in flight under the MFMAs
#endif
        if (c & 1) tg_compute_chunk(As1, Bs1, acc); else tg_compute_chunk(As0, Bs0, acc);
        if (more) { if (c & 1) tg_store_chunk(As0, Bs0, r); else tg_store_chunk(As1, Bs1, r); }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same engine with LDS-DMA staging (global_load_lds_dwordx4): the operand chunks go from global memory straight
// into LDS -- no VGPR round trip, no ds_write, no staging registers -- which the register-staged loop above pays with
// ~13 % of its rate (scripts/gemm_bench.py ablations).  One wave instruction writes 1 KiB contiguously (wave-uniform
// base + lane x 16 B), so padding cannot go inside it; instead each instruction fetches TWO k-rows that are never read
// by the same half-wave: lanes 0..31 row 4s + par, lanes 32..63 row 4s + par + 2 (64 doubles each), par = segment & 1.
// Segments are 1152 B apart (1 KiB + 128 B), so for a fragment read (lane = (kq, ij), k = 4s + kq) the half-wave
// {kq 0, 1} finds its two rows at byte offsets 0 and 1152 = 128 (mod 256) and {kq 2, 3} at 512 and 1664 = 128 (mod 256):
// each half-wave covers all 64 banks exactly once, as with the padded image.
// ---------------------------------------------------------------------------------------------------------------
constexpr int TD_SEG = 144;                               // doubles between segments
constexpr int TD_STEP = 2 * TD_SEG;                       // doubles per k-step (4 rows)
constexpr int TD_OPER_DOUBLES = (TG_KC / 2) * TD_SEG;     // one operand, one chunk: 16 segments
constexpr int TD_LDS_DOUBLES = 4 * TD_OPER_DOUBLES;       // A, B x 2 buffers = 72 KiB
typedef __attribute__((address_space(3))) void tg_lds_void;
typedef __attribute__((address_space(1))) const void tg_glb_void;

// Per-lane part of a DMA source address (bytes): rows (2 (lane & 31), +1) of k-row 2 (lane >> 5) -- loop invariant, so
// the address of a segment is a wave-uniform 64-bit base (SALU) plus this 32-bit VGPR offset: the `saddr` form of
// global_load_lds, no VALU per transfer.  (The first version rebuilt the 64-bit per-lane address with a quarter-rate
// 64-bit multiply for each of the 8 transfers of a chunk: ~15 % of the loop, scripts/probes/gemm_loop.hip.)
__device__ __forceinline__ unsigned td_lane_offset(long ld)
{
    const int lane = threadIdx.x & 63;
    return (unsigned)((2 * (lane & 31) + 2 * (lane >> 5) * ld) * (long)sizeof(double));
}

// segment sg (k-step sg >> 1, parity sg & 1) of the chunk that starts at column k0, both operands
// AUX: cache policy of the transfers (16 = sc1: operands another workgroup of the same launch has just published)
template <int AUX = 0>
__device__ __forceinline__ void td_issue_seg(const double* __restrict__ A, long lda, unsigned la, const double* __restrict__ B, long ldb, unsigned lb,
                                             int k0, int sg, double* As, double* Bs)
{
    const long ku = k0 + 4 * (sg >> 1) + (sg & 1);        // wave-uniform
    const char* ga = reinterpret_cast<const char*>(A + ku * lda) + la;
    const char* gb = reinterpret_cast<const char*>(B + ku * ldb) + lb;
    __builtin_amdgcn_global_load_lds((tg_glb_void*)ga, (tg_lds_void*)(As + sg * TD_SEG), 16, 0, AUX);
    __builtin_amdgcn_global_load_lds((tg_glb_void*)gb, (tg_lds_void*)(Bs + sg * TD_SEG), 16, 0, AUX);
}

// chunk k0 .. k0 + TG_KC of both operands -> LDS images As, Bs; four segments per wave and operand
// (wave: index 0..3 of the wave inside its four-wave engine)
template <int AUX = 0>
__device__ __forceinline__ void td_issue_chunk_w(const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb,
                                                 int k0, double* As, double* Bs, int wave)
{
    const unsigned la = td_lane_offset(lda), lb = td_lane_offset(ldb);
#pragma unroll
    for (int q = 0; q < TG_KC / 8; ++q) td_issue_seg<AUX>(A, lda, la, B, ldb, lb, k0, wave + 4 * q, As, Bs);
}
__device__ __forceinline__ void td_issue_chunk(const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb,
                                               int k0, double* As, double* Bs)
{
    td_issue_chunk_w<0>(A, lda, B, ldb, k0, As, Bs, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
}

// Fragment read addresses of a wave inside an operand image (k-step 0; k-step s is TD_STEP * s further)
struct TdFragPtr { const double* a; const double* b0; const double* b1; const double* b2; const double* b3; };
__device__ __forceinline__ TdFragPtr td_frag_ptr(const double* As, const double* Bs, int wave)
{
    const int lane = threadIdx.x & 63;
    const int wm = wave / TG_WN, wn = wave % TG_WN;
    const int kq = lane >> 4, ij = lane & 15;
    const int koff = (kq & 1) * TD_SEG + (kq >> 1) * 64;
    const double* bb = Bs + koff + wn * 16 * TG_NI;
    return TdFragPtr{ As + koff + wm * 16 * TG_MI + ij, bb + ij, bb + ((ij - 4) & 15), bb + ((ij - 8) & 15), bb + ((ij - 12) & 15) };
}
__device__ __forceinline__ TdFragPtr td_frag_ptr(const double* As, const double* Bs)
{
    return td_frag_ptr(As, Bs, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
}
__device__ __forceinline__ void td_read_frags(const TdFragPtr& p, int o, double (&a)[TG_MI], BFrag (&b)[TG_NI])
{
#pragma unroll
    for (int mi = 0; mi < TG_MI; ++mi) a[mi] = p.a[o + 16 * mi];
#pragma unroll
    for (int ni = 0; ni < TG_NI; ++ni) b[ni] = BFrag{ p.b0[o + 16 * ni], p.b1[o + 16 * ni], p.b2[o + 16 * ni], p.b3[o + 16 * ni] };
}

// One staged chunk: 8 k-steps of 16 MFMAs; on entry (a, b) hold the fragments of its k-step 0.
//  * The fragments of k-step s+1 are read while the MFMAs of k-step s run, and the instruction order is pinned
//    (sched_group_barrier: one LDS read after every second MFMA; sched_barrier between k-steps).  Left to itself the
//    compiler sinks the reads to their uses or issues the 16 reads of two k-steps in one burst and waits for all of
//    them, which starves the matrix pipe (probe: 57 -> 68.8 TFLOP/s for the loop without staging = the bare MFMA rate).
//  * NEXT: the wave's four segment pairs of the next chunk are issued one per k-step instead of in a burst, and the
//    chunk barrier sits BEFORE the MFMAs of the last k-step: the first fragments of the next chunk are read under them,
//    so no wave starts a chunk with an exposed LDS round trip.  (The other buffer is free again when every wave has
//    passed that barrier: its last reads were waited for in front of it.)
//  * wave: index 0..3 inside the four-wave engine (two engines share a 512-thread workgroup in the factor sweep's tile
//    workers); AUX: cache policy of the transfers; NEGA: acc -= A B^T.  (No run-time switch for the transfers: a branch
//    inside the k loop would split the scheduling region the pinned order lives in.  An engine that has no next chunk of
//    its own while its sibling has one passes any valid source.)
template <bool NEXT, int AUX = 0, int NEGA = 0>
__device__ __forceinline__ void td_compute_chunk_w(const double* As, const double* Bs, TgAcc& acc, double (&a)[TG_MI], BFrag (&b)[TG_NI],
                                                   const double* __restrict__ A, long lda, unsigned la, const double* __restrict__ B, long ldb, unsigned lb,
                                                   int k0n, double* An, double* Bn, int wave)
{
    const TdFragPtr cur = td_frag_ptr(As, Bs, wave);
#pragma unroll
    for (int kk = 0; kk < TG_KC; kk += 4) {
        double na[TG_MI];
        BFrag nb[TG_NI];
        const bool dma = NEXT && (kk / 4 < TG_KC / 8);
        const bool last = (kk + 4 >= TG_KC);
        if (dma) td_issue_seg<AUX>(A, lda, la, B, ldb, lb, k0n, wave + 4 * (kk / 4), An, Bn);
        if (!last) {
            td_read_frags(cur, (kk / 4 + 1) * TD_STEP, na, nb);
        } else if (NEXT) {
            __syncthreads();                                // next chunk landed (vmcnt(0) in front of the barrier), this one read
            __builtin_amdgcn_sched_barrier(0);
            td_read_frags(td_frag_ptr(An, Bn, wave), 0, na, nb);
        }
#pragma unroll
        for (int mi = 0; mi < TG_MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < TG_NI; ++ni) tg_mma_16x16x4<NEGA>(a[mi], b[ni], acc[mi][ni]);
        if (dma) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);          // the two transfers
        if (!last || NEXT) {
#pragma unroll
            for (int i = 0; i < 4 * TG_MI * TG_NI / 2; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);          // 2 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);          // 1 LDS read
            }
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * TG_MI * TG_NI, 0);
        }
        __builtin_amdgcn_sched_barrier(0);                  // nothing crosses a k-step: the reads stay one step ahead of their use
        if (!last || NEXT) {
#pragma unroll
            for (int mi = 0; mi < TG_MI; ++mi) a[mi] = na[mi];
#pragma unroll
            for (int ni = 0; ni < TG_NI; ++ni) b[ni] = nb[ni];
        }
    }
}

template <bool NEXT>
__device__ __forceinline__ void td_compute_chunk(const double* As, const double* Bs, TgAcc& acc, double (&a)[TG_MI], BFrag (&b)[TG_NI],
                                                 const double* __restrict__ A, long lda, unsigned la, const double* __restrict__ B, long ldb, unsigned lb,
                                                 int k0n, double* An, double* Bn)
{
    td_compute_chunk_w<NEXT, 0, 0>(As, Bs, acc, a, b, A, lda, la, B, ldb, lb, k0n, An, Bn, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
}

// acc += A(64 x K) * B(64 x K)^T, LDS-DMA staged.  lds holds TD_LDS_DOUBLES.  Ends with a barrier.
__device__ __forceinline__ void tile_gemm_nt_dma(const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb,
                                                 int K, double* lds, TgAcc& acc)
{
    double* As0 = lds;
    double* Bs0 = lds + TD_OPER_DOUBLES;
    double* As1 = lds + 2 * TD_OPER_DOUBLES;
    double* Bs1 = lds + 3 * TD_OPER_DOUBLES;
    const int nchunks = K / TG_KC;
    if (nchunks <= 0) return;
    const unsigned la = td_lane_offset(lda), lb = td_lane_offset(ldb);
    td_issue_chunk(A, lda, B, ldb, 0, As0, Bs0);
    __syncthreads();                                      // (a barrier waits for this wave's DMA: vmcnt(0))
    double a[TG_MI];
    BFrag b[TG_NI];
    td_read_frags(td_frag_ptr(As0, Bs0), 0, a, b);
    for (int c = 0; c + 1 < nchunks; ++c) {
        // chunk c+1 flies into the other buffer under the MFMAs of chunk c
        if (c & 1) td_compute_chunk<true>(As1, Bs1, acc, a, b, A, lda, la, B, ldb, lb, (c + 1) * TG_KC, As0, Bs0);
        else       td_compute_chunk<true>(As0, Bs0, acc, a, b, A, lda, la, B, ldb, lb, (c + 1) * TG_KC, As1, Bs1);
    }
    if ((nchunks - 1) & 1) td_compute_chunk<false>(As1, Bs1, acc, a, b, A, lda, la, B, ldb, lb, 0, As0, Bs0);
    else                   td_compute_chunk<false>(As0, Bs0, acc, a, b, A, lda, la, B, ldb, lb, 0, As1, Bs1);
    __syncthreads();
}

__device__ __forceinline__ void tg_zero(TgAcc& acc)
{
#pragma unroll
    for (int i = 0; i < TG_MI; ++i)
#pragma unroll
        for (int j = 0; j < TG_NI; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
}

// Spill the accumulators into an LDS tile Cs[col][row] (ld = TS_LD) so that the
// epilogue can read columns (coalesced global stores) or rows (transposes).
constexpr int TS_LD = 65;
constexpr int TS_DOUBLES = 64 * TS_LD;
static_assert(TG_LDS_DOUBLES >= 2 * TS_DOUBLES, "epilogues stage two 64 x 65 tiles in the operand buffers");
static_assert(TD_LDS_DOUBLES >= 2 * TS_DOUBLES, "epilogues stage two 64 x 65 tiles in the operand buffers");

template <int LD = 65>
__device__ __forceinline__ void tg_acc_to_lds_w(const TgAcc& acc, double* Cs, double scale, int wave)
{
    const int lane = threadIdx.x & 63;
    const int wm = wave / TG_WN, wn = wave % TG_WN;
    const int i = lane >> 4, blk = (lane >> 2) & 3, j = lane & 3;
#pragma unroll
    for (int mi = 0; mi < TG_MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < TG_NI; ++ni)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = (wm * TG_MI + mi) * 16 + 4 * blk + i;
                const int col = (wn * TG_NI + ni) * 16 + 4 * ((blk - t) & 3) + j;
                Cs[col * LD + row] = acc[mi][ni][t] * scale;
            }
}
template <int LD = 65>
__device__ __forceinline__ void tg_acc_to_lds(const TgAcc& acc, double* Cs, double scale)
{
    tg_acc_to_lds_w<LD>(acc, Cs, scale, (int)(threadIdx.x >> 6));
}

// A 64 x 64 operand from global memory into two consecutive operand buffers (k = 0..31, 32..63).
__device__ __forceinline__ void tg_fill64(const double* __restrict__ G, long ldg, double* S)
{
    const int t = threadIdx.x;
    const int rp = t & 31, cq = t >> 5;
    d2 r[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) r[q] = *reinterpret_cast<const d2*>(G + 2 * rp + (long)(cq + 8 * q) * ldg);
#pragma unroll
    for (int q = 0; q < 8; ++q) *reinterpret_cast<d2*>(S + (cq + 8 * q) * TG_LD + 2 * rp) = r[q];
}

// acc += A * B^T for two 64 x 64 operands already resident in LDS in operand layout ([k][TG_LD], i.e. a
// 64 x 64 matrix stored column-major with leading dimension TG_LD spans two consecutive operand buffers)
__device__ __forceinline__ void tg_gemm64_lds(const double* As, const double* Bs, TgAcc& acc)
{
    tg_compute_chunk(As, Bs, acc);
    tg_compute_chunk(As + TG_OPER_DOUBLES, Bs + TG_OPER_DOUBLES, acc);
}

}  // namespace rslam
