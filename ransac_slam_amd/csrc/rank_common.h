// rank_common.h -- pieces of the covariance rank update shared by its 64 x 64 form (kernels.hip, rank_update_kernel) and its
// 128 x 128 macro-tile form (rank_macro.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "tile_gemm.h"

namespace rslam {

// The Jnorm congruence (ExtendKF.cpp:629-634) on an LDS tile Cs[col][row] of the first block column: tile (bi, 0) has its
// columns 3..6 mixed, tile (0, 0) rows and columns (its 4 x 4 block symmetrised afterwards, as the reader of an immediate P_li
// would see it); every thread of the 256 calls it; ends behind a barrier.
__device__ __forceinline__ void k11_lds(double* Cs, const double (&T)[16], int bi)
{
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
            double cb[4][4], out[4][4];         // cb = J * P44 ; out = cb * J^T
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
            // (the pass that reads an immediate P_li takes 1/2 (P + P^T) of what the congruence left: the same here)
            for (int i = 0; i < 4; ++i)
                for (int c = 0; c < 4; ++c) Cs[(3 + c) * TS_LD + (3 + i)] = 0.5 * out[i][c] + 0.5 * out[c][i];
        }
    }
    __syncthreads();
}

}  // namespace rslam
