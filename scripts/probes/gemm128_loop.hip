// The 128 x 128 macro-tile engine (ransac_slam_amd/csrc/tile_gemm128.h) on a full chip: is its K loop faster than the
// 64 x 64 engine's (tile_gemm.h: 54 TFLOP/s in the C5 rank update), and is it right?
//   1. one macro tile, K = 192, against a host reference (every element);
//   2. the rank-update shape without epilogue: Y (6016 x K), lower-triangle macro tiles, K = 512 and 1600, grids of whole rounds.
// build: hipcc --offload-arch=gfx950 -O3 -Iransac_slam_amd/csrc scripts/probes/gemm128_loop.hip -o scripts/probes/gemm128_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "tile_gemm128.h"
using namespace rslam;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) tile_check_kernel(const double* A, const double* B, long ld, int K, double* C)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    T8Src s{{A, A + 64, B, B + 64}, ld};
    T8Acc acc;
    t8_zero(acc);
    tile_gemm128_nt<0, 0>(s, K, lds, acc);
    const int row = threadIdx.x & 63, g = threadIdx.x >> 6;
    double* Cs = lds;
#define QUAD(QI, QJ) do { \
        t8_quadrant_to_lds<QI>(acc, QJ, Cs, 1.0); __syncthreads(); \
        for (int q = 0; q < 16; ++q) { const int c = g + 4 * q; C[64 * QI + row + (long)(64 * QJ + c) * 128] = Cs[c * TS_LD + row]; } \
        __syncthreads(); } while (0)
    QUAD(0, 0); QUAD(1, 0); QUAD(0, 1); QUAD(1, 1);
#undef QUAD
}

// macro tile t of the lower triangle (row-major), no epilogue
template <int MODE>
__global__ void __launch_bounds__(256) loop_kernel(const double* __restrict__ Y, long ldy, int K, int nM, double* out)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int t = blockIdx.x;
    int bi = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((long)bi * (bi + 1) / 2 > t) --bi;
    while ((long)(bi + 1) * (bi + 2) / 2 <= t) ++bi;
    int bj = t - bi * (bi + 1) / 2;
    bi %= nM; bj %= nM;
    const double* A = Y + 128L * bi;
    const double* B = Y + 128L * bj;
    T8Src s{{A, A + 64, B, B + 64}, ldy};
    T8Acc acc;
    t8_zero(acc);
    tile_gemm128_nt<0, MODE>(s, K, lds, acc);
    double sum = 0;
    for (int mi = 0; mi < T8_MI; ++mi) for (int ni = 0; ni < T8_NI; ++ni) for (int q = 0; q < 4; ++q) sum += acc[mi][ni][q];
    if (sum == -1.2345) out[blockIdx.x * 256 + threadIdx.x] = sum;
}

int main(int argc, char** argv)
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const size_t lds_bytes = sizeof(double) * T8_LDS_DOUBLES;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(tile_check_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(loop_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(loop_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    {   // 1. correctness of one macro tile
        const int K = 192, ld = 200;
        std::vector<double> hA((size_t)ld * K), hB((size_t)ld * K), hC(128 * 128), ref(128 * 128, 0.0);
        srand(1);
        for (auto& v : hA) v = (rand() % 2001 - 1000) / 1000.0;
        for (auto& v : hB) v = (rand() % 2001 - 1000) / 1000.0;
        for (int i = 0; i < 128; ++i) for (int j = 0; j < 128; ++j) { double a = 0; for (int k = 0; k < K; ++k) a += hA[i + (size_t)k * ld] * hB[j + (size_t)k * ld]; ref[i + 128 * j] = a; }
        double *dA, *dB, *dC;
        CK(hipMalloc(&dA, hA.size() * 8)); CK(hipMalloc(&dB, hB.size() * 8)); CK(hipMalloc(&dC, hC.size() * 8));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 8, hipMemcpyHostToDevice));
        tile_check_kernel<<<1, 256, lds_bytes>>>(dA, dB, ld, K, dC);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hC.data(), dC, hC.size() * 8, hipMemcpyDeviceToHost));
        double worst = 0; int wi = 0;
        for (int i = 0; i < 128 * 128; ++i) { const double d = fabs(hC[i] - ref[i]); if (d > worst) { worst = d; wi = i; } }
        printf("macro tile vs host reference (K = %d): max abs diff %.3e at (%d, %d)%s\n", K, worst, wi % 128, wi / 128, worst < 1e-12 ? "  OK" : "  WRONG");
        if (!(worst < 1e-12)) return 2;
    }
    const int n = 6016, nM = n / 128;
    double *Y, *out;
    CK(hipMalloc(&Y, sizeof(double) * (size_t)n * 1664)); CK(hipMalloc(&out, sizeof(double) * 256 * 4096));
    CK(hipMemset(Y, 0, sizeof(double) * (size_t)n * 1664));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int K : {512, 1600}) {
        for (int grid : {256, 1024, 1128}) {
            for (int i = 0; i < 2; ++i) loop_kernel<1><<<grid, 256, lds_bytes>>>(Y, n, K, nM, out);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            const int reps = 4;
            for (int i = 0; i < reps; ++i) loop_kernel<1><<<grid, 256, lds_bytes>>>(Y, n, K, nM, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            const double t = ms * 1e-3 / reps, fl = 2.0 * 128 * 128 * K * grid;
            printf("K = %4d, %4d macro tiles (%.2f per CU): %8.1f us  %6.1f TFLOP/s (%.3f of 78.6)\n", K, grid, grid / 256.0, t * 1e6, fl / t * 1e-12, fl / t * 1e-12 / 78.6);
        }
    }
    printf("done\n");
    return 0;
}
