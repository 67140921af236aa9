// Yardstick only (never linked into the product): what does the vendor library reach on the two dense shapes of the
// large-system route -- C = C - Y Y^T (dsyrk, n = 6016, k = 1600) and Y = T M^T (dgemm NT, 6080 x 512 x 512 / 1600)?
// The hand-written engine (tile_gemm.h) is priced against the FP64 MFMA peak in bench.py; this says how much of the gap
// a production DGEMM closes on the same box.
// build: hipcc -O2 scripts/probes/rocblas_yardstick.cpp -o scripts/probes/rocblas_yardstick -lrocblas
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <vector>
#define CK(x) do { auto e_ = (x); if ((int)e_ != 0) { printf("%s:%d %s -> %d\n", __FILE__, __LINE__, #x, (int)e_); return 1; } } while (0)

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    rocblas_handle h; CK(rocblas_create_handle(&h));
    const int n = 6016, r = 1600;
    double *P, *Y, *T, *M;
    CK(hipMalloc(&P, sizeof(double) * (size_t)n * n)); CK(hipMalloc(&Y, sizeof(double) * (size_t)n * r));
    CK(hipMalloc(&T, sizeof(double) * (size_t)n * r)); CK(hipMalloc(&M, sizeof(double) * (size_t)r * r));
    CK(hipMemset(P, 0, sizeof(double) * (size_t)n * n)); CK(hipMemset(Y, 0, sizeof(double) * (size_t)n * r));
    CK(hipMemset(T, 0, sizeof(double) * (size_t)n * r)); CK(hipMemset(M, 0, sizeof(double) * (size_t)r * r));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double m1 = -1.0, one = 1.0, zero = 0.0;
    auto timeit = [&](const char* what, double flops, auto&& fn) -> int {
        for (int i = 0; i < 2; ++i) fn();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        const int reps = 5;
        for (int i = 0; i < reps; ++i) fn();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-58s %8.1f us  %6.1f TFLOP/s (%.3f of 78.6)\n", what, ms / reps * 1e3, flops / (ms / reps * 1e-3) * 1e-12, flops / (ms / reps * 1e-3) * 1e-12 / 78.6);
        return 0;
    };
    for (int k : {256, 512, 1600}) {
        char buf[128];
        snprintf(buf, sizeof buf, "dsyrk lower, n = %d, k = %d (n(n+1)k flop)", n, k);
        if (timeit(buf, (double)n * (n + 1) * k, [&] { rocblas_dsyrk(h, rocblas_fill_lower, rocblas_operation_none, n, k, &m1, Y, n, &one, P, n); })) return 1;
        snprintf(buf, sizeof buf, "dgemm NT full square, n = %d, k = %d (2 n^2 k flop)", n, k);
        if (timeit(buf, 2.0 * n * n * k, [&] { rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_transpose, n, n, k, &m1, Y, n, Y, n, &one, P, n); })) return 1;
    }
    for (int k : {512, 1600}) {
        char buf[128];
        snprintf(buf, sizeof buf, "dgemm NT Y = T M^T, %d x 512 x %d", n, k);
        if (timeit(buf, 2.0 * n * 512 * k, [&] { rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_transpose, n, 512, k, &one, T, n, M, r, &zero, Y, n); })) return 1;
    }
    if (timeit("dtrsm right lower trans, 6016 x 1600", (double)n * r * r,
               [&] { rocblas_dtrsm(h, rocblas_side_right, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit, n, r, &one, M, r, T, n); })) { /* zero matrix: singular, timing only */ }
    printf("done\n");
    return 0;
}
