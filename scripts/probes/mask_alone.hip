// Does a kernel on a CU-masked stream run slower when no other queue is active?  (seen in the traces of the staged and the
// look-ahead route: the same launches take 3-4x longer once the other masked stream has gone idle)
// A fixed amount of FMA work per workgroup (duration ~ 1 / shader clock) on a stream masked to 216 of 256 CUs:
//   (a) alone, (b) while a second stream (masked to 32 CUs) runs one long workgroup, (c) on an unmasked stream alone.
// build: hipcc --offload-arch=gfx950 -O3 scripts/probes/mask_alone.hip -o scripts/probes/mask_alone
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(512) work_kernel(double* out, int iters)
{
    extern __shared__ double lds[];
    double a = threadIdx.x * 1e-9, b = 1.0000001, c = 0.5;
    for (int i = 0; i < iters; ++i) { a = fma(a, b, c); c = fma(c, b, a); }
    lds[threadIdx.x] = a + c;
    __syncthreads();
    if (lds[(threadIdx.x + 1) & 511] == -1.0) out[blockIdx.x] = a;
}
__global__ void hold_kernel(unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    CK(hipSetDevice(0));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(work_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    std::vector<uint32_t> mS(8, 0), mR(8, 0), all(8, 0xffffffffu);
    for (int i = 0; i < 256; ++i) { if (i < 32) mS[i / 32] |= 1u << (i % 32); else if (i >= 40) mR[i / 32] |= 1u << (i % 32); }
    hipStream_t sS, sR, sU;
    CK(hipExtStreamCreateWithCUMask(&sS, 8, mS.data()));
    CK(hipExtStreamCreateWithCUMask(&sR, 8, mR.data()));
    CK(hipStreamCreateWithFlags(&sU, hipStreamNonBlocking));
    double* out; CK(hipMalloc(&out, 8 * 4096));
    hipEvent_t e[32];
    for (auto& x : e) CK(hipEventCreate(&x));
    auto series = [&](const char* what, hipStream_t s, bool hold) -> int {
        CK(hipDeviceSynchronize());
        if (hold) hold_kernel<<<1, 64, 0, sS>>>(100ull * 3000);          // 3 ms on the other masked stream
        for (int i = 0; i < 12; ++i) {
            CK(hipEventRecord(e[2 * i], s));
            work_kernel<<<216, 512, 120 * 1024, s>>>(out, 4000);
            CK(hipEventRecord(e[2 * i + 1], s));
        }
        CK(hipDeviceSynchronize());
        printf("%-52s", what);
        for (int i = 0; i < 12; ++i) { float ms = 0; CK(hipEventElapsedTime(&ms, e[2 * i], e[2 * i + 1])); printf(" %6.1f", ms * 1e3); }
        printf(" us\n");
        return 0;
    };
    for (int rep = 0; rep < 2; ++rep) {
        if (series("masked (216 CUs), alone", sR, false)) return 1;
        if (series("masked (216 CUs), other masked stream busy", sR, true)) return 1;
        if (series("unmasked, alone", sU, false)) return 1;
        if (series("unmasked, masked stream busy", sU, true)) return 1;
    }
    printf("done\n");
    return 0;
}
