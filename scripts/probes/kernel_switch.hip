// What does it cost a launch to follow a DIFFERENT kernel?  Two kernels of a few instructions each (so that code size is not in it),
// 256 workgroups, launched back to back on one stream: A A A A ... against A B A B ...; wall clock over 2000 launches.
//   hipcc --offload-arch=gfx950 -O3 kernel_switch.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void ka(int* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1; }
__global__ void kb(int* p) { if (threadIdx.x == 0) p[blockIdx.x] += 2; }
// the same pair with ~6 KB of straight-line code each in front of the store (never executed: a run-time condition)
template <int SALT> __global__ void big(int* p, int n)
{
    int v = p[blockIdx.x];
    if (n > 0) {
#pragma unroll
        for (int i = 0; i < 400; ++i) v = v * (SALT + 3 + i) + (v >> (i & 7)) + i;
    }
    if (threadIdx.x == 0) p[blockIdx.x] = v + SALT;
}
int main()
{
    int* p; hipMalloc(&p, 4096 * 4); hipMemset(p, 0, 4096 * 4);
    const int N = 2000;
    for (int mode = 0; mode < 4; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {
                const bool second = (mode & 1) && (i & 1);
                if (mode < 2) { if (second) kb<<<256, 256>>>(p); else ka<<<256, 256>>>(p); }
                else { if (second) big<2><<<256, 256>>>(p, 0); else big<1><<<256, 256>>>(p, 0); }
            }
            hipDeviceSynchronize();
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            if (rep == 2) printf("%s, %s: %.2f us per launch\n", mode < 2 ? "two tiny kernels" : "two kernels of ~6 KB", (mode & 1) ? "A B A B" : "A A A A", us);
        }
    return 0;
}
