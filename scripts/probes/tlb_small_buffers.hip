// Does it matter for a small, latency-bound launch whether its dozen small operands live in a dozen allocations (a page each) or in
// one?  One workgroup per compute unit reads one word from each of N operands (dependent on nothing), sums, stores; between two
// timed launches a 64 MB fill evicts caches (and, if small, translations).   hipcc --offload-arch=gfx950 -O3 tlb_small_buffers.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
constexpr int N = 16;
struct Ptrs { const int* p[N]; };
__global__ void touch(Ptrs ps, int* out)
{
    int s = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) s += ps.p[i][threadIdx.x & 63];
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}
__global__ void fill(int* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (int)i; }
int main()
{
    int* big; hipMalloc(&big, 64u << 20);
    int* out; hipMalloc(&out, 4096);
    std::vector<int*> sep(N);
    for (int i = 0; i < N; ++i) { hipMalloc(&sep[i], 4096); hipMemset(sep[i], 0, 4096); }
    int* one; hipMalloc(&one, N * 4096); hipMemset(one, 0, N * 4096);
    Ptrs a{}, b{};
    for (int i = 0; i < N; ++i) { a.p[i] = sep[i]; b.p[i] = one + i * 256; }        // b: 1 KB apart inside ONE allocation
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        std::vector<float> ts;
        for (int rep = 0; rep < 60; ++rep) {
            fill<<<1024, 256>>>(big, (64u << 20) / 4);
            hipEventRecord(e0);
            touch<<<256, 256>>>(which ? b : a, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms * 1e3f);
        }
        std::sort(ts.begin(), ts.end());
        printf("%s: median %.2f us  min %.2f us\n", which ? "16 operands in ONE allocation " : "16 operands in 16 allocations", ts[ts.size() / 2], ts[0]);
    }
    return 0;
}
