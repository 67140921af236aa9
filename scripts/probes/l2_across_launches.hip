// Does data written by one launch wait in the writer's XCD L2 for the next launch?  Launch W: workgroup b writes chunk b (8 KB) of an
// 8 MB buffer.  Launch R(shift): workgroup b reads chunk (b + shift) mod N.  Workgroup b runs on XCD b mod 8: shift 0 reads what
// the same XCD wrote, shift 1 what the neighbour XCD wrote, shift 8 the same XCD again (another compute unit).  Timed with hipEvents
// around R alone (their overhead is the same in every variant), medians of 80.   hipcc --offload-arch=gfx950 -O3 l2_across_launches.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
constexpr int NWG = 1024, PER = 1024;                 // doubles per chunk
__global__ void writer(double* x, double v) { x[(size_t)blockIdx.x * PER + threadIdx.x] = v + threadIdx.x; x[(size_t)blockIdx.x * PER + 256 + threadIdx.x] = v; x[(size_t)blockIdx.x * PER + 512 + threadIdx.x] = v; x[(size_t)blockIdx.x * PER + 768 + threadIdx.x] = v; }
__global__ void reader(const double* x, int shift, double* out)
{
    const size_t c = (size_t)((blockIdx.x + shift) % NWG) * PER;
    double s = x[c + threadIdx.x] + x[c + 256 + threadIdx.x] + x[c + 512 + threadIdx.x] + x[c + 768 + threadIdx.x];
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;
}
__global__ void fill(double* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0; }
int main()
{
    double *x, *out, *big;
    hipMalloc(&x, sizeof(double) * NWG * PER); hipMalloc(&out, sizeof(double) * NWG * 4); hipMalloc(&big, 512u << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int shifts[4] = {0, 1, 8, 9};
    for (int mode = 0; mode < 3; ++mode)            // 0: W then R; 1: W, a 512 MB fill, R; 2: W, R untimed, R timed
        for (int si = 0; si < 4; ++si) {
            std::vector<float> ts;
            for (int rep = 0; rep < 80; ++rep) {
                writer<<<NWG, 256>>>(x, (double)rep);
                if (mode == 1) fill<<<2048, 256>>>(big, (512u << 20) / 8);
                if (mode == 2) reader<<<NWG, 256>>>(x, shifts[si], out);
                hipEventRecord(e0);
                reader<<<NWG, 256>>>(x, shifts[si], out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms * 1e3f);
            }
            std::sort(ts.begin(), ts.end());
            printf("%-34s shift %d: median %.2f us  min %.2f\n", mode == 0 ? "read right behind the write" : mode == 1 ? "read behind a 512 MB fill" : "read a second time", shifts[si], ts[ts.size() / 2], ts[0]);
        }
    return 0;
}
