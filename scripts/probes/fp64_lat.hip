// Latency / accuracy probe for the FP64 ops on the pivot chain of chol_diag_kernel (gfx950).
// build: hipcc --offload-arch=gfx950 -O3 scripts/probes/fp64_lat.hip -o scripts/probes/fp64_lat
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <cstdio>
#include <cmath>
#include <vector>

__global__ void lat_kernel(double* out, unsigned long long* cyc, double seed, double m, double c0)
{
    double x = seed + threadIdx.x * 1e-9;
    unsigned long long t0, t1;
    // dependent FMA chain
    asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    asm volatile("" : "+v"(x));
#pragma unroll
    for (int i = 0; i < 256; ++i) x = fma(x, m, c0);
    asm volatile("" :: "v"(x));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = x;
    // dependent rsq chain
    double y = seed + 2.0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma unroll
    for (int i = 0; i < 64; ++i) y = __builtin_amdgcn_rsq(y) + 1.5;
    asm volatile("" :: "v"(y));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (threadIdx.x == 0) cyc[1] = t1 - t0;
    out[64 + threadIdx.x] = y;
    // dependent rcp chain
    double z = seed + 2.0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma unroll
    for (int i = 0; i < 64; ++i) z = __builtin_amdgcn_rcp(z) + 1.5;
    asm volatile("" :: "v"(z));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (threadIdx.x == 0) cyc[2] = t1 - t0;
    out[128 + threadIdx.x] = z;
    // dependent mul chain
    double w = seed;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    asm volatile("" : "+v"(w));
#pragma unroll
    for (int i = 0; i < 256; ++i) w = w * m;
    asm volatile("" :: "v"(w));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (threadIdx.x == 0) cyc[3] = t1 - t0;
    out[192 + threadIdx.x] = w;
    // independent FMA x4 chains (issue rate)
    double a = seed, b = seed + 1, c = seed + 2, d = seed + 3;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#pragma unroll
    for (int i = 0; i < 64; ++i) { a = fma(a, m, c0); b = fma(b, m, c0); c = fma(c, m, c0); d = fma(d, m, c0); }
    asm volatile("" :: "v"(a), "v"(b), "v"(c), "v"(d));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (threadIdx.x == 0) cyc[4] = t1 - t0;
    out[256 + threadIdx.x] = a + b + c + d;
    // s_memtime back-to-back, and a 1 us sleep, to calibrate the counter
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (threadIdx.x == 0) cyc[5] = t1 - t0;
    // LDS round trip: dependent ds_read chain
    __shared__ int idx[64];
    idx[threadIdx.x] = (threadIdx.x + 1) & 63;
    __syncthreads();
    int k = threadIdx.x;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma unroll
    for (int i = 0; i < 64; ++i) k = ((volatile int*)idx)[k];
    asm volatile("" :: "v"(k));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (threadIdx.x == 0) cyc[6] = t1 - t0;
    out[320 + threadIdx.x] = k;
}

__global__ void acc_kernel(const double* in, double* rsq, double* rcp, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { rsq[i] = __builtin_amdgcn_rsq(in[i]); rcp[i] = __builtin_amdgcn_rcp(in[i]); }
}

int main()
{
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 8 * 512); hipMalloc(&cyc, 8 * 8);
    hipMemset(cyc, 0, 64);
    for (int rep = 0; rep < 3; ++rep) lat_kernel<<<1, 64>>>(out, cyc, 1.0, 0.999999, 1e-7);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("s_memtime ticks: fma dep %.1f/op, rsq dep (+add) %.1f/op, rcp dep (+add) %.1f/op, mul dep %.1f/op, fma indep x4 %.1f/op, empty %llu, lds dep read %.1f/op\n",
           h[0] / 256.0, h[1] / 64.0, h[2] / 64.0, h[3] / 256.0, h[4] / 256.0, h[5], h[6] / 64.0);
    // clock calibration: time a long dependent chain with events
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 1 << 16;
    std::vector<double> in(n), r1(n), r2(n);
    for (int i = 0; i < n; ++i) in[i] = std::exp((i / (double)n) * 40.0 - 20.0) * (1.0 + (i % 97) * 1e-3);
    double *din, *d1, *d2;
    hipMalloc(&din, 8 * n); hipMalloc(&d1, 8 * n); hipMalloc(&d2, 8 * n);
    hipMemcpy(din, in.data(), 8 * n, hipMemcpyHostToDevice);
    acc_kernel<<<n / 256, 256>>>(din, d1, d2, n);
    hipMemcpy(r1.data(), d1, 8 * n, hipMemcpyDeviceToHost);
    hipMemcpy(r2.data(), d2, 8 * n, hipMemcpyDeviceToHost);
    double e_rsq = 0, e_rcp = 0;
    for (int i = 0; i < n; ++i) {
        e_rsq = std::fmax(e_rsq, std::fabs(r1[i] * std::sqrt(in[i]) - 1.0));
        e_rcp = std::fmax(e_rcp, std::fabs(r2[i] * in[i] - 1.0));
    }
    printf("max rel err: v_rsq_f64 %.3e (2^%.1f), v_rcp_f64 %.3e (2^%.1f)\n", e_rsq, std::log2(e_rsq), e_rcp, std::log2(e_rcp));
    return 0;
}
