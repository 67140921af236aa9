// LDS read cost probe for the panel wave of chol_diag_kernel (gfx950): per-lane and uniform
// (broadcast) 64-bit reads, alone and with other waves of the workgroup polling LDS.
// build: hipcc --offload-arch=gfx950 -O3 scripts/probes/lds_lat.hip -o scripts/probes/lds_lat
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <cstdio>

#define STAMP(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")

template <int MODE>   // 0: alone; 1: other waves poll LDS without sleep; 2: with s_sleep 1; 3: other waves idle at a barrier
__global__ void __launch_bounds__(640) lds_kernel(double* out, unsigned long long* cyc, int p0)
{
    __shared__ double T[256];
    __shared__ double X[256];
    __shared__ int flag[16];
    const int t = threadIdx.x, l = t & 63;
    if (t < 256) { T[t] = t * 0.5; X[t] = 1.0 / (1 + t); }
    if (t < 16) flag[t] = 0;
    __syncthreads();
    if (t < 64) {
        unsigned long long acc[4] = {0, 0, 0, 0};
        double sum = 0;
        for (int it = 0; it < 64; ++it) {
            STAMP(s0);
            double a0 = T[l], a1 = T[64 + l], a2 = T[128 + l], a3 = T[192 + l];
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            STAMP(s1);
            double u[16];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) u[4 * j + k] = X[j * 64 + p0 + k];
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(u[j]));
            STAMP(s2);
#pragma unroll
            for (int j = 0; j < 4; ++j) { a0 = fma(-a1, u[4 * j], a0); a1 = fma(-a2, u[4 * j + 1], a1); a2 = fma(-a3, u[4 * j + 2], a2); a3 = fma(-a0, u[4 * j + 3], a3); }
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            STAMP(s3);
            T[l] = a0; T[64 + l] = a1; T[128 + l] = a2; T[192 + l] = a3;
            STAMP(s4);
            acc[0] += s1 - s0; acc[1] += s2 - s1; acc[2] += s3 - s2; acc[3] += s4 - s3;
            sum += a0;
        }
        if (t == 0) for (int k = 0; k < 4; ++k) cyc[k] = acc[k];
        out[l] = sum;
        __hip_atomic_store(&flag[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (MODE == 1 || MODE == 2) {
        int spins = 0;
        while (__hip_atomic_load(&flag[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0 && ++spins < (1 << 22)) {
            if (MODE == 2) __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

int main()
{
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 8 * 64); hipMalloc(&cyc, 8 * 8);
    const char* names[4] = {"1 wave alone (64 threads)", "10 waves, 9 polling LDS", "10 waves, 9 polling with s_sleep 1", "10 waves, 9 waiting at a barrier"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) lds_kernel<0><<<1, 64>>>(out, cyc, 12);
            if (mode == 1) lds_kernel<1><<<1, 640>>>(out, cyc, 12);
            if (mode == 2) lds_kernel<2><<<1, 640>>>(out, cyc, 12);
            if (mode == 3) lds_kernel<3><<<1, 640>>>(out, cyc, 12);
        }
        hipDeviceSynchronize();
        unsigned long long h[4];
        hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
        printf("%-38s per-lane 4 x b64 read %6.1f | 16 uniform b64 reads %6.1f | 16 dependent FMA %6.1f | 4 x b64 write %6.1f   (s_memtime ticks, incl. ~40 per stamp)\n",
               names[mode], h[0] / 64.0, h[1] / 64.0, h[2] / 64.0, h[3] / 64.0);
    }
    return 0;
}
