// Can two streams own disjoint sets of compute units on this box?  (hipExtStreamCreateWithCUMask)
//   1. where do the workgroups of a masked stream land (XCC / SE / CU ids from the hardware registers);
//   2. do a masked "chain" stream (few big-LDS workgroups, launch after launch) and a masked "bulk" stream (thousands of
//      72 KiB workgroups) run side by side without the placement stall of round 4 (a 120 KiB workgroup never finds a
//      compute unit while 72 KiB workgroups refill every slot);
//   3. does hipStreamWaitValue32 gate a stream on a word a running kernel writes.
// build: hipcc --offload-arch=gfx950 -O3 scripts/probes/cu_mask.hip -o scripts/probes/cu_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <set>
#include <map>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return 1; } } while (0)



// every workgroup records where it ran and spins for `ticks` of the 100 MHz clock
__global__ void where_kernel(unsigned* out, unsigned long long ticks)
{
    extern __shared__ double lds[];
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
        lds[0] = 1.0;
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
}

// stamps: start / end of the kernel (first thread of block 0)
__global__ void spin_kernel(unsigned long long* stamps, int slot, unsigned long long ticks)
{
    extern __shared__ double lds[];
    if (threadIdx.x == 0) {
        lds[0] = 1.0;
        const unsigned long long t0 = wall_clock64();
        if (blockIdx.x == 0) stamps[2 * slot] = t0;
        while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
        if (blockIdx.x == 0) stamps[2 * slot + 1] = wall_clock64();
    }
    __syncthreads();
}

__global__ void post_kernel(unsigned* flag, unsigned value, unsigned long long ticks_before)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < ticks_before) __builtin_amdgcn_s_sleep(8);
        __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        // stays alive a while: the waiter must get through while this kernel is still running
        while (wall_clock64() - t0 < 4 * ticks_before) __builtin_amdgcn_s_sleep(8);
    }
}

static void describe(const std::vector<unsigned>& v, int n)
{
    std::map<unsigned, std::set<unsigned>> per_xcc;
    for (int i = 0; i < n; ++i) {
        const unsigned hw = v[2 * i], xcc = v[2 * i + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
    }
    int total = 0;
    for (auto& kv : per_xcc) {
        printf("   xcc %u: %zu CUs:", kv.first, kv.second.size());
        for (unsigned c : kv.second) printf(" %u.%u.%u", c >> 8, (c >> 4) & 1, c & 0xf);
        printf("\n");
        total += (int)kv.second.size();
    }
    printf("   distinct compute units: %d\n", total);
}

int main(int argc, char** argv)
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    int dev = 0;
    CK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, cus);
    const int words = (cus + 31) / 32;

    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(where_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(spin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));

    unsigned* d_out; CK(hipMalloc(&d_out, sizeof(unsigned) * 2 * 8192));
    std::vector<unsigned> h_out(2 * 8192);

    // ---- 1. placement under masks
    auto run_where = [&](const char* what, const std::vector<uint32_t>& mask, int wgs) -> int {
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
        if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask -> %s\n", what, hipGetErrorString(e)); return 1; }
        CK(hipMemsetAsync(d_out, 0, sizeof(unsigned) * 2 * 8192, s));
        where_kernel<<<dim3(wgs), dim3(64), 100 * 1024, s>>>(d_out, 2000);        // 20 us each, one per CU (100 KiB of LDS)
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h_out.data(), d_out, sizeof(unsigned) * 2 * wgs, hipMemcpyDeviceToHost));
        printf("%s (%d workgroups of 100 KiB):\n", what, wgs);
        describe(h_out, wgs);
        CK(hipStreamDestroy(s));
        return 0;
    };
    std::vector<uint32_t> all(words, 0xffffffffu), low(words, 0), high(words, 0), evens(words, 0);
    const int n_low = argc > 1 ? atoi(argv[1]) : 64;
    for (int i = 0; i < cus; ++i) {
        if (i < n_low) low[i / 32] |= 1u << (i % 32); else high[i / 32] |= 1u << (i % 32);
        if ((i & 3) == 0) evens[i / 32] |= 1u << (i % 32);
    }
    if (run_where("mask: all", all, 512)) return 1;
    if (run_where("mask: first n_low bits", low, 512)) return 1;
    if (run_where("mask: the other bits", high, 512)) return 1;
    if (argc > 3 && run_where("mask: every fourth bit", evens, 512)) return 1;

    // ---- 2. chain stream beside bulk stream
    unsigned long long* d_st; CK(hipMalloc(&d_st, sizeof(unsigned long long) * 2 * 256));
    std::vector<unsigned long long> st(2 * 256);
    auto side_by_side = [&](const char* what, const std::vector<uint32_t>& mchain, const std::vector<uint32_t>& mbulk) -> int {
        hipStream_t sc, sb;
        CK(hipExtStreamCreateWithCUMask(&sc, (uint32_t)mchain.size(), mchain.data()));
        CK(hipExtStreamCreateWithCUMask(&sb, (uint32_t)mbulk.size(), mbulk.data()));
        CK(hipMemset(d_st, 0, sizeof(unsigned long long) * 2 * 256));
        CK(hipDeviceSynchronize());
        // bulk: 3 launches of 4000 workgroups x 72 KiB x 30 us  (2 per CU: ~4000 / (2 x CUs) x 30 us each)
        for (int i = 0; i < 3; ++i) spin_kernel<<<dim3(4000), dim3(256), 72 * 1024, sb>>>(d_st, 100 + i, 3000);
        // chain: 20 launches of 24 workgroups x 120 KiB x 15 us
        for (int i = 0; i < 20; ++i) spin_kernel<<<dim3(24), dim3(512), 120 * 1024, sc>>>(d_st, i, 1500);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(st.data(), d_st, sizeof(unsigned long long) * 2 * 256, hipMemcpyDeviceToHost));
        const unsigned long long t0 = st[200] < st[0] ? st[200] : st[0];
        printf("%s:\n   bulk launches:", what);
        for (int i = 0; i < 3; ++i) printf("  [%.0f .. %.0f us]", (st[200 + 2 * i] - t0) / 100.0, (st[201 + 2 * i] - t0) / 100.0);
        printf("\n   chain launches (start of block 0):");
        for (int i = 0; i < 20; ++i) printf(" %.0f", (st[2 * i] - t0) / 100.0);
        printf("\n   chain last end %.0f us\n", (st[39] - t0) / 100.0);
        CK(hipStreamDestroy(sc)); CK(hipStreamDestroy(sb));
        return 0;
    };
    if (side_by_side("chain and bulk both unmasked", all, all)) return 1;
    if (side_by_side("chain on the low bits, bulk on the others", low, high)) return 1;

    // ---- 3. hipStreamWaitValue32 on a word a running kernel writes
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    if (can && argc > 2 && atoi(argv[2]) == 1) {
        unsigned* sig = nullptr;
        hipError_t e = hipExtMallocWithFlags((void**)&sig, 64, hipMallocSignalMemory);
        if (e != hipSuccess) { printf("hipExtMallocWithFlags(signal) -> %s\n", hipGetErrorString(e)); return 0; }
        hipStream_t s1, s2;
        CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
        CK(hipMemset(sig, 0, 64));
        CK(hipMemset(d_st, 0, sizeof(unsigned long long) * 2 * 256));
        CK(hipDeviceSynchronize());
        post_kernel<<<1, 64, 0, s1>>>(sig, 7, 5000);                  // posts after 50 us, lives 200 us
        e = hipStreamWaitValue32(s2, sig, 7, hipStreamWaitValueGte, 0xffffffffu);
        if (e != hipSuccess) { printf("hipStreamWaitValue32 -> %s\n", hipGetErrorString(e)); return 0; }
        spin_kernel<<<1, 64, 1024, s2>>>(d_st, 0, 100);
        spin_kernel<<<1, 64, 1024, s1>>>(d_st, 1, 100);                // behind the poster on its own stream
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(st.data(), d_st, sizeof(unsigned long long) * 4, hipMemcpyDeviceToHost));
        printf("waiter started %.1f us before the poster's kernel ended (poster posts at 50 us, ends at 200 us: expect ~150)\n",
               ((double)st[2] - (double)st[0]) / 100.0);
    }
    printf("done\n");
    return 0;
}
