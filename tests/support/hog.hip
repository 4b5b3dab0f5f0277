// Test support only (not part of libmtbc_hip.so): a kernel that HOLDS compute units for a fixed time, the way a resident
// RCCL collective on another stream does while the backward pass runs.  One 1024-thread workgroup with all of a CU's LDS
// occupies that CU alone; every wave spins on the real-time clock (100 MHz) until `ticks` have passed -- an exit condition
// every wave reaches -- and touches no memory but one word at the end.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(1024) void hog_kernel(unsigned long long ticks, unsigned* out) {
    extern __shared__ char lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned n = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) { __builtin_amdgcn_s_sleep(32); ++n; }
    if (threadIdx.x == 0 && blockIdx.x == 0) { lds[0] = (char)n; out[0] = n + (unsigned)lds[0]; }
}

extern "C" int hog_launch(int blocks, double milliseconds, void* out_word, void* stream) {
    if (blocks < 1 || blocks > 256 || milliseconds <= 0.0 || milliseconds > 5000.0 || !out_word) return -1;
    static bool attr = false;
    if (!attr) { if (hipFuncSetAttribute((const void*)hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -2; attr = true; }
    hipLaunchKernelGGL(hog_kernel, dim3(blocks), dim3(1024), 160 * 1024, (hipStream_t)stream, (unsigned long long)(milliseconds * 1e5), (unsigned*)out_word);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
