// Experiment (round 3): what does an LDS-DMA fill (`buffer_load_dwordx4 ... lds`, 64 lanes x 16 B per instruction) sustain per CU,
// as a function of  waves per CU  x  instructions in flight per wave  x  contiguity of an instruction's 64 pieces  x  where the
// bytes live (2 MB = every XCD's L2, 64 MB = memory-side cache, 2 GB = HBM)?  The 3x3 kernels of this code base all measured
// ~7.2 TB/s (28 GB/s per CU) of fill whatever the layer; this probe says which of those knobs that number belongs to.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/dma_fill_probe.hip -o tools/experiments/bin/dmafill && tools/experiments/bin/dmafill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef int i32x4 __attribute__((ext_vector_type(4)));
// LDS-DMA from inline assembly: the compiler does not know it stores to LDS, so it inserts no `s_waitcnt vmcnt(0)` in front of later LDS reads
__device__ __forceinline__ i32x4 dma_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    return (i32x4){(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p; }
__device__ __forceinline__ void dma16(const i32x4 rsrc, unsigned lds, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds), "v"(voff), "s"(rsrc) : "memory", "m0");
}

// one wave = NI instructions per round into its own NI KiB of LDS; PIPE = 1: round r + 1 is issued before round r is waited for
template <int NI, int PIPE>
__global__ __launch_bounds__(1024) void fill(const char* src, unsigned footprint, int rounds, int P, unsigned S, unsigned span) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wpb = blockDim.x >> 6;
    const unsigned gw = blockIdx.x * wpb + wv, nwaves = gridDim.x * wpb;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 0xffffffffu, 0x00020000);
    const unsigned loff = (unsigned)(lane / P) * S + (unsigned)(lane % P) * 16u;
    const unsigned runs = P == 64 ? 1u : P == 32 ? 8u : 16u, runs_log = P == 64 ? 0u : P == 32 ? 3u : 4u, fmask = (footprint >> 1) - 1;   // (half the buffer: base + span stays inside)
    char* my = smem + (size_t)wv * NI * 1024 * (PIPE ? 2 : 1);
    auto issue = [&](int r) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const unsigned c = ((unsigned)r * nwaves + gw) * NI + i;
            // neighbouring waves read neighbouring runs of the same rows (strips of one image), so that every row is read completely;
            // shifts and masks only (runs, span and the footprint are powers of two): the address arithmetic must not be the measurement
            const unsigned base = (((c >> runs_log) * span) + (c & (runs - 1)) * (unsigned)(P * 16)) & fmask;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(my + ((PIPE ? (r & 1) * NI : 0) + i) * 1024), 16, (base & ~15u) + loff, 0, 0, 0);
        }
    };
    if (PIPE) {
        issue(0);
        for (int r = 0; r < rounds; ++r) {
            if (r + 1 < rounds) { issue(r + 1); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else {
        for (int r = 0; r < rounds; ++r) { issue(r); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
}

// the same address stream into REGISTERS (buffer_load_dwordx4), NI loads in flight per wave, xor-folded so nothing is dropped
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int NI>
__global__ __launch_bounds__(1024) void fill_reg(const char* src, unsigned footprint, int rounds, int P, unsigned S, unsigned span, unsigned* sink) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wpb = blockDim.x >> 6;
    const unsigned gw = blockIdx.x * wpb + wv, nwaves = gridDim.x * wpb;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 0xffffffffu, 0x00020000);
    const unsigned loff = (unsigned)(lane / P) * S + (unsigned)(lane % P) * 16u;
    const unsigned runs = P == 64 ? 1u : P == 32 ? 8u : 16u, runs_log = P == 64 ? 0u : P == 32 ? 3u : 4u, fmask = (footprint >> 1) - 1;   // (half the buffer: base + span stays inside)
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < rounds; ++r) {
        u32x4 v[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const unsigned c = ((unsigned)r * nwaves + gw) * NI + i;
            const unsigned base = (((c >> runs_log) * span) + (c & (runs - 1)) * (unsigned)(P * 16)) & fmask;
            v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (base & ~15u) + loff, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) acc ^= v[i];
    }
    if (acc[0] == 0x12345678u && acc[1] == 77u) sink[threadIdx.x] = acc[2] ^ acc[3];
}
template <int NI>
static int run_reg(const char* d, unsigned footprint, int bpc, int wpb, int P, unsigned S, const char* what, unsigned* sink) {
    const unsigned span = P == 64 ? 1024u : ((64 + P - 1) / P) * S;
    const int blocks = 256 * bpc;
    int rounds = (int)(512e6 / ((double)blocks * wpb * NI * 1024));
    if (rounds < 4) rounds = 4;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    fill_reg<NI><<<blocks, wpb * 64>>>(d, footprint, rounds, P, S, span, sink);
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < 3; ++rep) fill_reg<NI><<<blocks, wpb * 64>>>(d, footprint, rounds, P, S, span, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    const double bytes = (double)rounds * blocks * wpb * NI * 1024;
    printf("%-5s P=%2d  %2d waves/CU (%d x %2d)  NI=%2d REG   in flight/CU %4.0f KB : %6.1f GB/s per CU  %5.2f TB/s\n", what, P, bpc * wpb, bpc, wpb, NI,
           (double)bpc * wpb * NI, bytes / ms / 1e6 / 256, bytes / ms / 1e9);
    return 0;
}

// The access pattern of the channel-blocked weight gradient (conv3x3_wgrad_c8w_kernel) with nothing but the DMA: X = [32 images][18 groups]
// [256 x 256 pixels][16 B]; a block = (image, one of 8 strips of 32 columns, one of 2 row segments) walks down 32 steps of 4 rows; wave w
// brings groups w and w + 8 of the block's 10 (2 instructions of 2 rows x 512 B per group and step).  BAR = 1: s_barrier per step, as in the kernel.
template <int BAR, int DEPTH>
__global__ __launch_bounds__(512) void conv_like(const char* src, int rounds_unused) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int k = blockIdx.x;
    const int img0 = k & 7; k >>= 3;
    const int cib = k & 1; k >>= 1;
    const int tx = k & 7; k >>= 3;
    const int seg = k & 1; k >>= 1;
    const int n = k * 8 + img0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 0xffffffffu, 0x00020000);
    const unsigned plane = 1u << 20, img = 18u << 20;
    char* my = smem + wv * (8 * 1024);
    const unsigned lo = (unsigned)(lane >> 5) * 4096u + (unsigned)(lane & 31) * 16u + (unsigned)tx * 512u;
    for (int t = 0; t < 32; ++t) {
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
            const int g = wv + 8 * g2;
            if (g >= 10) continue;
            const unsigned base = (unsigned)n * img + (unsigned)(cib * 8 + g) * plane + (unsigned)(seg * 128 + 4 * t) * 4096u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(my + ((t % DEPTH) * 4 + g2 * 2) * 1024), 16, base + lo, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(my + ((t % DEPTH) * 4 + g2 * 2 + 1) * 1024), 16, base + 8192u + lo, 0, 0, 0);
        }
        if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (wv < 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        if (BAR) __builtin_amdgcn_s_barrier();
    }
}
// The same stream with the kernel's OTHER LDS traffic beside it: every wave reads NR x 512 B of LDS (ds_read_b64, conflict-free) and issues
// NM MFMAs between issuing the loads of step t + 1 and waiting for them.  STAGE 0 = LDS-DMA (builtin); 1 = buffer_load into registers, ds_write_b128 after the reads;
// 2 = LDS-DMA issued from inline assembly (no compiler-inserted vmcnt(0) in front of the LDS reads).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int STAGE, int NR, int NM>
__global__ __launch_bounds__(512) void conv_busy(const char* src, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int k = blockIdx.x;
    const int img0 = k & 7; k >>= 3;
    const int cib = k & 1; k >>= 1;
    const int tx = k & 7; k >>= 3;
    const int seg = k & 1; k >>= 1;
    const int n = k * 8 + img0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 0xffffffffu, 0x00020000);
    const i32x4 rs2 = dma_rsrc(src, 0xffffffffu);
    const unsigned plane = 1u << 20, img = 18u << 20;
    char* my = smem + wv * (8 * 1024);
    const unsigned lo = (unsigned)(lane >> 5) * 4096u + (unsigned)(lane & 31) * 16u + (unsigned)tx * 512u;
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    unsigned long long x = 0;
    for (int t = 0; t < 32; ++t) {
        u32x4 st[4];
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
            const int g = wv + 8 * g2;
            if (g >= 10) continue;
            const unsigned base = (unsigned)n * img + (unsigned)(cib * 8 + g) * plane + (unsigned)(seg * 128 + 4 * t) * 4096u;
            if (STAGE == 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(my + ((t & 1) * 4 + g2 * 2) * 1024), 16, base + lo, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(my + ((t & 1) * 4 + g2 * 2 + 1) * 1024), 16, base + 8192u + lo, 0, 0, 0);
            } else if (STAGE == 2) {
                dma16(rs2, lds_addr(my) + ((t & 1) * 4 + g2 * 2) * 1024, base + lo);
                dma16(rs2, lds_addr(my) + ((t & 1) * 4 + g2 * 2 + 1) * 1024, base + 8192u + lo);
            } else {
                st[g2 * 2] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + lo, 0, 0);
                st[g2 * 2 + 1] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + 8192u + lo, 0, 0);
            }
        }
        // the "compute" of step t: LDS reads of the other stage + MFMAs
        const char* rd = my + ((t & 1) ^ 1) * 4096;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            unsigned long long vv;
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(vv) : "v"(lds_addr(rd) + lane * 8), "n"((i & 7) * 512));      // (reads stay in flight: waited for in groups below)
            if ((i & 7) == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            x ^= vv;
            if (NM && (i % ((NR + NM - 1) / NM)) == 0) {
                bf16x8 a, b;
#pragma unroll
                for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(x & 3); b[e] = (__bf16)1.0f; }
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m], 0, 0, 0);
            }
        }
        if (STAGE == 1) {
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                if (wv + 8 * g2 >= 10) continue;
                *reinterpret_cast<u32x4*>(my + ((t & 1) * 4 + g2 * 2) * 1024 + lane * 16) = st[g2 * 2];
                *reinterpret_cast<u32x4*>(my + ((t & 1) * 4 + g2 * 2 + 1) * 1024 + lane * 16) = st[g2 * 2 + 1];
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (x == 0x123456789ull) sink[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
template <int STAGE, int NR, int NM>
static int run_busy(const char* d, float* sink) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_busy<STAGE, NR, NM>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    conv_busy<STAGE, NR, NM><<<512, 512, 8 * 8 * 1024>>>(d, sink);
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < 5; ++rep) conv_busy<STAGE, NR, NM><<<512, 512, 8 * 8 * 1024>>>(d, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("conv-like stream + per wave and step %2d LDS reads of 512 B, %2d MFMAs, %s: %.1f us\n", NR, NM * 4, STAGE == 1 ? "register-staged (buffer_load + ds_write_b128)" : STAGE == 2 ? "LDS-DMA from inline asm" : "LDS-DMA (builtin)", ms * 1e3);
    return 0;
}

template <int BAR, int DEPTH>
static int run_conv(const char* d) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_like<BAR, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    conv_like<BAR, DEPTH><<<512, 512, 8 * 8 * 1024>>>(d, 0);
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < 5; ++rep) conv_like<BAR, DEPTH><<<512, 512, 8 * 8 * 1024>>>(d, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double bytes = 512.0 * 32 * 10 * 2048;
    printf("conv-like X stream (512 blocks x 8 waves, 10 groups, 32 steps of 4 rows x 512 B): barrier %d depth %d : %.1f us  %.2f TB/s\n", BAR, DEPTH, ms * 1e3, bytes / ms / 1e9);
    return 0;
}

template <int NI, int PIPE>
static int run(const char* d, unsigned footprint, int bpc, int wpb, int P, unsigned S, const char* what) {
    const unsigned span = P == 64 ? 1024u : ((64 + P - 1) / P) * S;
    const int blocks = 256 * bpc;
    const size_t lds = (size_t)wpb * NI * 1024 * (PIPE ? 2 : 1);
    if (lds * bpc > 160 * 1024) return 0;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fill<NI, PIPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const double target = 512e6;                      // bytes per launch
    int rounds = (int)(target / ((double)blocks * wpb * NI * 1024));
    if (rounds < 4) rounds = 4;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    fill<NI, PIPE><<<blocks, wpb * 64, lds>>>(d, footprint, rounds, P, S, span);
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < 3; ++rep) fill<NI, PIPE><<<blocks, wpb * 64, lds>>>(d, footprint, rounds, P, S, span);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    const double bytes = (double)rounds * blocks * wpb * NI * 1024;
    printf("%-5s P=%2d  %2d waves/CU (%d x %2d)  NI=%2d %s  in flight/CU %4.0f KB : %6.1f GB/s per CU  %5.2f TB/s\n", what, P, bpc * wpb, bpc, wpb, NI,
           PIPE ? "pipe" : "sync", (double)bpc * wpb * NI * (PIPE ? 2 : 1), bytes / ms / 1e6 / 256, bytes / ms / 1e9);
    return 0;
}

int main() {
    char* d; const size_t big = 2049ull << 20;
    CK(hipMalloc(&d, big)); CK(hipMemset(d, 1, big));
    struct { unsigned fp; const char* name; } fps[] = {{2u << 20, "L2"}, {64u << 20, "MALL"}, {2048u << 20, "HBM"}};
    unsigned* sink; CK(hipMalloc(&sink, 4096));
    if (getenv("DMAFILL_CONV")) {
        if (run_conv<0, 1>(d)) return 1;
        if (run_conv<1, 1>(d)) return 1;
        if (run_conv<0, 2>(d)) return 1;
        if (run_conv<1, 2>(d)) return 1;
        float* fs = reinterpret_cast<float*>(sink);
        if (run_busy<0, 0, 0>(d, fs)) return 1;
        if (run_busy<0, 40, 0>(d, fs)) return 1;
        if (run_busy<0, 80, 0>(d, fs)) return 1;
        if (run_busy<0, 40, 12>(d, fs)) return 1;
        if (run_busy<0, 0, 12>(d, fs)) return 1;
        if (run_busy<1, 0, 0>(d, fs)) return 1;
        if (run_busy<1, 40, 0>(d, fs)) return 1;
        if (run_busy<1, 80, 0>(d, fs)) return 1;
        if (run_busy<1, 40, 12>(d, fs)) return 1;
        if (run_busy<2, 0, 0>(d, fs)) return 1;
        if (run_busy<2, 40, 0>(d, fs)) return 1;
        if (run_busy<2, 80, 0>(d, fs)) return 1;
        if (run_busy<2, 40, 12>(d, fs)) return 1;
        if (run_busy<2, 160, 12>(d, fs)) return 1;
        if (run_busy<0, 160, 12>(d, fs)) return 1;
        return 0;
    }
    if (getenv("DMAFILL_REG")) {
        for (auto& f : fps)
            for (int P : {64, 32}) {
                if (run_reg<8>(d, f.fp, 1, 4, P, 4096, f.name, sink)) return 1;
                if (run_reg<8>(d, f.fp, 1, 8, P, 4096, f.name, sink)) return 1;
                if (run_reg<8>(d, f.fp, 2, 8, P, 4096, f.name, sink)) return 1;
                if (run_reg<4>(d, f.fp, 2, 16, P, 4096, f.name, sink)) return 1;
                if (run_reg<8>(d, f.fp, 2, 16, P, 4096, f.name, sink)) return 1;
                if (run_reg<16>(d, f.fp, 2, 8, P, 4096, f.name, sink)) return 1;
            }
        return 0;
    }
    for (auto& f : fps) {
        for (int P : {64, 32, 18}) {
            const unsigned S = 4096;
            // waves per CU sweep at NI = 8, sync and pipelined
            for (auto cfg : {std::pair<int, int>{1, 4}, {1, 8}, {2, 8}, {4, 4}, {2, 16}}) {
                if (run<8, 0>(d, f.fp, cfg.first, cfg.second, P, S, f.name)) return 1;
                if (run<8, 1>(d, f.fp, cfg.first, cfg.second, P, S, f.name)) return 1;
            }
            // depth sweep at 16 waves per CU
            if (run<2, 1>(d, f.fp, 2, 8, P, S, f.name)) return 1;
            if (run<4, 1>(d, f.fp, 2, 8, P, S, f.name)) return 1;
            if (run<16, 0>(d, f.fp, 1, 8, P, S, f.name)) return 1;
            if (run<4, 1>(d, f.fp, 4, 8, P, S, f.name)) return 1;
        }
    }
    return 0;
}
