// Experiment (not part of the library): the 3x3 weight gradient with BOTH operands already stored as bf16 in the
// channel-blocked layout [C/8][H*W][8].  The contraction runs over pixels, so the
// MFMA fragments (8 consecutive pixels of one channel per lane) are the transpose of the stored pieces (8 channels of
// one pixel): they come from `ds_read_b64_tr_b16` on a [group][pixel][8 ch] LDS image that LDS-DMA fills without
// touching a VGPR.  Same block structure as conv3x3_wgrad_lp2_kernel (32 co x 32 ci per block, 4x32-pixel tiles,
// waves = ci-tile x row-half, dx taps cut with v_alignbit), only the staging differs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DDBUF] [-DBPC=4] tools/experiments/c8_wgrad_probe.hip -o /tmp/c8w && /tmp/c8w 144 24
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#ifndef BPC
#ifdef DBUF
#define BPC 3
#else
#define BPC 4
#endif
#endif

constexpr int TH = 4, TW = 32, HR = TH + 2, LW = TW + 2;
constexpr int XG = HR * LW;                 // 204 slots per channel group: 3264 B = 192 (mod 256) -> the two groups of a
constexpr int ZG = TH * TW + 4;             // 132 slots: 2112 B = 64 (mod 256)      transposed read land on disjoint banks
constexpr int XI = (4 * XG + 63) / 64;      // 13 DMA instructions of 64 slots x 16 bytes
constexpr int ZI = (4 * ZG + 63) / 64;      // 9
constexpr int BUF16 = (XI + ZI) * 64 * 8;   // 16-bit elements per stage buffer (22.5 KB)

__global__ void to_c8_kernel(const float* __restrict__ x, __bf16* __restrict__ y, int N, int C, int HW) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;          // (n, grp, px)
    const long long total = (long long)N * (C / 8) * HW;
    if (idx >= total) return;
    const int px = idx % HW; const long long t = idx / HW;
    const int grp = t % (C / 8), n = t / (C / 8);
    bf16x8 v;
    for (int e = 0; e < 8; ++e) v[e] = (__bf16)x[((size_t)n * C + grp * 8 + e) * HW + px];
    *reinterpret_cast<bf16x8*>(y + idx * 8) = v;
}

struct P {
    int N, H, W, Cin, Cout, tiles_x, tiles_y, total_tiles, tiles_per_split, ciblocks;
    const __bf16* x8; const __bf16* z8; float* partial;
};

__global__ __launch_bounds__(256, BPC) void wgrad_c8_kernel(const P p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = p.H * p.W;
    const int co0 = (blockIdx.y / p.ciblocks) * 32, ci0 = (blockIdx.y % p.ciblocks) * 32;
    const int split = blockIdx.x;
    const int t_begin = split * p.tiles_per_split, t_end = min(p.total_tiles, t_begin + p.tiles_per_split);
    const int cig = p.Cin / 8, cog = p.Cout / 8;

    // DMA slots of this lane: instruction inst = wv + 4k covers slots 64*inst .. +63 of the X (then Z) image
    int xgrp[4], xrow[4], xcol[4], zgrp[3], zpx[3];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int s = (wv + 4 * k) * 64 + lane, g = s / XG, hp = s % XG;
        xgrp[k] = (g < 4 && ci0 / 8 + g < cig) ? ci0 / 8 + g : -1;
        xrow[k] = hp / LW - 1; xcol[k] = hp % LW - 1;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int s = (wv + 4 * k) * 64 + lane, g = s / ZG, px = s % ZG;
        zgrp[k] = (g < 4 && px < TH * TW && co0 / 8 + g < cog) ? co0 / 8 + g : -1;
        zpx[k] = (px / TW) * p.W + px % TW;
    }
    auto issue = [&](int tile, unsigned short* buf) {
        int t = tile;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y; t /= p.tiles_y;
        const int n = t, x0 = tx * TW, y0 = ty * TH;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.x8 + (size_t)n * cig * HW * 8), 0, cig * HW * 16, 0x00020000);
        const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.z8 + (size_t)n * cog * HW * 8), 0, cog * HW * 16, 0x00020000);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int inst = wv + 4 * k;
            if (inst < XI) {
                const int y = y0 + xrow[k], x = x0 + xcol[k];
                const bool ok = xgrp[k] >= 0 && y >= 0 && y < p.H && x >= 0 && x < p.W;
                const unsigned voff = ok ? 16u * (unsigned)(xgrp[k] * HW + y * p.W + x) : 0xfffffff0u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(buf + inst * 512), 16, voff, 0, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int inst = wv + 4 * k;
            if (inst < ZI) {
                const unsigned voff = zgrp[k] >= 0 ? 16u * (unsigned)(zgrp[k] * HW + y0 * p.W + x0 + zpx[k]) : 0xfffffff0u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(zr, (lds_ptr_t)(buf + (XI + inst) * 512), 16, voff, 0, 0, 0);
            }
        }
    };

    const int j = lane & 15, kg = lane >> 4, q = j >> 2, pp = j & 3;
    const int it = wv & 1, kh = wv >> 1;
    // transposed-read addresses (16-bit element offsets inside a stage buffer): lane 4q+pp of a 16-lane group gives
    // row q (= pixel) and columns 4pp..4pp+3 (= channels, two 8-channel groups side by side) of the 4x16 block
    const int zoff = ((pp >> 1) * ZG + 8 * kg + q) * 8 + 4 * (pp & 1) + XI * 512;          // + c*2*ZG*8 + (row*32 + 4*blk)*8
    const int xoff = ((2 * it + (pp >> 1)) * XG + 8 * kg + q) * 8 + 4 * (pp & 1);         // + ((row+r)*LW + 4*blk)*8

    f32x4 acc[2][9];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 9; ++i) acc[c][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto compute = [&](const unsigned short* buf) {
#pragma unroll
        for (int ksl = 0; ksl < 2; ++ksl) {
            const int row = 2 * kh + ksl;
            bf16x8 a[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(buf + zoff + c * 2 * ZG * 8 + (row * TW) * 8));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(buf + zoff + c * 2 * ZG * 8 + (row * TW + 4) * 8));
                const short e[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                a[c] = __builtin_bit_cast(bf16x8, e);
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                unsigned d[6];
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(buf + xoff + ((row + r) * LW + 4 * b) * 8));
                    const uint2 u = __builtin_bit_cast(uint2, v);
                    d[2 * b] = u.x; d[2 * b + 1] = u.y;
                }
                u32x4 w0, w1, w2;           // halo columns 8kg + e + s: taps s = 0, 1, 2 start at elements 0, 1, 2
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    w0[k] = d[k];
                    w1[k] = __builtin_amdgcn_alignbit(d[k + 1], d[k], 16);
                    w2[k] = d[k + 1];
                }
                const bf16x8 b0 = __builtin_bit_cast(bf16x8, w0), b1 = __builtin_bit_cast(bf16x8, w1), b2 = __builtin_bit_cast(bf16x8, w2);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    acc[c][r * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[c], b0, acc[c][r * 3 + 0], 0, 0, 0);
                    acc[c][r * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[c], b1, acc[c][r * 3 + 1], 0, 0, 0);
                    acc[c][r * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[c], b2, acc[c][r * 3 + 2], 0, 0, 0);
                }
            }
        }
    };

#ifdef DBUF
    if (t_begin < t_end) issue(t_begin, smem);
    int cur = 0;
    for (int tile = t_begin; tile < t_end; ++tile) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // tile landed; previous tile's reads are done
        if (tile + 1 < t_end) issue(tile + 1, smem + (cur ^ 1) * BUF16);
        compute(smem + cur * BUF16);
        cur ^= 1;
    }
#else
    for (int tile = t_begin; tile < t_end; ++tile) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        issue(tile, smem);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        compute(smem);
    }
#endif
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(smem);        // [it][18][64] f32x4 = 36 KB
    if (kh == 1) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 9; ++i) red[(it * 18 + c * 9 + i) * 64 + lane] = acc[c][i];
    }
    __syncthreads();
    if (kh == 1) return;
    const int ci = ci0 + it * 16 + j;
    if (ci >= p.Cin) return;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + c * 16 + kg * 4 + r;
            if (co >= p.Cout) continue;
            float* dst = p.partial + (((size_t)split * p.Cout + co) * p.Cin + ci) * 9;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) dst[tap] = acc[c][tap][r] + red[(it * 18 + c * 9 + tap) * 64 + lane][r];
        }
}

static float bf(float v) { return (float)(__bf16)v; }

int main(int argc, char** argv) {
    const int Cin = argc > 1 ? atoi(argv[1]) : 144, Cout = argc > 2 ? atoi(argv[2]) : 24;
    const int S = argc > 3 ? atoi(argv[3]) : 256, N = argc > 4 ? atoi(argv[4]) : 32;
    const int H = S, W = S, HW = H * W;
    if (Cin % 8 || Cout % 8 || W % TW || H % TH) { printf("shape\n"); return 1; }
    std::vector<float> hx((size_t)N * Cin * HW), hz((size_t)N * Cout * HW);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hz) v = rnd();
    float *dx, *dz; __bf16 *dx8, *dz8; float* dpart;
    CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dz, hz.size() * 4));
    CK(hipMalloc(&dx8, hx.size() * 2)); CK(hipMalloc(&dz8, hz.size() * 2));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dz, hz.data(), hz.size() * 4, hipMemcpyHostToDevice));
    to_c8_kernel<<<(unsigned)(((size_t)N * (Cin / 8) * HW + 255) / 256), 256>>>(dx, dx8, N, Cin, HW);
    to_c8_kernel<<<(unsigned)(((size_t)N * (Cout / 8) * HW + 255) / 256), 256>>>(dz, dz8, N, Cout, HW);
    P p{};
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.tiles_x = W / TW; p.tiles_y = H / TH;
    p.total_tiles = p.tiles_x * p.tiles_y * N;
    const int coblocks = (Cout + 31) / 32; p.ciblocks = (Cin + 31) / 32;
    const int pairs = coblocks * p.ciblocks;
    int ns = (256 * BPC) / pairs; if (ns > p.total_tiles) ns = p.total_tiles; if (ns < 1) ns = 1;
    p.tiles_per_split = (p.total_tiles + ns - 1) / ns;
    const int nsplit = (p.total_tiles + p.tiles_per_split - 1) / p.tiles_per_split;
    CK(hipMalloc(&dpart, (size_t)nsplit * Cout * Cin * 9 * 4));
    p.x8 = dx8; p.z8 = dz8; p.partial = dpart;
#ifdef DBUF
    const size_t lds = 2 * BUF16 * 2;
#else
    const size_t lds = 2 * 18 * 64 * sizeof(f32x4);
#endif
    CK(hipFuncSetAttribute((const void*)wgrad_c8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid(nsplit, pairs);
    wgrad_c8_kernel<<<grid, 256, lds>>>(p);
    CK(hipDeviceSynchronize());
    std::vector<float> hp((size_t)nsplit * Cout * Cin * 9);
    CK(hipMemcpy(hp.data(), dpart, hp.size() * 4, hipMemcpyDeviceToHost));
    // check a sample of (co, ci, tap) against fp64 on the rounded operands
    double maxerr = 0, maxref = 0;
    for (int t = 0; t < 24; ++t) {
        const int co = (t * 7 + 3) % Cout, ci = (t * 37 + 5) % Cin, tap = t % 9, dy = tap / 3 - 1, dxx = tap % 3 - 1;
        double ref = 0;
        for (int n = 0; n < N; ++n)
            for (int y = 0; y < H; ++y) {
                const int yy = y + dy; if (yy < 0 || yy >= H) continue;
                for (int x = 0; x < W; ++x) {
                    const int xx = x + dxx; if (xx < 0 || xx >= W) continue;
                    ref += (double)bf(hz[((size_t)n * Cout + co) * HW + y * W + x]) * (double)bf(hx[((size_t)n * Cin + ci) * HW + yy * W + xx]);
                }
            }
        double got = 0;
        for (int sp = 0; sp < nsplit; ++sp) got += hp[(((size_t)sp * Cout + co) * Cin + ci) * 9 + tap];
        maxerr = fmax(maxerr, fabs(got - ref)); maxref = fmax(maxref, fabs(ref));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) wgrad_c8_kernel<<<grid, 256, lds>>>(p);
    CK(hipEventRecord(e0));
    const int R = 20;
    for (int i = 0; i < R; ++i) wgrad_c8_kernel<<<grid, 256, lds>>>(p);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= R;
    const double gf = 2.0 * N * HW * Cin * Cout * 9 / 1e9, gb = ((double)N * (Cin + Cout) * HW * 2) / 1e9;
    printf("c8 wgrad %d->%d @%dx%d N=%d: %.3f ms  %.1f TF  %.2f TB/s  max|err| %.3e (max|ref| %.1f)  grid %dx%d lds %zu blocks/CU %d\n",
           Cin, Cout, H, W, N, ms, gf / ms, gb / ms, maxerr, maxref, nsplit, pairs, lds, BPC);
    return 0;
}
