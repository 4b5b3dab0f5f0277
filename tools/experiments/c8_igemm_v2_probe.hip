// Experiment (not part of the library): phase overlap in the channel-blocked 16-bit implicit GEMM (round 2).
// The production kernel runs  [DMA X,W] -> vmcnt(0) -> barrier -> [9 taps of reads + MFMA] -> [16-byte stores]  per (tile, chunk)
// with ONE LDS buffer: a block's load, MFMA and store phases never overlap, and because vmcnt counts stores too, the
// wait for the next tile's loads also drains the previous tile's stores.  Variants here (same LDS images, same MFMA order):
//   VAR 0  production structure (baseline, for A/B inside one binary)
//   VAR 1  the NEXT item's DMA is issued as soon as every wave has finished reading the buffer -- for the last chunk of a
//          tile that is BEFORE the epilogue, so the stores are younger than the loads and the wait at the top of the next
//          tile is a counted vmcnt(#stores): loads fly under the epilogue, stores drain under the next tile's MFMAs
//   VAR 2  VAR 1 + a second X slot: chunk c+1 (or the next tile's chunk 0) is in flight under the MFMAs of chunk c
// Geometry: TH x 32 pixel tiles, NW waves of 2 rows x 32 pixels (TH = 2 NW).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/c8_igemm_v2_probe.hip -o /tmp/c8v2 && /tmp/c8v2 144 24 256
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) f32x4 gf32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int WROW = 32;     // 64-byte weight rows, pieces XOR-swizzled (production layout)

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
// weights image [mtile][chunk32][tap][16 rows][4 pieces of 8], piece kq of row i at position kq ^ ((i >> 1) & 3)
__global__ void pack_w_kernel(const float* __restrict__ w, __bf16* __restrict__ p, int Cin, int Cout, long long total8) {
    const long long idx8 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx8 >= total8) return;
    const int nch = (Cin + 31) / 32;
    const int kq = idx8 % 4; long long t = idx8 / 4;
    const int i = t % 16; t /= 16; const int tap = t % 9; t /= 9; const int cb = t % nch; const int mt = t / nch;
    const int r = mt * 16 + i;
    bf16x8 v;
    for (int e = 0; e < 8; ++e) { const int k = cb * 32 + kq * 8 + e; v[e] = (__bf16)((r < Cout && k < Cin) ? w[((size_t)r * Cin + k) * 9 + tap] : 0.f); }
    const long long dst8 = idx8 - kq + (kq ^ ((i >> 1) & 3));
    *reinterpret_cast<bf16x8*>(p + dst8 * 8) = v;
}
struct P { int N, H, W, Cin, Cout, tiles_x, tiles_y, ntiles, mtiles; const __bf16* x8; const __bf16* wp; float* out; };

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most `n` (wave-uniform, 0..12) vector-memory operations of this wave are outstanding
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
    switch (n) {
        case 0: wait_vmcnt<0>(); break; case 1: wait_vmcnt<1>(); break; case 2: wait_vmcnt<2>(); break; case 3: wait_vmcnt<3>(); break;
        case 4: wait_vmcnt<4>(); break; case 5: wait_vmcnt<5>(); break; case 6: wait_vmcnt<6>(); break; case 7: wait_vmcnt<7>(); break;
        case 8: wait_vmcnt<8>(); break; case 9: wait_vmcnt<9>(); break; case 10: wait_vmcnt<10>(); break; case 11: wait_vmcnt<11>(); break;
        default: wait_vmcnt<12>(); break;
    }
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ABL (timing ablations, results wrong): 1 = halo pixels are not loaded (centre only), 2 = no stores, 4 = no X loads, 8 = no MFMAs,
// 16 = W loaded once per block (not per chunk), 32 = every tile reads the pixels of tiles 0..7 (X served by L2)
// -DASM_DMA (round 3): issue the LDS-DMA from inline assembly -- behind the builtin hipcc may put `s_waitcnt vmcnt(0)` in front of the next
// LDS read (it knows the builtin stores to LDS and cannot tell which part), which turns VAR 1 / VAR 2's prefetch into a plain load.
#ifdef ASM_DMA
typedef int i32x4_ __attribute__((ext_vector_type(4)));
typedef i32x4_ rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, int bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    return (i32x4_){(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), bytes, 0x00020000};
}
__device__ __forceinline__ void dma16_asm(rsrc_t rs, const void* lds, unsigned voff) {
    const unsigned la = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)lds;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(la), "v"(voff), "s"(rs) : "memory", "m0");
}
#define DMA16(rsrc, ldsp, off) dma16_asm(rsrc, ldsp, off)
#else
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, int bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000); }
#define DMA16(rsrc, ldsp, off) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(ldsp), 16, off, 0, 0, 0)
#endif
template <int MT, int NW, int VAR, int ABL = 0>
__global__ __launch_bounds__(64 * NW, (NW == 4 ? 3 : 4)) void conv_c8_v(const P p) {
    constexpr int TH = 2 * NW, TW = 32, HR = TH + 2, HC = TW + 2, HP = HR * HC, HPP = (HP + 15) / 16 * 16;
    constexpr int XB = 4 * HPP * 8, WB = MT * 9 * 16 * WROW;           // 16-bit elements
    constexpr int NSLOT = VAR == 2 ? 2 : 1;
    constexpr int XI = (HPP + 63) / 64;                                 // DMA instructions per channel group
    constexpr int XPW = NW == 4 ? XI : (XI + 1) / 2;                    // ... per wave (8 waves: half a group each)
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    unsigned short* Xs = smem;                  // [slot][4 groups][HPP halo px][8 ch]
    unsigned short* Ws = smem + NSLOT * XB;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kg = lane >> 4, HW = p.H * p.W;
    const int mt0 = blockIdx.y * MT, nchunks = (p.Cin + 31) / 32;
    int bpix[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bpix[g] = (2 * wv + (g >> 1)) * HC + 16 * (g & 1) + j;
    const rsrc_t wrsrc = make_rsrc(p.wp, (int)((size_t)p.mtiles * nchunks * 9 * 16 * WROW * 2));
    const int per = (p.ntiles + 7) >> 3, xcd = blockIdx.x & 7;
    const int tstep = gridDim.x >> 3, tend = min(p.ntiles, (xcd + 1) * per);
    const int grp_w = NW == 4 ? wv : (wv >> 1), half = NW == 4 ? 0 : (wv & 1);      // which channel group / half of it this wave brings

    auto issue_x = [&](int tile, int ch, int slot) {
        int t = (ABL & 32) ? (tile & 7) : tile; const int tx = t % p.tiles_x; t /= p.tiles_x; const int ty = t % p.tiles_y; t /= p.tiles_y;
        const int n = t, x0 = tx * TW, y0 = ty * TH;
        const int g8 = ch * 4 + grp_w;
        const __bf16* base = p.x8 + ((size_t)n * (p.Cin / 8) + (g8 < p.Cin / 8 ? g8 : 0)) * HW * 8;
        const rsrc_t xrsrc = make_rsrc(base, g8 < p.Cin / 8 ? HW * 16 : 0);
#pragma unroll
        for (int q = 0; q < XPW; ++q) {
            const int qi = half * XPW + q;
            const int hp = lane + 64 * qi, row = hp / HC, col = hp % HC, y = y0 + row - 1, x = x0 + col - 1;
            bool okx = hp < HP && y >= 0 && y < p.H && x >= 0 && x < p.W;
            if (ABL & 1) okx = okx && row >= 1 && row <= TH && col >= 1 && col <= TW;
            if (ABL & 4) okx = false;
            const unsigned off = okx ? 16u * (unsigned)(y * p.W + x) : 0xfffffff0u;
            if (qi < XI && hp < HPP)
                DMA16(xrsrc, Xs + slot * XB + (grp_w * HPP + 64 * qi) * 8, off);
        }
    };
    auto issue_w = [&](int ch) {
        constexpr int W16 = WB / 8, WI = (W16 + 63) / 64;
#pragma unroll
        for (int k = 0; k < (WI + NW - 1) / NW; ++k) {
            const int inst = wv + NW * k;
            if (inst < WI) {
                const int idx = inst * 64 + lane, mt = idx / 576, r = idx % 576;
                const bool ok = idx < W16 && (mt0 + mt) < p.mtiles;
                const unsigned voff = ok ? (unsigned)((((mt0 + mt) * nchunks + ch) * 576 + r) * 16) : 0xfffffff0u;
                if (idx < W16) DMA16(wrsrc, Ws + inst * 512, voff);
            }
        }
    };
    auto compute = [&](f32x4 (&acc)[MT][4], int slot) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = (tap / 3) * HC + tap % 3;
            bf16x8 a[MT], b[4];
#pragma unroll
            for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const bf16x8*>(Ws + ((m * 9 + tap) * 16 + j) * WROW + 8 * (kg ^ ((j >> 1) & 3)));
#pragma unroll
            for (int g = 0; g < 4; ++g) b[g] = *reinterpret_cast<const bf16x8*>(Xs + slot * XB + (kg * HPP + bpix[g] + toff) * 8);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (ABL & 8) { acc[m][g][0] += (float)b[g][0] * (float)a[m][0]; }
                    else acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[g], a[m], acc[m][g], 0, 0, 0);
                }
        }
    };
    // epilogue: fp32 planar output, one 16-byte store per accumulator tile; returns the number of store INSTRUCTIONS that had an
    // active lane (a lower bound of the vector-memory operations issued: the counted wait of the next tile relies on it)
    auto epilogue = [&](const f32x4 (&acc)[MT][4], int tile) -> int {
        int t = tile; const int tx = t % p.tiles_x; t /= p.tiles_x; const int ty = t % p.tiles_y; t /= p.tiles_y;
        const int n = t, x0 = tx * TW, y0 = ty * TH;
        int nst = 0;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int co = (mt0 + m) * 16 + j;
            gf32x4* cb = (gf32x4*)(p.out + ((size_t)n * p.Cout + co) * HW);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int y = y0 + 2 * wv + (g >> 1), x = x0 + 16 * (g & 1) + 4 * kg;
                bool ok = co < p.Cout && y < p.H && x < p.W;
                if (ABL & 2) ok = ok && acc[m][g][0] == 12345.678f;
                nst += __ballot(ok) != 0ull ? 1 : 0;
                if (ok) cb[(y * p.W + x) >> 2] = acc[m][g];
            }
        }
        return nst;
    };

    int tile = xcd * per + (blockIdx.x >> 3);
    if (VAR == 0) {
        int w_have = -1;
        for (; tile < tend; tile += tstep) {
            f32x4 acc[MT][4];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[m][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int ch = 0; ch < nchunks; ++ch) {
                lds_barrier();
                issue_x(tile, ch, 0);
                if (w_have != ch && !((ABL & 16) && w_have >= 0)) { issue_w(ch); w_have = ch; }
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                compute(acc, 0);
            }
            epilogue(acc, tile);
        }
    } else {
        // flattened (tile, chunk) items; the DMA of item i+1 is issued when the buffer it lands in is free:
        //   VAR 1: after the barrier that ends item i's reads (same slot);  VAR 2: before item i's MFMAs (other slot)
        if (tile >= tend) return;
        int pend_stores = 0;              // stores issued AFTER the DMA that is waited for next (0 = nothing younger than it)
        bool younger_stores = false;
        issue_x(tile, 0, 0);
        issue_w(0);
        int slot = 0;
        for (; tile < tend; tile += tstep) {
            f32x4 acc[MT][4];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[m][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int ch = 0; ch < nchunks; ++ch) {
                const bool last = ch + 1 == nchunks;
                const int ntile = tile + tstep;
                const bool more = !last || ntile < tend;
                if (younger_stores) wait_vmcnt_dyn(pend_stores); else wait_vmcnt<0>();      // this item's X (and W) landed
                younger_stores = false;
                asm volatile("s_barrier" ::: "memory");
                if (VAR == 2 && more) {                              // other slot: its last reader was item i-1 (barrier passed)
                    if (last) issue_x(ntile, 0, slot ^ 1); else issue_x(tile, ch + 1, slot ^ 1);
                }
                compute(acc, slot);
                lds_barrier();                                       // every wave has its fragments: X slot and W are free
                if (more) {
                    if (VAR == 1) { if (last) issue_x(ntile, 0, 0); else issue_x(tile, ch + 1, 0); }
                    if (nchunks > 1) issue_w(last ? 0 : ch + 1);
                }
                if (VAR == 2) slot ^= 1;
            }
            asm volatile("" ::: "memory");
            pend_stores = epilogue(acc, tile);                       // younger than the DMA above
            younger_stores = true;
            asm volatile("" ::: "memory");
        }
    }
}

template <int MT, int NW, int VAR, int ABL = 0>
static float run(const P& p0, int reps) {
    P p = p0;
    constexpr int TH = 2 * NW, HP = (TH + 2) * 34, HPP = (HP + 15) / 16 * 16;
    p.tiles_x = p.W / 32; p.tiles_y = (p.H + TH - 1) / TH; p.ntiles = p.tiles_x * p.tiles_y * p.N;
    const size_t lds = ((size_t)(VAR == 2 ? 2 : 1) * 4 * HPP * 8 + MT * 9 * 16 * WROW) * 2;
    const int per_cu = (int)(160 * 1024 / lds) > (NW == 4 ? 3 : 2) ? (NW == 4 ? 3 : 2) : (int)(160 * 1024 / lds);
    const int mblocks = (p.mtiles + MT - 1) / MT;
    int gx = (256 * per_cu / mblocks) / 8 * 8;
    if (gx > (p.ntiles + 7) / 8 * 8) gx = (p.ntiles + 7) / 8 * 8;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_c8_v<MT, NW, VAR, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) conv_c8_v<MT, NW, VAR, ABL><<<dim3(gx, mblocks), 64 * NW, lds>>>(p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) conv_c8_v<MT, NW, VAR, ABL><<<dim3(gx, mblocks), 64 * NW, lds>>>(p);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("    [MT%d NW%d VAR%d ABL%d grid %dx%d lds %zu (%d/CU)]", MT, NW, VAR, ABL, gx, mblocks, lds, per_cu);
    return ms / reps;
}

int main(int argc, char** argv) {
    const int Cin = argc > 1 ? atoi(argv[1]) : 144, Cout = argc > 2 ? atoi(argv[2]) : 24, H = argc > 3 ? atoi(argv[3]) : 256, W = H;
    const int N = argc > 4 ? atoi(argv[4]) : 32, HW = H * W;
    std::vector<float> hx((size_t)N * Cin * HW), hw((size_t)Cout * Cin * 9);
    srand(1);
    for (auto& v : hx) v = (rand() % 2001 - 1000) / 1000.f;
    for (auto& v : hw) v = (rand() % 2001 - 1000) / 5000.f;
    float *dx, *dw, *dout; __bf16 *dx8, *dwp;
    const int mtiles = (Cout + 15) / 16, nch = (Cin + 31) / 32;
    const long long wtotal8 = (long long)mtiles * nch * 9 * 16 * 4;
    CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&dout, (size_t)N * Cout * HW * 4));
    CK(hipMalloc(&dx8, hx.size() * 2)); CK(hipMalloc(&dwp, wtotal8 * 16));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    const long long tot8 = (long long)N * (Cin / 8) * HW;
    to_c8_kernel<<<(unsigned)((tot8 + 255) / 256), 256>>>(dx, dx8, N, Cin, HW);
    pack_w_kernel<<<(unsigned)((wtotal8 + 255) / 256), 256>>>(dw, dwp, Cin, Cout, wtotal8);
    CK(hipDeviceSynchronize());
    P p{N, H, W, Cin, Cout, 0, 0, 0, mtiles, dx8, dwp, dout};
    auto rb = [](float v) { __bf16 h = (__bf16)v; return (float)h; };
    std::vector<float> ho((size_t)N * Cout * HW);
    auto verify = [&]() {
        CK(hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemset(dout, 0xff, ho.size() * 4));
        double maxerr = 0;
        srand(7);
        for (int s = 0; s < 600; ++s) {
            const int n = rand() % N, co = rand() % Cout, y = (s % 7 == 0) ? 0 : (s % 11 == 0 ? H - 1 : rand() % H), x = (s % 5 == 0) ? W - 1 : rand() % W;
            double ref = 0;
            for (int ci = 0; ci < Cin; ++ci) for (int t = 0; t < 9; ++t) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                ref += (double)rb(hw[((size_t)co * Cin + ci) * 9 + t]) * rb(hx[((size_t)n * Cin + ci) * HW + yy * W + xx]);
            }
            const double e = fabs(ref - ho[((size_t)n * Cout + co) * HW + y * W + x]);
            maxerr = e == e ? fmax(maxerr, e) : 1e30;
        }
        return maxerr;
    };
    const double gf = 2.0 * N * HW * Cin * Cout * 9 / 1e9, gb = ((double)N * Cin * HW * 2 + (double)N * Cout * HW * 4) / 1e9;
    printf("c8 igemm %d->%d @%dx%d N=%d  (%.1f GFLOP, %.1f MB algorithmic)\n", Cin, Cout, H, W, N, gf, gb * 1e3);
#define RUN(MT_, NW_, VAR_) do { const float ms = run<MT_, NW_, VAR_>(p, 20); const double err = verify(); \
        printf("  %.4f ms  %.1f TF  %.2f TB/s  max|err| %.2e\n", ms, gf / ms, gb / ms, err); } while (0)
#define RUNA(MT_, NW_, VAR_, ABL_) do { const float ms = run<MT_, NW_, VAR_, ABL_>(p, 20); CK(hipMemset(dout, 0, 16)); \
        printf("  %.4f ms  %.2f TB/s (algorithmic bytes)\n", ms, gb / ms); } while (0)
    if (argc > 5) {          // ablations of the production structure
        RUN(2, 4, 0); RUNA(2, 4, 0, 1); RUNA(2, 4, 0, 2); RUNA(2, 4, 0, 4); RUNA(2, 4, 0, 8); RUNA(2, 4, 0, 3); RUNA(2, 4, 0, 6); RUNA(2, 4, 0, 10); RUNA(2, 4, 0, 12);
        RUNA(2, 4, 0, 16); RUNA(2, 4, 0, 32); RUNA(2, 4, 0, 48); RUNA(2, 4, 0, 18); RUNA(2, 4, 0, 50);
        RUN(2, 8, 0); RUNA(2, 8, 0, 16); RUNA(2, 8, 0, 18);
        return 0;
    }
    if (Cout <= 16) { RUN(1, 4, 0); RUN(1, 4, 1); RUN(1, 4, 2); RUN(1, 8, 0); RUN(1, 8, 1); RUN(1, 8, 2); }
    else { RUN(2, 4, 0); RUN(2, 4, 1); RUN(2, 4, 2); RUN(2, 8, 0); RUN(2, 8, 1); RUN(2, 8, 2); }
    return 0;
}
