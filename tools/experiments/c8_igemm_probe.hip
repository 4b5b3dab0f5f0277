// Experiment (not part of the library): what would the bf16 implicit-GEMM forward cost if its input were ALREADY
// stored as bf16 in the channel-blocked layout [C/8][H*W][8] (one 16-byte piece = 8 channels of one pixel)?
// Staging then is pure LDS-DMA (no staging VGPRs, no conversions): DESIGN.md section 7, item 1.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/c8_igemm_probe.hip -o /tmp/c8probe && /tmp/c8probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#ifdef OUT_C8
#define MMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, C, 0, 0, 0)      /* rows = channels */
#else
#define MMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(B, A, C, 0, 0, 0)      /* rows = pixels   */
#endif
constexpr int LPROW = 40, TH = 8, TW = 32, HR = TH + 2, HC = TW + 2, HP = HR * HC, HPP = 352;   // 340 halo pixels, padded

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
// weights image [mtile][chunk32][tap][16][40] bf16, value = w[co][ci][tap]
__global__ void pack_w_kernel(const float* __restrict__ w, __bf16* __restrict__ p, int Cin, int Cout, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int nch = (Cin + 31) / 32;
    const int kk = idx % LPROW; long long t = idx / LPROW;
    const int i = t % 16; t /= 16; const int tap = t % 9; t /= 9; const int cb = t % nch; const int mt = t / nch;
    const int r = mt * 16 + i, k = cb * 32 + kk;
    p[idx] = (__bf16)((kk < 32 && r < Cout && k < Cin) ? w[((size_t)r * Cin + k) * 9 + tap] : 0.f);
}
struct P { int N, H, W, Cin, Cout, tiles_x, tiles_y, ntiles, mtiles; const __bf16* x8; const __bf16* wp; float* out; __bf16* out8; };

template <int MT>
__global__ __launch_bounds__(256, 3) void conv_c8_kernel(const P p) {
    constexpr int XB = 4 * HPP * 8, WB = MT * 9 * 16 * LPROW;           // 16-bit elements
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    unsigned short* Xs = smem;                  // [4 groups][352 halo px][8 ch]
#ifdef PREFETCH
    unsigned short* Ws = smem + 2 * XB;
#else
    unsigned short* Ws = smem + XB;
#endif
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kg = lane >> 4, HW = p.H * p.W;
    const int mt0 = blockIdx.y * MT, nchunks = (p.Cin + 31) / 32;
    int bpix[4];
    for (int g = 0; g < 4; ++g) bpix[g] = (2 * wv + (g >> 1)) * HC + 16 * (g & 1) + j;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.wp), 0, (int)((size_t)p.mtiles * nchunks * 9 * 16 * LPROW * 2), 0x00020000);
    const int per = (p.ntiles + 7) >> 3, xcd = blockIdx.x & 7;
    const int tstep = gridDim.x >> 3, tend = min(p.ntiles, (xcd + 1) * per);
    for (int tile = xcd * per + (blockIdx.x >> 3); tile < tend; tile += tstep) {
        int t = tile; const int tx = t % p.tiles_x; t /= p.tiles_x; const int ty = t % p.tiles_y; t /= p.tiles_y;
        const int n = t, x0 = tx * TW, y0 = ty * TH;
        unsigned pixo[6];
        for (int q = 0; q < 6; ++q) {
            const int hp = lane + 64 * q, row = hp / HC, col = hp % HC, y = y0 + row - 1, x = x0 + col - 1;
            pixo[q] = (hp < HP && y >= 0 && y < p.H && x >= 0 && x < p.W) ? 16u * (unsigned)(y * p.W + x) : 0xfffffff0u;
        }
        f32x4 acc[MT][4];
        for (int m = 0; m < MT; ++m) for (int g = 0; g < 4; ++g) acc[m][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifndef PREFETCH
        for (int ch = 0; ch < nchunks; ++ch) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // X: wave w = channel group w of the chunk, 6 DMA instructions of 1 KB (64 halo pixels x 16 bytes)
            const int g8 = ch * 4 + wv;
            const __bf16* base = p.x8 + ((size_t)n * (p.Cin / 8) + (g8 < p.Cin / 8 ? g8 : 0)) * HW * 8;
            const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(base), 0, g8 < p.Cin / 8 ? HW * 16 : 0, 0x00020000);
#pragma unroll
            for (int q = 0; q < 6; ++q)
                if (lane + 64 * q < HPP)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr_t)(Xs + (wv * HPP + 64 * q) * 8), 16, pixo[q], 0, 0, 0);
            {
                constexpr int W16 = WB / 8, WI = (W16 + 63) / 64;
#pragma unroll
                for (int k = 0; k < (WI + 3) / 4; ++k) {
                    const int inst = wv + 4 * k;
                    if (inst < WI) {
                        const int idx = inst * 64 + lane, mt = idx / 720, r = idx % 720;
                        const bool ok = idx < W16 && (mt0 + mt) < p.mtiles;
                        const unsigned voff = ok ? (unsigned)((((mt0 + mt) * nchunks + ch) * 720 + r) * 16) : 0xfffffff0u;
                        if (idx < W16) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(Ws + inst * 512), 16, voff, 0, 0, 0);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int toff = (tap / 3) * HC + tap % 3;
                bf16x8 a[MT], b[4];
#pragma unroll
                for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const bf16x8*>(Ws + ((m * 9 + tap) * 16 + j) * LPROW + 8 * kg);
#pragma unroll
                for (int g = 0; g < 4; ++g) b[g] = *reinterpret_cast<const bf16x8*>(Xs + (kg * HPP + bpix[g] + toff) * 8);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[m][g] = MMA(a[m], b[g], acc[m][g]);
            }
        }
#else
        // X ring of 2 (the DMA of chunk c+1 flies under the MFMAs of chunk c), W single-buffered
        auto issue_x = [&](int ch, int slot) {
            const int g8 = ch * 4 + wv;
            const __bf16* base = p.x8 + ((size_t)n * (p.Cin / 8) + (g8 < p.Cin / 8 ? g8 : 0)) * HW * 8;
            const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(base), 0, g8 < p.Cin / 8 ? HW * 16 : 0, 0x00020000);
#pragma unroll
            for (int q = 0; q < 6; ++q)
                if (lane + 64 * q < HPP)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr_t)(Xs + slot * XB + (wv * HPP + 64 * q) * 8), 16, pixo[q], 0, 0, 0);
        };
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        issue_x(0, 0);
        for (int ch = 0; ch < nchunks; ++ch) {
            const int slot = ch & 1;
            {
                constexpr int W16 = WB / 8, WI = (W16 + 63) / 64;
#pragma unroll
                for (int k = 0; k < (WI + 3) / 4; ++k) {
                    const int inst = wv + 4 * k;
                    if (inst < WI) {
                        const int idx = inst * 64 + lane, mt = idx / 720, r = idx % 720;
                        const bool ok = idx < W16 && (mt0 + mt) < p.mtiles;
                        const unsigned voff = ok ? (unsigned)((((mt0 + mt) * nchunks + ch) * 720 + r) * 16) : 0xfffffff0u;
                        if (idx < W16) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(Ws + inst * 512), 16, voff, 0, 0, 0);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");        // X(ch) and W(ch) landed
            if (ch + 1 < nchunks) issue_x(ch + 1, slot ^ 1);                       // other slot: last read during chunk ch-1
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int toff = (tap / 3) * HC + tap % 3;
                bf16x8 a[MT], b[4];
#pragma unroll
                for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const bf16x8*>(Ws + ((m * 9 + tap) * 16 + j) * LPROW + 8 * kg);
#pragma unroll
                for (int g = 0; g < 4; ++g) b[g] = *reinterpret_cast<const bf16x8*>(Xs + slot * XB + (kg * HPP + bpix[g] + toff) * 8);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[m][g] = MMA(a[m], b[g], acc[m][g]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // everyone done with Ws / this X slot
        }
#endif
#ifndef OUT_C8
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int co = (mt0 + m) * 16 + j;
            if (co >= p.Cout) continue;
            float* cb = p.out + ((size_t)n * p.Cout + co) * HW;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int y = y0 + 2 * wv + (g >> 1), x = x0 + 16 * (g & 1) + 4 * kg;
                if (y < p.H && x < p.W) *reinterpret_cast<f32x4*>(cb + y * p.W + x) = acc[m][g];
            }
        }
    #else
        // bf16 channel-blocked OUTPUT: channels on the MFMA rows (see the mfma call), lane (j, kg) holds channels
        // 16m + 4kg .. +3 of pixel j: one 8-byte store = half of a 16-byte piece, the lane pair (kg, kg^1) completes it
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int c4 = (mt0 + m) * 16 + 4 * kg;
            if (c4 >= p.Cout) continue;
            __bf16* ob = p.out8 + ((size_t)n * (p.Cout / 8) + (c4 >> 3)) * HW * 8 + 4 * (kg & 1);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int y = y0 + 2 * wv + (g >> 1), x = x0 + 16 * (g & 1) + j;
                if (y < p.H && x < p.W) {
                    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                    bf16x4 v;
                    for (int r = 0; r < 4; ++r) v[r] = (__bf16)acc[m][g][r];
                    *reinterpret_cast<bf16x4*>(ob + (size_t)(y * p.W + x) * 8) = v;
                }
            }
        }
#endif
    }
}

int main(int argc, char** argv) {
    const int N = 32, H = 256, W = 256, Cin = argc > 1 ? atoi(argv[1]) : 144, Cout = argc > 2 ? atoi(argv[2]) : 24, HW = H * W;
    std::vector<float> hx((size_t)N * Cin * HW), hw((size_t)Cout * Cin * 9);
    srand(1);
    for (auto& v : hx) v = (rand() % 2001 - 1000) / 1000.f;
    for (auto& v : hw) v = (rand() % 2001 - 1000) / 5000.f;
    float *dx, *dw, *dout; __bf16 *dx8, *dwp;
    const int mtiles = (Cout + 15) / 16, nch = (Cin + 31) / 32;
    const long long wtotal = (long long)mtiles * nch * 9 * 16 * LPROW;
    CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&dout, (size_t)N * Cout * HW * 4));
    CK(hipMalloc(&dx8, hx.size() * 2)); CK(hipMalloc(&dwp, wtotal * 2));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    const long long tot8 = (long long)N * (Cin / 8) * HW;
    to_c8_kernel<<<(unsigned)((tot8 + 255) / 256), 256>>>(dx, dx8, N, Cin, HW);
    pack_w_kernel<<<(unsigned)((wtotal + 255) / 256), 256>>>(dw, dwp, Cin, Cout, wtotal);
    __bf16* dout8; CK(hipMalloc(&dout8, (size_t)N * ((Cout + 7) / 8) * 8 * HW * 2));
    P p{N, H, W, Cin, Cout, W / TW, H / TH, (W / TW) * (H / TH) * N, mtiles, dx8, dwp, dout, dout8};
    constexpr int MT = 2;
    const int mblocks = (mtiles + MT - 1) / MT;
#ifdef PREFETCH
    const size_t lds = (2 * 4 * HPP * 8 + MT * 9 * 16 * LPROW) * 2;
    const int resident = 512;
#else
    const size_t lds = (4 * HPP * 8 + MT * 9 * 16 * LPROW) * 2;
    const int resident = 768;
#endif
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_c8_kernel<MT>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    const int gx = (resident / mblocks) / 8 * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) conv_c8_kernel<MT><<<dim3(gx, mblocks), 256, lds>>>(p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) conv_c8_kernel<MT><<<dim3(gx, mblocks), 256, lds>>>(p);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    // verify a sample of outputs against a host reference on bf16-rounded operands
    std::vector<float> ho((size_t)N * Cout * HW);
#ifdef OUT_C8
    {
        std::vector<__bf16> h8((size_t)N * (Cout / 8) * HW * 8);
        CK(hipMemcpy(h8.data(), dout8, h8.size() * 2, hipMemcpyDeviceToHost));
        for (int n = 0; n < N; ++n) for (int c = 0; c < Cout; ++c) for (int px = 0; px < HW; ++px)
            ho[((size_t)n * Cout + c) * HW + px] = (float)h8[(((size_t)n * (Cout / 8) + c / 8) * HW + px) * 8 + c % 8];
    }
#else
    CK(hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost));
#endif
    auto rb = [](float v) { __bf16 h = (__bf16)v; return (float)h; };
    double maxerr = 0;
    for (int s = 0; s < 400; ++s) {
        const int n = rand() % N, co = rand() % Cout, y = (s % 7 == 0) ? 0 : rand() % H, x = (s % 5 == 0) ? W - 1 : rand() % W;
        double ref = 0;
        for (int ci = 0; ci < Cin; ++ci) for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
            ref += (double)rb(hw[((size_t)co * Cin + ci) * 9 + t]) * rb(hx[((size_t)n * Cin + ci) * HW + yy * W + xx]);
        }
        maxerr = fmax(maxerr, fabs(ref - ho[((size_t)n * Cout + co) * HW + y * W + x]));
    }
    #ifdef OUT_C8
    const int obytes = 2;
#else
    const int obytes = 4;
#endif
    const double gf = 2.0 * N * HW * Cin * Cout * 9 / 1e9, gb = ((double)N * Cin * HW * 2 + (double)N * Cout * HW * obytes) / 1e9;
    printf("c8 igemm %d->%d @%dx%d N=%d: %.3f ms  %.1f TF  %.2f TB/s (bf16 in, %d-byte out)  max|err| %.2e  grid %dx%d lds %zu\n", Cin, Cout, H, W, N, ms,
           gf / ms, gb / ms, obytes, maxerr, gx, mblocks, lds);
    return 0;
}
