// conv3x3 (stride 1, pad 1) forward / dgrad / wgrad for gfx950.
//
// Hot path: implicit GEMM on the fp32 MFMA v_mfma_f32_16x16x4_f32 (exact fp32 == fmaf chain).
//   fwd  : Z[co][pix] = sum_{ci,tap} Wp[co][ci,tap] * X[ci][pix+tap]   M=Cout, N=pixels, K=9*Cin
//   dgrad: the same kernel on dZ with the flipped/transposed packed image
//   wgrad: dW[co][ci][tap] = sum_pix dZ[co][pix] * X[ci][pix+tap]      M=Cout, N=Cin, K=pixels (split-K)
// NCHW planes are staged through LDS as halo tiles with 16-byte coalesced global reads; the
// packed weight image is already the LDS image ([mtile][ci][tap][16]) so its staging is a copy.
// Replaces nn.Conv2d(k=3,padding=1) of MTnnUNet.py:12-16 and MONAI Convolution (MTUNetPlusPlus.py:47-81).
#include "common.h"
#include <type_traits>
#include <utility>
#include <cstdlib>

namespace {

// ------------------------------------------------------------------ packing
__device__ __forceinline__ void pack_fwd_elem(const float* __restrict__ w, float* __restrict__ p, int Cin, int Cout, int idx) {
    int i = idx & 15, tap = (idx >> 4) % 9, ci = (idx / 144) % Cin, mt = idx / (144 * Cin);
    int co = mt * 16 + i;
    p[idx] = co < Cout ? w[((size_t)co * Cin + ci) * 9 + tap] : 0.f;
}
// dgrad image: rows = input channels of the forward conv, K = (co, flipped tap)
__device__ __forceinline__ void pack_dgrad_elem(const float* __restrict__ w, float* __restrict__ p, int Cin, int Cout, int idx) {
    int i = idx & 15, tap = (idx >> 4) % 9, co = (idx / 144) % Cout, mt = idx / (144 * Cout);
    int ci = mt * 16 + i;
    p[idx] = ci < Cin ? w[((size_t)co * Cin + ci) * 9 + (8 - tap)] : 0.f;
}
__global__ void pack_fwd_kernel(const float* __restrict__ w, float* __restrict__ p, int Cin, int Cout, int total) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < total) pack_fwd_elem(w, p, Cin, Cout, idx);
}
__global__ void pack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ p, int Cin, int Cout, int total) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < total) pack_dgrad_elem(w, p, Cin, Cout, idx);
}

// ------------------------------------------------------------------ geometry
// 256 output pixels per block = 16 groups of 16 pixels; wave w owns groups 4w..4w+3.
template <int GEO> struct Geo;
template <> struct Geo<0> {   // wide maps: 8 rows x 32 cols
    static constexpr int TH = 8, TW = 32, ROWS = 10, LW = 40, IMG = 1, IMGS = 400, PS = 400;
};
template <> struct Geo<1> {   // 16-wide maps: 16 rows x 16 cols
    static constexpr int TH = 16, TW = 16, ROWS = 18, LW = 24, IMG = 1, IMGS = 432, PS = 432;
};
template <> struct Geo<2> {   // 8x8 maps: 4 images per block
    static constexpr int TH = 8, TW = 8, ROWS = 10, LW = 16, IMG = 4, IMGS = 160, PS = 648;
};

struct ConvP {
    int N, H, W, Cin, Cout;     // Cout = rows of this GEMM (dgrad: the forward conv's Cin)
    SegTable in, out;
    const float* wp;
    const float* bias;
    int tiles_x, tiles_y, ntiles;
    int mtiles;
    int dbg;      // timing probes only (env MTBC_DBG): 1 = no global loads, 2 = no epilogue, 4 = no LDS stores
    float* stats;               // channel-blocked 16-bit output only, or nullptr: [N][slots][Cout][2] per-wave {sum, sum of squares} of the stored values
    // O8 == 2 (gathered dgrad that prepares the InstanceNorm backward of its output tensor): stats = {sum g, sum g * xhat}
    const float* extra;         // fp32 planar partial gradient added before the rounding, or nullptr
    const unsigned short* nz;   // the tensor's conv output z, channel-blocked like the output
    const float* nmean; const float* nrstd; const float* ngamma; const float* nbeta;
    float nslope;
#ifdef MTBC_PROBES
    unsigned long long* ts;     // phase timestamps of the ring kernel (MTBC_RING_TS=1): [block][16] ticks of the 100 MHz clock
#endif
};
#ifdef MTBC_PROBES
#define MTBC_TS(p, k) do { if ((p).ts && threadIdx.x == 0) (p).ts[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (k)] = wall_clock64(); } while (0)
#else
#define MTBC_TS(p, k) do { } while (0)
#endif

constexpr int KC = 8;           // input channels per LDS chunk

// Segment tables copied from the kernel arguments into LDS once per block: looking a channel up from LDS keeps
// ~66 SGPRs (two by-value tables) from staying live across the MFMA loop, where they spilled.
struct SegL { float* ptr; long long bs; int cb; int acc; };
constexpr int SEGL_FLOATS = 2 * MTBC_MAX_SEGS * (int)(sizeof(SegL) / sizeof(float));
__device__ __forceinline__ void segl_fill(SegL* dst, const SegTable& t) {
    // thread k writes entry k, one exec-masked store per entry with COMPILE-TIME indices into the by-value table.  (Round 4: written as "e = entry 0; if (i == k) e =
    // entry k; dst[i] = e", hipcc recognised a dynamic index, copied the whole kernel-argument table to SCRATCH and loaded entry i back: 36 scratch stores, 3 scratch
    // loads and two vmcnt waits in the prologue of every block of every igemm kernel -- a memory round trip before the first DMA could be issued.)
#pragma unroll
    for (int k = 0; k < MTBC_MAX_SEGS; ++k)
        if (threadIdx.x == k) {
            SegL e;
            // entries past t.n get cb = INT_MAX so that the scan never selects them
            e.ptr = t.ptr[k]; e.bs = t.bstride[k]; e.cb = (k == 0) ? 0 : (k < t.n ? t.cbegin[k] : 0x7fffffff); e.acc = t.accumulate[k];
            dst[k] = e;
        }
}
__device__ __forceinline__ SegL segl_ref(const SegL* t, int c) {
    SegL r = t[0];
#pragma unroll
    for (int i = 1; i < MTBC_MAX_SEGS; ++i) {
        const SegL e = t[i];
        if (c >= e.cb) r = e;
    }
    return r;
}
// ... knowing the table's length n (uniform): a one-segment table -- every conv output, every single-input conv -- is ONE LDS read instead
// of six reads and a select chain of ~60 instructions, per chunk and per channel tile of every pixel tile
__device__ __forceinline__ SegL segl_ref_n(const SegL* t, int c, int n) {
    SegL r = t[0];
    for (int i = 1; i < n; ++i) {
        const SegL e = t[i];
        if (c >= e.cb) r = e;
    }
    return r;
}

// Persistent: block (bx, by) walks pixel tiles bx, bx+gridDim.x, ... for its channel block by.  The work list is the
// flattened sequence of (tile, chunk) items; item i+1 is prefetched into registers while item i feeds the MFMAs, so a
// tile's prologue and epilogue overlap the neighbouring tiles' matrix work (one barrier per item).
template <int MT, int GEO>
__global__ __launch_bounds__(256, MT >= 3 ? 2 : 3) void conv3x3_igemm_kernel(const ConvP p) {
    using G = Geo<GEO>;
    constexpr int XS = KC * G::PS;                 // floats
    constexpr int WS = MT * KC * 144;
    constexpr int BUF = XS + WS;
    constexpr int XF4_PER_CH = G::IMG * G::ROWS * G::LW / 4;
    constexpr int XF4 = KC * XF4_PER_CH;
    constexpr int XSLOTS = (XF4 + 255) / 256;
    constexpr int WF4 = MT * KC * 36;
    constexpr int WSLOTS = (WF4 + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    SegL* seg_in = reinterpret_cast<SegL*>(smem + 2 * BUF);
    SegL* seg_out = seg_in + MTBC_MAX_SEGS;
    segl_fill(seg_in, p.in);
    segl_fill(seg_out, p.out);
    __syncthreads();

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int HW = p.H * p.W;
    const int mt0 = blockIdx.y * MT;
    const int nchunks = p.Cin / KC;
    const int my_tiles = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_tiles * nchunks;
    if (total <= 0) return;

    // ---- staging: per tile, each slot keeps ONE 32-bit element offset (-1 = outside the image -> zero fill);
    //      per chunk only a wave-uniform base pointer changes, so the prefetch costs ~3 VALU per 16-byte load
    int pn0 = 0;
    int x_off[XSLOTS], w_off[WSLOTS];
#pragma unroll
    for (int s = 0; s < WSLOTS; ++s) {
        const int idx = tid + s * 256;
        const int mt = idx / (KC * 36), r = idx % (KC * 36);
        w_off[s] = (idx < WF4 && (mt0 + mt) < p.mtiles) ? ((mt0 + mt) * p.Cin) * 144 + r * 4 : -1;
    }
    auto set_tile = [&](int tile) {
        int t = tile;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y; t /= p.tiles_y;
        pn0 = t * G::IMG;
        const int px0 = tx * G::TW, py0 = ty * G::TH;
#pragma unroll
        for (int s = 0; s < XSLOTS; ++s) {
            const int idx = tid + s * 256;
            const int c = idx / XF4_PER_CH;
            int rem = idx % XF4_PER_CH;
            const int img = rem / (G::ROWS * G::LW / 4); rem %= (G::ROWS * G::LW / 4);
            const int row = rem / (G::LW / 4), c4 = rem % (G::LW / 4);
            const int y = py0 + row - 1, x = px0 - 4 + c4 * 4;
            const bool ok = idx < XF4 && (pn0 + img) < p.N && y >= 0 && y < p.H && x >= 0 && x < p.W;
            // image index folded in per chunk (batch stride differs per segment): keep img in the low bits
            x_off[s] = ok ? ((c * HW + y * p.W + x) << 2) | img : -1;
        }
    };
    float4 xr[XSLOTS], wr[WSLOTS];
#pragma unroll
    for (int s = 0; s < XSLOTS; ++s) xr[s] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int s = 0; s < WSLOTS; ++s) wr[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_chunk = [&](int chunk) {
        const int ci0 = chunk * KC;
        const SegL sr = segl_ref(seg_in, ci0);
        const float* base = sr.ptr + (size_t)(ci0 - sr.cb) * HW + (size_t)pn0 * sr.bs;
        const long long bs = sr.bs;
#pragma unroll
        for (int s = 0; s < XSLOTS; ++s) {
            const bool ok = x_off[s] >= 0;
            const size_t off = ok ? (size_t)(x_off[s] >> 2) + (G::IMG > 1 ? (size_t)(x_off[s] & 3) * bs : 0) : 0;
            const float4 v = *reinterpret_cast<const float4*>(base + off);
            xr[s] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float* wbase = p.wp + (size_t)ci0 * 144;
#pragma unroll
        for (int s = 0; s < WSLOTS; ++s) {
            const bool ok = w_off[s] >= 0;
            const float4 v = *reinterpret_cast<const float4*>(wbase + (ok ? w_off[s] : 0));
            wr[s] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_chunk = [&](int boff) {       // LDS is addressed as smem[offset]: no generic pointers in the loop
#pragma unroll
        for (int s = 0; s < XSLOTS; ++s) {
            const int idx = tid + s * 256;
            if (idx < XF4) {
                const int c = idx / XF4_PER_CH;
                int rem = idx % XF4_PER_CH;
                const int img = rem / (G::ROWS * G::LW / 4); rem %= (G::ROWS * G::LW / 4);
                const int row = rem / (G::LW / 4), c4 = rem % (G::LW / 4);
                *reinterpret_cast<float4*>(&smem[boff + c * G::PS + img * G::IMGS + row * G::LW + c4 * 4]) = xr[s];
            }
        }
#pragma unroll
        for (int s = 0; s < WSLOTS; ++s) {
            const int idx = tid + s * 256;
            if (idx < WF4) *reinterpret_cast<float4*>(&smem[boff + XS + idx * 4]) = wr[s];
        }
    };

    // ---- per-lane fragment bases
    const int j = lane & 15, kk = lane >> 4;
    int laneB, gbase;
    if (GEO == 0) { laneB = kk * G::PS + j; gbase = (2 * wv) * G::LW; }
    else if (GEO == 1) { laneB = kk * G::PS + j; gbase = (4 * wv) * G::LW; }
    else { laneB = kk * G::PS + (j >> 3) * G::LW + (j & 7); gbase = wv * G::IMGS; }
    const int bBase = laneB + gbase + 3;
    const int aBase = XS + kk * 144 + j;

    f32x4 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[m][g] = (f32x4){0.f, 0.f, 0.f, 0.f};

    set_tile(blockIdx.x);
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    int chunk = 0, tile = blockIdx.x;
    for (int it = 0; it < total; ++it) {
        const int cur = (it & 1) * BUF;
        const bool more = it + 1 < total;
        const bool last = chunk + 1 == nchunks;
        // 6 groups of 3 taps (fixed cs, kernel row r); the fragments of group q+1 are read from LDS before the MFMAs
        // of group q issue.  sched_barriers keep hipcc from hoisting every read of the chunk to the top (~90 VGPRs).
        float fa[2][3][MT], fb[2][3][4];
        auto read_group = [&](int q, int slot) {
            const int cs = q / 3, r = q % 3;
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
                const int tap = r * 3 + s3;
#pragma unroll
                for (int m = 0; m < MT; ++m) fa[slot][s3][m] = smem[cur + aBase + m * (KC * 144) + cs * 576 + tap * 16];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    int toff;
                    if (GEO == 0) toff = (g >> 1) * G::LW + 16 * (g & 1);
                    else if (GEO == 1) toff = g * G::LW;
                    else toff = 2 * g * G::LW;
                    fb[slot][s3][g] = smem[cur + bBase + toff + cs * 4 * G::PS + r * G::LW + s3];
                }
            }
        };
        read_group(0, 0);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            if (q + 1 < 6) read_group(q + 1, (q + 1) & 1);
            if (q == 0 && more) {
                // next item's global loads: issued here so that their address arithmetic runs in the shadow of the
                // MFMAs below (24 of every 32 MFMA cycles leave the vector issue port free) instead of ahead of them
                if (last) set_tile(tile + gridDim.x);
                if (!MTBC_DBG_BIT(p, 1)) load_chunk(last ? 0 : chunk + 1);
            } else {
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[q & 1][s3][m], fb[q & 1][s3][g], acc[m][g], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more && !MTBC_DBG_BIT(p, 4)) store_chunk(((it + 1) & 1) * BUF);     // prefetch registers die here, before the epilogue
        if (last) {
            // ---- epilogue of `tile`: D row = (lane>>4)*4 + reg (channel), col = lane&15 (pixel)
            int t = tile;
            const int tx = t % p.tiles_x; t /= p.tiles_x;
            const int ty = t % p.tiles_y; t /= p.tiles_y;
            const int n0 = t * G::IMG, x0 = tx * G::TW, y0 = ty * G::TH;
            int poff[4];
            const int n = GEO == 2 ? n0 + wv : n0;
            bool all_px = n < p.N;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int y, x;
                if (GEO == 0) { y = y0 + 2 * wv + (g >> 1); x = x0 + 16 * (g & 1) + j; }
                else if (GEO == 1) { y = y0 + 4 * wv + g; x = x0 + j; }
                else { y = y0 + 2 * g + (j >> 3); x = x0 + (j & 7); }
                const bool ok = n < p.N && y < p.H && x < p.W && !MTBC_DBG_BIT(p, 2);
                all_px = all_px && ok;
                poff[g] = ok ? y * p.W + x : -1;
            }
            // wave-uniform fast path: every pixel and every channel row of this wave's fragment is in range
            const bool fast = __all(all_px) && (mt0 + MT) * 16 <= p.Cout;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                // the 4 rows a lane holds (channels co4 .. co4+3) sit in one segment: segment sizes are multiples of 8
                // on this path (host check), so one table lookup serves all four
                const int co4 = (mt0 + m) * 16 + kk * 4;
                const SegL so = segl_ref(seg_out, co4 < p.Cout ? co4 : 0);
                gfloat* cb0 = (gfloat*)so.ptr + (size_t)n * so.bs + (size_t)((co4 < p.Cout ? co4 : 0) - so.cb) * HW;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co4 + r;
                    const bool row_ok = fast || co < p.Cout;
                    gfloat* cb = cb0 + (size_t)r * HW;
                    const float bv = (p.bias && row_ok) ? p.bias[co] : 0.f;
                    if (fast) {
                        float old[4];
#pragma unroll
                        for (int g = 0; g < 4; ++g) old[g] = so.acc ? cb[poff[g]] : 0.f;
#pragma unroll
                        for (int g = 0; g < 4; ++g) cb[poff[g]] = acc[m][g][r] + bv + old[g];
                    } else if (row_ok) {
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            if (poff[g] >= 0) cb[poff[g]] = acc[m][g][r] + bv + (so.acc ? cb[poff[g]] : 0.f);
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[m][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            tile += gridDim.x;
            chunk = 0;
        } else {
            ++chunk;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ igemm with LDS-DMA staging (wide / 16-wide maps)
// Same tiling and fragment maps as conv3x3_igemm_kernel, but the halo tile and the packed weights go HBM -> LDS
// directly (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction, out-of-image lanes zero-filled by the buffer
// bounds check), into a 3-slot LDS ring: the DMA for item i+2 is issued while item i feeds the MFMAs, a counted
// s_waitcnt vmcnt leaves it in flight across the raw s_barrier that publishes item i+1.  No staging VGPRs, no ds_write.
template <int MT, int GEO> struct DmaCount {
    using G = Geo<GEO>;
    static constexpr int XF4 = KC * G::IMG * G::ROWS * G::LW / 4;
    static constexpr int WF4 = MT * KC * 36;
    static constexpr int XI = (XF4 + 63) / 64, WI = (WF4 + 63) / 64;     // wave-instructions per item
    // wave w issues the instructions k = w, w+4, ... of each stream
    static constexpr int of(int w) { return (XI - w + 3) / 4 + (WI - w + 3) / 4; }
};
// workgroup barrier that publishes LDS only: unlike __syncthreads() it does not drain vmcnt, so global loads issued
// before it (register prefetch, LDS-DMA) stay in flight across it
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <bool B> struct BoolC { static constexpr bool value = B; };
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int MT, int GEO> __device__ __forceinline__ void wait_newest_in_flight(int wv) {
    using D = DmaCount<MT, GEO>;
    switch (wv) {                         // wave-uniform; the immediate must be a literal
        case 0: wait_vmcnt<D::of(0)>(); break;
        case 1: wait_vmcnt<D::of(1)>(); break;
        case 2: wait_vmcnt<D::of(2)>(); break;
        default: wait_vmcnt<D::of(3)>(); break;
    }
}
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// RING = LDS slots: 3 -> DMA runs two items ahead (2 blocks/CU); 2 -> one item ahead, smaller footprint (3 blocks/CU)
template <int MT, int GEO, int RING>
__global__ __launch_bounds__(256, RING == 3 ? 2 : 3) void conv3x3_igemm_dma_kernel(const ConvP p) {
    using G = Geo<GEO>;
    using D = DmaCount<MT, GEO>;
    static_assert(G::IMG == 1 && G::PS == G::ROWS * G::LW, "LDS image must be linear in the staging index");
    constexpr int XS = KC * G::PS, WS = MT * KC * 144, BUF = XS + WS;
    constexpr int XF4_PER_CH = G::ROWS * G::LW / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    SegL* seg_in = reinterpret_cast<SegL*>(smem + RING * BUF);
    SegL* seg_out = seg_in + MTBC_MAX_SEGS;
    float* bias_s = reinterpret_cast<float*>(seg_out + MTBC_MAX_SEGS);      // MT*16 floats
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = p.H * p.W;
    const int mt0 = blockIdx.y * MT;
    segl_fill(seg_in, p.in);
    segl_fill(seg_out, p.out);
    if (tid < MT * 16) { const int co = mt0 * 16 + tid; bias_s[tid] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f; }
    __syncthreads();

    const int nchunks = p.Cin / KC;
    // XCD-aware walk: workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with its own L2.  Every
    // XCD takes one contiguous eighth of the tile list and its blocks sweep it side by side, so the halo rows / the
    // partly used cache lines two neighbouring tiles share are fetched into ONE L2 instead of two.
    int tile0, tstep, tend;
    if ((gridDim.x & 7) == 0 && !MTBC_DBG_BIT(p, 16)) {
        const int per = (p.ntiles + 7) >> 3, xcd = blockIdx.x & 7;
        tile0 = xcd * per + (blockIdx.x >> 3); tstep = gridDim.x >> 3; tend = min(p.ntiles, (xcd + 1) * per);
    } else { tile0 = blockIdx.x; tstep = gridDim.x; tend = p.ntiles; }
    const int my_tiles = tile0 < tend ? (tend - tile0 + tstep - 1) / tstep : 0;
    const int total = my_tiles * nchunks;
    if (total <= 0) return;

    // ---- prefetch cursor (two items ahead of the compute cursor)
    int ptile = tile0, pchunk = 0, pitem = 0;
    int pn0 = 0, py0 = 0, px0 = 0;
    auto set_ptile = [&]() {
        int t = ptile;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y; t /= p.tiles_y;
        pn0 = t; px0 = tx * G::TW; py0 = ty * G::TH;
    };
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.wp), 0, (int)((size_t)p.mtiles * p.Cin * 144 * 4), 0x00020000);
    auto issue = [&]() {            // DMA of item `pitem` into ring slot pitem % 3, then advance the cursor
        const int slot = pitem % RING;
        const int ci0 = pchunk * KC;
        const SegL sr = segl_ref(seg_in, ci0);
        const float* base = sr.ptr + (size_t)(ci0 - sr.cb) * HW + (size_t)pn0 * sr.bs;
        // records: the KC channel planes of image pn0 starting at `base` (always inside the segment tensor)
        // the segment entry came through LDS: tell the compiler the descriptor is wave-uniform (no waterfall loops)
        const unsigned long long bp = reinterpret_cast<unsigned long long>(base);
        const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)bp), bhi = __builtin_amdgcn_readfirstlane((unsigned)(bp >> 32));
        float* ubase = reinterpret_cast<float*>(((unsigned long long)bhi << 32) | blo);
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(ubase, 0, KC * HW * 4, 0x00020000);
#pragma unroll
        for (int k = 0; k < (D::XI + 3) / 4; ++k) {
            const int inst = wv + 4 * k;                 // wave-uniform instruction index
            if (inst < D::XI) {
                const int idx = inst * 64 + lane;
                const int c = idx / XF4_PER_CH, rem = idx % XF4_PER_CH;
                const int row = rem / (G::LW / 4), c4 = rem % (G::LW / 4);
                const int y = py0 + row - 1, x = px0 - 4 + c4 * 4;
                const bool ok = idx < D::XF4 && y >= 0 && y < p.H && x >= 0 && x < p.W;
                const unsigned voff = ok ? (unsigned)((c * HW + y * p.W + x) * 4) : 0xfffffff0u;   // OOB -> zeros
                if (idx < D::XF4)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr_t)(smem + slot * BUF + inst * 256), 16, voff, 0, 0, 0);
            }
        }
        const unsigned wbase = (unsigned)(ci0 * 144 * 4);
#pragma unroll
        for (int k = 0; k < (D::WI + 3) / 4; ++k) {
            const int inst = wv + 4 * k;
            if (inst < D::WI) {
                const int idx = inst * 64 + lane;
                const int mt = idx / (KC * 36), r = idx % (KC * 36);
                const bool ok = idx < D::WF4 && (mt0 + mt) < p.mtiles;
                const unsigned voff = ok ? (unsigned)(((mt0 + mt) * p.Cin) * 144 * 4 + r * 16) + wbase : 0xfffffff0u;
                if (idx < D::WF4)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(smem + slot * BUF + XS + inst * 256), 16, voff, 0, 0, 0);
            }
        }
        ++pitem;
        if (++pchunk == nchunks) { pchunk = 0; ptile += tstep; set_ptile(); }
    };

    // ---- per-lane fragment bases
    const int j = lane & 15, kk = lane >> 4;
    const int laneB = kk * G::PS + j;
    const int gbase = GEO == 0 ? (2 * wv) * G::LW : (4 * wv) * G::LW;
    const int bBase = laneB + gbase + 3;
    const int aBase = XS + kk * 144 + j;

    f32x4 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[m][g] = (f32x4){0.f, 0.f, 0.f, 0.f};

    set_ptile();
    issue();                                   // item 0
    if (RING == 3 && total > 1) { issue(); wait_newest_in_flight<MT, GEO>(wv); } else { wait_vmcnt<0>(); }
    __builtin_amdgcn_s_barrier();

    int chunk = 0, tile = tile0;
    for (int it = 0; it < total; ++it) {
        const int cur = (it % RING) * BUF;
        const bool last = chunk + 1 == nchunks;
        const bool have2 = it + (RING - 1) < total;     // is there an item to prefetch now?
        float fa[2][3][MT], fb[2][3][4];
        auto read_group = [&](int q, int slot) {
            const int cs = q / 3, r = q % 3;
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
                const int tap = r * 3 + s3;
#pragma unroll
                for (int m = 0; m < MT; ++m) fa[slot][s3][m] = smem[cur + aBase + m * (KC * 144) + cs * 576 + tap * 16];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int toff = GEO == 0 ? (g >> 1) * G::LW + 16 * (g & 1) : g * G::LW;
                    fb[slot][s3][g] = smem[cur + bBase + toff + cs * 4 * G::PS + r * G::LW + s3];
                }
            }
        };
        read_group(0, 0);
        if (have2) issue();                    // item it+2 -> slot (it+2)%3, last read during item it-1
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            if (q + 1 < 6) read_group(q + 1, (q + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[q & 1][s3][g], fa[q & 1][s3][m], acc[m][g], 0, 0, 0);      // rows = pixels
            __builtin_amdgcn_sched_barrier(0);
        }
        // item it+1 must have landed before anyone reads it; item it+2 (RING 3) may stay in flight.  Waiting HERE, in
        // front of the epilogue, keeps the epilogue's stores out of the wait: they drain under the next item's MFMAs.
        if (RING == 3 && have2) wait_newest_in_flight<MT, GEO>(wv); else wait_vmcnt<0>();
        if (last) {
            int t = tile;
            const int tx = t % p.tiles_x; t /= p.tiles_x;
            const int ty = t % p.tiles_y; t /= p.tiles_y;
            const int n = t, x0 = tx * G::TW, y0 = ty * G::TH;
            // D = X^T W^T (pixels on the MFMA rows): a lane holds 4 consecutive pixels (4kk .. 4kk+3 of the 16-pixel
            // group) of ONE channel (16m + j) -> one 16-byte store per accumulator tile; the bias comes from LDS; a fan-in
            // segment of dgrad is a 16-byte read-modify-write (its loads are younger than the DMA waited for above).
            int poff[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int y, x;
                if (GEO == 0) { y = y0 + 2 * wv + (g >> 1); x = x0 + 16 * (g & 1) + 4 * kk; }
                else { y = y0 + 4 * wv + g; x = x0 + 4 * kk; }
                poff[g] = (y < p.H && x < p.W) ? y * p.W + x : -1;          // W % 4 == 0: x < W covers x + 3
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int co = (mt0 + m) * 16 + j;
                if (co < p.Cout) {
                    const SegL so = segl_ref(seg_out, co);
                    gfloat* cb = (gfloat*)so.ptr + (size_t)n * so.bs + (size_t)(co - so.cb) * HW;
                    const float bv = bias_s[m * 16 + j];
                    if (so.acc) {
                        f32x4 old[4];
#pragma unroll
                        for (int g = 0; g < 4; ++g) old[g] = poff[g] >= 0 ? *(const gf32x4*)(cb + poff[g]) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            if (poff[g] >= 0) *(gf32x4*)(cb + poff[g]) = old[g] + (acc[m][g] + bv);
                    } else {
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            if (poff[g] >= 0) *(gf32x4*)(cb + poff[g]) = acc[m][g] + bv;
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[m][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            tile += tstep;
            chunk = 0;
        } else {
            ++chunk;
        }
        __builtin_amdgcn_s_barrier();
    }
}

// 16-bit MFMA operand helpers (bf16 / fp16), shared by the low-precision igemm and wgrad kernels
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <bool F16> struct LP;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <> struct LP<false> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ frag pack(const float* f) { frag r;      // four v_cvt_pk_bf16_f32
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16x2 h = __builtin_convertvector((f32x2){f[2 * i], f[2 * i + 1]}, bf16x2);
            r[2 * i] = h[0]; r[2 * i + 1] = h[1];
        }
        return r; }
    static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ float lo(unsigned u) { return __uint_as_float(u << 16); }       // the two stored values of a dword, exact
    static __device__ __forceinline__ float hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
};
template <> struct LP<true> {
    typedef f16x8 frag;
    static __device__ __forceinline__ frag pack(const float* f) { frag r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f16x2 h = __builtin_convertvector((f32x2){f[2 * i], f[2 * i + 1]}, f16x2);
            r[2 * i] = h[0]; r[2 * i + 1] = h[1];
        }
        return r; }
    static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ float lo(unsigned u) { const f16x2 t = __builtin_bit_cast(f16x2, u); return (float)t[0]; }
    static __device__ __forceinline__ float hi(unsigned u) { const f16x2 t = __builtin_bit_cast(f16x2, u); return (float)t[1]; }
};

// ------------------------------------------------------------------ igemm on the 16-bit MFMA (optional compute mode)
// fwd / dgrad with bf16 (or fp16) MFMA operands, fp32 accumulate, fp32 tensors in HBM.  v_mfma_f32_16x16x32 wants 8
// consecutive K per lane, K = input channels at one tap, so the LDS image is channel-interleaved: [halo pixel][32 ch]
// as 16-bit, 80-byte pixel stride (5*px + kg spreads the 16-B slots of a ds_read_b128).  Staging: each lane owns one
// halo pixel and one 8-channel group, reads its 8 channels with 8 dword loads (a wave reads 256 contiguous bytes of
// one plane per load), converts, and writes ONE ds_write_b128.  Weights come pre-converted: [mtile][chunk32][tap][16][32].
// With the matrix pipe 16x faster these convs are load-bound; the structure is therefore the simple one (single LDS
// buffer, 3 blocks per CU overlap each other).
constexpr int LPKC = 32;                 // channels per chunk = K of one MFMA
constexpr int LPROW = 48;                // 16-bit elements per X row of the planar-operand kernel's LDS image (32 + 16 pad) = 96 bytes: row stride = 2 (mod 4) pieces, conflict-free b128 reads (80 bytes was 2-way)
constexpr int WROW = 32;                 // 16-bit elements per row of the weight images (unpadded; pieces XOR-swizzled, see pack_lp_elem8)
template <int GEO> struct GeoLP;
template <> struct GeoLP<0> { static constexpr int TH = 8, TW = 32, IMG = 1; };
template <> struct GeoLP<1> { static constexpr int TH = 16, TW = 16, IMG = 1; };
template <> struct GeoLP<2> { static constexpr int TH = 8, TW = 8, IMG = 4; };

// one thread = 8 consecutive K of one row of a 16-bit image (one 16-byte store); idx8 counts those groups
__device__ __forceinline__ void pack_lp_elem8(const float* __restrict__ w, unsigned short* __restrict__ p, int Cin, int Cout,
                                              int dgrad, int f16, long long idx8) {
    // fwd  : rows = co (Cout), K = ci      value = w[co][ci][tap]
    // dgrad: rows = ci (Cin),  K = co      value = w[co][ci][8-tap]
    const int rows = dgrad ? Cin : Cout, red = dgrad ? Cout : Cin;
    const int nch = (red + LPKC - 1) / LPKC;
    const int kq = idx8 % (WROW / 8); long long t = idx8 / (WROW / 8);
    const int i = t % 16; t /= 16;
    const int tap = t % 9; t /= 9;
    const int cb = t % nch; const int mt = t / nch;
    const int r = mt * 16 + i;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int kk = kq * 8 + e, k = cb * LPKC + kk;
        v[e] = 0.f;
        if (r < rows && k < red)
            v[e] = dgrad ? w[((size_t)k * Cin + r) * 9 + (8 - tap)] : w[((size_t)r * Cin + k) * 9 + tap];
    }
    // 64-byte rows, 16-byte piece kq of row i stored at position kq ^ ((i >> 1) & 3): with that XOR the A-fragment read of the
    // igemm kernels (lane (j, kg) -> row j, piece kg) is conflict-free for ds_read_b128's lane groups; plain 64- or 80-byte
    // rows are 2-way (SQ_LDS_BANK_CONFLICT / brute force over the groups of MI355X_MICROARCH.md)
    const long long dst8 = idx8 - kq + (kq ^ ((i >> 1) & 3));
    if (f16) *reinterpret_cast<f16x8*>(p + dst8 * 8) = LP<true>::pack(v);
    else *reinterpret_cast<bf16x8*>(p + dst8 * 8) = LP<false>::pack(v);
}
// The same image, one (16-row tile, 32-K chunk) UNIT per 256-thread block, through LDS (round 3).  pack_lp_elem8 gathers 8 floats that lie
// 36 B (forward) or 36 Cin B (dgrad) apart: every load instruction of a wave touches 64 cache lines, and the ~70 images of a step took
// 105 us for 120 MB.  Here the unit's slab of the weight tensor -- 16 rows x 288 contiguous floats (forward) / 32 rows x 144 (dgrad) -- is
// read with consecutive lanes on consecutive floats, and the 576 pieces of the unit are cut from LDS.  Same values, same places.
constexpr int PACK_UNIT_SM = 32 * 145;
__device__ __forceinline__ void pack_lp_unit(const float* __restrict__ w, unsigned short* __restrict__ p, int Cin, int Cout, int dgrad,
                                             int f16, int unit, float* __restrict__ sm) {
    const int rows = dgrad ? Cin : Cout, red = dgrad ? Cout : Cin;
    const int nch = (red + LPKC - 1) / LPKC;
    const int cb = unit % nch, mt = unit / nch, r0 = mt * 16, k0 = cb * LPKC;
    const int tid = threadIdx.x;
    if (!dgrad) {          // sm[i][kk * 9 + tap] = w[r0 + i][k0 + kk][tap]
        for (int t = tid; t < 16 * 288; t += 256) {
            const int i = t / 288, c = t - i * 288;
            sm[i * 289 + c] = (r0 + i < rows && k0 + c / 9 < red) ? w[((size_t)(r0 + i) * Cin + k0) * 9 + c] : 0.f;
        }
    } else {               // sm[kk][i * 9 + tap] = w[k0 + kk][r0 + i][tap]
        for (int t = tid; t < 32 * 144; t += 256) {
            const int kk = t / 144, c = t - kk * 144;
            sm[kk * 145 + c] = (k0 + kk < red && r0 + c / 9 < rows) ? w[((size_t)(k0 + kk) * Cin + r0) * 9 + c] : 0.f;
        }
    }
    __syncthreads();
    for (int q = tid; q < 576; q += 256) {
        const int kq = q & 3, i = (q >> 2) & 15, tap = q >> 6;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kk = kq * 8 + e;
            v[e] = dgrad ? sm[kk * 145 + i * 9 + (8 - tap)] : sm[i * 289 + kk * 9 + tap];
        }
        const long long dst8 = (long long)unit * 576 + q - kq + (kq ^ ((i >> 1) & 3));      // (the XOR placement of pack_lp_elem8)
        if (f16) *reinterpret_cast<f16x8*>(p + dst8 * 8) = LP<true>::pack(v);
        else *reinterpret_cast<bf16x8*>(p + dst8 * 8) = LP<false>::pack(v);
    }
}
__global__ void pack_lp_kernel(const float* __restrict__ w, unsigned short* __restrict__ p, int Cin, int Cout, int dgrad,
                               int f16, long long total8) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < total8) pack_lp_elem8(w, p, Cin, Cout, dgrad, f16, idx);
}

// Every weight image of a step in ONE launch (a step re-packs ~70 small tensors after each optimizer update; as 70
// launches that is ~0.8 ms of launch latency).  The descriptors travel by value in the kernel arguments.
constexpr int PACK_MANY = 96;
struct PackManyP {
    const float* w[PACK_MANY];
    void* dst[PACK_MANY];
    int Cin[PACK_MANY], Cout[PACK_MANY];
    unsigned char kind[PACK_MANY];          // 0 fp32 fwd, 1 fp32 dgrad, 2 16-bit fwd, 3 16-bit dgrad
    int first_block[PACK_MANY + 1];
    int n, f16;
};
__global__ void pack_many_kernel(const PackManyP q) {
    int d = 0;
    while (d + 1 < q.n && (int)blockIdx.x >= q.first_block[d + 1]) ++d;          // scalar, uniform
    const long long idx = (long long)((int)blockIdx.x - q.first_block[d]) * blockDim.x + threadIdx.x;
    const int Cin = q.Cin[d], Cout = q.Cout[d], kind = q.kind[d];
    const float* w = q.w[d];
    if (kind < 2) {
        const long long total = kind == 0 ? (long long)((Cout + 15) / 16) * Cin * 144 : (long long)((Cin + 15) / 16) * Cout * 144;
        if (kind == 0) {
            // forward image = per 16-channel tile the transpose of a [16 co][9*Cin] slab: read 16 consecutive K of a
            // row (64 contiguous bytes; element-wise the 16 lanes of a tile row hit 16 different weight rows), turn
            // the 16x16 patch through LDS, write 1 KB contiguous
            __shared__ float tile[16][17];
            const int K9 = 9 * Cin;
            const long long g = (long long)((int)blockIdx.x - q.first_block[d]) * 16 + (threadIdx.x & 15);
            const int ir = threadIdx.x >> 4;
            const int mt = (int)(g / K9), k = (int)(g % K9), co = mt * 16 + ir;
            tile[ir][threadIdx.x & 15] = (g * 16 < total && co < Cout) ? w[(size_t)co * K9 + k] : 0.f;
            __syncthreads();
            if (idx < total) static_cast<float*>(q.dst[d])[idx] = tile[threadIdx.x & 15][threadIdx.x >> 4];
            return;
        }
        if (idx >= total) return;
        pack_dgrad_elem(w, static_cast<float*>(q.dst[d]), Cin, Cout, (int)idx);
    } else {               // 16-bit images: one block per (row tile, chunk) unit
        __shared__ float sm[PACK_UNIT_SM];
        pack_lp_unit(w, static_cast<unsigned short*>(q.dst[d]), Cin, Cout, kind == 3, q.f16, (int)blockIdx.x - q.first_block[d], sm);
    }
}

template <int MT, int GEO, bool F16>
__global__ __launch_bounds__(256, 3) void conv3x3_igemm_lp_kernel(const ConvP p) {
    using G = GeoLP<GEO>;
    using T = LP<F16>;
    constexpr int HR = G::TH + 2, HC = G::TW + 2, HP = G::IMG * HR * HC;       // halo pixels
    constexpr int XB = HP * LPROW;                                             // 16-bit elements
    constexpr int WB = MT * 9 * 16 * WROW;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    unsigned short* Xs = smem16;
    unsigned short* Ws = smem16 + XB;
    SegL* seg_in = reinterpret_cast<SegL*>(smem16 + XB + WB);
    SegL* seg_out = seg_in + MTBC_MAX_SEGS;
    float* bias_s = reinterpret_cast<float*>(seg_out + MTBC_MAX_SEGS);      // MT*16 floats
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int HW = p.H * p.W;
    const int mt0 = blockIdx.y * MT;
    segl_fill(seg_in, p.in);
    segl_fill(seg_out, p.out);
    if (tid < MT * 16) { const int co = mt0 * 16 + tid; bias_s[tid] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f; }
    __syncthreads();

    const int nchunks = (p.Cin + LPKC - 1) / LPKC;
    const int j = lane & 15, kg = lane >> 4;

    // fragment read bases (16-bit element offsets)
    int bpix[4];                      // halo-pixel index of this lane's output pixel in group g, tap (0,0)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        int img = 0, y, x;
        if (GEO == 0) { y = 2 * wv + (g >> 1); x = 16 * (g & 1) + j; }
        else if (GEO == 1) { y = 4 * wv + g; x = j; }
        else { img = wv; y = 2 * g + (j >> 3); x = j & 7; }
        bpix[g] = (img * HR + y) * HC + x;
    }

    // The block walks the flattened (tile, chunk) list; the X values of step s+1 travel in registers while step s
    // feeds the MFMAs, across tile boundaries too (HBM latency must not sit between barriers).  Staging is arranged
    // so that almost all of its address arithmetic is scalar: wave w owns channel group w of the chunk (8 channels,
    // one segment lookup per chunk, plane bases in SGPRs) and its lanes own halo pixels lane + 64q, whose byte
    // offsets are computed once per tile -- each load is then one saddr + 32-bit-voffset instruction.
    constexpr int XQ = (HP + 63) / 64;
    float xf[XQ][8];
    unsigned pixb[XQ];                            // byte offset y*W + x of the halo pixel inside a plane (0 if outside)
    unsigned okm = 0;                             // bit q: halo pixel q of the fetched tile lies inside the image
    const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
    int fn0 = 0;                                  // first image of the tile pixb/okm describe
    auto tile_origin = [&](int tile, int& n0, int& y0, int& x0) {
        int t = tile;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y; t /= p.tiles_y;
        n0 = t * G::IMG; x0 = tx * G::TW; y0 = ty * G::TH;
    };
    auto tile_geom = [&](int tile) {
        int y0, x0;
        tile_origin(tile, fn0, y0, x0);
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int hp = lane + 64 * q;
            const int img = hp / (HR * HC), rem = hp % (HR * HC);
            const int row = rem / HC, col = rem % HC;
            const int y = y0 + row - 1, x = x0 + col - 1;
            const bool ok = hp < HP && fn0 + img < p.N && y >= 0 && y < p.H && x >= 0 && x < p.W && !MTBC_DBG_BIT(p, 1);
            pixb[q] = ok ? 4u * (unsigned)(y * p.W + x) : 0xfffffff0u;      // out of range for the buffer: reads 0
            if (G::IMG > 1) okm |= ok ? (1u << q) : 0u;
        }
    };
    // fetch_x only ISSUES loads -- any use of a loaded register here would put the s_waitcnt in front of the MFMAs
    // and serialize the pipeline.  They are raw buffer loads over the 8 channel planes of the group: the plane
    // stride rides in the scalar offset, a halo pixel outside the image carries an out-of-range offset and an
    // absent channel group gets an empty buffer, so the hardware returns the zeros and a load costs no VALU work.
    auto fetch_x = [&](int chn) {
        const int c0 = chn * LPKC + 8 * wvu;      // wave-uniform
        const bool xgrp = c0 < p.Cin;
        const SegL sr = segl_ref(seg_in, xgrp ? c0 : 0);
        const unsigned long long pu = reinterpret_cast<unsigned long long>(sr.ptr);
        const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pu), phi = __builtin_amdgcn_readfirstlane((unsigned)(pu >> 32));
        const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)sr.bs), bhi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)sr.bs >> 32));
        const int cb = __builtin_amdgcn_readfirstlane(sr.cb);
        const long long bs = (long long)(((unsigned long long)bhi << 32) | blo);
        float* base = reinterpret_cast<float*>(((unsigned long long)phi << 32) | plo) +
                      ((size_t)fn0 * bs + (size_t)((xgrp ? c0 : 0) - cb) * HW);
        // records: 8 planes of image fn0 (GEO2: of the tile's 4 images, reached through the batch stride)
        const unsigned span = G::IMG > 1 ? (unsigned)((G::IMG - 1) * bs + 8 * HW) * 4u : (unsigned)(8 * HW) * 4u;
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, xgrp ? (int)span : 0, 0x00020000);
        const int plane = HW * 4;
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            unsigned off = pixb[q];
            if (G::IMG > 1) off += ((okm >> q) & 1u) ? 4u * (unsigned)(((lane + 64 * q) / (HR * HC)) * bs) : 0u;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                xf[q][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, (int)off, e * plane, 0));
        }
    };
    int w_have = -1;                              // chunk whose weights sit in Ws (mt0 is fixed per block)
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.wp), 0, (int)((size_t)p.mtiles * nchunks * (9 * 16 * WROW) * 2), 0x00020000);
    // XCD-aware walk: workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with its own L2.  Every
    // XCD takes one contiguous eighth of the tile list and its blocks sweep it side by side, so the halo rows / the
    // partly used cache lines two neighbouring tiles share are fetched into ONE L2 instead of two.
    int tile, tstep, tend;
    if ((gridDim.x & 7) == 0 && !MTBC_DBG_BIT(p, 16)) {
        const int per = (p.ntiles + 7) >> 3, xcd = blockIdx.x & 7;
        tile = xcd * per + (blockIdx.x >> 3); tstep = gridDim.x >> 3; tend = min(p.ntiles, (xcd + 1) * per);
    } else { tile = blockIdx.x; tstep = gridDim.x; tend = p.ntiles; }
    if (tile < tend) { tile_geom(tile); fetch_x(0); }
    for (; tile < tend; tile += tstep) {
        int n0, y0, x0;
        tile_origin(tile, n0, y0, x0);
        f32x4 acc[MT][4];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[m][g] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int ch = 0; ch < nchunks; ++ch) {
            lds_barrier();                        // previous step's fragments are consumed
            // ---- W: LDS-DMA of the pre-converted image (L2-resident, shared by every block), 1 KB per wave
            //      instruction, issued first so that waiting for it later leaves the X prefetch in flight; a
            //      single-chunk conv keeps its weights in LDS for the whole launch
            const bool wload = w_have != ch && !MTBC_DBG_BIT(p, 8);
            if (wload) {
                constexpr int W16 = WB / 8;       // 16-byte pieces
                constexpr int WI = (W16 + 63) / 64;
#pragma unroll
                for (int k = 0; k < (WI + 3) / 4; ++k) {
                    const int inst = wvu + 4 * k;
                    if (inst < WI) {
                        const int idx = inst * 64 + lane;
                        const int mt = idx / (9 * 16 * WROW / 8), r = idx % (9 * 16 * WROW / 8);
                        const bool ok = idx < W16 && (mt0 + mt) < p.mtiles;
                        const unsigned voff = ok ? (unsigned)((((mt0 + mt) * nchunks + ch) * (9 * 16 * WROW / 8) + r) * 16) : 0xfffffff0u;
                        if (idx < W16)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(Ws + inst * 512), 16, voff, 0, 0, 0);
                    }
                }
                w_have = ch;
            }
#pragma unroll
            for (int q = 0; q < XQ; ++q) {
                const int hp = lane + 64 * q;
                if (hp < HP) *reinterpret_cast<typename T::frag*>(Xs + hp * LPROW + 8 * wvu) = T::pack(xf[q]);
            }
            // next step's X: in flight under the MFMAs and the epilogue
            bool fetched = false;
            if (ch + 1 < nchunks) { fetch_x(ch + 1); fetched = true; }
            else if (tile + tstep < tend) { tile_geom(tile + tstep); fetch_x(0); fetched = true; }
            if (wload) {                          // the weight DMA is older than the X loads just issued
                if (fetched) wait_vmcnt<8 * XQ>(); else wait_vmcnt<0>();
            }
            lds_barrier();
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int toff = ((tap / 3) * HC + tap % 3) * LPROW + 8 * kg;
                typename T::frag a[MT], b[4];
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    a[m] = *reinterpret_cast<const typename T::frag*>(Ws + ((m * 9 + tap) * 16 + j) * WROW + 8 * (kg ^ ((j >> 1) & 3)));
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    b[g] = *reinterpret_cast<const typename T::frag*>(Xs + bpix[g] * LPROW + toff);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[m][g] = T::mfma(b[g], a[m], acc[m][g]);      // rows = pixels, cols = channels
            }
        }
        // ---- epilogue.  The MFMAs ran as D = X^T W^T (pixels on the rows): a lane holds FOUR CONSECUTIVE PIXELS
        //      (4kg .. 4kg+3 of the 16-pixel group) of ONE output channel (16m + j), i.e. one 16-byte store per
        //      accumulator tile -- the epilogue is store-issue bound.  The bias comes from LDS (a global load here would
        //      be waited for together with the whole X prefetch: vmcnt retires in order).  A fan-in segment of dgrad is a
        //      16-byte read-modify-write: measured against no-return float atomics (4 per store, coalesced only with
        //      channels on the rows) it is 7 % faster over the step's dgrads even though its loads queue behind the prefetch.
        const int n = GEO == 2 ? n0 + wv : n0;
        if MTBC_DBG_BIT(p, 2) { if (acc[0][0][0] != 12345.678f) continue; }
        int poff[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            int y, x;
            if (GEO == 0) { y = y0 + 2 * wv + (g >> 1); x = x0 + 16 * (g & 1) + 4 * kg; }
            else if (GEO == 1) { y = y0 + 4 * wv + g; x = x0 + 4 * kg; }
            else { y = y0 + 2 * g + (kg >> 1); x = x0 + 4 * (kg & 1); }
            poff[g] = (n < p.N && y < p.H && x < p.W) ? y * p.W + x : -1;        // W % 4 == 0: x < W covers x + 3
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int co = (mt0 + m) * 16 + j;
            if (co >= p.Cout) continue;
            const SegL so = segl_ref(seg_out, co);
            gfloat* cb = (gfloat*)so.ptr + (size_t)n * so.bs + (size_t)(co - so.cb) * HW;
            const float bv = bias_s[m * 16 + j];
            if (so.acc) {
                f32x4 old[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) old[g] = poff[g] >= 0 ? *(const gf32x4*)(cb + poff[g]) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (poff[g] >= 0) *(gf32x4*)(cb + poff[g]) = old[g] + (acc[m][g] + bv);
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (poff[g] >= 0) *(gf32x4*)(cb + poff[g]) = acc[m][g] + bv;
            }
        }
    }
}

// ------------------------------------------------------------------ 16-bit channel-blocked operands ("c8")
// The tensor the MFMA reads is ALREADY stored in the MFMA's 16-bit type as [n][C/8][H*W][8]: one 16-byte piece = 8
// channels of one pixel = one lane's share of a 16x16x32 fragment.  Staging is then pure LDS-DMA -- no staging VGPRs, no
// conversions, half the bytes -- and the fragment reads are the same ds_read_b128 as above on a [4 groups][halo pixel][8]
// image.  Everything after the staging (MFMA order, epilogue, fp32 planar output with fan-in read-modify-write) is the
// kernel above, so results are bit-identical to it on the same rounded operands.  SegL.ptr carries the 16-bit base and
// SegL.bs the batch stride in 16-bit elements; every segment holds a multiple of 8 channels.
// Measured stand-alone (tools/experiments/c8_igemm_probe.hip): fwd 144->24 @256x256 N=32 0.27 ms against 0.46 ms.
// NW = waves per block: 4 (256 pixels: 8 x 32, 16 x 16 or four 8 x 8 images) or, wide maps only, 8 (512 pixels: 16 x 32 --
// 19.5 % halo instead of 33 % and each weight chunk shared by twice the pixels; 58 KB of LDS = 2 blocks per CU.  Measured
// (tools/experiments/c8_igemm_v2_probe.hip): 144->24 @256x256 -6..-17 %, 24->24 -3 %, 64x64 maps +15 %: chosen by run_igemm).
// sum over the 16 lanes of a DPP row, result in lane 15 of the row: four v_add_f32 with row_shr modifiers (no LDS crossbar)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));      // row_shr:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));      // row_shr:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));      // row_shr:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));      // row_shr:1
    return v;
}

// sum over the wave, on the VALU: DPP row sums, then the four rows' last lanes (fixed order); every lane gets the result
__device__ __forceinline__ float wave_sum_rows(float v) {
    const int r = __builtin_bit_cast(int, row16_sum(v));
    return ((__builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 15)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 31))) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 47))) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 63));
}

// Epilogue of the channel-blocked igemm kernels (one pixel tile's accumulators -> HBM), shared by the single-buffer kernel and
// the ring kernel of the deep levels.  O8 = 0: fp32 planar segments (store / read-modify-write / 16-bit planes); 1: 16-bit
// channel-blocked (+ forward InstanceNorm statistics); 2: that + the norm-backward reductions (zpre = the tensor's z, prefetched).
// FULL (round 3, late): the tile lies inside the image and the batch -- no per-pixel validity, no predicated stores; `wv` is the wave index
// as a SCALAR (readfirstlane): with it the pixel offsets, the statistics slot and their 64-bit addresses are scalar arithmetic + one vector
// add per lane instead of vector multiplies per accumulator tile.  (Phase timestamps, MTBC_C8_TS: a 24 -> 24 tile of 7.6 us spent 2.8 us in
// this epilogue and 2.3 us between its start and its last DMA instruction -- ~1000 vector / scalar instructions beside 72 MFMAs per wave,
// on SIMDs that four waves share: instruction issue, not bytes and not the matrix pipe, is what the level-0 launches wait for.)
template <int MT, int GEO, bool F16, int NW, int O8, bool FULL, typename ZPRE>
__device__ __forceinline__ void c8_epilogue(const ConvP& p, f32x4 (&acc)[MT][4], const SegL* seg_out, const float* bias_s, const int n0,
                                            const int y0, const int x0, const int tx, const int ty, const int mt0, const int wv,
                                            const int j, const int kg, const int HW, ZPRE& zpre) {
    using T = LP<F16>;
    using TO = LP<F16 || O8 == 3>;       // type of a channel-blocked output: the operands' type, or fp16 for the conv output z of the bf16 mode (O8 == 3)
    typedef unsigned pre_u32x2 __attribute__((ext_vector_type(2)));
    const int n = GEO == 2 ? n0 + wv : n0;
    if MTBC_DBG_BIT(p, 2) { if (acc[0][0][0] != 12345.678f) return; }
    if constexpr (O8 != 0) {
        // ---- epilogue, channel-blocked 16-bit output: lane (j, kg) holds channels 16m + 4kg .. + 3 of pixel j of group g
        typedef unsigned ep_u32x2 __attribute__((ext_vector_type(2)));
        typedef unsigned ep_u32x4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(1))) ep_u32x2 guint2;
        int pix[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            int y, x;
            if (GEO == 0) { y = y0 + 2 * wv + (g >> 1); x = x0 + 16 * (g & 1) + j; }
            else if (GEO == 1) { y = y0 + 4 * wv + g; x = x0 + j; }
            else { y = y0 + 2 * g + (j >> 3); x = x0 + (j & 7); }
            pix[g] = (FULL || (n < p.N && y < p.H && x < p.W)) ? y * p.W + x : -1;
        }
        const bool want_stats = O8 == 2 || p.stats != nullptr;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int co = (mt0 + m) * 16 + 4 * kg;
            if (co >= p.Cout) continue;
            const SegL so = segl_ref_n(seg_out, co, p.out.n);
            // piece (n, group, pixel) of the segment's tensor; this lane owns channels (co - cb) % 8 .. + 3 of it
            const size_t poff8 = 2 * ((size_t)n * so.bs + (size_t)((co - so.cb) >> 3) * HW * 8) + 2 * ((co - so.cb) & 7);
            gchar* cb = (gchar*)so.ptr + poff8;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + m * 16 + 4 * kg);
            f32x4 ss = (f32x4){0.f, 0.f, 0.f, 0.f}, sq = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (O8 == 2) {
                // gathered dgrad + the reductions of the InstanceNorm / LeakyReLU backward of the tensor it differentiates
                // (one output segment = the whole tensor): dy = sum over the 3x3 consumers (+ the other readers' fp32 partial),
                // rounded once and stored; g = dy_stored * lrelu'(gamma * xhat + beta); the wave's {sum g, sum g * xhat}
                const size_t plane = (size_t)n * p.Cout + co;
                f32x4 mean4 = (f32x4){0.f, 0.f, 0.f, 0.f}, rstd4 = mean4, ga4 = (f32x4){1.f, 1.f, 1.f, 1.f}, be4 = mean4;
                if (n < p.N) { mean4 = *reinterpret_cast<const f32x4*>(p.nmean + plane); rstd4 = *reinterpret_cast<const f32x4*>(p.nrstd + plane); }
                if (p.ngamma) { ga4 = *reinterpret_cast<const f32x4*>(p.ngamma + co); be4 = *reinterpret_cast<const f32x4*>(p.nbeta + co); }
                const float* eb = p.extra ? p.extra + plane * HW : nullptr;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (FULL || pix[g] >= 0) {
                        const pre_u32x2 zw = zpre[m][g];
                        f32x4 ex = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (eb) ex = (f32x4){eb[pix[g]], eb[(size_t)HW + pix[g]], eb[2 * (size_t)HW + pix[g]], eb[3 * (size_t)HW + pix[g]]};
                        const f32x4 r = acc[m][g] + ex;
                        const float q[8] = {r[0], r[1], r[2], r[3], 0.f, 0.f, 0.f, 0.f};
                        const ep_u32x4 u = __builtin_bit_cast(ep_u32x4, T::pack(q));
                        *(guint2*)(cb + 16 * (size_t)pix[g]) = (ep_u32x2){u[0], u[1]};
                        const unsigned u0 = u[0], u1 = u[1], z0 = zw[0], z1 = zw[1];
                        const f32x4 dyv = (f32x4){T::lo(u0), T::hi(u0), T::lo(u1), T::hi(u1)};
                        const f32x4 xh = ((f32x4){T::lo(z0), T::hi(z0), T::lo(z1), T::hi(z1)} - mean4) * rstd4;
                        const f32x4 pre = xh * ga4 + be4;
                        f32x4 gg;
#pragma unroll
                        for (int e = 0; e < 4; ++e) gg[e] = dyv[e] * (pre[e] > 0.f ? 1.f : p.nslope);
                        ss += gg; sq += gg * xh;
                    }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (FULL || pix[g] >= 0) {
                        f32x4 r = acc[m][g] + bv;
                        if constexpr (O8 == 3) {          // fp16 storage of a bf16-mode conv output: saturate (an inf would poison the norm behind it)
#pragma unroll
                            for (int e = 0; e < 4; ++e) r[e] = __builtin_amdgcn_fmed3f(r[e], -65504.f, 65504.f);
                        }
                        const float q[8] = {r[0], r[1], r[2], r[3], 0.f, 0.f, 0.f, 0.f};
                        const ep_u32x4 u = __builtin_bit_cast(ep_u32x4, TO::pack(q));
                        *(guint2*)(cb + 16 * (size_t)pix[g]) = (ep_u32x2){u[0], u[1]};
                        if (p.stats) {          // InstanceNorm statistics of the STORED values
                            const unsigned u0 = u[0], u1 = u[1];
                            const f32x4 v = (f32x4){TO::lo(u0), TO::hi(u0), TO::lo(u1), TO::hi(u1)};
                            ss += v; sq += v * v;
                        }
                    }
            }
            if (want_stats) {
                // this wave's pixels: the 16 lanes of a row group hold 16 pixels of the same 4 channels -> DPP row sums
#pragma unroll
                for (int e = 0; e < 4; ++e) { ss[e] = row16_sum(ss[e]); sq[e] = row16_sum(sq[e]); }
                if (j == 15 && (FULL || n < p.N)) {
                    const int slots = GEO == 2 ? 1 : p.tiles_x * p.tiles_y * NW;
                    const int slot = GEO == 2 ? 0 : (ty * p.tiles_x + tx) * NW + wv;
                    float* sp = p.stats + (((size_t)n * slots + slot) * p.Cout + co) * 2;
                    *reinterpret_cast<f32x4*>(sp) = (f32x4){ss[0], sq[0], ss[1], sq[1]};
                    *reinterpret_cast<f32x4*>(sp + 4) = (f32x4){ss[2], sq[2], ss[3], sq[3]};
                }
            }
        }
        return;
    }
    // ---- epilogue: identical to conv3x3_igemm_lp_kernel (fp32 planar output, 16-byte stores / read-modify-write)
    int poff[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        int y, x;
        if (GEO == 0) { y = y0 + 2 * wv + (g >> 1); x = x0 + 16 * (g & 1) + 4 * kg; }
        else if (GEO == 1) { y = y0 + 4 * wv + g; x = x0 + 4 * kg; }
        else { y = y0 + 2 * g + (kg >> 1); x = x0 + 4 * (kg & 1); }
        poff[g] = (FULL || (n < p.N && y < p.H && x < p.W)) ? y * p.W + x : -1;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int co = (mt0 + m) * 16 + j;
        if (co >= p.Cout) continue;
        const SegL so = segl_ref_n(seg_out, co, p.out.n);
        gfloat* cb = (gfloat*)so.ptr + (size_t)n * so.bs + (size_t)(co - so.cb) * HW;
        const float bv = bias_s[m * 16 + j];
        if (so.acc == 2) {          // 16-bit planar segment: the lane's 4 consecutive pixels of its channel = one 8-byte store
            typedef unsigned ep_u32x2 __attribute__((ext_vector_type(2)));
            typedef unsigned ep_u32x4 __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(1))) ep_u32x2 guint2;
            gchar* c16 = (gchar*)so.ptr + 2 * ((size_t)n * so.bs + (size_t)(co - so.cb) * HW);
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (FULL || poff[g] >= 0) {
                    const float q[8] = {acc[m][g][0] + bv, acc[m][g][1] + bv, acc[m][g][2] + bv, acc[m][g][3] + bv, 0.f, 0.f, 0.f, 0.f};
                    const typename T::frag h = T::pack(q);          // RNE, the conversion every consumer's staging would apply
                    const ep_u32x4 u = __builtin_bit_cast(ep_u32x4, h);
                    *(guint2*)(c16 + 2 * (size_t)poff[g]) = (ep_u32x2){u[0], u[1]};
                }
        } else if (so.acc) {
            f32x4 old[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) old[g] = (FULL || poff[g] >= 0) ? *(const gf32x4*)(cb + poff[g]) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (FULL || poff[g] >= 0) *(gf32x4*)(cb + poff[g]) = old[g] + (acc[m][g] + bv);
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (FULL || poff[g] >= 0) *(gf32x4*)(cb + poff[g]) = acc[m][g] + bv;
        }
    }
}

// halo-pixel offset of pixel group g from group 0 of the same lane (GEO 0: 2 rows x two 16-column halves; 1: 4 rows; 2: 4 row pairs of an 8 x 8 map)
template <int GEO> __device__ __forceinline__ constexpr int c8_goff(int g, int HC) { return GEO == 0 ? (g >> 1) * HC + 16 * (g & 1) : GEO == 1 ? g * HC : 2 * g * HC; }

// O8 = the OUTPUT is 16-bit channel-blocked as well (conv outputs z / single-writer gradients of the 16-bit modes): the MFMAs
// run as D = W X (channels on the rows), so a lane holds 4 consecutive channels of ONE pixel = half a 16-byte piece; the 16
// lanes of a row group write 16 consecutive pixels, and the lane groups kg = 2q, 2q + 1 the two halves of the same pieces
// (256 contiguous bytes per channel group and instruction).  fp32 accumulate + bias, one RNE.
template <int MT, int GEO, bool F16, int NW, int O8>      // O8: 0 = fp32 planar output, 1 = 16-bit channel-blocked (+ forward statistics), 2 = that + the norm-backward epilogue, 3 = 1 stored as fp16 (bf16 operands)
__global__ __launch_bounds__(64 * NW, NW == 4 ? 3 : 4) void conv3x3_igemm_c8_kernel(const ConvP p) {
    using G = GeoLP<GEO>;
    using T = LP<F16>;
    static_assert(NW == 4 || (NW == 8 && GEO == 0), "8-wave blocks: wide-map geometry only");
    constexpr int TH = GEO == 0 ? 2 * NW : G::TH;
    constexpr int HR = TH + 2, HC = G::TW + 2, HP = G::IMG * HR * HC;       // halo pixels
    constexpr int HPP = (HP + 15) / 16 * 16;       // group stride = 0 (mod 256 B): the four 16-lane groups of a ds_read_b128 hit disjoint banks (PMC-checked)
    constexpr int XB = 4 * HPP * 8;                                            // 16-bit elements
    constexpr int WB = MT * 9 * 16 * WROW;
    constexpr int XQ = (HPP + 63) / 64;                                        // DMA instructions per channel group ...
    constexpr int XPW = NW == 4 ? XQ : (XQ + 1) / 2;                           // ... per wave (8 waves: two waves share a group)
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    unsigned short* Xs = smem16;
    unsigned short* Ws = smem16 + XB;
    SegL* seg_in = reinterpret_cast<SegL*>(smem16 + XB + WB);
    SegL* seg_out = seg_in + MTBC_MAX_SEGS;
    float* bias_s = reinterpret_cast<float*>(seg_out + MTBC_MAX_SEGS);      // MT*16 floats
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = p.H * p.W;
    const int mt0 = blockIdx.y * MT;
    // the bias is REQUESTED now and lands in LDS right in front of the tile loop: its memory round trip runs under the block's set-up arithmetic
    // instead of in front of it (round 4; with the scratch-free segl_fill no other memory access is left in the prologue)
    float bias_v = 0.f;
    if (tid < MT * 16) { const int co = mt0 * 16 + tid; if (p.bias && co < p.Cout) bias_v = p.bias[co]; }
    segl_fill(seg_in, p.in);
    segl_fill(seg_out, p.out);

    const int nchunks = (p.Cin + LPKC - 1) / LPKC;
    const int j = lane & 15, kg = lane >> 4;
    // Halo-pixel index of this lane's output pixel in group 0, tap (0,0); group g and tap t sit at COMPILE-TIME offsets from it
    // (c8_goff), so the 36 fragment reads of a chunk are one address register + the ds_read immediate.  (Written as an array bpix[g]
    // the compiler kept 36 separate address registers, ran out of the 128 this occupancy allows, and spilled the per-tile halo
    // constants: every tile then began with scratch reloads whose vmcnt(0) also drained the previous tile's output stores.)
    int bpix0;
    if (GEO == 0) bpix0 = (2 * wvu) * HC + j;
    else if (GEO == 1) bpix0 = (4 * wvu) * HC + j;
    else bpix0 = (wvu * HR + (j >> 3)) * HC + (j & 7);
    const unsigned short* const xlane = Xs + (kg * HPP + bpix0) * 8;
    int w_have = -1;
    // weights: what a lane loads does not depend on the chunk except through a constant stride (one chunk of one channel tile = 9216 B), so the
    // per-instruction index arithmetic is done once per block and the chunk goes into the load's SCALAR offset (round 3, late)
    constexpr int W16 = WB / 8, WI = (W16 + 63) / 64, WK = (WI + NW - 1) / NW;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.wp), 0, (int)((size_t)p.mtiles * nchunks * (9 * 16 * WROW) * 2), 0x00020000);
    unsigned wvoff[WK];
#pragma unroll
    for (int k = 0; k < WK; ++k) {
        const int inst = wvu + NW * k, idx = inst * 64 + lane;
        const int mt = idx / (9 * 16 * WROW / 8), r = idx % (9 * 16 * WROW / 8);
        const bool ok = inst < WI && idx < W16 && (mt0 + mt) < p.mtiles;
        wvoff[k] = ok ? (unsigned)(((mt0 + mt) * nchunks * (9 * 16 * WROW / 8) + r) * 16) : 0xfffffff0u;
    }
    int tile, tstep, tend;            // XCD-aware walk, as above
    if ((gridDim.x & 7) == 0 && !MTBC_DBG_BIT(p, 16)) {
        const int per = (p.ntiles + 7) >> 3, xcd = blockIdx.x & 7;
        tile = xcd * per + (blockIdx.x >> 3); tstep = gridDim.x >> 3; tend = min(p.ntiles, (xcd + 1) * per);
    } else { tile = blockIdx.x; tstep = gridDim.x; tend = p.ntiles; }
    // what a lane's halo pieces are does not depend on the tile: (image, row, column) of piece q inside the halo image, once per block
    const int xgrp_w = NW == 4 ? wvu : (wvu >> 1), xq0 = NW == 4 ? 0 : (wvu & 1) * XPW;      // this wave's channel group / first piece row
    int hrc[XPW];                     // (image << 20) | (row << 10) | column, -1 past the halo image
#pragma unroll
    for (int q = 0; q < XPW; ++q) {
        const int hp = lane + 64 * (xq0 + q);
        const int img = hp / (HR * HC), rem = hp % (HR * HC);
        hrc[q] = (hp < HP && !MTBC_DBG_BIT(p, 1)) ? ((img << 20) | ((rem / HC) << 10) | (rem % HC)) : -1;
    }
    // tile index -> (column, row, image): shifts when the tile counts are powers of two (every size of the BASELINE configurations)
    const int txs = (p.tiles_x & (p.tiles_x - 1)) == 0 ? __builtin_ctz(p.tiles_x) : -1;
    const int tys = (p.tiles_y & (p.tiles_y - 1)) == 0 ? __builtin_ctz(p.tiles_y) : -1;
    int tsk = 0;      // (probes: tile counter of the phase stamps)
    if (tid < MT * 16) bias_s[tid] = bias_v;
    __syncthreads();                     // segment tables + bias are in LDS
    for (; tile < tend; tile += tstep) {
        int t = tile, tx, ty;
        if (txs >= 0 && tys >= 0) { tx = t & (p.tiles_x - 1); t >>= txs; ty = t & (p.tiles_y - 1); t >>= tys; }
        else { tx = t % p.tiles_x; t /= p.tiles_x; ty = t % p.tiles_y; t /= p.tiles_y; }
        const int n0 = t * G::IMG, x0 = tx * G::TW, y0 = ty * TH;
        if (tsk < 3) MTBC_TS(p, 4 * tsk);
        unsigned pixo[XPW];           // byte offset of the halo pixel's piece inside a channel group (out of range: reads 0)
        int pimg[XPW];
#pragma unroll
        for (int q = 0; q < XPW; ++q) {
            const int rc = hrc[q];
            const int img = G::IMG > 1 ? (rc >> 20) : 0;
            const int y = y0 + ((rc >> 10) & 1023) - 1, x = x0 + (rc & 1023) - 1;
            const bool ok = rc >= 0 && (G::IMG == 1 || n0 + img < p.N) && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            pixo[q] = ok ? 16u * (unsigned)(y * p.W + x) : 0xfffffff0u;
            pimg[q] = ok ? img : 0;
        }
        f32x4 acc[MT][4];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[m][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // O8 == 2: the tensor's own z at this lane's output positions, requested NOW so that it arrives under the MFMAs (loaded in
        // the epilogue it is a full memory latency per tile with the accumulators parked)
        typedef unsigned pre_u32x2 __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(1))) pre_u32x2 gpre2;
        pre_u32x2 zpre[O8 == 2 ? MT : 1][4];
        if constexpr (O8 == 2) {
            const int nn = GEO == 2 ? n0 + wvu : n0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int y, x;
                if (GEO == 0) { y = y0 + 2 * wvu + (g >> 1); x = x0 + 16 * (g & 1) + j; }
                else if (GEO == 1) { y = y0 + 4 * wvu + g; x = x0 + j; }
                else { y = y0 + 2 * g + (j >> 3); x = x0 + (j & 7); }
                const bool ok = nn < p.N && y < p.H && x < p.W;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int co = (mt0 + m) * 16 + 4 * kg;
                    zpre[m][g] = (pre_u32x2){0u, 0u};
                    if (ok && co < p.Cout)
                        zpre[m][g] = *(const gpre2*)((const gchar*)p.nz + 2 * ((size_t)nn * p.Cout * HW + (size_t)(co >> 3) * HW * 8) + 2 * (co & 7) + 16 * (size_t)(y * p.W + x));
                }
            }
        }

        for (int ch = 0; ch < nchunks; ++ch) {
            lds_barrier();                        // the previous step's fragments are consumed
            {   // ---- X: wave w (wave pair w/2 for 8 waves) brings one channel group of the chunk, 64 halo pixels x 16 bytes per instruction
                const int c0 = ch * LPKC + 8 * xgrp_w;
                const bool xgrp = c0 < p.Cin;
                const SegL sr = segl_ref_n(seg_in, xgrp ? c0 : 0, p.in.n);
                const unsigned long long pu = reinterpret_cast<unsigned long long>(sr.ptr);
                const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pu), phi = __builtin_amdgcn_readfirstlane((unsigned)(pu >> 32));
                const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)sr.bs), bhi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)sr.bs >> 32));
                const int cb = __builtin_amdgcn_readfirstlane(sr.cb);
                const long long bs = (long long)(((unsigned long long)bhi << 32) | blo);
                unsigned short* base = reinterpret_cast<unsigned short*>(((unsigned long long)phi << 32) | plo) +
                                       ((size_t)n0 * bs + (size_t)((xgrp ? c0 : 0) - cb) * HW);
                const unsigned span = (G::IMG > 1 ? (unsigned)((G::IMG - 1) * bs) * 2u : 0u) + (unsigned)HW * 16u;
                const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, xgrp ? (int)span : 0, 0x00020000);
#pragma unroll
                for (int q = 0; q < XPW; ++q) {
                    unsigned off = pixo[q];
                    if (G::IMG > 1) off += 2u * (unsigned)(pimg[q] * bs);
                    if (xq0 + q < XQ && lane + 64 * (xq0 + q) < HPP)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr_t)(Xs + (xgrp_w * HPP + 64 * (xq0 + q)) * 8), 16, off, 0, 0, 0);
                }
            }
            if (w_have != ch && !MTBC_DBG_BIT(p, 8)) {   // ---- W: a single-chunk conv keeps its weights in LDS for the whole launch
                const int adv = ch * (9 * 16 * WROW * 2);      // (range check: voffset alone, as for the plane strides of the planar kernel)
#pragma unroll
                for (int k = 0; k < WK; ++k) {
                    const int inst = wvu + NW * k;
                    if (inst < WI && inst * 64 + lane < W16)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(Ws + inst * 512), 16, (int)wvoff[k], adv, 0, 0);
                }
                w_have = ch;
            }
            if (ch == 0 && tsk < 3) MTBC_TS(p, 4 * tsk + 1);
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (ch == 0 && tsk < 3) MTBC_TS(p, 4 * tsk + 2);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int toff = (tap / 3) * HC + tap % 3;
                typename T::frag a[MT], b[4];
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    a[m] = *reinterpret_cast<const typename T::frag*>(Ws + ((m * 9 + tap) * 16 + j) * WROW + 8 * (kg ^ ((j >> 1) & 3)));
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    b[g] = *reinterpret_cast<const typename T::frag*>(xlane + (c8_goff<GEO>(g, HC) + toff) * 8);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if constexpr (O8 != 0) acc[m][g] = T::mfma(a[m], b[g], acc[m][g]);      // rows = channels, cols = pixels
                        else acc[m][g] = T::mfma(b[g], a[m], acc[m][g]);                   // rows = pixels, cols = channels
                    }
            }
        }
        if (tsk < 3) MTBC_TS(p, 4 * tsk + 3);
        if (n0 + G::IMG <= p.N && y0 + TH <= p.H && x0 + G::TW <= p.W) c8_epilogue<MT, GEO, F16, NW, O8, true>(p, acc, seg_out, bias_s, n0, y0, x0, tx, ty, mt0, wvu, j, kg, HW, zpre);
        else c8_epilogue<MT, GEO, F16, NW, O8, false>(p, acc, seg_out, bias_s, n0, y0, x0, tx, ty, mt0, wvu, j, kg, HW, zpre);
        if (tsk < 3) MTBC_TS(p, 12 + tsk);
        ++tsk;
    }
}

// ------------------------------------------------------------------ deep levels: the same igemm with an LDS ring
// Maps <= 32 x 32 give a launch at most two blocks per CU (few pixel tiles, K = 9 x 96 .. 1152): resident blocks cannot hide each
// other's DMA latency, and the single-buffer kernel spends ~3 us per 32-channel chunk of which 0.7 are MFMAs.  Same LDS images,
// fragment reads, MFMA order and epilogues as conv3x3_igemm_c8_kernel -- bit-identical results -- but R slots of {X chunk, W chunk}
// (one block per CU, 147 KB for R = 3): the DMA of chunk c + R - 1 is issued before the MFMAs of chunk c, a COUNTED vmcnt waits for
// chunk c only, one raw barrier per chunk (it publishes chunk c and retires the slot chunk c + R - 1 overwrites).  Every wave
// issues the same number of DMA instructions per chunk (surplus weight instructions read out of range into a pad), so the count
// is a compile-time constant.
// LDS-DMA from inline assembly (the wait the compiler puts behind the builtin is described at conv3x3_wgrad_c8w_kernel below)
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 dma_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    return (i32x4){(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p; }
// 64 lanes x 16 B: lane l's piece lands at LDS byte address `lds` + 16 l; out-of-range `voff` -> zeros
__device__ __forceinline__ void dma16(const i32x4 rsrc, unsigned lds, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds), "v"(voff), "s"(rsrc) : "memory", "m0");
}
// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <typename F, int... I> __device__ __forceinline__ void sfor_c_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void sfor_c(F&& f) { sfor_c_impl(f, std::make_integer_sequence<int, N>{}); }
template <int MT, int GEO> struct RingGeo {
    using G = GeoLP<GEO>;
    static constexpr int HR = G::TH + 2, HC = G::TW + 2, HP = HR * HC, HPP = (HP + 15) / 16 * 16;
    static constexpr int XB = 4 * HPP * 8, WB = MT * 9 * 16 * WROW, PAD = 512;      // 16-bit elements
    static constexpr int XQ = (HPP + 63) / 64, W16 = WB / 8, WI = (W16 + 63) / 64, WPW = (WI + 3) / 4;
    static constexpr int CNT = XQ + WPW;                                             // vector-memory instructions per wave and chunk
    static constexpr int SLOT = XB + WB + PAD;
};
template <int MT, int GEO, bool F16, int O8, int R>
__global__ __launch_bounds__(256, 1) void conv3x3_igemm_c8_ring_kernel(const ConvP p) {
    using G = GeoLP<GEO>;
    using T = LP<F16>;
    using RG = RingGeo<MT, GEO>;
    static_assert(GEO != 2 && (O8 == 0 || O8 == 1 || O8 == 3) && R >= 2 && R <= 3, "wide / 16-wide maps, plain epilogues");
    constexpr int NW = 4, TH = G::TH, HR = RG::HR, HC = RG::HC, HP = RG::HP, HPP = RG::HPP, XB = RG::XB, WB = RG::WB, XQ = RG::XQ;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    SegL* seg_in = reinterpret_cast<SegL*>(smem16 + R * RG::SLOT);
    SegL* seg_out = seg_in + MTBC_MAX_SEGS;
    float* bias_s = reinterpret_cast<float*>(seg_out + MTBC_MAX_SEGS);      // MT*16 floats
    unsigned long long* xbase_s = reinterpret_cast<unsigned long long*>(bias_s + MT * 16);      // [chunk][channel group]: base of the group's pieces, this tile's image (0 = absent)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = p.H * p.W;
    const int mt0 = blockIdx.y * MT;
    MTBC_TS(p, 0);
    // the bias: requested now, written to LDS in front of the first tile's barrier (behind a __syncthreads() here its whole memory round trip
    // was the kernel's first microsecond)
    float bias_v = 0.f;
    if (tid < MT * 16) { const int co = mt0 * 16 + tid; if (p.bias && co < p.Cout) bias_v = p.bias[co]; }
    segl_fill(seg_in, p.in);
    segl_fill(seg_out, p.out);
    lds_barrier();
    MTBC_TS(p, 1);

    const int nchunks = (p.Cin + LPKC - 1) / LPKC;
    const int j = lane & 15, kg = lane >> 4;
    const int bpix0 = (GEO == 0 ? 2 * wv : 4 * wv) * HC + j;      // group g / tap t at compile-time offsets (c8_goff), as in the kernel above
    // ---- DMA issue (round 3, late).  Phase timestamps of this kernel (MTBC_RING_TS in the probes build; 384 -> 384 @16 x 16: 33 us) showed
    // 2.14 us per chunk of which 1.33 are the tap loop and 0.8 the ISSUE of the 13 DMA instructions of chunk c + 2: with one wave per SIMD
    // the segment lookup, 64-bit address arithmetic, descriptor words and per-instruction index divisions in front of the MFMAs are all
    // exposed, and so is the 60 - 185 cycles a `buffer_load ... lds` costs the wave that issues it.  Now: what does not depend on the chunk is
    // computed once (weight offsets per lane: the chunk only moves the descriptor's base), the X base of every (chunk, group) of the tile
    // comes out of a small LDS table filled by 4 nchunks threads, and the 13 instructions go out ONE AT A TIME between the MFMA groups of the
    // tap loop (inline assembly: the compiler neither waits for them nor moves them), under the matrix pipe: 1.78 us per chunk.  (Four
    // producer waves beside four consumer waves -- tools/experiments/ring_producer_consumer.patch -- reach 1.61 us per chunk and lose it
    // again in the prologue of a 512-thread block: 28.2 against 28.8 us for 384 -> 384, the step 0.04 ms slower than this form.)
    constexpr int WPW = RG::WPW, CNT = RG::CNT;
    const unsigned lds0 = lds_addr(smem16);
    const unsigned wbytes = (unsigned)((size_t)p.mtiles * nchunks * (9 * 16 * WROW) * 2);
    unsigned wvoff[WPW];
#pragma unroll
    for (int k = 0; k < WPW; ++k) {
        const int inst = wvu + NW * k, idx = inst * 64 + lane;
        const int mt = idx / (9 * 16 * WROW / 8), r = idx % (9 * 16 * WROW / 8);
        const bool ok = inst < RG::WI && idx < RG::W16 && (mt0 + mt) < p.mtiles;
        wvoff[k] = ok ? (unsigned)(((mt0 + mt) * nchunks * (9 * 16 * WROW / 8) + r) * 16) : 0xfffffff0u;      // + chunk * 9216 through the descriptor base
    }
    static_assert(RG::W16 % 64 == 0, "whole weight instructions: every wave issues CNT instructions per chunk");
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        int t = tile;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y; t /= p.tiles_y;
        const int n0 = t, x0 = tx * G::TW, y0 = ty * TH;
        unsigned pixo[XQ];
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int hp = lane + 64 * q;
            const int row = hp / HC, col = hp % HC;
            const int y = y0 + row - 1, x = x0 + col - 1;
            const bool ok = hp < HP && y >= 0 && y < p.H && x >= 0 && x < p.W;
            pixo[q] = ok ? 16u * (unsigned)(y * p.W + x) : 0xfffffff0u;
        }
        f32x4 acc[MT][4];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[m][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (tid < 4 * nchunks) {                  // (nchunks <= 64: the launcher's condition)
            const int c0 = (tid >> 2) * LPKC + 8 * (tid & 3);
            unsigned long long b = 0;
            if (c0 < p.Cin) {
                const SegL sr = segl_ref(seg_in, c0);
                b = reinterpret_cast<unsigned long long>(reinterpret_cast<unsigned short*>(sr.ptr) + ((size_t)n0 * sr.bs + (size_t)(c0 - sr.cb) * HW));
            }
            xbase_s[tid] = b;
        }
        if (tid < MT * 16) bias_s[tid] = bias_v;
        lds_barrier();                            // the table (and the bias) are there; the previous tile's fragments are consumed (its stores were waited for below)
        // descriptors of chunk `c` for this wave: X = its channel group's pieces of the tile's image, W = the packed image moved on by c chunks
        auto chunk_rsrc = [&](int c, i32x4& xr, i32x4& wr) {
            const unsigned long long b = xbase_s[4 * c + wvu];
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
            xr = (i32x4){(int)lo, (int)(hi & 0xffffu), (lo | hi) ? (int)((unsigned)HW * 16u) : 0, 0x00020000};
            const unsigned adv = (unsigned)c * (unsigned)(9 * 16 * WROW * 2);
            wr = dma_rsrc(reinterpret_cast<const char*>(p.wp) + adv, wbytes - adv);
        };
        // instruction i of the CNT a wave issues per chunk: the first XQ bring halo pixels 64 i .. of its channel group, the rest its share of W
        auto piece = [&](auto i_, int sl, const i32x4& xr, const i32x4& wr) {
            constexpr int i = decltype(i_)::value;
            const unsigned slot0 = lds0 + 2u * (unsigned)(sl * RG::SLOT);
            if constexpr (i < XQ) {
                if (lane + 64 * i < HPP) dma16(xr, slot0 + 2u * (unsigned)((wvu * HPP + 64 * i) * 8), pixo[i]);
            } else {
                constexpr int k = i - XQ;
                const int inst = wvu + NW * k;      // a wave without a real instruction left still issues one (zeros into the slot's pad)
                dma16(wr, slot0 + 2u * (unsigned)(XB + (inst < RG::WI ? inst * 512 : WB)), wvoff[k]);
            }
        };
#pragma unroll
        for (int s0 = 0; s0 < R - 1; ++s0)
            if (s0 < nchunks) {
                i32x4 xr, wr;
                chunk_rsrc(s0, xr, wr);
                sfor_c<CNT>([&](auto I) { piece(I, s0, xr, wr); });
            }
        MTBC_TS(p, 2);
        for (int ch = 0; ch < nchunks; ++ch) {
            // chunks ch + 1 .. ch + R - 2 may stay in flight
            const int rem = min(R - 2, nchunks - 1 - ch);
            if (rem >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RG::CNT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // chunk ch is in LDS for everybody; everybody is done with chunk ch - 1
            if (ch == 0) MTBC_TS(p, 3);
            if (ch == 1) MTBC_TS(p, 4);
            if (ch == nchunks - 1) MTBC_TS(p, 5);
            const bool more = ch + R - 1 < nchunks;      // (uniform) chunk ch + R - 1 goes out under this chunk's MFMAs
            const int sl_next = (ch + R - 1) % R;
            i32x4 xr_n = (i32x4){0, 0, 0, 0}, wr_n = xr_n;
            if (more) chunk_rsrc(ch + R - 1, xr_n, wr_n);
            const unsigned short* Xs = smem16 + (ch % R) * RG::SLOT;
            const unsigned short* Ws = Xs + XB;
            // ONE wave per SIMD: nobody else covers a fragment read's latency.  Left to the scheduler the taps ran as `2 reads, wait for
            // the first, 4 MFMAs, ...` (ISA: `r2 [lgkmcnt(1)] M4 r2 [lgkmcnt(1)] M4`), the matrix pipe a third busy; here the fragments of
            // tap t + 1 are read into a second register set BEFORE the 4 MT MFMAs of tap t (56 VGPRs of fragments; the block has 512).
            typename T::frag fa[2][MT], fb[2][4];
            auto load_tap = [&](int tap, int set) {
                const int toff = (tap / 3) * HC + tap % 3;
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    fa[set][m] = *reinterpret_cast<const typename T::frag*>(Ws + ((m * 9 + tap) * 16 + j) * WROW + 8 * (kg ^ ((j >> 1) & 3)));
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    fb[set][g] = *reinterpret_cast<const typename T::frag*>(Xs + (kg * HPP + bpix0) * 8 + (c8_goff<GEO>(g, HC) + toff) * 8);
            };
            load_tap(0, 0);
            constexpr int PPS = (CNT + 9 * MT - 1) / (9 * MT);      // DMA instructions per MFMA group (1 for every instantiation in use)
            sfor_c<9>([&](auto TAP) {
                constexpr int tap = decltype(TAP)::value, set = tap & 1;
                if (tap + 1 < 9) load_tap(tap + 1, set ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                sfor_c<MT>([&](auto M) {
                    constexpr int m = decltype(M)::value;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if constexpr (O8 != 0) acc[m][g] = T::mfma(fa[set][m], fb[set][g], acc[m][g]);
                        else acc[m][g] = T::mfma(fb[set][g], fa[set][m], acc[m][g]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (more)
                        sfor_c<PPS>([&](auto U) {
                            constexpr int i = (tap * MT + m) * PPS + decltype(U)::value;
                            if constexpr (i < CNT) piece(std::integral_constant<int, i>{}, sl_next, xr_n, wr_n);
                        });
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        }
        int zpre = 0;
        MTBC_TS(p, 6);
        if (y0 + TH <= p.H && x0 + G::TW <= p.W) c8_epilogue<MT, GEO, F16, NW, O8, true>(p, acc, seg_out, bias_s, n0, y0, x0, tx, ty, mt0, wvu, j, kg, HW, zpre);
        else c8_epilogue<MT, GEO, F16, NW, O8, false>(p, acc, seg_out, bias_s, n0, y0, x0, tx, ty, mt0, wvu, j, kg, HW, zpre);
        MTBC_TS(p, 7);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the stores: the next tile counts its DMA instructions from zero
        MTBC_TS(p, 8);
    }
}

// ------------------------------------------------------------------ wgrad (MFMA, split-K)
template <int GEO> struct WGeo;
template <> struct WGeo<0> { static constexpr int TH = 4, TW = 32, ROWS = 6, LW = 40, IMG = 1, IMGS = 240, PSX = 258; };
template <> struct WGeo<1> { static constexpr int TH = 8, TW = 16, ROWS = 10, LW = 24, IMG = 1, IMGS = 240, PSX = 258; };
template <> struct WGeo<2> { static constexpr int TH = 8, TW = 8, ROWS = 10, LW = 16, IMG = 2, IMGS = 160, PSX = 322; };
constexpr int PSZ = 130;

struct WgP {
    int N, H, W, Cin, Cout;
    SegTable in;
    const float* dz;
    float* partial;            // [nsplit][Cout][Cin][9]
    int tiles_x, tiles_y, total_tiles, tiles_per_split;
    int ciblocks;
    int dbg;      // timing probes (env MTBC_DBG): 1 = no global loads, 4 = no LDS commits
#ifdef MTBC_PROBES
    unsigned long long* ts;    // phase timestamps (MTBC_WG_TS=1): [block][16] ticks of the 100 MHz clock
#endif
};

// COT = output-channel tiles per block (2 or 3): 128*COT threads, wave w = (co-tile w/2, half w%2 of the column tiles).  COT = 3 serves
// Cout = 48 (U-Net++ level 1) without padding the second 32-channel block half empty.
// Columns (round 3): the N dimension of the MFMA is the (tap, ci) PAIR, 9 * cib columns per block in tiles of 16, instead of one 16-channel
// tile per tap.  With 32 input channels per block that is the same 18 tiles; with 24 (PACK: every level-0 / level-1 conv of the U-Net++ has
// Cin = 24 k, which blocks of 32 cut into 32 + 32 + 8 or pad 24 -> 32) it is 13.5 -> 14 tiles, 7 per wave instead of 9: -22 % MFMAs on
// 60 % of the weight-gradient FLOPs.  A lane's column decides its channel AND its tap, so its LDS offset is per lane and per tile
// (`boff`); a tile that straddles two taps reads 2-way conflicted (the LDS is not what bounds this kernel).
template <int GEO, int COT, bool PACK>
__global__ __launch_bounds__(128 * COT, 3) void conv3x3_wgrad_mfma_kernel(const WgP p) {
    using G = WGeo<GEO>;
    constexpr int XS = 32 * G::PSX;
    constexpr int XF4_PER_CH = G::IMG * G::ROWS * G::LW / 4;      // float4 per channel of the halo tile
    constexpr int XSLOTS = (XF4_PER_CH + 7) / 8;                   // 8 threads share one channel
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;
    float* Zs = smem + XS;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int HW = p.H * p.W;
    constexpr int CIB = PACK ? 24 : 32, NTW = PACK ? 7 : 9;          // input channels per block, column tiles per wave
    const int co0 = (blockIdx.y / p.ciblocks) * (16 * COT), ci0 = (blockIdx.y % p.ciblocks) * CIB;
    const int split = blockIdx.x;
    const int t_begin = split * p.tiles_per_split;
    const int t_end = min(p.total_tiles, t_begin + p.tiles_per_split);

    // ---- staging role: thread -> (channel ch = tid/8, lane-in-channel q = tid%8), fixed for the block
    const int ch = tid >> 3, q = tid & 7;                 // ch < 16*COT; only ch < 32 stage X
    const int my_ci = ci0 + ch, my_co = co0 + ch;
    const bool ci_ok = ch < CIB && my_ci < p.Cin, co_ok = my_co < p.Cout;
    const SegRef sr = seg_ref(p.in, ci_ok ? my_ci : 0);
    const float* xplane = sr.ptr + (size_t)((ci_ok ? my_ci : 0) - sr.cb) * HW;       // + n*bs + y*W + x
    const long long xbs = sr.bs;
    const float* zplane = p.dz + (size_t)(co_ok ? my_co : 0) * HW;                    // + n*Cout*HW + y*W + x
    float4 xr[XSLOTS], zr[4];
    unsigned live = 0;                      // bit s: slot s of the prefetched tile holds real data
    // Slot offsets relative to the tile origin are tile-invariant: computed once.  A tile that does not touch the
    // image border then costs one 64-bit add per operand and one address instruction per load (the generic path
    // below spends ~25 VALU instructions per slot on div/mod, bounds and 64-bit index math).
    unsigned slots_ok = 0;
    if (G::IMG == 1) {
#pragma unroll
        for (int s = 0; s < XSLOTS; ++s) slots_ok |= (q + 8 * s < XF4_PER_CH && ci_ok) ? (1u << s) : 0u;
#pragma unroll
        for (int s = 0; s < 4; ++s) slots_ok |= co_ok ? (1u << (XSLOTS + s)) : 0u;
    }

    auto prefetch = [&](int tile) {
        int t = tile;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y; t /= p.tiles_y;
        const int n0 = t * G::IMG, x0 = tx * G::TW, y0 = ty * G::TH;
        if (G::IMG == 1) {
            const bool interior = y0 >= 1 && y0 + G::TH + 1 <= p.H && x0 >= 4 && x0 + G::TW + 4 <= p.W && !MTBC_DBG_BIT(p, 1);   // uniform
            if (interior) {
                live = slots_ok;
                // the slot offsets are recomputed per tile from an opaque copy of q: kept live across the MFMA loop they were spilled
                // (the packed-column variant has 7 per-lane column offsets more), and every scratch reload between these loads waits
                // for the loads already issued
                int qq = q;
                asm volatile("" : "+v"(qq));
                // ONE base pointer per operand and an integer offset that is 0 for an absent slot.  (Round 4, found with the phase stamps + the ISA:
                // written as `ok ? tile_pointer + offset : plane_pointer`, hipcc turned the pointer select into control flow -- a load from
                // the plane for every lane, then, under an exec mask, the address computed INTO the first load's destination registers and
                // a second load: two `s_waitcnt vmcnt(0)` in the middle of every prefetch, i.e. two exposed memory round trips per tile,
                // 6.4 of a tile's 16.2 us at 24 -> 24 @256 x 256, profiles/r04_f32_wgrad_phase_stamps.txt.)
                const long long xo = (long long)((size_t)n0 * xbs) + y0 * p.W + x0, zo = (long long)((size_t)n0 * p.Cout * HW) + y0 * p.W + x0;
#pragma unroll
                for (int s = 0; s < XSLOTS; ++s) {
                    const int f = qq + 8 * s, row = f / (G::LW / 4), c4 = f % (G::LW / 4);
                    long long off = ((slots_ok >> s) & 1u) ? xo + ((row - 1) * p.W + (c4 * 4 - 4)) : 0;
                    asm volatile("" : "+v"(off));          // opaque: the select stays a select (hipcc otherwise splits the LOAD over the two arms)
                    xr[s] = *reinterpret_cast<const float4*>(xplane + off);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int px = (qq + 8 * s) * 4;
                    long long off = co_ok ? zo + ((px / G::TW) * p.W + px % G::TW) : 0;
                    asm volatile("" : "+v"(off));
                    zr[s] = *reinterpret_cast<const float4*>(zplane + off);
                }
                return;
            }
        }
        live = 0;
#pragma unroll
        for (int s = 0; s < XSLOTS; ++s) {
            const int f = q + 8 * s;
            const int img = f / (G::ROWS * G::LW / 4), rem = f % (G::ROWS * G::LW / 4);
            const int row = rem / (G::LW / 4), c4 = rem % (G::LW / 4);
            const int y = y0 + row - 1, x = x0 - 4 + c4 * 4, n = n0 + img;
            const bool ok = f < XF4_PER_CH && ci_ok && n < p.N && y >= 0 && y < p.H && x >= 0 && x < p.W;
            xr[s] = *reinterpret_cast<const float4*>(xplane + (ok ? (size_t)n * xbs + y * p.W + x : 0));
            live |= ok ? (1u << s) : 0u;          // zeroed in commit(): touching xr here would wait for the load
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int px = (q + 8 * s) * 4;
            int n, y, x;
            if (GEO == 0) { n = n0; y = y0 + px / 32; x = x0 + px % 32; }
            else if (GEO == 1) { n = n0; y = y0 + px / 16; x = x0 + px % 16; }
            else { n = n0 + px / 64; y = y0 + (px % 64) / 8; x = x0 + px % 8; }
            const bool ok = co_ok && n < p.N && y < p.H && x < p.W;
            zr[s] = *reinterpret_cast<const float4*>(zplane + (ok ? (size_t)n * p.Cout * HW + y * p.W + x : 0));
            live |= ok ? (1u << (XSLOTS + s)) : 0u;
        }
    };
    auto commit = [&]() {       // registers -> LDS (8-byte aligned rows: stride PSX/PSZ == 2 mod 32)
#pragma unroll
        for (int s = 0; s < XSLOTS; ++s) {        // component-wise: a float4 ?: takes the array's address (scratch)
            const bool l = (live >> s) & 1u;
            xr[s].x = l ? xr[s].x : 0.f; xr[s].y = l ? xr[s].y : 0.f; xr[s].z = l ? xr[s].z : 0.f; xr[s].w = l ? xr[s].w : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool l = (live >> (XSLOTS + s)) & 1u;
            zr[s].x = l ? zr[s].x : 0.f; zr[s].y = l ? zr[s].y : 0.f; zr[s].z = l ? zr[s].z : 0.f; zr[s].w = l ? zr[s].w : 0.f;
        }
#pragma unroll
        for (int s = 0; s < XSLOTS; ++s) {
            const int f = q + 8 * s;
            if (f < XF4_PER_CH && ch < CIB) {
                const int img = f / (G::ROWS * G::LW / 4), rem = f % (G::ROWS * G::LW / 4);
                const int row = rem / (G::LW / 4), c4 = rem % (G::LW / 4);
                float* d = Xs + ch * G::PSX + img * G::IMGS + row * G::LW + c4 * 4;
                *reinterpret_cast<float2*>(d) = make_float2(xr[s].x, xr[s].y);
                *reinterpret_cast<float2*>(d + 2) = make_float2(xr[s].z, xr[s].w);
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float* d = Zs + ch * PSZ + (q + 8 * s) * 4;
            *reinterpret_cast<float2*>(d) = make_float2(zr[s].x, zr[s].y);
            *reinterpret_cast<float2*>(d + 2) = make_float2(zr[s].z, zr[s].w);
        }
    };

    const int j = lane & 15, kk = lane >> 4;
    const int aBase = ((wv >> 1) * 16 + j) * PSZ + kk;
    // column tile i of this wave = tile 2 i + (wv & 1); lane j's column c -> (tap, channel); columns past 9 * CIB (the second half of the
    // 14th tile of a 24-channel block) read a valid address and are never stored
    // (32-channel blocks: tile i IS tap i of the wave's 16-channel tile -- one base + compile-time tap offsets, as before)
    const int bBase = ((wv & 1) * 16 + j) * G::PSX + kk + 3;
    int boff[PACK ? NTW : 1];
    if constexpr (PACK) {
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const int c = 16 * (2 * i + (wv & 1)) + j;
            const int tap = min(c / CIB, 8), cc = min(c - (c / CIB) * CIB, CIB - 1);
            boff[i] = cc * G::PSX + kk + 3 + (tap / 3) * G::LW + tap % 3;
        }
    }

    f32x4 acc[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < XSLOTS; ++s) xr[s] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int s = 0; s < 4; ++s) zr[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    MTBC_TS(p, 0);
    if (t_begin < t_end && !MTBC_DBG_BIT(p, 1)) prefetch(t_begin);
    for (int tile = t_begin; tile < t_end; ++tile) {
        __syncthreads();                       // everyone is done reading the previous tile
        if (tile - t_begin < 3) MTBC_TS(p, 1 + 4 * (tile - t_begin));          // previous tile's MFMAs done by every wave
        if (!MTBC_DBG_BIT(p, 4)) commit();
        if (tile - t_begin < 3) MTBC_TS(p, 2 + 4 * (tile - t_begin));          // this wave's loads have landed and are written to LDS
        __syncthreads();
        if (tile - t_begin < 3) MTBC_TS(p, 3 + 4 * (tile - t_begin));          // ... everybody's
        if (tile + 1 < t_end && !MTBC_DBG_BIT(p, 1)) prefetch(tile + 1);   // in flight under the MFMAs below
        if (tile - t_begin < 3) MTBC_TS(p, 4 + 4 * (tile - t_begin));          // next tile's loads issued
        // 32 k-steps of 4 pixels; fragments of step s+1 are read before the MFMAs of step s (bounded live ranges:
        // without the sched_barriers hipcc hoists all 320 LDS reads and needs >180 VGPRs, i.e. 2 waves/SIMD)
        float fa[2], fb[2][NTW];
        auto read_step = [&](int p4, int slot) {
            const int px = p4 * 4;
            int xoff;
            if (GEO == 0) xoff = (px / 32) * G::LW + px % 32;
            else if (GEO == 1) xoff = (px / 16) * G::LW + px % 16;
            else xoff = (px / 64) * G::IMGS + ((px % 64) / 8) * G::LW + px % 8;
            fa[slot] = Zs[aBase + px];
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                if constexpr (PACK) fb[slot][i] = Xs[boff[i] + xoff];
                else fb[slot][i] = Xs[bBase + xoff + (i / 3) * G::LW + i % 3];
            }
        };
        read_step(0, 0);
#pragma unroll
        for (int p4 = 0; p4 < 32; ++p4) {
            if (p4 + 1 < 32) read_step(p4 + 1, (p4 + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NTW; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[p4 & 1], fb[p4 & 1][i], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    MTBC_TS(p, 13);
    // partial[split][co][ci][tap]; D row = co (kk*4+r), column = (tap, ci) of tile 2 i + (wv & 1)
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int c = PACK ? 16 * (2 * i + (wv & 1)) + j : i * CIB + (wv & 1) * 16 + j;      // (32-channel blocks: tile i = tap i)
        const int tap = c / CIB, ci = ci0 + c - tap * CIB;
        if (tap >= 9 || ci >= p.Cin) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + (wv >> 1) * 16 + kk * 4 + r;
            if (co < p.Cout) p.partial[(((size_t)split * p.Cout + co) * p.Cin + ci) * 9 + tap] = acc[i][r];
        }
    }
    MTBC_TS(p, 14);
}

// ------------------------------------------------------------------ wgrad on the 16-bit MFMA (optional compute mode)
// Same block structure as conv3x3_wgrad_mfma_kernel, but the contraction runs on v_mfma_f32_16x16x32_{bf16,f16}
// (fp32 accumulate).  K = pixels is the contiguous dimension of an NCHW plane, so a lane's 8 K-values are 8 consecutive
// pixels: two aligned ds_read_b128 for dZ, and one aligned 16-float window per kernel row of X from which the three
// horizontally shifted taps are cut in registers (v_cvt_pk_* packs pairs, so the shift costs nothing).  fp32 stays the
// storage type in HBM and LDS; only the MFMA operands are rounded.
template <int GEO> struct WGeoLP;      // 16-byte aligned channel strides, stride/4 odd (b128 reads spread over slots)
template <> struct WGeoLP<0> { static constexpr int PSX = 244; };
template <> struct WGeoLP<1> { static constexpr int PSX = 244; };
template <> struct WGeoLP<2> { static constexpr int PSX = 324; };
constexpr int PSZ_LP = 132;

template <int GEO, int COT, bool F16>
__global__ __launch_bounds__(128 * COT, 3) void conv3x3_wgrad_lp_kernel(const WgP p) {
    using G = WGeo<GEO>;
    using T = LP<F16>;
    constexpr int PSX = WGeoLP<GEO>::PSX;
    constexpr int XS = 32 * PSX;
    constexpr int XF4_PER_CH = G::IMG * G::ROWS * G::LW / 4;
    constexpr int XSLOTS = (XF4_PER_CH + 7) / 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;
    float* Zs = smem + XS;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int HW = p.H * p.W;
    const int co0 = (blockIdx.y / p.ciblocks) * (16 * COT), ci0 = (blockIdx.y % p.ciblocks) * 32;
    const int split = blockIdx.x;
    const int t_begin = split * p.tiles_per_split;
    const int t_end = min(p.total_tiles, t_begin + p.tiles_per_split);

    const int ch = tid >> 3, q = tid & 7;
    const int my_ci = ci0 + ch, my_co = co0 + ch;
    const bool ci_ok = ch < 32 && my_ci < p.Cin, co_ok = my_co < p.Cout;
    const SegRef sr = seg_ref(p.in, ci_ok ? my_ci : 0);
    const float* xplane = sr.ptr + (size_t)((ci_ok ? my_ci : 0) - sr.cb) * HW;
    const long long xbs = sr.bs;
    const float* zplane = p.dz + (size_t)(co_ok ? my_co : 0) * HW;
    float4 xr[XSLOTS], zr[4];
    unsigned live = 0;                      // bit s: slot s of the prefetched tile holds real data
#pragma unroll
    for (int s = 0; s < XSLOTS; ++s) xr[s] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int s = 0; s < 4; ++s) zr[s] = make_float4(0.f, 0.f, 0.f, 0.f);

    auto prefetch = [&](int tile) {
        int t = tile;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y; t /= p.tiles_y;
        const int n0 = t * G::IMG, x0 = tx * G::TW, y0 = ty * G::TH;
        live = 0;
#pragma unroll
        for (int s = 0; s < XSLOTS; ++s) {
            const int f = q + 8 * s;
            const int img = f / (G::ROWS * G::LW / 4), rem = f % (G::ROWS * G::LW / 4);
            const int row = rem / (G::LW / 4), c4 = rem % (G::LW / 4);
            const int y = y0 + row - 1, x = x0 - 4 + c4 * 4, n = n0 + img;
            const bool ok = f < XF4_PER_CH && ci_ok && n < p.N && y >= 0 && y < p.H && x >= 0 && x < p.W;
            xr[s] = *reinterpret_cast<const float4*>(xplane + (ok ? (size_t)n * xbs + y * p.W + x : 0));
            live |= ok ? (1u << s) : 0u;          // zeroed in commit(): touching xr here would wait for the load
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int px = (q + 8 * s) * 4;
            int n, y, x;
            if (GEO == 0) { n = n0; y = y0 + px / 32; x = x0 + px % 32; }
            else if (GEO == 1) { n = n0; y = y0 + px / 16; x = x0 + px % 16; }
            else { n = n0 + px / 64; y = y0 + (px % 64) / 8; x = x0 + px % 8; }
            const bool ok = co_ok && n < p.N && y < p.H && x < p.W;
            zr[s] = *reinterpret_cast<const float4*>(zplane + (ok ? (size_t)n * p.Cout * HW + y * p.W + x : 0));
            live |= ok ? (1u << (XSLOTS + s)) : 0u;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int s = 0; s < XSLOTS; ++s) {        // component-wise: a float4 ?: takes the array's address (scratch)
            const bool l = (live >> s) & 1u;
            xr[s].x = l ? xr[s].x : 0.f; xr[s].y = l ? xr[s].y : 0.f; xr[s].z = l ? xr[s].z : 0.f; xr[s].w = l ? xr[s].w : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool l = (live >> (XSLOTS + s)) & 1u;
            zr[s].x = l ? zr[s].x : 0.f; zr[s].y = l ? zr[s].y : 0.f; zr[s].z = l ? zr[s].z : 0.f; zr[s].w = l ? zr[s].w : 0.f;
        }
#pragma unroll
        for (int s = 0; s < XSLOTS; ++s) {
            const int f = q + 8 * s;
            if (f < XF4_PER_CH && ch < 32) {
                const int img = f / (G::ROWS * G::LW / 4), rem = f % (G::ROWS * G::LW / 4);
                const int row = rem / (G::LW / 4), c4 = rem % (G::LW / 4);
                *reinterpret_cast<float4*>(Xs + ch * PSX + img * G::IMGS + row * G::LW + c4 * 4) = xr[s];
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) *reinterpret_cast<float4*>(Zs + ch * PSZ_LP + (q + 8 * s) * 4) = zr[s];
    };

    const int j = lane & 15, kg = lane >> 4;
    const int aBase = ((wv >> 1) * 16 + j) * PSZ_LP + 8 * kg;
    int bLane;
    if (GEO == 0) bLane = 8 * kg;
    else if (GEO == 1) bLane = (kg >> 1) * G::LW + 8 * (kg & 1);
    else bLane = kg * G::LW;
    const int bBase = ((wv & 1) * 16 + j) * PSX + bLane;

    f32x4 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (t_begin < t_end) prefetch(t_begin);
    for (int tile = t_begin; tile < t_end; ++tile) {
        __syncthreads();
        commit();
        __syncthreads();
        if (tile + 1 < t_end) prefetch(tile + 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {            // 32 pixels per step
            int xoff;
            if (GEO == 0) xoff = ks * G::LW;
            else if (GEO == 1) xoff = 2 * ks * G::LW;
            else xoff = (ks >> 1) * G::IMGS + 4 * (ks & 1) * G::LW;
            float fa[8];
            *reinterpret_cast<float4*>(fa) = *reinterpret_cast<const float4*>(Zs + aBase + 32 * ks);
            *reinterpret_cast<float4*>(fa + 4) = *reinterpret_cast<const float4*>(Zs + aBase + 32 * ks + 4);
            const typename T::frag a = T::pack(fa);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                float f[16];                        // aligned window: columns x .. x+15 of halo row (row + r)
#pragma unroll
                for (int v4 = 0; v4 < 4; ++v4)
                    *reinterpret_cast<float4*>(f + 4 * v4) = *reinterpret_cast<const float4*>(Xs + bBase + xoff + r * G::LW + 4 * v4);
#pragma unroll
                for (int s3 = 0; s3 < 3; ++s3)      // tap (r, s3): input pixel x + s3 - 1 -> halo column x + 3 + s3
                    acc[r * 3 + s3] = T::mfma(a, T::pack(f + 3 + s3), acc[r * 3 + s3]);
            }
        }
    }
    const int ci = ci0 + (wv & 1) * 16 + j;
    if (ci < p.Cin) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + (wv >> 1) * 16 + kg * 4 + r;
            if (co >= p.Cout) continue;
            float* d = p.partial + (((size_t)split * p.Cout + co) * p.Cin + ci) * 9;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) d[tap] = acc[tap][r];
        }
    }
}

// ------------------------------------------------------------------ wgrad on the 16-bit MFMA, 16-bit LDS image (wide maps)
// Second generation of conv3x3_wgrad_lp_kernel for maps >= 32 wide.  That kernel keeps fp32 in LDS and converts at
// fragment time: 4.4 v_cvt_pk per MFMA, every X value converted 3x (once per horizontal tap) by 2 waves -- it is
// VALU-bound.  Here the tile is converted ONCE when it is committed to LDS (16-bit image, half the LDS bytes, one
// ds_read_b128 per fragment) and the three horizontally shifted taps are cut from two aligned reads with
// v_alignbit_b32 (a 16-bit funnel shift: 8 VALU per kernel row instead of 12 conversions + packs).  A wave owns TWO
// output-channel tiles x one input-channel tile (the shifted X windows are shared by both), and the 4 waves are
// (ci tile 0/1) x (tile rows 0-1 / 2-3); the two row halves are summed through LDS once, at the end of the block.
//   LDS: Xs[32 ci][6 halo rows x 40] + Zs[32 co][128 + 16], 16-bit; channel strides == 16 mod 32 elements make every
//   ds_read_b128 fragment conflict-free (brute-forced over the b128 lane groups of MI355X_MICROARCH.md).
constexpr int W2_PSX = 240, W2_PSZ = 144;
template <bool F16>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_lp2_kernel(const WgP p) {
    using T = LP<F16>;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int TH = 4, TW = 32, ROWS = 6, LW = 40;
    constexpr int XF4 = ROWS * LW / 4;                    // 60 float4 per channel
    constexpr int XSLOTS = (XF4 + 7) / 8;                 // 8
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16w[];
    unsigned short* Xs = smem16w;
    unsigned short* Zs = smem16w + 32 * W2_PSX;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int HW = p.H * p.W;
    const int co0 = (blockIdx.y / p.ciblocks) * 32, ci0 = (blockIdx.y % p.ciblocks) * 32;
    const int split = blockIdx.x;
    const int t_begin = split * p.tiles_per_split;
    const int t_end = min(p.total_tiles, t_begin + p.tiles_per_split);

    // staging: thread = (channel tid/8, float4 slots q + 8s)
    const int ch = tid >> 3, q = tid & 7;
    const int my_ci = ci0 + ch, my_co = co0 + ch;
    const bool ci_ok = my_ci < p.Cin, co_ok = my_co < p.Cout;
    const SegRef sr = seg_ref(p.in, ci_ok ? my_ci : 0);
    const float* xplane = sr.ptr + (size_t)((ci_ok ? my_ci : 0) - sr.cb) * HW;
    const long long xbs = sr.bs;
    const float* zplane = p.dz + (size_t)(co_ok ? my_co : 0) * HW;
    float4 xr[XSLOTS], zr[4];
    unsigned live = 0;                      // bit s: slot s of the prefetched tile holds real data
    // Offsets of this thread's slots relative to the tile origin do not depend on the tile: computed once.  A tile
    // that does not touch the image border (73 % of them at 256x256) then costs one 64-bit add per operand and one
    // address instruction per load; only border tiles run the per-slot bounds checks.
    int xrel[XSLOTS], zrel[4];
    unsigned slots_ok = 0;                  // slots that exist for this thread (f < XF4, channel inside the tensor)
#pragma unroll
    for (int s = 0; s < XSLOTS; ++s) {
        const int f = q + 8 * s, row = f / (LW / 4), c4 = f % (LW / 4);
        xrel[s] = (row - 1) * p.W + (c4 * 4 - 4);
        slots_ok |= (f < XF4 && ci_ok) ? (1u << s) : 0u;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int px = (q + 8 * s) * 4;
        zrel[s] = (px / TW) * p.W + px % TW;
        slots_ok |= co_ok ? (1u << (XSLOTS + s)) : 0u;
    }
    auto prefetch = [&](int tile) {         // issue only: nothing may touch xr / zr before commit()
        int t = tile;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y; t /= p.tiles_y;
        const int n = t, x0 = tx * TW, y0 = ty * TH;
        // (one base pointer + an integer offset that is 0 for an absent slot: a SELECT OF POINTERS in front of a load becomes control flow with
        //  two loads into the same registers and a vmcnt(0) between them -- see conv3x3_wgrad_mfma_kernel)
        const long long xo = (long long)((size_t)n * xbs) + y0 * p.W + x0, zo = (long long)((size_t)n * p.Cout * HW) + y0 * p.W + x0;
        const bool interior = y0 >= 1 && y0 + TH + 1 <= p.H && x0 >= 4 && x0 + TW + 4 <= p.W && !MTBC_DBG_BIT(p, 1);   // uniform
        if (interior) {
            live = slots_ok;
#pragma unroll
            for (int s = 0; s < XSLOTS; ++s) xr[s] = *reinterpret_cast<const float4*>(xplane + (((slots_ok >> s) & 1u) ? xo + xrel[s] : 0));
#pragma unroll
            for (int s = 0; s < 4; ++s) zr[s] = *reinterpret_cast<const float4*>(zplane + (co_ok ? zo + zrel[s] : 0));
            return;
        }
        live = 0;
#pragma unroll
        for (int s = 0; s < XSLOTS; ++s) {
            const int f = q + 8 * s;
            const int row = f / (LW / 4), c4 = f % (LW / 4);
            const int y = y0 + row - 1, x = x0 - 4 + c4 * 4;
            const bool ok = f < XF4 && ci_ok && y >= 0 && y < p.H && x >= 0 && x < p.W && !MTBC_DBG_BIT(p, 1);
            xr[s] = *reinterpret_cast<const float4*>(xplane + (ok ? xo + xrel[s] : 0));
            live |= ok ? (1u << s) : 0u;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int px = (q + 8 * s) * 4;
            const int y = y0 + px / TW, x = x0 + px % TW;
            const bool ok = co_ok && y < p.H && x < p.W && !MTBC_DBG_BIT(p, 1);
            zr[s] = *reinterpret_cast<const float4*>(zplane + (ok ? zo + zrel[s] : 0));
            live |= ok ? (1u << (XSLOTS + s)) : 0u;
        }
    };
    auto cvt4 = [&](const float4& v, bool l) {            // 4 floats -> 4 x 16 bit (zeros if the slot is dead)
        float f[8] = {l ? v.x : 0.f, l ? v.y : 0.f, l ? v.z : 0.f, l ? v.w : 0.f, 0.f, 0.f, 0.f, 0.f};
        const typename T::frag h = T::pack(f);
        const u32x4 u = __builtin_bit_cast(u32x4, h);
        return make_uint2(u[0], u[1]);
    };
    auto commit = [&]() {
#pragma unroll
        for (int s = 0; s < XSLOTS; ++s) {
            const int f = q + 8 * s;
            if (f < XF4) *reinterpret_cast<uint2*>(Xs + ch * W2_PSX + f * 4) = cvt4(xr[s], (live >> s) & 1u);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
            *reinterpret_cast<uint2*>(Zs + ch * W2_PSZ + (q + 8 * s) * 4) = cvt4(zr[s], (live >> (XSLOTS + s)) & 1u);
    };

    const int j = lane & 15, kg = lane >> 4;
    const int it = wv & 1, kh = wv >> 1;                  // input-channel tile, row half of the 4-row tile
    const unsigned short* zb = Zs + j * W2_PSZ + 8 * kg;                        // + ct*16*PSZ + row*32
    const unsigned short* xb = Xs + (it * 16 + j) * W2_PSX + 8 * kg;            // + halo_row*40

    f32x4 acc[2][9];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 9; ++i) acc[c][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (t_begin < t_end) prefetch(t_begin);
    for (int tile = t_begin; tile < t_end; ++tile) {
        __syncthreads();                       // everyone is done reading the previous tile
        if (!MTBC_DBG_BIT(p, 4)) commit();
        __syncthreads();
        if (tile + 1 < t_end) prefetch(tile + 1);          // in flight under the MFMAs below
#pragma unroll
        for (int ksl = 0; ksl < 2; ++ksl) {
            const int row = 2 * kh + ksl;                  // 32 pixels of one tile row per step
            typename T::frag a[2];
#pragma unroll
            for (int c = 0; c < 2; ++c)
                a[c] = *reinterpret_cast<const typename T::frag*>(zb + c * 16 * W2_PSZ + row * TW);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                // elements e0..e15 of halo row (row + r) from column 8kg: pixel x0 + 8kg - 4 + e
                const u32x4 lo = *reinterpret_cast<const u32x4*>(xb + (row + r) * LW);
                const u32x4 hi = *reinterpret_cast<const u32x4*>(xb + (row + r) * LW + 8);
                const unsigned d[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                u32x4 w0, w1, w2;                          // taps s3 = 0, 1, 2 start at elements 3, 4, 5
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    w0[k] = __builtin_amdgcn_alignbit(d[2 + k], d[1 + k], 16);
                    w1[k] = d[2 + k];
                    w2[k] = __builtin_amdgcn_alignbit(d[3 + k], d[2 + k], 16);
                }
                const typename T::frag b0 = __builtin_bit_cast(typename T::frag, w0);
                const typename T::frag b1 = __builtin_bit_cast(typename T::frag, w1);
                const typename T::frag b2 = __builtin_bit_cast(typename T::frag, w2);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    acc[c][r * 3 + 0] = T::mfma(a[c], b0, acc[c][r * 3 + 0]);
                    acc[c][r * 3 + 1] = T::mfma(a[c], b1, acc[c][r * 3 + 1]);
                    acc[c][r * 3 + 2] = T::mfma(a[c], b2, acc[c][r * 3 + 2]);
                }
            }
        }
    }
    // sum the two row halves (waves 2,3 -> waves 0,1) through LDS, then one partial per block
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(smem16w);        // [it][18][64] f32x4 = 36 KB
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
            float* d = p.partial + (((size_t)split * p.Cout + co) * p.Cin + ci) * 9;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) d[tap] = acc[c][tap][r] + red[(it * 18 + c * 9 + tap) * 64 + lane][r];
        }
}
constexpr size_t W2_LDS = 2 * 18 * 64 * sizeof(f32x4);      // the end-of-block reduction is the larger user (36 KB)

// ------------------------------------------------------------------ wgrad on 16-bit channel-blocked operands ("c8")
// Both operands are already stored as [n][C/8][H*W][8] in the MFMA's 16-bit type (see conv3x3_igemm_c8_kernel).  The
// contraction runs over PIXELS, so a lane's fragment (8 consecutive pixels of one channel) is the transpose of a stored
// piece (8 channels of one pixel): LDS-DMA copies the pieces into a [group][pixel][8 ch] image without touching a VGPR
// and `ds_read_b64_tr_b16` delivers 4 pixels x 16 channels column-major -- its 16 columns are two 8-channel groups that
// sit in different images (each lane of the instruction supplies its own row address).  Group strides of 204 / 132
// pieces (3264 / 2112 B = 192 / 64 mod 256) put the two groups, and the two 16-lane blocks of a 32-lane half 8 pixels
// apart, on disjoint banks.  The +-1 pixel taps: three 4-pixel blocks per halo row give elements e0..e11 from halo
// column 8kg; tap s uses e_s..e_(s+7) (s = 1 through v_alignbit).  Block structure as conv3x3_wgrad_lp2_kernel
// (32 co x 32 ci, 4x32-pixel tiles, waves = ci-tile x row-half); no staging registers -> 4 blocks per CU.
// dbias rides along on the matrix pipe: dz x ones on the ci-block-0 / ci-tile-0 waves.
// Measured stand-alone (tools/experiments/c8_wgrad_probe.hip): 144->24 @256x256 N=32 0.30 ms against 0.65 ms.
// Tile geometries (128 pixels either way = 4 K-steps of 32 pixels): GEO 0 = 4 rows x 32 columns, a K-step is one tile row
// and lane group kg owns columns 8kg..8kg+7; GEO 1 (maps <= 16 wide) = 8 rows x 16 columns, a K-step is two tile rows and
// kg owns row kg>>1, columns 8(kg&1)..+7 -- the 4x32 tile wasted half of every MFMA on the 16x16 levels.
template <int GEO> struct C8WGeo;
template <> struct C8WGeo<0> { static constexpr int TH = 4, TW = 32, HR = 6, LW = 34; };
template <> struct C8WGeo<1> { static constexpr int TH = 8, TW = 16, HR = 10, LW = 18; };
constexpr int C8W_ZG = 128 + 4;                  // 132 pieces per output-channel group (2112 B = 64 mod 256)
constexpr size_t C8W_LDS = (2 * 18 + 2) * 64 * sizeof(f32x4);      // the end-of-block reduction is the larger user (38 KB)
static_assert((4 * C8WGeo<0>::HR * C8WGeo<0>::LW + 16 + 4 * C8W_ZG) * 16 <= (int)C8W_LDS, "stage buffer must fit the reduction area");
static_assert((C8WGeo<0>::HR * C8WGeo<0>::LW * 16) % 256 == 192 && (C8WGeo<1>::HR * C8WGeo<1>::LW * 16) % 256 == 64, "group strides: 64/192 mod 256 B");
// ------------------------------------------------------------------ split-K partials reduced by the LAST ARRIVER, inside the weight-gradient kernel (round 4)
// Until round 3 every weight-gradient launch was followed by a reduction launch over its split-K partials (55 `splitk_reduce_k` launches,
// 0.41 ms per step).  Now the block that stores the LAST partial of a group of rows sums that group -- no second launch, and the sums of the
// groups that finish early run under the MFMAs of the blocks still working.  No block ever waits for another one: a block stores its partial
// with AGENT-SCOPE stores (sc1: written through the XCD's L2 to memory), waits for them (vmcnt), bumps the group's counter with an agent-scope
// atomic and EXITS unless the counter says it was the last of the group; the last one reads the group's rows with agent-scope loads and sums
// them in ROW ORDER (not arrival order: bit-reproducible run to run) into the group's first row and arrives, the same way, at the next level.
// (First version, measured: ordinary stores + __threadfence() = a write-back of the XCD's whole L2 per wave: +200 .. 250 us on EVERY launch,
// 2.48 -> 9.98 ms over the step's 35 weight gradients, profiles/r04_wgrad_fixup_probe.txt.  The L2s of the 8 XCDs are not coherent with each
// other for ordinary accesses; sc1 accesses are, at the price of going to the memory side every time.)  A tree of fan-in G and 1 - 3 levels (1024 splits: 11 x 11 x 9; a deep-level
// tile with 6 splits: one level), so that no single block pulls more than G x (its tile of the gradient) through one CU; the top level writes
// dw (+ bias gradient), adding to what is there when the module is shared.  No spin, no co-residency requirement (safe beside RCCL's resident
// kernels and for grids larger than the chip).  Counters live in a small zeroed buffer of the caller's (mtbc_conv3x3_args.wgrad_sync): the
// winner of a group resets its counter, so the buffer is all zeros again when the launch ends.
// Leaf order: the kernels hand in a `leaf` index that puts the splits an XCD runs next to each other (workgroups go to the 8 XCDs round-robin by
// linear id), so that a first-level group is mostly the work of ONE XCD: its rows are still in that XCD's L2 when the winner reads them.
constexpr int SPLITK_MAXG = 16;
template <bool AGENT> __device__ __forceinline__ void st_partial(float* p, float v) {
    if constexpr (AGENT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
struct SplitKFix {
    int* ctr;                // nullptr: partials are left to a reduction launch
    float* dw; float* dbias; // destination of the top level ([Cout][Cin][9], [Cout] or nullptr)
    int accumulate;          // top level adds to dw / dbias
    int G, levels, nsplit;   // fan-in, tree depth (G^levels >= nsplit), leaves per tile
    int ctr_per_tile;        // counters of one (co block, ci block) tile: off[l] = first counter of level l
    int off[3];
};
// counters a tile needs + the level offsets (host)
static inline int splitk_fix_plan(int nsplit, SplitKFix* f) {
    f->nsplit = nsplit;
    f->levels = nsplit <= 8 ? 1 : nsplit <= 64 ? 2 : 3;
    int G = 1;
    for (;;) { long long c = 1; for (int l = 0; l < f->levels; ++l) c *= G; if (c >= nsplit) break; ++G; }
    if (G > SPLITK_MAXG) { f->levels = 3; G = SPLITK_MAXG; }            // (nsplit <= 4096 = 16^3: every plan of plan_wgrad is <= 1024)
    f->G = G < 2 ? 2 : G;
    int n = nsplit, tot = 0;
    for (int l = 0; l < 3; ++l) { f->off[l] = tot; if (l < f->levels) { n = (n + f->G - 1) / f->G; tot += n; } }
    f->ctr_per_tile = tot;
    return tot;
}
// splits with the same (split % 8) first: leaf index of `split` among `nsplit` (the classes keep their order of sizes: class x has
// (nsplit - x + 7) / 8 members)
__device__ __forceinline__ int splitk_leaf_mod8(int split, int nsplit) {
    const int x = split & 7, q = nsplit >> 3, r = nsplit & 7;
    return x * q + min(x, r) + (split >> 3);
}
// Called by EVERY thread of the block after the block's partial has been stored to row `leaf` (all stores issued; no thread may have
// returned).  Tile region of a row: `nrun` runs of `runlen` floats at base + r * runstride (one run per output channel: its input channels
// x 9 taps are contiguous), plus `nb` bias-gradient floats at boff (nb = 0: none).  `flag`: one int of LDS.
template <int THREADS>
__device__ __forceinline__ void splitk_fixup(const SplitKFix& f, float* __restrict__ partial, long long prow, int tile, int leaf,
                                             size_t base, int nrun, int runlen, size_t runstride, size_t boff, int b0, int nb, int* flag) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int NW = THREADS / 64;
    int idx = leaf, n_here = f.nsplit;
    long long stride = prow;                       // floats between consecutive members of a group at this level
    for (int l = 0; l < f.levels; ++l) {
        const int gid = idx / f.G, cnt = min(f.G, n_here - gid * f.G);
        const bool top = l + 1 == f.levels;
        if (cnt > 1) {
            // this block's rows (its own partial, or the group sum it has just written: agent-scope stores) have reached memory before
            // the counter moves: every wave waits for its stores, then the barrier
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                int* c = f.ctr + (size_t)tile * f.ctr_per_tile + f.off[l] + gid;
                const int old = __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = old == cnt - 1;
                if (last) __hip_atomic_store(c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // every member has arrived: the counter is free again
                *flag = last;
            }
            __syncthreads();
            const int last = *flag;
            if (!last) return;
        } else if (top) {
            __syncthreads();                       // a single split: the row is this block's own partial
        }
        if (cnt > 1 || top) {
            float* __restrict__ src = partial + (size_t)gid * f.G * stride;
            for (int r = wv; r < nrun; r += NW) {
                const size_t ro = base + (size_t)r * runstride;
                for (int k = lane; k < runlen; k += 64) {
                    float v[SPLITK_MAXG];
#pragma unroll
                    for (int m = 0; m < SPLITK_MAXG; ++m) v[m] = m < cnt ? ld_agent(src + (size_t)m * stride + ro + k) : 0.f;
                    float sacc = v[0];
#pragma unroll
                    for (int m = 1; m < SPLITK_MAXG; ++m) sacc += v[m];            // row order; absent members add +0
                    if (top) { float* d = f.dw + ro + k; *d = f.accumulate ? *d + sacc : sacc; }
                    else st_partial<true>(src + ro + k, sacc);
                }
            }
            if (tid < nb) {
                float sacc = ld_agent(src + boff + b0 + tid);
                for (int m = 1; m < cnt; ++m) sacc += ld_agent(src + (size_t)m * stride + boff + b0 + tid);
                if (top) { float* d = f.dbias + b0 + tid; *d = f.accumulate ? *d + sacc : sacc; }
                else st_partial<true>(src + boff + b0 + tid, sacc);
            }
        }
        idx = gid; stride *= f.G; n_here = (n_here + f.G - 1) / f.G;
    }
}

struct WgC8P {
    int N, H, W, Cin, Cout;
    SegTable in;                       // ptr = 16-bit base, bstride in 16-bit elements, channels % 8 == 0
    const unsigned short* dz;          // [N][Cout/8][HW][8]
    float* partial;                    // [nsplit] rows of `prow` floats: [Cout][Cin][9], then [Cout] bias-gradient partials when asked for
    long long prow;
    int want_bias;
    int tiles_x, tiles_y, total_tiles, tiles_per_split, ciblocks;
    int hack;
    int coblocks, cit, segs, seg_tiles, depth;   // conv3x3_wgrad_c8w_kernel: input-channel tiles of 16 per block; row segments per strip, steps (4 rows) per segment
    SplitKFix fix;                     // in-kernel reduction of the split-K partials (fix.ctr != nullptr), see splitk_fixup
#ifdef MTBC_PROBES
    unsigned long long* ts;            // phase timestamps (MTBC_WG_TS=1): [block][16] ticks of the 100 MHz clock
#endif
};
#ifdef MTBC_PROBES
#define MTBC_WTS(p, k) do { if ((p).ts && threadIdx.x == 0) (p).ts[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (k)] = wall_clock64(); } while (0)
#else
#define MTBC_WTS(p, k) do { } while (0)
#endif
template <bool F16, int GEO>
__global__ __launch_bounds__(256, 4) void conv3x3_wgrad_c8_kernel(const WgC8P p) {
    using G = C8WGeo<GEO>;
    using T = LP<F16>;
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;
    constexpr int TW = G::TW, LW = G::LW, XG = G::HR * G::LW, ZG = C8W_ZG, ZBASE = (4 * XG + 16) * 8;
    constexpr int XI = (XG + 63) / 64;                  // DMA instructions per input-channel group (the last one partly masked)
    extern __shared__ __attribute__((aligned(16))) unsigned short smemc8[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = p.H * p.W;
    const int cib = blockIdx.y % p.ciblocks;
    const int co0 = (blockIdx.y / p.ciblocks) * 32, ci0 = cib * 32;
    const int split = blockIdx.x;
    // tiles dealt evenly over the splits (tiles_per_split < 0: split s owns [s T / S, (s + 1) T / S)), so that the plan can ask for ANY number of splits
    // -- a multiple of 8, see plan_wgrad -- without a short last split
    const int t_begin = p.tiles_per_split < 0 ? (int)((long long)split * p.total_tiles / (int)gridDim.x) : split * p.tiles_per_split;
    const int t_end = p.tiles_per_split < 0 ? (int)((long long)(split + 1) * p.total_tiles / (int)gridDim.x) : min(p.total_tiles, t_begin + p.tiles_per_split);

    // DMA: wave w brings input-channel group w (4 instructions) and output-channel group w (2 instructions) of the block
    const int cx = ci0 + 8 * wv, cz = co0 + 8 * wv;
    const bool xg_ok = cx < p.Cin, zg_ok = cz < p.Cout;
    const SegRef sr = seg_ref(p.in, xg_ok ? cx : 0);
    const long long xbs = sr.bs;
    const unsigned short* xgrp = reinterpret_cast<const unsigned short*>(sr.ptr) + (size_t)((xg_ok ? cx : 0) - sr.cb) * HW;
    const unsigned short* zgrp = p.dz + (size_t)(zg_ok ? cz : 0) * HW;
    int xrow[XI], xcol[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) { const int s = 64 * i + lane; xrow[i] = s / LW - 1; xcol[i] = s % LW - 1; }
    const int zrow = lane / TW, zcol = lane % TW;          // instruction i adds 64 / TW rows
    auto issue = [&](int tile) {
        int t = tile;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y; t /= p.tiles_y;
        const int n = t, x0 = tx * TW, y0 = ty * G::TH;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(xgrp + (size_t)n * xbs), 0, xg_ok ? HW * 16 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(zgrp + (size_t)n * p.Cout * HW), 0, zg_ok ? HW * 16 : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int y = y0 + xrow[i], x = x0 + xcol[i];
            const bool ok = y >= 0 && y < p.H && x >= 0 && x < p.W;
            const unsigned voff = ok ? 16u * (unsigned)(y * p.W + x) : 0xfffffff0u;
            if (64 * i + lane < XG)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(smemc8 + (wv * XG + 64 * i) * 8), 16, voff, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int y = y0 + zrow + (64 / TW) * i, x = x0 + zcol;
            const unsigned voff = (y < p.H && x < p.W) ? 16u * (unsigned)(y * p.W + x) : 0xfffffff0u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(zr, (lds_ptr_t)(smemc8 + ZBASE + (wv * ZG + 64 * i) * 8), 16, voff, 0, 0, 0);
        }
    };

    const int j = lane & 15, kg = lane >> 4, q = j >> 2, pp = j & 3;
    const int it = wv & 1, kh = wv >> 1;
    // transposed-read addresses: lane 4q+pp of a 16-lane group supplies row q (a pixel) and columns 4pp..4pp+3 (channels;
    // columns 0-7 from the first 8-channel group of the tile, 8-15 from the second) of the 4x16 block
    const int lrow = GEO == 0 ? 0 : (kg >> 1), lcol = GEO == 0 ? 8 * kg : 8 * (kg & 1);      // this lane group's place in a K-step
    const int zoff = ZBASE + ((pp >> 1) * ZG + lrow * TW + lcol + q) * 8 + 4 * (pp & 1);     // + c*2*ZG*8 + (step*32 + 4*blk)*8
    const int xoff = ((2 * it + (pp >> 1)) * XG + lrow * LW + lcol + q) * 8 + 4 * (pp & 1);  // + ((row0+r)*LW + 4*blk)*8
    const bool do_bias = p.want_bias && cib == 0 && it == 0;         // wave-uniform

    f32x4 acc[2][9], accb[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        accb[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 9; ++i) acc[c][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    typename T::frag ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = 1.0f;

    MTBC_WTS(p, 0);
    for (int tile = t_begin; tile < t_end; ++tile) {
        lds_barrier();                         // everyone is done reading the previous tile
        if (tile - t_begin < 3) MTBC_WTS(p, 1 + 3 * (tile - t_begin));
        issue(tile);
        if (tile - t_begin < 3) MTBC_WTS(p, 2 + 3 * (tile - t_begin));
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (tile - t_begin < 3) MTBC_WTS(p, 3 + 3 * (tile - t_begin));
#pragma unroll
        for (int ksl = 0; ksl < 2; ++ksl) {
            const int step = 2 * kh + ksl;     // 32 pixels per step: one tile row (GEO 0) or two (GEO 1)
            const int row = GEO == 0 ? step : 2 * step;
            typename T::frag a[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(smemc8 + zoff + c * 2 * ZG * 8 + (step * 32) * 8));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(smemc8 + zoff + c * 2 * ZG * 8 + (step * 32 + 4) * 8));
                const short e[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                a[c] = __builtin_bit_cast(typename T::frag, e);
            }
            if (do_bias) {
#pragma unroll
                for (int c = 0; c < 2; ++c) accb[c] = T::mfma(a[c], ones, accb[c]);
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                unsigned d[6];
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(smemc8 + xoff + ((row + r) * LW + 4 * b) * 8));
                    const uint2 u = __builtin_bit_cast(uint2, v);
                    d[2 * b] = u.x; d[2 * b + 1] = u.y;
                }
                u32x4 w0, w1, w2;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    w0[k] = d[k];
                    w1[k] = __builtin_amdgcn_alignbit(d[k + 1], d[k], 16);
                    w2[k] = d[k + 1];
                }
                const typename T::frag b0 = __builtin_bit_cast(typename T::frag, w0);
                const typename T::frag b1 = __builtin_bit_cast(typename T::frag, w1);
                const typename T::frag b2 = __builtin_bit_cast(typename T::frag, w2);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    acc[c][r * 3 + 0] = T::mfma(a[c], b0, acc[c][r * 3 + 0]);
                    acc[c][r * 3 + 1] = T::mfma(a[c], b1, acc[c][r * 3 + 1]);
                    acc[c][r * 3 + 2] = T::mfma(a[c], b2, acc[c][r * 3 + 2]);
                }
            }
        }
    }
    // sum the two row halves (waves 2,3 -> waves 0,1) through LDS, then one partial per block
    MTBC_WTS(p, 10);
    __syncthreads();
    MTBC_WTS(p, 11);
    f32x4* red = reinterpret_cast<f32x4*>(smemc8);        // [it][18][64] + [2][64] f32x4
    if (kh == 1) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 9; ++i) red[(it * 18 + c * 9 + i) * 64 + lane] = acc[c][i];
        if (do_bias) { red[(36 + 0) * 64 + lane] = accb[0]; red[(36 + 1) * 64 + lane] = accb[1]; }
    }
    __syncthreads();
    // the row this block's partial goes to: with the in-kernel reduction, the splits that ran on one XCD side by side (splitk_fixup)
    const int prow_idx = p.fix.ctr ? splitk_leaf_mod8(split, (int)gridDim.x) : split;
    auto store_partial = [&](auto agent_) {           // agent-scope stores when the partials are summed inside this launch (splitk_fixup)
        constexpr bool AG = decltype(agent_)::value;
        if (do_bias && j == 0) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + c * 16 + kg * 4 + r;
                    if (co < p.Cout) st_partial<AG>(p.partial + (size_t)prow_idx * p.prow + (size_t)p.Cout * p.Cin * 9 + co, accb[c][r] + red[(36 + c) * 64 + lane][r]);
                }
        }
        const int ci = ci0 + it * 16 + j;
        if (ci < p.Cin) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + c * 16 + kg * 4 + r;
                    if (co >= p.Cout) continue;
                    float* dst = p.partial + (size_t)prow_idx * p.prow + ((size_t)co * p.Cin + ci) * 9;
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) st_partial<AG>(dst + tap, acc[c][tap][r] + red[(it * 18 + c * 9 + tap) * 64 + lane][r]);
                }
        }
    };
    if (kh == 0) { if (p.fix.ctr) store_partial(std::true_type{}); else store_partial(std::false_type{}); }
    MTBC_WTS(p, 12);
    if (!p.fix.ctr) return;
    int* fix_flag = reinterpret_cast<int*>(smemc8);      // (the LDS images are dead: splitk_fixup's first barrier is behind every wave's last read)
    const int nco = min(32, p.Cout - co0), nci = min(32, p.Cin - ci0);
    splitk_fixup<256>(p.fix, p.partial, p.prow, (int)blockIdx.y, prow_idx, ((size_t)co0 * p.Cin + ci0) * 9, nco, nci * 9, (size_t)p.Cin * 9,
                      (size_t)p.Cout * p.Cin * 9, co0, (p.want_bias && cib == 0) ? nco : 0, fix_flag);
}

// ------------------------------------------------------------------ wgrad on channel-blocked operands, wide blocks + rolling rows ("c8w", round 3)
// conv3x3_wgrad_c8_kernel stages (L2 -> LDS, by DMA) 1.59 x 32 input channels + 32 output channels per 128-pixel tile and
// (32 co x 32 ci) block: dz once per 32 input channels (five times for the 144 -> 24 conv), X once per 32 output channels
// (twice for Cout = 48, with 64 x 64 MFMA padding around 48 x 48), and a third of every X tile is halo that was staged before.
// Its measured HBM traffic is 2.2x the algorithmic bytes (profiles/r02c_hbm_traffic_bf16.json) at a byte rate that is already
// what this code base reaches, so the lever is bytes.  This kernel:
//   * a block owns ALL output channels of a 24 / 48-channel conv (COT = 2 or 3 tiles of 16; wider outputs in blocks of 48 or 32)
//     and up to 5 input-channel tiles (80 channels);
//   * it walks DOWN a 32-column strip of one image and keeps the rows it has already staged: per group an LDS ring of 10 halo
//     rows x 34 columns; a step multiplies 4 rows x 32 pixels (4 K-steps, 6 live ring rows) while the DMA of the NEXT 4 rows
//     (and of the next step's dz, second stage) is in flight -- every X row is staged once (34 / 32 = 1.06x), nothing waits for
//     a load that was not issued a whole step earlier, ONE barrier per step;
//   * 8 waves; the work items are (ci tile, tap row) PAIRS dealt round-robin to the waves (at most U = 2 per wave): a pair is
//     one set of three shifted X windows (three transposed reads + v_alignbit, as before) times all COT dz fragments, which a
//     wave reads once per K-step for both of its pairs -- (2 COT + 6) LDS reads for 6 COT MFMAs;
//   * every accumulator has ONE owner (no row-half split): no end-of-block LDS reduction, the partial is stored straight from
//     the accumulators.  2 blocks (16 waves) per CU, one wave of <= 512 blocks; a block = (strip, row segment, channel block).
// Same LDS piece layout, transposed reads, MFMA operand order as conv3x3_wgrad_c8_kernel; the K order and the split-K
// partition differ, so results agree to fp32 re-association, not bit for bit.
// Depth D = 1 .. 3 steps of DMA in flight per block (ring of 6 + 4 D rows, D + 1 dz stages, counted vmcnt): the plan takes the
// (blocks per CU, D) that puts the most bytes in flight within the 160 KB of LDS -- a 24 -> 24 step is only 12.7 KB.
// LDS-DMA by hand (round 3).  hipcc knows that `buffer_load ... lds` stores to LDS and cannot tell WHICH part of a dynamic LDS block a
// later ds_read touches, so behind the builtin it puts `s_waitcnt vmcnt(0)` in front of the next LDS read -- the DMA of step t + 1,
// issued before the MFMAs of step t precisely to run under them, was waited for before the first fragment read of step t
// (ISA: `[vmcnt(0)] r14 ...` at the top of the step body; ablation: DMA-only 107 us + compute-only 123 us = full 235 us).  Issued from
// inline assembly the load is an opaque instruction: no alias tracking, no inserted wait; the kernel's own counted `s_waitcnt vmcnt`
// + barrier are the ordering (they were all along).  The descriptor is the four dwords make_buffer_rsrc builds.
// (dma_rsrc / lds_addr / dma16: defined in front of conv3x3_igemm_c8_ring_kernel, their first user)
#ifdef MTBC_PROBES
#define MTBC_DBG_HACK(p) ((p).hack)
#else
#define MTBC_DBG_HACK(p) (0)
#endif
#define MTBC_C8W_ADVANCE() do { b4 = b4 + 4 >= RING ? b4 + 4 - RING : b4 + 4; bn = bn + 4 >= RING ? bn + 4 - RING : bn + 4; zs = zs == D ? 0 : zs + 1; zn = zn == D ? 0 : zn + 1; } while (0)
constexpr int C8WW_TH = 4, C8WW_TW = 32, C8WW_LW = 34, C8WW_ZG = C8WW_TH * C8WW_TW + 4;
constexpr int C8WW_MAXCIT = 5;            // ci tiles of 16 per block: 3 * 5 = 15 pairs <= 8 waves x 2
static inline int c8w_ring(int depth) { return 6 + 4 * depth; }
static_assert((10 * C8WW_LW * 16) % 256 == 64 && (14 * C8WW_LW * 16) % 256 == 192 && (18 * C8WW_LW * 16) % 256 == 64 && (C8WW_ZG * 16) % 256 == 64,
              "group strides: 64 / 192 mod 256 B keep the transposed reads conflict-free");
// X ring of every input-channel group, 16 pieces of pad, D + 1 stages of dz
static inline size_t c8w_lds_bytes(int cit, int cot, int depth) { return (size_t)(2 * cit * c8w_ring(depth) * C8WW_LW + 16 + (depth + 1) * 2 * cot * C8WW_ZG) * 16; }
static inline size_t c8w_step_bytes(int cit, int cot) { return (size_t)(2 * cit * C8WW_TH * C8WW_LW + 2 * cot * C8WW_TH * C8WW_TW) * 16; }
__device__ __forceinline__ void c8w_wait_vm(int c) {       // s_waitcnt vmcnt(c), c wave-uniform and even, <= 20
    switch (c) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    }
}
template <bool F16, int COT, bool BIAS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv3x3_wgrad_c8w_kernel(const WgC8P p) {
    using T = LP<F16>;
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;
    constexpr int TW = C8WW_TW, TH = C8WW_TH, LW = C8WW_LW, ZG = C8WW_ZG, U = 2;
    const int D = p.depth, RING = 6 + 4 * D, XG = RING * LW;      // steps in flight, ring rows, pieces per input-channel group
    extern __shared__ __attribute__((aligned(16))) unsigned short smemw[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = p.H * p.W;
    // block id -> (image, row segment, strip, channel block).  Workgroups go to the 8 XCDs round-robin by linear id: with N % 8 == 0
    // image n is handled by XCD n % 8 alone, and there the channel blocks of one (strip, segment) and then the strips of one segment
    // are neighbours in dispatch order -- what they share (dz between input-channel blocks, X between output-channel blocks, the
    // cache lines a strip's halo columns have in common with the next strip) is fetched into ONE L2, by blocks that run side by side
    int yb, seg, tx, n;
    {
        const int ny = p.ciblocks * p.coblocks;
        int k = blockIdx.x;
        int img0 = 0, imgs = 1;
        if (p.N % 8 == 0) { img0 = k & 7; imgs = 8; k >>= 3; }
        yb = k % ny; k /= ny;
        tx = k % p.tiles_x; k /= p.tiles_x;
        seg = k % p.segs; k /= p.segs;
        n = k * imgs + img0;
    }
    const int cib = yb % p.ciblocks, co0 = (yb / p.ciblocks) * 16 * COT;
    const int cit0 = cib * p.cit, ncit = min(p.cit, ((p.Cin + 15) >> 4) - cit0), ci0 = cit0 * 16;
    const int ZBASE = (2 * p.cit * XG + 16) * 8;           // 16-bit elements
    constexpr int ZSTAGE = 2 * COT * ZG * 8;
    // this block's share of the pixels: rows [y0, y0 + 4 nt) of the 32-column strip x0 of image n
    const int split = (n * p.tiles_x + tx) * p.segs + seg;
    const int x0 = tx * TW;
    const int ty0 = seg * p.seg_tiles, nt = min(p.seg_tiles, p.tiles_y - ty0), y0 = ty0 * TH;

    // DMA: wave w brings input-channel groups w and w + 8 of the block and output-channel group 7 - w; one image, so the
    // buffer descriptors are built once
    i32x4 xr[2]; bool xok[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int gx = wv + 8 * k, c = ci0 + 8 * gx;
        xok[k] = gx < 2 * ncit && c < p.Cin;             // (an absent group's LDS image is never read into a stored result)
        const SegRef sr = seg_ref(p.in, xok[k] ? c : 0);
        const unsigned short* base = reinterpret_cast<const unsigned short*>(sr.ptr) + (size_t)((xok[k] ? c : 0) - sr.cb) * HW + (size_t)n * sr.bs;
        xr[k] = dma_rsrc(base, HW * 16);
    }
    const int gz = 7 - wv, cz = co0 + 8 * gz;
    const bool zok = gz < 2 * COT && cz < p.Cout;
    const i32x4 zr = dma_rsrc(p.dz + ((size_t)n * p.Cout + (zok ? cz : 0)) * HW, HW * 16);
    const unsigned lds0 = lds_addr(smemw);
    // X rows: one instruction per (group, halo row), lanes 0..33 = halo columns x0 - 1 .. x0 + 32; row `rel` counts from y0 - 1
    // and lives in ring slot rel % 10
    auto issue_x = [&](int rel_first, int slot_first, int count) {
        int ln = lane;
        asm volatile("" : "+v"(ln));                       // (keeps the per-lane offsets out of the registers that live across the loop)
        const int x = x0 - 1 + ln;
        const bool colok = ln < LW && x >= 0 && x < p.W;
#ifdef MTBC_PROBES
        if (p.hack & 1) {      // TIMING ONLY (wrong results): the same rows as aligned 2-row x 32-column instructions of a full 1 KiB, no halo columns
            const int xx = x0 + (ln & 31);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (!xok[k]) continue;
                for (int rr = 0; rr < count; rr += 2) {
                    const int y = y0 - 1 + rel_first + rr + (ln >> 5);
                    const unsigned voff = (xx < p.W && y >= 0 && y < p.H) ? 16u * (unsigned)(y * p.W + xx) : 0xfffffff0u;
                    dma16(xr[k], lds0 + 2u * (unsigned)(((wv + 8 * k) * XG + ((slot_first + rr) % (RING - 2)) * LW) * 8), voff);
                }
            }
            return;
        }
#endif
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (!xok[k]) continue;                         // wave-uniform
            int slot = slot_first;
            for (int rr = 0; rr < count; ++rr) {
                const int y = y0 - 1 + rel_first + rr;
                const unsigned voff = (colok && y >= 0 && y < p.H) ? 16u * (unsigned)(y * p.W + x) : 0xfffffff0u;
                if (ln < LW)
                    dma16(xr[k], lds0 + 2u * (unsigned)(((wv + 8 * k) * XG + slot * LW) * 8), voff);
                slot = slot + 1 == RING ? 0 : slot + 1;
            }
        }
    };
    auto issue_z = [&](int t, int stg) {                   // dz of step t: 4 rows x 32 pixels into stage t % (D + 1)
        if (!zok) return;
        const int x = x0 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int y = y0 + TH * t + 2 * i + (lane >> 5);
            const unsigned voff = (y < p.H && x < p.W) ? 16u * (unsigned)(y * p.W + x) : 0xfffffff0u;
            dma16(zr, lds0 + 2u * (unsigned)(ZBASE + stg * ZSTAGE + (gz * ZG + 64 * i) * 8), voff);
        }
    };

    const int j = lane & 15, kg = lane >> 4, q = j >> 2, pp = j & 3;
    // transposed-read addresses (see conv3x3_wgrad_c8_kernel): a K-step is one row of 32 pixels, lane group kg owns columns 8kg .. 8kg + 7
    const int zoff = ZBASE + ((pp >> 1) * ZG + 8 * kg + q) * 8 + 4 * (pp & 1);        // + stage + c * 2 * ZG * 8 + (step * 32 + 4 * half) * 8
    int xoff[U], pcit[U], prow[U]; bool pok[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int pi = wv + 8 * i;                          // pair = (ci tile, tap row)
        pok[i] = pi < 3 * ncit;
        pcit[i] = pi / 3; prow[i] = pi - 3 * pcit[i];
        xoff[i] = ((2 * pcit[i] + (pp >> 1)) * XG + 8 * kg + q) * 8 + 4 * (pp & 1);   // + ring slot * LW * 8 + 4 * blk * 8
    }
    // BIAS (dz x ones on the matrix pipe; the training step never asks: its conv-bias gradient comes out of the InstanceNorm backward): every
    // wave of the input-channel block 0 carries the sums, wave 7 stores them
    const bool do_bias = BIAS && cib == 0 && wv == 7;

    f32x4 acc[U][COT][3], accb[COT];
#pragma unroll
    for (int c = 0; c < COT; ++c) {
        accb[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < U; ++i)
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_) acc[i][c][s_] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    typename T::frag ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = 1.0f;

    // step t multiplies rows rel 4t .. 4t + 5 (ring slots (4t + k) % RING) while the rows of steps t + 1 .. t + D arrive in the slots
    // step t does not read.  ONE barrier per step: behind it every wave's share of step t has landed (each wave waited until at most
    // the instructions of the later batches were outstanding: vmcnt counts down in issue order) and every wave has finished step
    // t - 1, whose four oldest rows / dz stage the batch issued next overwrites.
    const int nw = ((p.hack & 1) ? 2 : 4) * ((xok[0] ? 1 : 0) + (xok[1] ? 1 : 0)) + (zok ? 2 : 0);       // DMA instructions of this wave per steady batch
    MTBC_WTS(p, 0);
    issue_x(0, 0, 6);
    issue_z(0, 0);
    for (int k = 1; k < D && k < nt; ++k) { issue_x(TH * k + 2, TH * k + 2, 4); issue_z(k, k); }       // (4k + 2 + 3 < RING for k < D)
    MTBC_WTS(p, 1);
    int b4 = 0, bn = (TH * D + 2) % RING, zs = 0, zn = D % (D + 1);     // (4 t) % RING, slot of the first row of batch t + D, dz stage of t / of t + D
    for (int t = 0; t < nt; ++t) {
        const int later = min(nt - 1 - t, D - 1);
        c8w_wait_vm(later * nw);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (t >= 2 && t < 5) MTBC_WTS(p, 2 + 3 * (t - 2));          // (steps 2 .. 4: the steady state)
        if (t + D < nt && !(MTBC_DBG_HACK(p) & 4)) {
            issue_x(TH * (t + D) + 2, bn, 4);
            issue_z(t + D, zn);
        }
        if (t >= 2 && t < 5) MTBC_WTS(p, 3 + 3 * (t - 2));
        const unsigned short* smz = smemw + zs * ZSTAGE;
        if (MTBC_DBG_HACK(p) & 2) { MTBC_C8W_ADVANCE(); continue; }        // TIMING ONLY: no fragment reads, no MFMAs
        // The four K-steps of a step as ONE straight-line body per (valid pair slots of this wave, bias wave or not): a wave-uniform
        // branch inside the body ends the scheduler's view, and each K-step then runs read -> wait -> align -> MFMA back to back
        // (measured: 3.5x the MFMA time at 4 waves per SIMD); chosen once per step.
        auto body = [&](auto ns_) {
            constexpr int NS = decltype(ns_)::value;
            // Software pipeline in the source, fenced with sched_barriers (left alone the scheduler hoists ALL LDS reads of the step to
            // the top and spills ~300 registers): the dz fragments of K-step s + 1 are read into a second set before the MFMAs of
            // K-step s; the X windows of pair slot i for K-step s + 1 are read right behind that slot's MFMAs of K-step s, into
            // the registers those MFMAs have just consumed (the other slot's MFMAs cover the latency).
            s16x4 ra[2][COT][2], rb[NS][3];
            auto load_a = [&](int step, int set) {
#pragma unroll
                for (int c = 0; c < COT; ++c) {
                    ra[set][c][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(smz + zoff + c * 2 * ZG * 8 + (step * 32) * 8));
                    ra[set][c][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(smz + zoff + c * 2 * ZG * 8 + (step * 32 + 4) * 8));
                }
            };
            auto load_b = [&](int step, int i) {
                const int v = b4 + step + prow[i];                       // ring slot of halo row (step + tap row): scalar
                const unsigned short* smx = smemw + (v >= RING ? v - RING : v) * (LW * 8) + xoff[i];
#pragma unroll
                for (int b = 0; b < 3; ++b) rb[i][b] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(smx + 4 * b * 8));
            };
            load_a(0, 0);
#pragma unroll
            for (int i = 0; i < NS; ++i) load_b(0, i);
#pragma unroll
            for (int step = 0; step < TH; ++step) {
                const int set = step & 1;
                if (step + 1 < TH) load_a(step + 1, set ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                typename T::frag a[COT];
#pragma unroll
                for (int c = 0; c < COT; ++c) {
                    const s16x4 lo = ra[set][c][0], hi = ra[set][c][1];
                    const short e[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    a[c] = __builtin_bit_cast(typename T::frag, e);
                }
                if constexpr (BIAS) {
#pragma unroll
                    for (int c = 0; c < COT; ++c) accb[c] = T::mfma(a[c], ones, accb[c]);
                }
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    unsigned d[6];
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        const uint2 u = __builtin_bit_cast(uint2, rb[i][b]);
                        d[2 * b] = u.x; d[2 * b + 1] = u.y;
                    }
                    u32x4 w0, w1, w2;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        w0[k] = d[k];
                        w1[k] = __builtin_amdgcn_alignbit(d[k + 1], d[k], 16);
                        w2[k] = d[k + 1];
                    }
                    const typename T::frag b0 = __builtin_bit_cast(typename T::frag, w0);
                    const typename T::frag b1 = __builtin_bit_cast(typename T::frag, w1);
                    const typename T::frag b2 = __builtin_bit_cast(typename T::frag, w2);
#pragma unroll
                    for (int c = 0; c < COT; ++c) {
                        acc[i][c][0] = T::mfma(a[c], b0, acc[i][c][0]);
                        acc[i][c][1] = T::mfma(a[c], b1, acc[i][c][1]);
                        acc[i][c][2] = T::mfma(a[c], b2, acc[i][c][2]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (step + 1 < TH) load_b(step + 1, i);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        // (pair slots this wave does not own are multiplied too -- their accumulators are never stored: one straight-line body
        //  per kernel keeps the register allocation within 128; COT = 2 affords a second body for the waves with ONE pair)
        if (COT == 2 && !pok[1]) body(I1{}); else body(I2{});
        if (t >= 2 && t < 5) MTBC_WTS(p, 4 + 3 * (t - 2));
        MTBC_C8W_ADVANCE();
    }
    MTBC_WTS(p, 11);
    // the row of this block's partial: with the in-kernel reduction the images of one XCD (n % 8, see the block order above) are neighbours
    const int per_img = p.tiles_x * p.segs;
    const int prow_idx = (p.fix.ctr && p.N % 8 == 0) ? ((n & 7) * (p.N >> 3) + (n >> 3)) * per_img + tx * p.segs + seg : split;
    float* prow_base = p.partial + (size_t)prow_idx * p.prow;
    auto store_partial = [&](auto agent_) {           // agent-scope stores when the partials are summed inside this launch (splitk_fixup)
        constexpr bool AG = decltype(agent_)::value;
        if (do_bias && j == 0) {
#pragma unroll
            for (int c = 0; c < COT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + c * 16 + kg * 4 + r;
                    if (co < p.Cout) st_partial<AG>(prow_base + (size_t)p.Cout * p.Cin * 9 + co, accb[c][r]);
                }
        }
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int ci = ci0 + pcit[i] * 16 + j;
            if (!pok[i] || ci >= p.Cin) continue;
#pragma unroll
            for (int c = 0; c < COT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + c * 16 + kg * 4 + r;
                    if (co >= p.Cout) continue;
                    float* dst = prow_base + ((size_t)co * p.Cin + ci) * 9 + 3 * prow[i];
                    st_partial<AG>(dst, acc[i][c][0][r]); st_partial<AG>(dst + 1, acc[i][c][1][r]); st_partial<AG>(dst + 2, acc[i][c][2][r]);
                }
        }
    };
    if (p.fix.ctr) store_partial(std::true_type{}); else store_partial(std::false_type{});
    MTBC_WTS(p, 12);
    if (!p.fix.ctr) return;
    {
        int* fix_flag = reinterpret_cast<int*>(smemw);      // (a static LDS word would not fit beside 160 KB of dynamic LDS; the ring is dead by now)
        const int nco = min(16 * COT, p.Cout - co0), nci = min(16 * ncit, p.Cin - ci0);
        splitk_fixup<512>(p.fix, p.partial, p.prow, yb, prow_idx, ((size_t)co0 * p.Cin + ci0) * 9, nco, nci * 9, (size_t)p.Cin * 9,
                          (size_t)p.Cout * p.Cin * 9, co0, (BIAS && cib == 0) ? nco : 0, fix_flag);
    }
}

// ------------------------------------------------------------------ wgrad on channel-blocked operands, 16 x 16 maps ("c8i", round 3)
// The deep level of the U-Net++ (16 x 16 maps, 192 .. 1152 -> 384 / 512 channels): K = 8192 pixels only, a weight gradient of 1.3 .. 5.3 M
// elements.  conv3x3_wgrad_c8_kernel runs it as 576 .. 1024 blocks of 32 co x 32 ci that each stage the pixels in 128-pixel tiles of 21 KB
// for 36 MFMAs per wave, single-buffered: 149 us for the 1152 -> 512 conv whose MFMAs take 35 us.  Here a block owns 32 / 48 output x
// <= 80 input channels (the pairs / MFMA body of conv3x3_wgrad_c8w_kernel) and a range of IMAGES; a tile is one whole image with its
// zero halo (18 x 18 pieces per input-channel group: the padding comes out of the buffer bounds check, nothing is staged twice), eight
// K-steps of 2 rows x 16 pixels, 135 MFMAs per wave and tile; two stages, the DMA of image n + 1 (inline assembly: no compiler-inserted
// vmcnt(0) in front of the fragment reads) under the MFMAs of image n, ONE barrier per image.  One block of 8 waves per CU (up to 154 KB
// of LDS), <= 256 blocks; everything it reads sits in L2 / the memory-side cache.
constexpr int C8I_LW = 18, C8I_XG = C8I_LW * C8I_LW, C8I_ZG = 256 + 4;
static_assert((C8I_XG * 16) % 256 == 64 && (C8I_ZG * 16) % 256 == 64, "group strides: 64 / 192 mod 256 B keep the transposed reads conflict-free");
static inline size_t c8i_lds_bytes(int cit, int cot) { return 2 * (size_t)(2 * cit * C8I_XG + 2 * cot * C8I_ZG) * 16 + 256; }
template <bool F16, int COT, bool BIAS>
__global__ __launch_bounds__(512) void conv3x3_wgrad_c8i_kernel(const WgC8P p) {
    using T = LP<F16>;
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;
    constexpr int LW = C8I_LW, XG = C8I_XG, ZG = C8I_ZG, U = 2, KS = 8, XI = (XG + 63) / 64, ZI = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned short smemi[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = 256;
    // block id = split + nsplit * (channel block): with nsplit a multiple of 8 an image range is read by ONE XCD (its L2 holds it)
    const int split = blockIdx.x % p.segs, yb = blockIdx.x / p.segs;          // (segs = number of image ranges)
    const int cib = yb % p.ciblocks, co0 = (yb / p.ciblocks) * 16 * COT;
    const int cit0 = cib * p.cit, ncit = min(p.cit, ((p.Cin + 15) >> 4) - cit0), ci0 = cit0 * 16;
    const int n_begin = split * p.seg_tiles, n_end = min(p.N, n_begin + p.seg_tiles);        // (seg_tiles = images per range)
    const int ZBASE = 2 * p.cit * XG * 8;                  // 16-bit elements, within a stage
    const int SS = ZBASE + 2 * COT * ZG * 8;               // one stage

    // DMA: wave w brings input-channel groups w and w + 8 (6 instructions each) and output-channel group 7 - w (4) of every image
    const unsigned short* xbase[2]; long long xbs[2]; bool xok[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int gx = wv + 8 * k, c = ci0 + 8 * gx;
        xok[k] = gx < 2 * ncit && c < p.Cin;             // (an absent group's LDS image is never read into a stored result)
        const SegRef sr = seg_ref(p.in, xok[k] ? c : 0);
        xbase[k] = reinterpret_cast<const unsigned short*>(sr.ptr) + (size_t)((xok[k] ? c : 0) - sr.cb) * HW;
        xbs[k] = sr.bs;
    }
    const int gz = 7 - wv, cz = co0 + 8 * gz;
    const bool zok = gz < 2 * COT && cz < p.Cout;
    const unsigned short* zbase = p.dz + (size_t)(zok ? cz : 0) * HW;
    const unsigned lds0 = lds_addr(smemi);
    unsigned xvoff[XI];                                   // the same for every image: piece s of the 18 x 18 halo image <- pixel (row - 1, col - 1)
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int s_ = 64 * i + lane, row = s_ / LW - 1, col = s_ % LW - 1;
        xvoff[i] = (s_ < XG && row >= 0 && row < 16 && col >= 0 && col < 16) ? 16u * (unsigned)(row * 16 + col) : 0xfffffff0u;
    }
    auto issue = [&](int n, int stg) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (!xok[k]) continue;                         // wave-uniform
            const i32x4 xr = dma_rsrc(xbase[k] + (size_t)n * xbs[k], HW * 16);
#pragma unroll
            for (int i = 0; i < XI; ++i)
                if (64 * i + lane < XG) dma16(xr, lds0 + 2u * (unsigned)(stg * SS + ((wv + 8 * k) * XG + 64 * i) * 8), xvoff[i]);
        }
        if (zok) {
            const i32x4 zr = dma_rsrc(zbase + (size_t)n * p.Cout * HW, HW * 16);
#pragma unroll
            for (int i = 0; i < ZI; ++i) dma16(zr, lds0 + 2u * (unsigned)(stg * SS + ZBASE + (gz * ZG + 64 * i) * 8), 16u * (unsigned)(64 * i + lane));
        }
    };

    const int j = lane & 15, kg = lane >> 4, q = j >> 2, pp = j & 3;
    // a K-step = two image rows of 16 pixels: lane group kg owns row kg >> 1, columns 8 (kg & 1) .. + 7 -- in the linear dz image that is
    // pixels 32 step + 8 kg .. + 7, in the halo image row 2 step + (kg >> 1) + tap row, columns 8 (kg & 1) + {0, 4, 8} + q
    const int zoff = ZBASE + ((pp >> 1) * ZG + 8 * kg + q) * 8 + 4 * (pp & 1);
    int xoff[U], pcit[U], prow[U]; bool pok[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int pi = wv + 8 * i;                          // pair = (ci tile, tap row)
        pok[i] = pi < 3 * ncit;
        pcit[i] = pi / 3; prow[i] = pi - 3 * pcit[i];
        xoff[i] = ((2 * pcit[i] + (pp >> 1)) * XG + ((kg >> 1) + prow[i]) * LW + 8 * (kg & 1) + q) * 8 + 4 * (pp & 1);
    }
    const bool do_bias = BIAS && cib == 0 && wv == 7;

    f32x4 acc[U][COT][3], accb[COT];
#pragma unroll
    for (int c = 0; c < COT; ++c) {
        accb[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < U; ++i)
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_) acc[i][c][s_] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    typename T::frag ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = 1.0f;

    int stage = 0;
    MTBC_WTS(p, 0);
    if (n_begin < n_end) issue(n_begin, 0);
    MTBC_WTS(p, 1);
    for (int n = n_begin; n < n_end; ++n) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // image n has landed for everybody; everybody is done with image n - 1
        if (n - n_begin < 3) MTBC_WTS(p, 2 + 3 * (n - n_begin));
        if (n + 1 < n_end) issue(n + 1, stage ^ 1);
        if (n - n_begin < 3) MTBC_WTS(p, 3 + 3 * (n - n_begin));
        const unsigned short* sm = smemi + stage * SS;
        stage ^= 1;
        // the eight K-steps as one straight-line, software-pipelined body (see conv3x3_wgrad_c8w_kernel): pair slots this wave does not own
        // are multiplied too, their accumulators are never stored
        auto body = [&](auto ns_) {
            constexpr int NS = decltype(ns_)::value;
            s16x4 ra[2][COT][2], rb[NS][3];
            auto load_a = [&](int step, int set) {
#pragma unroll
                for (int c = 0; c < COT; ++c) {
                    ra[set][c][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(sm + zoff + c * 2 * ZG * 8 + (step * 32) * 8));
                    ra[set][c][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(sm + zoff + c * 2 * ZG * 8 + (step * 32 + 4) * 8));
                }
            };
            auto load_b = [&](int step, int i) {
#pragma unroll
                for (int b = 0; b < 3; ++b) rb[i][b] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(sm + xoff[i] + (2 * step * LW + 4 * b) * 8));
            };
            load_a(0, 0);
#pragma unroll
            for (int i = 0; i < NS; ++i) load_b(0, i);
#pragma unroll
            for (int step = 0; step < KS; ++step) {
                const int set = step & 1;
                if (step + 1 < KS) load_a(step + 1, set ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                typename T::frag a[COT];
#pragma unroll
                for (int c = 0; c < COT; ++c) {
                    const s16x4 lo = ra[set][c][0], hi = ra[set][c][1];
                    const short e[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    a[c] = __builtin_bit_cast(typename T::frag, e);
                }
                if constexpr (BIAS) {
#pragma unroll
                    for (int c = 0; c < COT; ++c) accb[c] = T::mfma(a[c], ones, accb[c]);
                }
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    unsigned d[6];
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        const uint2 u = __builtin_bit_cast(uint2, rb[i][b]);
                        d[2 * b] = u.x; d[2 * b + 1] = u.y;
                    }
                    u32x4 w0, w1, w2;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        w0[k] = d[k];
                        w1[k] = __builtin_amdgcn_alignbit(d[k + 1], d[k], 16);
                        w2[k] = d[k + 1];
                    }
                    const typename T::frag b0 = __builtin_bit_cast(typename T::frag, w0);
                    const typename T::frag b1 = __builtin_bit_cast(typename T::frag, w1);
                    const typename T::frag b2 = __builtin_bit_cast(typename T::frag, w2);
#pragma unroll
                    for (int c = 0; c < COT; ++c) {
                        acc[i][c][0] = T::mfma(a[c], b0, acc[i][c][0]);
                        acc[i][c][1] = T::mfma(a[c], b1, acc[i][c][1]);
                        acc[i][c][2] = T::mfma(a[c], b2, acc[i][c][2]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (step + 1 < KS) load_b(step + 1, i);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        if (COT == 2 && !pok[1]) body(I1{}); else body(I2{});
        if (n - n_begin < 3) MTBC_WTS(p, 4 + 3 * (n - n_begin));
    }
    MTBC_WTS(p, 11);
    // (block id = split + segs * channel block: with segs % 8 == 0 the splits of one (split % 8) class run on one XCD)
    const int prow_idx = (p.fix.ctr && p.segs % 8 == 0) ? splitk_leaf_mod8(split, p.segs) : split;
    float* prow_base = p.partial + (size_t)prow_idx * p.prow;
    auto store_partial = [&](auto agent_) {           // agent-scope stores when the partials are summed inside this launch (splitk_fixup)
        constexpr bool AG = decltype(agent_)::value;
        if (do_bias && j == 0) {
#pragma unroll
            for (int c = 0; c < COT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + c * 16 + kg * 4 + r;
                    if (co < p.Cout) st_partial<AG>(prow_base + (size_t)p.Cout * p.Cin * 9 + co, accb[c][r]);
                }
        }
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int ci = ci0 + pcit[i] * 16 + j;
            if (!pok[i] || ci >= p.Cin) continue;
#pragma unroll
            for (int c = 0; c < COT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + c * 16 + kg * 4 + r;
                    if (co >= p.Cout) continue;
                    float* dst = prow_base + ((size_t)co * p.Cin + ci) * 9 + 3 * prow[i];
                    st_partial<AG>(dst, acc[i][c][0][r]); st_partial<AG>(dst + 1, acc[i][c][1][r]); st_partial<AG>(dst + 2, acc[i][c][2][r]);
                }
        }
    };
    if (p.fix.ctr) store_partial(std::true_type{}); else store_partial(std::false_type{});
    MTBC_WTS(p, 12);
    if (!p.fix.ctr) return;
    {
        int* fix_flag = reinterpret_cast<int*>(smemi);
        const int nco = min(16 * COT, p.Cout - co0), nci = min(16 * ncit, p.Cin - ci0);
        splitk_fixup<512>(p.fix, p.partial, p.prow, yb, prow_idx, ((size_t)co0 * p.Cin + ci0) * 9, nco, nci * 9, (size_t)p.Cin * 9,
                          (size_t)p.Cout * p.Cin * 9, co0, (BIAS && cib == 0) ? nco : 0, fix_flag);
    }
}

// fp32 planar (N,C,H,W) <-> 16-bit channel-blocked [N][C/8][H*W][8]; one thread = one 16-byte piece
template <bool F16>
__global__ void c8_pack_kernel(const float* __restrict__ x, long long xbs, unsigned short* __restrict__ y, int C, int HW, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;          // (n, grp, px)
    if (idx >= total) return;
    const int px = (int)(idx % HW); const long long t = idx / HW;
    const int grp = (int)(t % (C / 8)); const long long n = t / (C / 8);
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = x[n * xbs + (size_t)(grp * 8 + e) * HW + px];
    *reinterpret_cast<typename LP<F16>::frag*>(y + idx * 8) = LP<F16>::pack(f);
}
template <bool F16>
__global__ void c8_unpack_kernel(const unsigned short* __restrict__ x, float* __restrict__ y, int C, int HW, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int px = (int)(idx % HW); const long long t = idx / HW;
    const int grp = (int)(t % (C / 8)); const long long n = t / (C / 8);
    const typename LP<F16>::frag v = *reinterpret_cast<const typename LP<F16>::frag*>(x + idx * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) y[(n * C + grp * 8 + e) * HW + px] = (float)v[e];
}

// 16-bit planar -> 16-bit channel-blocked: one thread = 4 pixels x 8 channels (8 loads of 8 bytes, 4 stores of 16)
__global__ void c8_pack16_kernel(const unsigned short* __restrict__ x, long long xbs, unsigned short* __restrict__ y, int C, int HW, long long total) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;          // (n, grp, px/4)
    if (idx >= total) return;
    const int hw4 = HW >> 2;
    const int q = (int)(idx % hw4); const long long t = idx / hw4;
    const int grp = (int)(t % (C / 8)); const long long n = t / (C / 8);
    const unsigned short* src = x + n * xbs + (size_t)(grp * 8) * HW + 4 * q;
    uint2 v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = *reinterpret_cast<const uint2*>(src + (size_t)c * HW);
    u32x4* dst = reinterpret_cast<u32x4*>(y + ((n * (C / 8) + grp) * HW + 4 * q) * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned a = e < 2 ? v[2 * k].x : v[2 * k].y, b = e < 2 ? v[2 * k + 1].x : v[2 * k + 1].y;
            o[k] = (e & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
        }
        dst[e] = o;
    }
}

// Weight views for the gathered backward (engine: one dgrad launch per INPUT tensor over the dz of all its 3x3
// consumers).  mode 0: channel slice  dst[co][ci][t] = w[co][off + ci][t]                    (Cout, cnt, 3, 3)
//              mode 1: the slice as the weight of the equivalent FORWARD conv over dz, placed at K offset koff of a
//                      (cnt, K, 3, 3) tensor:  dst[ci][koff + co][t] = w[co][off + ci][8 - t]
__global__ void wview_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout, int Cin, int off, int cnt, int mode, int koff, int K) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cout * cnt * 9) return;
    const int t = idx % 9, ci = (idx / 9) % cnt, co = idx / (9 * cnt);
    const float v = w[((size_t)co * Cin + off + ci) * 9 + (mode ? 8 - t : t)];
    if (mode) dst[((size_t)ci * K + koff + co) * 9 + t] = v;
    else dst[((size_t)co * cnt + ci) * 9 + t] = v;
}

// Every weight view of a step in ONE launch (a U-Net++ step rebuilds 25 of them after each optimizer update; as 25 launches
// of a few KB each that is ~0.25 ms of launch latency).  Descriptors travel by value in the kernel arguments.
constexpr int WVIEW_MANY = 48;
struct WViewManyP {
    const float* w[WVIEW_MANY];
    float* dst[WVIEW_MANY];
    int Cout[WVIEW_MANY], Cin[WVIEW_MANY], off[WVIEW_MANY], cnt[WVIEW_MANY], koff[WVIEW_MANY], K[WVIEW_MANY];
    unsigned char mode[WVIEW_MANY];
    int first_block[WVIEW_MANY + 1];
    int n;
};
__global__ void wview_many_kernel(const WViewManyP q) {
    int d = 0;
    while (d + 1 < q.n && (int)blockIdx.x >= q.first_block[d + 1]) ++d;          // scalar, uniform
    const int idx = ((int)blockIdx.x - q.first_block[d]) * blockDim.x + threadIdx.x;
    const int Cout = q.Cout[d], Cin = q.Cin[d], off = q.off[d], cnt = q.cnt[d], mode = q.mode[d], koff = q.koff[d], K = q.K[d];
    if (idx >= Cout * cnt * 9) return;
    const float* __restrict__ w = q.w[d];
    float* __restrict__ dst = q.dst[d];
    const int t = idx % 9, ci = (idx / 9) % cnt, co = idx / (9 * cnt);
    const float v = w[((size_t)co * Cin + off + ci) * 9 + (mode ? 8 - t : t)];
    if (mode) dst[((size_t)ci * K + koff + co) * 9 + t] = v;
    else dst[((size_t)co * cnt + ci) * 9 + t] = v;
}

// ------------------------------------------------------------------ direct (VALU) fallbacks
// One thread = one pixel x 8 output channels.  mode 0: fwd  (w[co][ci][tap])
//                                              mode 1: dgrad (w[ci_in][co_out][8-tap], in = dz)
struct DirP {
    int N, H, W, Cin, Cout;    // Cin = channels of the tensor being read, Cout = channels written
    SegTable in, out;
    const float* w;
    const float* bias;
    int mode, wCin;            // wCin = Cin of the torch weight tensor (forward sense)
};
__global__ void conv3x3_direct_kernel(const DirP p) {
    const int HW = p.H * p.W;
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    const int cog = blockIdx.y * 8, n = blockIdx.z;
    if (pix >= HW) return;
    const int y = pix / p.W, x = pix % p.W;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int ci = 0; ci < p.Cin; ++ci) {
        const SegRef si = seg_ref(p.in, ci);
        const float* src = si.ptr + (size_t)n * si.bs + (size_t)(ci - si.cb) * HW;
        float v[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            v[tap] = (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) ? src[yy * p.W + xx] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int co = cog + i;
            if (co >= p.Cout) break;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float wv = p.mode == 0 ? p.w[((size_t)co * p.wCin + ci) * 9 + tap]
                                             : p.w[((size_t)ci * p.wCin + co) * 9 + (8 - tap)];
                acc[i] = fmaf(wv, v[tap], acc[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int co = cog + i;
        if (co >= p.Cout) break;
        const SegRef so = seg_ref(p.out, co);
        float* dst = so.ptr + (size_t)n * so.bs + (size_t)(co - so.cb) * HW + pix;
        float v = acc[i] + (p.bias ? p.bias[co] : 0.f);
        if (so.acc) v += *dst;
        *dst = v;
    }
}

// Stem convolution (Cin == 1, e.g. 1 -> 24 @256x256): K = 9 is not a dense contraction, the layer is bound by writing
// Cout planes.  One thread = 4 consecutive pixels x 8 output channels: the 3 x 6 input window sits in registers and
// every plane gets 16-byte stores (the generic direct kernel wrote 4 bytes per thread per plane: 0.8 TB/s).  Same
// fmaf order over the taps as conv3x3_direct_kernel -> bit-identical results.
__global__ void conv3x3_stem_fwd_kernel(const float* __restrict__ x, long long xbs, const float* __restrict__ w,
                                        const float* __restrict__ bias, float* __restrict__ out, int N, int H, int W, int Cout) {
    const int w4 = W >> 2;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;           // float4 index inside a plane
    const int cog = blockIdx.y * 8, n = blockIdx.z;
    if (q >= H * w4) return;
    const int y = q / w4, x0 = (q % w4) * 4;
    const float* src = x + (size_t)n * xbs;
    float v[3][6];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int yy = y + r - 1;
        const bool rowok = yy >= 0 && yy < H;
        const float4 mid = rowok ? *reinterpret_cast<const float4*>(src + (size_t)yy * W + x0) : make_float4(0.f, 0.f, 0.f, 0.f);
        v[r][0] = (rowok && x0 > 0) ? src[(size_t)yy * W + x0 - 1] : 0.f;
        v[r][1] = mid.x; v[r][2] = mid.y; v[r][3] = mid.z; v[r][4] = mid.w;
        v[r][5] = (rowok && x0 + 4 < W) ? src[(size_t)yy * W + x0 + 4] : 0.f;
    }
    const size_t HW = (size_t)H * W;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int co = cog + i;
        if (co >= Cout) break;
        float wk[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wk[t] = w[co * 9 + t];
        float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = fmaf(wk[t], v[t / 3][e + t % 3], a[e]);
        const float b = bias ? bias[co] : 0.f;
        *reinterpret_cast<float4*>(out + ((size_t)n * Cout + co) * HW + (size_t)y * W + x0) = make_float4(a[0] + b, a[1] + b, a[2] + b, a[3] + b);
    }
}

// The stem in the 16-bit modes: same arithmetic (fp32 operands -- a 1-channel image has no 16-bit operand tensor -- same fmaf order),
// but the output goes out as the cell's conv output goes out everywhere else: channel-blocked 16-bit (OF16: fp16, saturated; else
// bf16), one 16-byte piece = the thread's 8 channels of one pixel, + the InstanceNorm statistics of the stored values:
// per block {sum, sum of squares} of its 128 x 4 pixels -> stats[n][blockIdx.x][co][2] (slots = gridDim.x).
template <bool OF16>
__global__ __launch_bounds__(128) void conv3x3_stem_fwd_c8_kernel(const float* __restrict__ x, long long xbs, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, unsigned short* __restrict__ out,
                                                                  float* __restrict__ stats, int N, int H, int W, int Cout) {
    using TO = LP<OF16>;
    typedef unsigned st_u32x4 __attribute__((ext_vector_type(4)));
    __shared__ float red[2][16];
    const int w4 = W >> 2;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;           // 4-pixel group inside a plane
    const int cog = blockIdx.y * 8, n = blockIdx.z;
    const bool live = q < H * w4;
    const int y = live ? q / w4 : 0, x0 = live ? (q % w4) * 4 : 0;
    const float* src = x + (size_t)n * xbs;
    float v[3][6];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int yy = y + r - 1;
        const bool rowok = live && yy >= 0 && yy < H;
        const float4 mid = rowok ? *reinterpret_cast<const float4*>(src + (size_t)yy * W + x0) : make_float4(0.f, 0.f, 0.f, 0.f);
        v[r][0] = (rowok && x0 > 0) ? src[(size_t)yy * W + x0 - 1] : 0.f;
        v[r][1] = mid.x; v[r][2] = mid.y; v[r][3] = mid.z; v[r][4] = mid.w;
        v[r][5] = (rowok && x0 + 4 < W) ? src[(size_t)yy * W + x0 + 4] : 0.f;
    }
    const size_t HW = (size_t)H * W;
    float o[4][8];                  // [pixel][channel]
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int co = cog + i;                      // Cout % 8 == 0
        float wk[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wk[t] = w[co * 9 + t];
        float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = fmaf(wk[t], v[t / 3][e + t % 3], a[e]);
        const float b = bias ? bias[co] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e][i] = OF16 ? __builtin_amdgcn_fmed3f(a[e] + b, -65504.f, 65504.f) : a[e] + b;
    }
    float ss[8], sq[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { ss[i] = 0.f; sq[i] = 0.f; }
    unsigned short* dst = out + (((size_t)n * (Cout / 8) + blockIdx.y) * HW + (size_t)y * W + x0) * 8;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const st_u32x4 u = __builtin_bit_cast(st_u32x4, TO::pack(o[e]));
        if (live) *reinterpret_cast<st_u32x4*>(dst + e * 8) = u;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const unsigned uu = u[h];
            const float lo = live ? TO::lo(uu) : 0.f, hi = live ? TO::hi(uu) : 0.f;
            ss[2 * h] += lo; sq[2 * h] += lo * lo; ss[2 * h + 1] += hi; sq[2 * h + 1] += hi * hi;
        }
    }
    if (stats) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { ss[i] = wave_sum_rows(ss[i]); sq[i] = wave_sum_rows(sq[i]); }      // (16 butterflies = 96 ds_bpermute round trips: more than the convolution)
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { red[wv][2 * i] = ss[i]; red[wv][2 * i + 1] = sq[i]; }
        }
        __syncthreads();
        if (threadIdx.x < 16)
            stats[(((size_t)n * gridDim.x + blockIdx.x) * Cout + cog) * 2 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x];
    }
}

// The stem's weight gradient with dz 16-bit channel-blocked (what the cell's InstanceNorm backward writes in the 16-bit modes) and the
// 1-channel input fp32: block = (image, 8-channel group, band of 4-pixel groups); a thread keeps 8 channels x 9 taps of sums over
// its pixels (one 16-byte piece of dz per pixel), block-reduced at the end.  partial[(n * S + band)][co][9], as the fp32 kernel's.
template <bool F16>
__global__ __launch_bounds__(256) void conv3x3_wgrad_stem_c8_kernel(const float* __restrict__ x, long long xbs, const unsigned short* __restrict__ dz8,
                                                                    float* __restrict__ partial, int N, int H, int W, int Cout, int S) {
    using T = LP<F16>;
    typedef unsigned st_u32x4 __attribute__((ext_vector_type(4)));
    __shared__ float red[32];
    const int G8 = Cout / 8;
    int b = blockIdx.x;
    const int band = b % S; b /= S;
    const int g = b % G8, n = b / G8;
    const int HW = H * W, W4 = W >> 2, n4 = HW >> 2;
    const int q_lo = (int)((long long)n4 * band / S), q_hi = (int)((long long)n4 * (band + 1) / S);
    const float* src = x + (size_t)n * xbs;
    const unsigned short* gz = dz8 + ((size_t)n * G8 + g) * (size_t)HW * 8;
    float acc[8][9];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[c][t] = 0.f;
    for (int q = q_lo + threadIdx.x; q < q_hi; q += blockDim.x) {
        const int y = q / W4, x4 = (q % W4) * 4;
        float xv[3][6];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int yy = y + r - 1;
#pragma unroll
            for (int e = 0; e < 6; ++e) xv[r][e] = 0.f;
            if (yy >= 0 && yy < H) {
                const float* row = src + (size_t)yy * W;
                const float4 c = *reinterpret_cast<const float4*>(row + x4);
                xv[r][1] = c.x; xv[r][2] = c.y; xv[r][3] = c.z; xv[r][4] = c.w;
                xv[r][0] = x4 > 0 ? row[x4 - 1] : 0.f;
                xv[r][5] = x4 + 4 < W ? row[x4 + 4] : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const st_u32x4 u = *reinterpret_cast<const st_u32x4*>(gz + ((size_t)y * W + x4 + e) * 8);
            float ge[8];
#pragma unroll
            for (int h = 0; h < 4; ++h) { const unsigned uu = u[h]; ge[2 * h] = T::lo(uu); ge[2 * h + 1] = T::hi(uu); }
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[c][t] = fmaf(ge[c], xv[t / 3][e + t % 3], acc[c][t]);
        }
    }
    // 72 sums per block: every wave reduces all of its own first, ONE barrier, then 72 threads add the four wave sums in wave order (72
    // block_sum calls = 144 barriers before)
    __shared__ float wsum[4][72];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // wave sum on the VALU: DPP row sums (lanes 15 / 31 / 47 / 63), four lane reads.  (wave_sum's six ds_bpermute rounds per value
            // -- 432 dependent LDS round trips for the 72 values -- took longer than the accumulation itself.)
            const float v = wave_sum_rows(acc[c][t]);
            if (lane == 0) wsum[wid][c * 9 + t] = v;
        }
    __syncthreads();
    if (threadIdx.x < 72) {
        const float tsum = ((wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) + wsum[2][threadIdx.x]) + wsum[3][threadIdx.x];
        partial[(((size_t)n * S + band) * Cout + 8 * g) * 9 + threadIdx.x] = tsum;
    }
}

// wgrad direct: block = (co, ci, split over n); 9 sums per thread, block-reduced.  partial[split][co][ci][9]
struct DirWgP {
    int N, H, W, Cin, Cout, nsplit;
    SegTable in;
    const float* dz;
    float* partial;
};
__global__ void conv3x3_wgrad_direct_kernel(const DirWgP p) {
    __shared__ float red[32];
    const int co = blockIdx.x / p.Cin, ci = blockIdx.x % p.Cin, split = blockIdx.y;
    const int HW = p.H * p.W;
    const SegRef si = seg_ref(p.in, ci);
    float acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = 0.f;
    for (int n = split; n < p.N; n += p.nsplit) {
        const float* src = si.ptr + (size_t)n * si.bs + (size_t)(ci - si.cb) * HW;
        const float* g = p.dz + ((size_t)n * p.Cout + co) * HW;
        for (int pix = threadIdx.x; pix < HW; pix += blockDim.x) {
            const int y = pix / p.W, x = pix % p.W;
            const float gv = g[pix];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
                if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) acc[tap] = fmaf(gv, src[yy * p.W + xx], acc[tap]);
            }
        }
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        float s = block_sum(acc[tap], red);
        if (threadIdx.x == 0) p.partial[(((size_t)split * p.Cout + co) * p.Cin + ci) * 9 + tap] = s;
    }
}

// wgrad for very few input channels (the Cin=1 first layer: K=9 is not a dense contraction, HBM-bound):
// block = one (n, co) plane pair; each thread walks float4 groups of dz and the 3x6 neighbourhood of x.
// partial[n][co][ci][9], summed over n by the split-K reduce.  Needs W % 4 == 0.
// S row bands per plane (S = nsplit / N): a 256x256 plane per block left 3 blocks per CU walking 64 iterations each
// (1.6 TB/s); partial[(n*S + band)][co][ci][9].
__global__ void conv3x3_wgrad_smallcin_kernel(const DirWgP p) {
    __shared__ float red[32];
    const int S = p.nsplit / p.N;
    const int band = blockIdx.x % S, plane = blockIdx.x / S;
    const int n = plane / p.Cout, co = plane % p.Cout;
    const int HW = p.H * p.W, W4 = p.W >> 2, n4 = HW >> 2;
    const int q_lo = (int)((long long)n4 * band / S), q_hi = (int)((long long)n4 * (band + 1) / S);
    const float* g = p.dz + ((size_t)n * p.Cout + co) * HW;
    for (int ci = 0; ci < p.Cin; ++ci) {
        const SegRef si = seg_ref(p.in, ci);
        const float* src = si.ptr + (size_t)n * si.bs + (size_t)(ci - si.cb) * HW;
        float acc[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) acc[i] = 0.f;
#pragma unroll 4
        for (int q = q_lo + threadIdx.x; q < q_hi; q += blockDim.x) {      // unrolled: the loads of 4 iterations in flight
            const int y = q / W4, x4 = (q % W4) * 4;
            const float4 gv = *reinterpret_cast<const float4*>(g + (size_t)y * p.W + x4);
            const float ge[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int yy = y + r - 1;
                float xv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (yy >= 0 && yy < p.H) {
                    const float* row = src + (size_t)yy * p.W;
                    const float4 c = *reinterpret_cast<const float4*>(row + x4);
                    xv[1] = c.x; xv[2] = c.y; xv[3] = c.z; xv[4] = c.w;
                    xv[0] = x4 > 0 ? row[x4 - 1] : 0.f;
                    xv[5] = x4 + 4 < p.W ? row[x4 + 4] : 0.f;
                }
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[r * 3 + s] = fmaf(ge[e], xv[e + s], acc[r * 3 + s]);
            }
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float t = block_sum(acc[tap], red);
            if (threadIdx.x == 0) p.partial[((((size_t)n * S + band) * p.Cout + co) * p.Cin + ci) * 9 + tap] = t;
        }
    }
}

// ------------------------------------------------------------------ host helpers
bool mfma_ok(const mtbc_seg* segs, int nseg, int H, int W) {
    if (W % 4 != 0 || W < 8 || H < 8) return false;
    for (int i = 0; i < nseg; ++i) {
        if (segs[i].channels % KC) return false;
        if (segs[i].batch_stride % 4) return false;
        if ((reinterpret_cast<uintptr_t>(segs[i].ptr) & 15) != 0) return false;
    }
    return true;
}
int pick_geo(int H, int W) { return (W == 8 && H == 8) ? 2 : (W <= 16 ? 1 : 0); }

template <int MT, int GEO>
int launch_igemm(const ConvP& p, int mblocks, hipStream_t st) {
    using G = Geo<GEO>;
    static const bool nodma = mtbc_probe_set("MTBC_NODMA");
    // persistent grid: gridDim.x a multiple of 8 so that the channel blocks of one pixel tile (same blockIdx.x)
    // land on one XCD and share its L2
    if constexpr (GEO != 2) {
        if (!nodma) {
            constexpr int RING = 2;     // measured: occupancy (3 blocks/CU) beats the deeper 3-slot prefetch on every layer
            static const int ring_env = mtbc_probe_int("MTBC_RING", 0);
            const int ring = ring_env ? ring_env : RING;
            const size_t lds = ((size_t)ring * (KC * G::PS + MT * KC * 144) + SEGL_FLOATS + MT * 16) * sizeof(float);
            MTBC_ENSURE_DYN_LDS((&conv3x3_igemm_dma_kernel<MT, GEO, 2>), 160 * 1024);      // > 64 KiB of dynamic LDS: opt-in per kernel and device
            MTBC_ENSURE_DYN_LDS((&conv3x3_igemm_dma_kernel<MT, GEO, 3>), 160 * 1024);
            int gx = ((ring == 3 ? 512 : 768) / mblocks) / 8 * 8;          // <= 2 or 3 resident blocks per CU, one wave of blocks
            if (gx < 8) gx = 8;
            if (gx > p.ntiles) gx = p.ntiles;
            if (ring == 3) hipLaunchKernelGGL((conv3x3_igemm_dma_kernel<MT, GEO, 3>), dim3(gx, mblocks), dim3(256), lds, st, p);
            else hipLaunchKernelGGL((conv3x3_igemm_dma_kernel<MT, GEO, 2>), dim3(gx, mblocks), dim3(256), lds, st, p);
            MTBC_CHECK_LAUNCH();
            return MTBC_OK;
        }
    }
    const size_t lds = (2ull * (KC * G::PS + MT * KC * 144) + SEGL_FLOATS) * sizeof(float);
    MTBC_ENSURE_DYN_LDS((&conv3x3_igemm_kernel<MT, GEO>), (int)lds);
    int gx = (768 / mblocks) / 8 * 8;
    if (gx < 8) gx = 8;
    if (gx > p.ntiles) gx = p.ntiles;
    hipLaunchKernelGGL((conv3x3_igemm_kernel<MT, GEO>), dim3(gx, mblocks), dim3(256), lds, st, p);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}
template <int GEO>
int launch_igemm_mt(int MT, const ConvP& p, int mblocks, hipStream_t st) {
    switch (MT) {
        case 1: return launch_igemm<1, GEO>(p, mblocks, st);
        case 2: return launch_igemm<2, GEO>(p, mblocks, st);
        default: return launch_igemm<3, GEO>(p, mblocks, st);
    }
}

template <int MT, int GEO>
int launch_igemm_lp(const ConvP& p, int mblocks, bool f16, hipStream_t st) {
    using G = GeoLP<GEO>;
    constexpr int HP = G::IMG * (G::TH + 2) * (G::TW + 2);
    const size_t lds = ((size_t)HP * LPROW + (size_t)MT * 9 * 16 * WROW) * 2 + (SEGL_FLOATS + MT * 16) * sizeof(float);
    MTBC_ENSURE_DYN_LDS((&conv3x3_igemm_lp_kernel<MT, GEO, false>), 160 * 1024);
    MTBC_ENSURE_DYN_LDS((&conv3x3_igemm_lp_kernel<MT, GEO, true>), 160 * 1024);
    // one wave of resident blocks: LDS allows 3 blocks per CU for MT <= 2 (51 KB) but only 2 for MT = 3 (62 KB); a grid
    // sized for 3 would run a second, two-thirds-empty round
    const int per_cu = lds * 3 <= 160 * 1024 ? 3 : 2;
    int gx = (256 * per_cu / mblocks) / 8 * 8;
    if (gx < 8) gx = 8;
    if (gx > p.ntiles) gx = p.ntiles;
    const dim3 grid(gx, mblocks);
    if (f16) hipLaunchKernelGGL((conv3x3_igemm_lp_kernel<MT, GEO, true>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((conv3x3_igemm_lp_kernel<MT, GEO, false>), grid, dim3(256), lds, st, p);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}
template <int GEO>
int launch_igemm_lp_mt(int MT, const ConvP& p, int mblocks, bool f16, hipStream_t st) {
    switch (MT) {
        case 1: return launch_igemm_lp<1, GEO>(p, mblocks, f16, st);
        case 2: return launch_igemm_lp<2, GEO>(p, mblocks, f16, st);
        default: return launch_igemm_lp<3, GEO>(p, mblocks, f16, st);
    }
}

template <int MT, int GEO, int NW, int O8>
int launch_igemm_c8(const ConvP& p, int mblocks, bool f16, hipStream_t st) {
    using G = GeoLP<GEO>;
    constexpr int TH = GEO == 0 ? 2 * NW : G::TH;
    constexpr int HP = G::IMG * (TH + 2) * (G::TW + 2), HPP = (HP + 15) / 16 * 16;
    const size_t lds = ((size_t)4 * HPP * 8 + (size_t)MT * 9 * 16 * WROW) * 2 + (SEGL_FLOATS + MT * 16) * sizeof(float);
    MTBC_ENSURE_DYN_LDS((&conv3x3_igemm_c8_kernel<MT, GEO, false, NW, O8>), 160 * 1024);
    MTBC_ENSURE_DYN_LDS((&conv3x3_igemm_c8_kernel<MT, GEO, true, NW, O8>), 160 * 1024);
    static const int per_cu_env = mtbc_probe_int("MTBC_C8_BLOCKS_PER_CU", 0);      // A/B
    // one wave of resident blocks.  The 4-wave image fits 4 per CU (40.6 KB) and the kernel compiles to 128 VGPRs, but measured
    // (v9): 4 resident blocks are no faster than 3 (16.52 vs 16.49 ms per step; single launches 3-15 % slower) -> 3.
    const int per_cu = per_cu_env ? per_cu_env : (NW == 8 ? 2 : (lds * 3 <= 160 * 1024 ? 3 : 2));
    int gx = (256 * per_cu / mblocks) / 8 * 8;
    if (gx < 8) gx = 8;
    if (gx > p.ntiles) gx = p.ntiles;
    const dim3 grid(gx, mblocks);
#ifdef MTBC_PROBES
    // MTBC_C8_TS=1: phase timestamps of the first three tiles of every block (thread 0) -- a timing microscope, not a product path
    static const int ts_env = mtbc_probe_int("MTBC_C8_TS", 0);
    if (ts_env && (size_t)gx * mblocks <= 4096) {
        ConvP q = p;
        const size_t nb = (size_t)gx * mblocks;
        static unsigned long long* dts = nullptr;
        if (!dts) (void)hipMalloc(&dts, 4096 * 16 * sizeof(unsigned long long));
        (void)hipMemsetAsync(dts, 0, nb * 16 * sizeof(unsigned long long), st);
        q.ts = dts;
        if (f16) hipLaunchKernelGGL((conv3x3_igemm_c8_kernel<MT, GEO, true, NW, O8>), grid, dim3(64 * NW), lds, st, q);
        else hipLaunchKernelGGL((conv3x3_igemm_c8_kernel<MT, GEO, false, NW, O8>), grid, dim3(64 * NW), lds, st, q);
        (void)hipStreamSynchronize(st);
        static unsigned long long hts[4096 * 16];
        (void)hipMemcpy(hts, dts, nb * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull;
        for (size_t b = 0; b < nb; ++b) if (hts[b * 16] && hts[b * 16] < t0) t0 = hts[b * 16];
        double mean[15] = {0}; int cnt[15] = {0};
        for (size_t b = 0; b < nb; ++b)
            for (int k = 0; k < 15; ++k) if (hts[b * 16 + k]) { mean[k] += (double)(hts[b * 16 + k] - t0) * 0.01; ++cnt[k]; }
        for (int k = 0; k < 15; ++k) if (cnt[k]) mean[k] /= cnt[k];
        fprintf(stderr, "c8_ts %d->%d @%dx%d MT%d NW%d O8=%d blocks %zu tiles %d | mean us since the first block's start:", p.Cin, p.Cout, p.H, p.W, MT, NW, O8, nb, p.ntiles);
        for (int k = 0; k < 3; ++k)
            fprintf(stderr, "  tile%d: start %.2f issued %.2f landed %.2f mfma done %.2f epilogue done %.2f", k, mean[4 * k], mean[4 * k + 1], mean[4 * k + 2], mean[4 * k + 3], mean[12 + k]);
        fprintf(stderr, "\n");
        return MTBC_OK;
    }
#endif
    if (f16) hipLaunchKernelGGL((conv3x3_igemm_c8_kernel<MT, GEO, true, NW, O8>), grid, dim3(64 * NW), lds, st, p);
    else hipLaunchKernelGGL((conv3x3_igemm_c8_kernel<MT, GEO, false, NW, O8>), grid, dim3(64 * NW), lds, st, p);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}
template <int GEO, int O8>
int launch_igemm_c8_mt(int MT, const ConvP& p, int mblocks, bool f16, hipStream_t st) {
    if (MT == 1) return launch_igemm_c8<1, GEO, 4, O8>(p, mblocks, f16, st);
    if (MT == 3) return launch_igemm_c8<3, GEO, 4, O8>(p, mblocks, f16, st);
    return launch_igemm_c8<2, GEO, 4, O8>(p, mblocks, f16, st);
}

template <int MT, int GEO, int O8>
int launch_igemm_c8_ring(const ConvP& p, int mblocks, bool f16, hipStream_t st) {
    constexpr int R = 3;
    using RG = RingGeo<MT, GEO>;
    const size_t lds = (size_t)R * RG::SLOT * 2 + (SEGL_FLOATS + MT * 16) * sizeof(float) + 64 * 4 * sizeof(unsigned long long);      // + the (chunk, group) base table
    static_assert((size_t)R * RG::SLOT * 2 + (SEGL_FLOATS + MT * 16) * sizeof(float) + 2048 <= 160 * 1024, "ring slots + tables must fit a CU's LDS");
    MTBC_ENSURE_DYN_LDS((&conv3x3_igemm_c8_ring_kernel<MT, GEO, false, O8, R>), 160 * 1024);
    MTBC_ENSURE_DYN_LDS((&conv3x3_igemm_c8_ring_kernel<MT, GEO, true, O8, R>), 160 * 1024);
    int gx = 256 / mblocks;                  // one resident block per CU (147 KB of LDS), one wave of blocks
    if (gx < 1) gx = 1;
    if (gx > p.ntiles) gx = p.ntiles;
    const dim3 grid(gx, mblocks);
#ifdef MTBC_PROBES
    // MTBC_RING_TS=1: phase timestamps of every block (thread 0), printed after a synchronisation -- a timing microscope, not a product path
    static const int ts_env = mtbc_probe_int("MTBC_RING_TS", 0);
    if (ts_env) {
        ConvP q = p;
        const size_t nb = (size_t)gx * mblocks;
        static unsigned long long* dts = nullptr;
        if (!dts) (void)hipMalloc(&dts, 4096 * 16 * sizeof(unsigned long long));
        (void)hipMemsetAsync(dts, 0, nb * 16 * sizeof(unsigned long long), st);
        q.ts = dts;
        if (f16) hipLaunchKernelGGL((conv3x3_igemm_c8_ring_kernel<MT, GEO, true, O8, R>), grid, dim3(256), lds, st, q);
        else hipLaunchKernelGGL((conv3x3_igemm_c8_ring_kernel<MT, GEO, false, O8, R>), grid, dim3(256), lds, st, q);
        (void)hipStreamSynchronize(st);
        static unsigned long long hts[4096 * 16];
        (void)hipMemcpy(hts, dts, nb * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t1 = 0;
        for (size_t b = 0; b < nb; ++b) { if (hts[b * 16] < t0) t0 = hts[b * 16]; if (hts[b * 16 + 8] > t1) t1 = hts[b * 16 + 8]; }
        double mean[9] = {0}, mx[9] = {0};
        for (size_t b = 0; b < nb; ++b)
            for (int k = 0; k < 9; ++k) { const double d = (double)(hts[b * 16 + k] - t0) * 0.01; mean[k] += d / nb; if (d > mx[k]) mx[k] = d; }
        fprintf(stderr, "ring_ts %d->%d @%dx%d MT%d O8=%d blocks %zu span %.2f us | mean (max) us since the first block's entry: entry %.2f (%.2f) prologue %.2f (%.2f) issued %.2f (%.2f) "
                        "chunk0 landed %.2f (%.2f) chunk1 %.2f (%.2f) last chunk %.2f (%.2f) mfma done %.2f (%.2f) stores issued %.2f (%.2f) stores done %.2f (%.2f)\n",
                p.Cin, p.Cout, p.H, p.W, MT, O8, nb, (double)(t1 - t0) * 0.01, mean[0], mx[0], mean[1], mx[1], mean[2], mx[2], mean[3], mx[3], mean[4], mx[4], mean[5], mx[5],
                mean[6], mx[6], mean[7], mx[7], mean[8], mx[8]);
        return MTBC_OK;
    }
#endif
    if (f16) hipLaunchKernelGGL((conv3x3_igemm_c8_ring_kernel<MT, GEO, true, O8, R>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((conv3x3_igemm_c8_ring_kernel<MT, GEO, false, O8, R>), grid, dim3(256), lds, st, p);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}
template <int GEO, int O8>
int launch_igemm_c8_ring_mt(int MT, const ConvP& p, int mblocks, bool f16, hipStream_t st) {
    if (MT == 1) return launch_igemm_c8_ring<1, GEO, O8>(p, mblocks, f16, st);
    if (MT == 3) return launch_igemm_c8_ring<3, GEO, O8>(p, mblocks, f16, st);
    return launch_igemm_c8_ring<2, GEO, O8>(p, mblocks, f16, st);
}

// geometry of an igemm launch: map geometry, pixel tiles, channel tiles per block, 8-wave (16 x 32 pixel) blocks, LDS-ring kernel
struct IgemmPlan { int geo, tiles_x, tiles_y, ntiles, mtiles, MT, mblocks; bool nw8, ring; };
IgemmPlan plan_igemm(int N, int H, int W, int rows, int compute, bool c8, bool allow_nw8 = true, int red = 0) {
    IgemmPlan q{};
    q.geo = pick_geo(H, W);
    q.mtiles = cdiv(rows, 16);
    if (q.geo == 0) { q.tiles_x = cdiv(W, 32); q.tiles_y = cdiv(H, 8); q.ntiles = q.tiles_x * q.tiles_y * N; }
    else if (q.geo == 1) { q.tiles_x = cdiv(W, 16); q.tiles_y = cdiv(H, 16); q.ntiles = q.tiles_x * q.tiles_y * N; }
    else { q.tiles_x = 1; q.tiles_y = 1; q.ntiles = cdiv(N, 4); }
    q.nw8 = false; q.ring = false;
    static const int ring_env = mtbc_probe_int("MTBC_C8_RING", -1);      // A/B
    if (c8 && allow_nw8 && q.geo != 2 && red >= 96 && red <= 64 * LPKC && ring_env != 0) {      // (<= 64 chunks: the kernel's base table)
        // deep levels: a launch of at most ONE block per CU -- the ring kernel (chunks prefetched two ahead) with as many channel
        // tiles per block as divide evenly (the pixel tile is staged once per block).  Measured (U-Net++ B=32, 16 x 16 maps):
        // 192->384 29 -> 23 us, 384->384 40 -> 35 us; with two rounds of blocks (512->512: 512 blocks) it LOSES, 50 -> 60 us, and
        // 1152->512 85 -> 121 us: these launches are bound by the L2 -> LDS fill rate (~4.3 TB/s over the chip, 2.4x the
        // algorithmic bytes), not by its latency, and one block per CU cannot overlap its rounds
        const int mtr = q.mtiles % 3 == 0 ? 3 : (q.mtiles >= 2 ? 2 : 1);
        const int mb = cdiv(q.mtiles, mtr);
        if ((long long)q.ntiles * mb <= 256) { q.ring = true; q.MT = mtr; q.mblocks = mb; return q; }
    }
    // channel tiles per block: up to 3 (4 spills past 256 VGPRs), fewer when the launch would not fill 256 CUs twice over
    static const int mtmax_env = mtbc_probe_int("MTBC_LP_MT", 0);      // A/B probe
    // 16-bit kernels: 2 tiles per block keep LDS at 51 KB = 3 blocks per CU; 3 tiles (62 KB, 2 blocks) measured slower
    // on every layer (dgrad 144->24: 0.71 -> 0.56 ms) although the pixel tile is staged once more per channel block
    // ... except on the channel-blocked kernel when the channel tiles come in threes (48 / 96 / 192 / 384 channels): 3 tiles
    // per block are 49 KB there (still 3 blocks per CU) and the pixel tile is staged once per 48 channels instead of once
    // per 32 (Cout = 48: X read once instead of twice).  Not on 8x8 maps: that instantiation spills.
    const int mtmax = compute != 0 ? (mtmax_env ? mtmax_env : ((c8 && q.geo != 2 && q.mtiles % 3 == 0) ? 3 : 2)) : 3;
    q.mblocks = cdiv(q.mtiles, mtmax);
    q.MT = cdiv(q.mtiles, q.mblocks);
    while (q.MT > 1 && (long long)q.ntiles * q.mblocks < 512) {
        --q.MT;
        q.mblocks = cdiv(q.mtiles, q.MT);
    }
    q.MT = cdiv(q.mtiles, q.mblocks);
    if (c8) {
        // wide maps with enough tiles for two 512-pixel blocks per CU several times over: 16 x 32 tiles, 8 waves
        static const int nw_env = mtbc_probe_int("MTBC_C8_NW", 0);      // A/B
        const int t16 = q.tiles_x * cdiv(H, 16) * N;
        if (allow_nw8 && q.geo == 0 && q.MT == 2 && (nw_env ? nw_env == 8 : (long long)t16 * q.mblocks >= 2048)) {
            q.nw8 = true; q.tiles_y = cdiv(H, 16); q.ntiles = t16;
        }
        // (round 4: the same 16 x 32 tiles for the 48-channel blocks (MT = 3) of the 128 x 128 maps -- 1024 tiles on 512 8-wave blocks = exactly two each, where
        //  2048 tiles of 8 x 32 on 768 4-wave blocks are 2.67 -- were built and measured: 11.53 / 11.53 / 11.54 ms per step against 11.46 / 11.45 / 11.45.
        //  Slower; removed.  profiles/r04_ab.txt)
    }
    return q;
}

// shared by fwd and dgrad: `rows` = channels written, `red` = channels read; compute: 0 fp32, 1 bf16, 2 fp16 operands;
// c8: the tensor read is 16-bit channel-blocked (MTBC_LAYOUT_C8); o8: so is the tensor written (+ optional epilogue statistics)
int run_igemm(int N, int H, int W, int red, int rows, const SegTable& in, const SegTable& out, const float* wp,
              const float* bias, int compute, hipStream_t st, bool c8 = false, int o8 = 0, float* stats = nullptr,
              const mtbc_conv3x3_args* nb = nullptr) {
    ConvP p;
#ifdef MTBC_PROBES
    p.ts = nullptr;
#endif
    p.N = N; p.H = H; p.W = W; p.Cin = red; p.Cout = rows; p.in = in; p.out = out; p.wp = wp; p.bias = bias; p.stats = stats;
    p.extra = nullptr; p.nz = nullptr; p.nmean = p.nrstd = p.ngamma = p.nbeta = nullptr; p.nslope = 0.f;
    if (o8 == 2) {
        p.extra = nb->out_partial; p.nz = reinterpret_cast<const unsigned short*>(nb->norm_z); p.nmean = nb->norm_mean; p.nrstd = nb->norm_rstd;
        p.ngamma = nb->norm_gamma; p.nbeta = nb->norm_beta; p.nslope = nb->norm_slope;
    }
    static const int dbg = mtbc_probe_int("MTBC_DBG", 0);
    p.dbg = dbg;
    const IgemmPlan q = plan_igemm(N, H, W, rows, compute, c8, o8 != 2, red);      // (the norm-backward epilogue: neither the 8-wave nor the ring variant)
    const int geo = q.geo, MT = q.MT, mblocks = q.mblocks;
    p.mtiles = q.mtiles; p.tiles_x = q.tiles_x; p.tiles_y = q.tiles_y; p.ntiles = q.ntiles;
    if (c8 && q.ring) {
        if (o8 == 3) return geo == 0 ? launch_igemm_c8_ring_mt<0, 3>(MT, p, mblocks, false, st) : launch_igemm_c8_ring_mt<1, 3>(MT, p, mblocks, false, st);
        if (o8 == 1) return geo == 0 ? launch_igemm_c8_ring_mt<0, 1>(MT, p, mblocks, compute == 2, st) : launch_igemm_c8_ring_mt<1, 1>(MT, p, mblocks, compute == 2, st);
        return geo == 0 ? launch_igemm_c8_ring_mt<0, 0>(MT, p, mblocks, compute == 2, st) : launch_igemm_c8_ring_mt<1, 0>(MT, p, mblocks, compute == 2, st);
    }
    if (c8) {
        if (q.nw8) return o8 == 3 ? launch_igemm_c8<2, 0, 8, 3>(p, mblocks, false, st)
                        : o8 == 1 ? launch_igemm_c8<2, 0, 8, 1>(p, mblocks, compute == 2, st) : launch_igemm_c8<2, 0, 8, 0>(p, mblocks, compute == 2, st);
        if (o8 == 3) {
            if (geo == 0) return launch_igemm_c8_mt<0, 3>(MT, p, mblocks, false, st);
            if (geo == 1) return launch_igemm_c8_mt<1, 3>(MT, p, mblocks, false, st);
            return launch_igemm_c8_mt<2, 3>(MT, p, mblocks, false, st);
        }
        if (o8 == 2) {
            if (geo == 0) return launch_igemm_c8_mt<0, 2>(MT, p, mblocks, compute == 2, st);
            if (geo == 1) return launch_igemm_c8_mt<1, 2>(MT, p, mblocks, compute == 2, st);
            return launch_igemm_c8_mt<2, 2>(MT, p, mblocks, compute == 2, st);
        }
        if (o8 == 1) {
            if (geo == 0) return launch_igemm_c8_mt<0, 1>(MT, p, mblocks, compute == 2, st);
            if (geo == 1) return launch_igemm_c8_mt<1, 1>(MT, p, mblocks, compute == 2, st);
            return launch_igemm_c8_mt<2, 1>(MT, p, mblocks, compute == 2, st);
        }
        if (geo == 0) return launch_igemm_c8_mt<0, 0>(MT, p, mblocks, compute == 2, st);
        if (geo == 1) return launch_igemm_c8_mt<1, 0>(MT, p, mblocks, compute == 2, st);
        return launch_igemm_c8_mt<2, 0>(MT, p, mblocks, compute == 2, st);
    }
    if (compute != 0) {
        if (geo == 0) return launch_igemm_lp_mt<0>(MT, p, mblocks, compute == 2, st);
        if (geo == 1) return launch_igemm_lp_mt<1>(MT, p, mblocks, compute == 2, st);
        return launch_igemm_lp_mt<2>(MT, p, mblocks, compute == 2, st);
    }
    if (geo == 0) return launch_igemm_mt<0>(MT, p, mblocks, st);
    if (geo == 1) return launch_igemm_mt<1>(MT, p, mblocks, st);
    return launch_igemm_mt<2>(MT, p, mblocks, st);
}

struct WgPlan { bool mfma; bool smallcin; int cot; int geo, tiles_x, tiles_y, total_tiles, nsplit, tiles_per_split, coblocks, ciblocks; size_t partial_elems, dbias_elems; bool c8w, c8i, pack24; int cit, segs, seg_tiles, depth; };
// channel-blocked weight gradient on 16 x 16 maps: the image-tile kernel (conv3x3_wgrad_c8i_kernel)
bool c8i_wanted(const mtbc_conv3x3_args* a) {
    if (a->Cin == 1 || a->W != 16 || a->H != 16 || a->Cin % 8 || a->Cout % 8) return false;
    static const int probe = mtbc_probe_int("MTBC_WGRAD_C8I", -1);      // probes build only: 0 = never
    if (probe >= 0) return probe != 0;
    return a->Cin >= 64 && a->Cout >= 32;
}
// channel-blocked weight gradient: which launches take the wide-block kernel (conv3x3_wgrad_c8w_kernel)
bool c8w_wanted(const mtbc_conv3x3_args* a) {
    if (a->Cin == 1 || a->W <= 16 || a->Cin % 8 || a->Cout % 8) return false;
    static const int probe = mtbc_probe_int("MTBC_WGRAD_C8W", -1);      // probes build only: 0 = never, 1 = wherever the kernel can run
    if (probe >= 0) return probe != 0;
    // measured per layer of the U-Net++ step (tools/wgrad_probe.py, profiles/r03_wgrad_probe.txt): the wide blocks win where dz was
    // staged three or more times and the maps are large (72 .. 144 -> 24 @256x256: -2 .. -30 us, 96 .. 192 -> 48 @128x128: -1 .. -25 us);
    // on single-input convs (Cin <= 48: both kernels stage everything once) and on maps <= 64 wide (few steps per block) the
    // 32 x 32 kernel's four small blocks per CU are as fast or faster
    return a->Cin >= 64 && a->W >= 128;
}
// operand_layout = MTBC_LAYOUT_C8: are the 16-bit channel-blocked operands well-formed?
bool c8_segs_ok(const mtbc_seg* segs, int nseg) {
    for (int i = 0; i < nseg; ++i)
        if (!segs[i].ptr || segs[i].channels % 8 || segs[i].batch_stride % 8 || (reinterpret_cast<uintptr_t>(segs[i].ptr) & 15)) return false;
    return true;
}
WgPlan plan_wgrad(const mtbc_conv3x3_args* a) {
    WgPlan w{};
    if (a->operand_layout == MTBC_LAYOUT_C8 && a->Cin == 1) {      // conv3x3_wgrad_stem_c8_kernel: (image, band) splits
        // (image, 8-channel group, band) blocks: 16 bands on 256 x 256 planes -- with 4 a step's stem launch was 384 blocks = 1.5 waves per SIMD,
        // each walking 16 iterations of {load, wait, 288 FMAs}: nobody to run while a wave waits (87 us for 110 MB)
        const int bands = (a->H * a->W >= 65536) ? 16 : (a->H * a->W >= 16384) ? 4 : 1;
        w.nsplit = a->N * bands;
        w.partial_elems = (size_t)w.nsplit * a->Cout * 9;
        w.dbias_elems = 0;
        return w;
    }
    if (a->operand_layout == MTBC_LAYOUT_C8 && c8i_wanted(a)) {
        // conv3x3_wgrad_c8i_kernel: 16 x 16 maps, a tile = one image; (32 | 48 output) x <= 80 input channels per block, image ranges as the
        // split-K dimension (a multiple of 8 ranges where the block budget allows: one XCD per range), one block of 8 waves per CU
        w.mfma = true; w.geo = 1; w.c8i = true;
        w.cot = a->Cout % 48 == 0 ? 3 : 2;
        const int T = cdiv(a->Cin, 16);
        w.coblocks = cdiv(a->Cout, 16 * w.cot); w.ciblocks = cdiv(T, C8WW_MAXCIT); w.cit = cdiv(T, w.ciblocks);
        int ns = 256 / (w.coblocks * w.ciblocks);
        if (ns >= 8) ns &= ~7;
        if (ns > a->N) ns = a->N;
        if (ns < 1) ns = 1;
        w.seg_tiles = cdiv(a->N, ns);                      // images per range
        w.segs = cdiv(a->N, w.seg_tiles);
        w.nsplit = w.segs; w.total_tiles = a->N; w.tiles_per_split = w.seg_tiles;
        w.partial_elems = (size_t)w.nsplit * a->Cout * a->Cin * 9;
        w.dbias_elems = a->dbias ? (size_t)w.nsplit * a->Cout : 0;
        return w;
    }
    if (a->operand_layout == MTBC_LAYOUT_C8 && c8w_wanted(a)) {
        // conv3x3_wgrad_c8w_kernel: (all of a 24 / 48-channel output | 32 / 48 of a wider one) x <= 80 input channels per block; a block
        // walks down a row segment of a 32-column strip of one image; 2 blocks of 8 waves per CU, one wave of <= 512 blocks
        w.mfma = true; w.geo = 0; w.c8w = true;
        w.cot = a->Cout % 48 == 0 ? 3 : 2;
        w.tiles_x = cdiv(a->W, C8WW_TW); w.tiles_y = cdiv(a->H, C8WW_TH);
        const int T = cdiv(a->Cin, 16);
        w.coblocks = cdiv(a->Cout, 16 * w.cot); w.ciblocks = cdiv(T, C8WW_MAXCIT); w.cit = cdiv(T, w.ciblocks);
        // (blocks per CU, depth): the most DMA bytes in flight that the LDS holds -- beyond ~96 KB per CU nothing is gained
        static const int bpc_probe = mtbc_probe_int("MTBC_C8W_BPC", 0), d_probe = mtbc_probe_int("MTBC_C8W_DEPTH", 0);
        int bpc = 2; w.depth = 1; size_t best = 0;
        for (int b = 2; b >= 2; --b)
            for (int d = 1; d <= 1; ++d) {
                if (c8w_lds_bytes(w.cit, w.cot, d) * b > 160 * 1024) continue;
                size_t fl = c8w_step_bytes(w.cit, w.cot) * b * d;
                if (fl > 96 * 1024) fl = 96 * 1024;
                if (fl > best) { best = fl; bpc = b; w.depth = d; }
            }
        if (bpc_probe && d_probe && c8w_lds_bytes(w.cit, w.cot, d_probe) * bpc_probe <= 160 * 1024) { bpc = bpc_probe; w.depth = d_probe; }
        const int strips = w.tiles_x * a->N, budget = 256 * bpc / (w.coblocks * w.ciblocks);
        int segs = budget / strips;                        // row segments per strip: as many as the block budget allows ...
        if (segs < 1) segs = 1;
        if (segs > w.tiles_y) segs = w.tiles_y;
        w.seg_tiles = cdiv(w.tiles_y, segs);
        if (w.seg_tiles < 2 && w.tiles_y >= 2) w.seg_tiles = 2;        // ... but a segment re-stages two halo rows: at least 8 rows of its own
        w.segs = cdiv(w.tiles_y, w.seg_tiles);
        w.nsplit = strips * w.segs;
        w.total_tiles = w.nsplit; w.tiles_per_split = 1;
        w.partial_elems = (size_t)w.nsplit * a->Cout * a->Cin * 9;
        w.dbias_elems = a->dbias ? (size_t)w.nsplit * a->Cout : 0;
        return w;
    }
    if (a->operand_layout == MTBC_LAYOUT_C8) {      // conv3x3_wgrad_c8_kernel: 32 x 32 channel blocks, 4 x 32 pixel tiles, 4 blocks per CU
        w.mfma = true; w.geo = a->W <= 16 ? 1 : 0; w.cot = 2;
        w.tiles_x = cdiv(a->W, w.geo ? C8WGeo<1>::TW : C8WGeo<0>::TW); w.tiles_y = cdiv(a->H, w.geo ? C8WGeo<1>::TH : C8WGeo<0>::TH);
        w.total_tiles = w.tiles_x * w.tiles_y * a->N;
        w.coblocks = cdiv(a->Cout, 32); w.ciblocks = cdiv(a->Cin, 32);
        int ns = 1024 / (w.coblocks * w.ciblocks);
        if (ns > w.total_tiles) ns = w.total_tiles;
        if (ns < 1) ns = 1;
        // The (co, ci) blocks of one split read the SAME pixels -- X once per output-channel block, dz once per input-channel block -- and all of a
        // launch's blocks are resident at once (one wave of <= 1024), walking their tiles side by side.  Block (split, pair) is workgroup
        // pair * nsplit + split, and workgroups go to the 8 XCDs round-robin: with nsplit a multiple of 8 every block of a split sits on XCD split % 8
        // and the re-reads are hits in THAT XCD's L2; with the split counts the plain division gives (103, 54, 37, 52, 26 on the 64 x 64 / 32 x 32 levels)
        // they are spread over all eight and every one of them is a miss (round 4: this kernel moved 4.2 GB per step for 1.95 algorithmic,
        // profiles/r04_hbm_traffic_bf16.json).  So: a multiple of 8 splits, tiles dealt evenly (tiles_per_split < 0 in the kernel).
        // Launch by launch (profiles/r04_wgrad_xcd_splits.txt): 96 -> 96 @64x64 46.7 -> 43.1 us, 192 -> 96 68.8 -> 66.1, 288 -> 96 97.8 -> 90.4; on 32 x 32 maps
        // (256 tiles: the rounding costs blocks) 1 - 2 us SLOWER, so only from 1024 tiles up.  The step: -0.03 ms.
        static const int ns8_probe = mtbc_probe_int("MTBC_WG_NS8", 16);      // probes build, A/B: smallest split count that is rounded (0 = never)
        if (ns8_probe > 0 && ns >= ns8_probe && ns >= 8 && w.total_tiles >= 1024) {
            w.nsplit = ns & ~7;
            w.tiles_per_split = -1;
        } else {
            w.tiles_per_split = cdiv(w.total_tiles, ns);
            w.nsplit = cdiv(w.total_tiles, w.tiles_per_split);
        }
        w.partial_elems = (size_t)w.nsplit * a->Cout * a->Cin * 9;
        w.dbias_elems = a->dbias ? (size_t)w.nsplit * a->Cout : 0;
        return w;
    }
    w.mfma = !a->force_direct && mfma_ok(a->in, a->n_in, a->H, a->W) && a->Cin >= 8 &&
             (reinterpret_cast<uintptr_t>(a->dout) & 15) == 0;
    const size_t wel = (size_t)a->Cout * a->Cin * 9;
    if (w.mfma) {
        w.geo = pick_geo(a->H, a->W);
        int tn;
        if (w.geo == 0) { w.tiles_x = cdiv(a->W, 32); w.tiles_y = cdiv(a->H, 4); tn = a->N; }
        else if (w.geo == 1) { w.tiles_x = cdiv(a->W, 16); w.tiles_y = cdiv(a->H, 8); tn = a->N; }
        else { w.tiles_x = 1; w.tiles_y = 1; tn = cdiv(a->N, 2); }
        w.total_tiles = w.tiles_x * w.tiles_y * tn;
        // 48-channel output blocks (3 tiles, 384 threads) when that pads less than 32-channel blocks (Cout = 48)
        const int pad2 = cdiv(a->Cout, 32) * 32, pad3 = cdiv(a->Cout, 48) * 48;
        w.cot = pad3 < pad2 ? 3 : 2;
        if (w.geo == 0 && a->compute != 0) w.cot = 2;      // conv3x3_wgrad_lp2_kernel: 32 x 32 channel blocks
        w.coblocks = cdiv(a->Cout, 16 * w.cot); w.ciblocks = cdiv(a->Cin, 32);
        // fp32 kernel: input-channel blocks of 24 where Cin is a multiple of 24 and not of 32 (every level-0 / level-1 conv of the U-Net++):
        // 13.5 (tap, ci) column tiles per block instead of 18 -- conv3x3_wgrad_mfma_kernel<.., PACK>
        // Round 3 measured "no faster" on whole-step per-op times (profiles/r03_f32_wgrad_pack24.txt) and kept it behind the probes build.  Round 4,
        // launch by launch (tools/experiments/f32_wgrad_one.py, profiles/r04_f32_wgrad.txt): 24 -> 24 @256 409 -> 346 us, 48 -> 48 @128 354 -> 292,
        // 72 -> 24 1069 -> 932, 144 -> 24 1749 -> 1684 -- the phase stamps say why it has to: three resident blocks keep the matrix pipe busy ~95 % of
        // a tile period, so the MFMAs a launch ISSUES (a third of them padding at Cin = 24) are its time.  On by default; MTBC_WGRAD_PACK24=0 in the
        // probes build restores 32-channel blocks.
        // (Round 4, measured and NOT kept: the two tensors swapped for the level-0 multi-input convs -- rows = the 72 .. 144 input channels, whose 16-row
        //  tiles are full, columns = (tap, the 24 output channels) packed: 13 - 25 % fewer MFMAs issued.  Launch by launch 72 -> 24 942 -> 907 us,
        //  144 -> 24 1680 -> 1584, but 120 -> 24 1432 -> 1556, and the fp32 step 47.38 -> 47.46 ms: with 48-row blocks (6 waves, 2 blocks per CU) the
        //  launches lose in phase overlap what they save in the matrix pipe.  profiles/r04_f32_wgrad.txt)
        static const int pack_probe = mtbc_probe_int("MTBC_WGRAD_PACK24", 1);
        w.pack24 = pack_probe && a->compute == 0 && a->Cin % 24 == 0 && a->Cin % 32 != 0;
        if (w.pack24) w.ciblocks = a->Cin / 24;
        const int pairs = w.coblocks * w.ciblocks;
        // resident blocks per CU: 3 (256 threads, 50 KB LDS) or 2 (384 threads, 58 KB) -- ONE wave of blocks, a
        // block beyond that would double the launch time
        const bool lp2 = w.geo == 0 && a->compute != 0;      // conv3x3_wgrad_lp2_kernel: 216 VGPRs -> 2 blocks per CU
        int ns = (w.cot == 3 || lp2 ? 512 : 768) / pairs;
        if (ns > w.total_tiles) ns = w.total_tiles;
        if (ns < 1) ns = 1;
        w.tiles_per_split = cdiv(w.total_tiles, ns);
        w.nsplit = cdiv(w.total_tiles, w.tiles_per_split);
    } else {
        bool al = a->W % 4 == 0 && (reinterpret_cast<uintptr_t>(a->dout) & 15) == 0;
        for (int i = 0; al && i < a->n_in; ++i)
            al = (reinterpret_cast<uintptr_t>(a->in[i].ptr) & 15) == 0 && a->in[i].batch_stride % 4 == 0;
        w.smallcin = !a->force_direct && a->Cin <= 4 && al;
        const int bands = (a->H * a->W >= 16384) ? 4 : 1;
        w.nsplit = w.smallcin ? a->N * bands : (a->N < 16 ? a->N : 16);
    }
    w.partial_elems = (size_t)w.nsplit * wel;
    w.dbias_elems = a->dbias ? (size_t)a->N * a->Cout : 0;
    return w;
}

}  // namespace

// ====================================================================== C ABI
extern "C" {

size_t mtbc_conv3x3_packed_elems(int32_t Cin, int32_t Cout) { return (size_t)cdiv(Cout, 16) * Cin * 144; }
size_t mtbc_conv3x3_packed_dgrad_elems(int32_t Cin, int32_t Cout) { return (size_t)cdiv(Cin, 16) * Cout * 144; }

int mtbc_conv3x3_pack_fwd(const float* w, float* packed, int32_t Cin, int32_t Cout, void* stream) {
    if (!w || !packed || Cin <= 0 || Cout <= 0) return MTBC_E_BADARG;
    const int total = (int)mtbc_conv3x3_packed_elems(Cin, Cout);
    hipLaunchKernelGGL(pack_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, packed, Cin, Cout, total);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}
int mtbc_conv3x3_pack_dgrad(const float* w, float* packed, int32_t Cin, int32_t Cout, void* stream) {
    if (!w || !packed || Cin <= 0 || Cout <= 0) return MTBC_E_BADARG;
    const int total = (int)mtbc_conv3x3_packed_dgrad_elems(Cin, Cout);
    hipLaunchKernelGGL(pack_dgrad_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, packed, Cin, Cout, total);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}

size_t mtbc_conv3x3_packed_lp_elems(int32_t Cin, int32_t Cout, int32_t dgrad) {
    const int rows = dgrad ? Cin : Cout, red = dgrad ? Cout : Cin;
    return (size_t)cdiv(rows, 16) * cdiv(red, LPKC) * 9 * 16 * WROW;      // 16-bit elements
}
int mtbc_conv3x3_pack_lp(const float* w, void* packed, int32_t Cin, int32_t Cout, int32_t dgrad, int32_t compute, void* stream) {
    if (!w || !packed || Cin <= 0 || Cout <= 0 || (compute != 1 && compute != 2)) return MTBC_E_BADARG;
    const long long total = (long long)mtbc_conv3x3_packed_lp_elems(Cin, Cout, dgrad);
    hipLaunchKernelGGL(pack_lp_kernel, dim3((unsigned)cdiv64(total / 8, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       reinterpret_cast<unsigned short*>(packed), Cin, Cout, dgrad, compute == 2 ? 1 : 0, total / 8);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}

int mtbc_conv3x3_pack_many(const mtbc_pack_desc* descs, int32_t n, void* stream) {
    if (n < 0 || (n > 0 && !descs)) return MTBC_E_BADARG;
    int done = 0;
    while (done < n) {
        PackManyP q;
        q.n = 0; q.f16 = 0;
        int blocks = 0, lp_mode = 0;
        while (done < n && q.n < PACK_MANY) {
            const mtbc_pack_desc& d = descs[done];
            if (!d.w || !d.packed || d.Cin <= 0 || d.Cout <= 0 || d.kind < 0 || d.kind > 3) return MTBC_E_BADARG;
            long long total;
            if (d.kind < 2) total = d.kind == 0 ? (long long)mtbc_conv3x3_packed_elems(d.Cin, d.Cout) : (long long)mtbc_conv3x3_packed_dgrad_elems(d.Cin, d.Cout);
            else {
                if (d.compute != 1 && d.compute != 2) return MTBC_E_BADARG;
                if (lp_mode && lp_mode != d.compute) break;          // one 16-bit format per launch
                lp_mode = d.compute;
                total = (long long)mtbc_conv3x3_packed_lp_elems(d.Cin, d.Cout, d.kind == 3) / 8;      // threads: 8 elements each
            }
            const int i = q.n++;
            q.w[i] = d.w; q.dst[i] = d.packed; q.Cin[i] = d.Cin; q.Cout[i] = d.Cout; q.kind[i] = (unsigned char)d.kind;
            q.first_block[i] = blocks;
            if (d.kind < 2) blocks += (int)cdiv64(total, 256);
            else blocks += (int)(total / 576);          // 16-bit images: one block per (16-row tile, 32-K chunk) unit of 9 x 16 x 4 pieces
            ++done;
        }
        q.first_block[q.n] = blocks;
        q.f16 = lp_mode == 2;
        hipLaunchKernelGGL(pack_many_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, q);
        MTBC_CHECK_LAUNCH();
    }
    return MTBC_OK;
}

static int check_conv(const mtbc_conv3x3_args* a) {
    if (!a) return MTBC_E_BADARG;
    if (a->N <= 0 || a->H <= 0 || a->W <= 0 || a->Cin <= 0 || a->Cout <= 0) return MTBC_E_BADSHAPE;
    return MTBC_OK;
}

int mtbc_conv3x3_fwd(const mtbc_conv3x3_args* a, void* stream) {
    int rc = check_conv(a); if (rc) return rc;
    if (!a->out || (!a->w && !a->w_packed)) return MTBC_E_BADARG;
    SegTable in, out;
    rc = make_segtable(a->in, a->n_in, a->Cin, &in); if (rc) return rc;
    mtbc_seg o{a->out, (int64_t)a->Cout * a->H * a->W, a->Cout, a->out_accumulate ? 1 : 0};
    rc = make_segtable(&o, 1, a->Cout, &out); if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (a->out_accumulate && a->operand_layout != MTBC_LAYOUT_C8) return MTBC_E_UNSUPPORTED;
    const bool stem = a->Cin == 1 && a->n_in == 1 && !a->force_direct && a->W % 4 == 0 && a->in[0].batch_stride % 4 == 0 &&
                      (reinterpret_cast<uintptr_t>(a->in[0].ptr) & 15) == 0 && (reinterpret_cast<uintptr_t>(a->out) & 15) == 0;
    if (a->out_layout == MTBC_LAYOUT_C8 && a->operand_layout == MTBC_LAYOUT_PLANAR) {
        // the stem (Cin == 1) of the 16-bit modes: fp32 operands, output channel-blocked 16-bit (+ InstanceNorm statistics)
        if (!stem || !a->w || a->Cout % 8 || (a->compute != 1 && a->compute != 2) || a->out_accumulate) return MTBC_E_UNSUPPORTED;
        if (a->out_type != 0 && a->out_type != a->compute && !(a->out_type == 2 && a->compute == 1)) return MTBC_E_BADARG;
        if (a->stats_partial && (reinterpret_cast<uintptr_t>(a->stats_partial) & 15)) return MTBC_E_BADARG;
        const dim3 grid(cdiv(a->H * (a->W / 4), 128), a->Cout / 8, a->N);
        unsigned short* o16 = reinterpret_cast<unsigned short*>(a->out);
        if (a->compute == 2 || a->out_type == 2)
            hipLaunchKernelGGL(conv3x3_stem_fwd_c8_kernel<true>, grid, dim3(128), 0, st, a->in[0].ptr, (long long)a->in[0].batch_stride, a->w, a->bias, o16, a->stats_partial, a->N, a->H, a->W, a->Cout);
        else
            hipLaunchKernelGGL(conv3x3_stem_fwd_c8_kernel<false>, grid, dim3(128), 0, st, a->in[0].ptr, (long long)a->in[0].batch_stride, a->w, a->bias, o16, a->stats_partial, a->N, a->H, a->W, a->Cout);
        MTBC_CHECK_LAUNCH();
        return MTBC_OK;
    }
    if (a->out_layout != MTBC_LAYOUT_PLANAR && (a->out_layout != MTBC_LAYOUT_C8 || a->operand_layout != MTBC_LAYOUT_C8)) return MTBC_E_UNSUPPORTED;
    if (a->operand_layout == MTBC_LAYOUT_C8) {
        if (!a->w_packed || (a->compute != 1 && a->compute != 2) || !c8_segs_ok(a->in, a->n_in)) return MTBC_E_BADARG;
        if (a->W % 4 || a->W < 8 || a->H < 8 || (reinterpret_cast<uintptr_t>(a->out) & 15)) return MTBC_E_UNSUPPORTED;
        const bool o8 = a->out_layout == MTBC_LAYOUT_C8;
        if (a->out_type != 0 && a->out_type != a->compute && !(o8 && a->out_type == 2 && a->compute == 1)) return MTBC_E_BADARG;
        const bool of16 = o8 && a->compute == 1 && a->out_type == 2;          // bf16 operands, output stored as fp16
        if (o8) {
            if (a->Cout % 8 || a->out_accumulate) return MTBC_E_BADARG;
            out.accumulate[0] = 3;
        }
        if (a->stats_partial && (!o8 || (reinterpret_cast<uintptr_t>(a->stats_partial) & 15))) return MTBC_E_BADARG;
        if (a->norm_z || a->out_partial) {       // gathered dgrad + the reductions of the norm backward
            if (!o8 || of16 || !a->norm_z || !a->norm_mean || !a->norm_rstd || !a->stats_partial || a->bias || (a->norm_gamma == nullptr) != (a->norm_beta == nullptr)) return MTBC_E_BADARG;
            if ((reinterpret_cast<uintptr_t>(a->norm_z) | reinterpret_cast<uintptr_t>(a->norm_mean) | reinterpret_cast<uintptr_t>(a->norm_rstd) |
                 reinterpret_cast<uintptr_t>(a->norm_gamma) | reinterpret_cast<uintptr_t>(a->norm_beta)) & 15) return MTBC_E_BADARG;
            return run_igemm(a->N, a->H, a->W, a->Cin, a->Cout, in, out, a->w_packed, nullptr, a->compute, st, true, 2, a->stats_partial, a);
        }
        return run_igemm(a->N, a->H, a->W, a->Cin, a->Cout, in, out, a->w_packed, a->bias, a->compute, st, true, of16 ? 3 : (o8 ? 1 : 0), a->stats_partial);
    }
    if (a->stats_partial || a->norm_z || a->out_partial) return MTBC_E_UNSUPPORTED;
    if (a->operand_layout != MTBC_LAYOUT_PLANAR) return MTBC_E_BADARG;
    if (a->w_packed && !a->force_direct && mfma_ok(a->in, a->n_in, a->H, a->W))
        return run_igemm(a->N, a->H, a->W, a->Cin, a->Cout, in, out, a->w_packed, a->bias, a->compute, st);
    if (!a->w) return MTBC_E_BADARG;
    if (stem) {
        hipLaunchKernelGGL(conv3x3_stem_fwd_kernel, dim3(cdiv(a->H * (a->W / 4), 128), cdiv(a->Cout, 8), a->N), dim3(128), 0, st,
                           a->in[0].ptr, (long long)a->in[0].batch_stride, a->w, a->bias, a->out, a->N, a->H, a->W, a->Cout);
        MTBC_CHECK_LAUNCH();
        return MTBC_OK;
    }
    DirP p; p.N = a->N; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout; p.in = in; p.out = out;
    p.w = a->w; p.bias = a->bias; p.mode = 0; p.wCin = a->Cin;
    hipLaunchKernelGGL(conv3x3_direct_kernel, dim3(cdiv(a->H * a->W, 128), cdiv(a->Cout, 8), a->N), dim3(128), 0, st, p);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}

int mtbc_conv3x3_dgrad(const mtbc_conv3x3_args* a, void* stream) {
    int rc = check_conv(a); if (rc) return rc;
    if (!a->dout || (!a->w && !a->w_packed)) return MTBC_E_BADARG;
    SegTable in, out;
    mtbc_seg g{const_cast<float*>(a->dout), (int64_t)a->Cout * a->H * a->W, a->Cout, 0};
    rc = make_segtable(&g, 1, a->Cout, &in); if (rc) return rc;
    rc = make_segtable(a->in, a->n_in, a->Cin, &out); if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (a->operand_layout == MTBC_LAYOUT_C8) {
        if (!a->w_packed || (a->compute != 1 && a->compute != 2) || !c8_segs_ok(&g, 1)) return MTBC_E_BADARG;
        if (!mfma_ok(a->in, a->n_in, a->H, a->W)) return MTBC_E_UNSUPPORTED;      // the fp32 planar dx segments: 16-byte stores
        int n3 = 0;
        for (int i = 0; i < a->n_in; ++i) {
            if (a->in[i].accumulate == 2 && ((reinterpret_cast<uintptr_t>(a->in[i].ptr) & 7) || (a->in[i].batch_stride & 3))) return MTBC_E_BADARG;
            if (a->in[i].accumulate == 3) ++n3;
        }
        if (n3 != 0 && (n3 != a->n_in || !c8_segs_ok(a->in, a->n_in))) return MTBC_E_BADARG;      // channel-blocked dx: every segment or none
        return run_igemm(a->N, a->H, a->W, a->Cout, a->Cin, in, out, a->w_packed, nullptr, a->compute, st, true, n3 != 0 ? 1 : 0);
    }
    if (a->operand_layout != MTBC_LAYOUT_PLANAR) return MTBC_E_BADARG;
    for (int i = 0; i < a->n_in; ++i)
        if (a->in[i].accumulate >= 2) return MTBC_E_UNSUPPORTED;      // 16-bit dx segments: channel-blocked dgrad only
    bool ok = a->w_packed && !a->force_direct && mfma_ok(&g, 1, a->H, a->W) && a->Cout % KC == 0;
    for (int i = 0; ok && i < a->n_in; ++i) ok = a->in[i].ptr != nullptr && a->in[i].channels % 4 == 0;
    if (ok) return run_igemm(a->N, a->H, a->W, a->Cout, a->Cin, in, out, a->w_packed, nullptr, a->compute, st);
    if (!a->w) return MTBC_E_BADARG;
    DirP p; p.N = a->N; p.H = a->H; p.W = a->W; p.Cin = a->Cout; p.Cout = a->Cin; p.in = in; p.out = out;
    p.w = a->w; p.bias = nullptr; p.mode = 1; p.wCin = a->Cin;
    hipLaunchKernelGGL(conv3x3_direct_kernel, dim3(cdiv(a->H * a->W, 128), cdiv(a->Cin, 8), a->N), dim3(128), 0, st, p);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}

int32_t mtbc_conv3x3_stats_slots(const mtbc_conv3x3_args* a) {
    if (check_conv(a) || a->out_layout != MTBC_LAYOUT_C8 || (a->compute != 1 && a->compute != 2)) return 0;
    if (a->operand_layout == MTBC_LAYOUT_PLANAR)          // the stem: one subset per 512-pixel block
        return (a->Cin == 1 && a->n_in == 1 && a->W % 4 == 0 && a->Cout % 8 == 0) ? cdiv(a->H * (a->W / 4), 128) : 0;
    if (a->operand_layout != MTBC_LAYOUT_C8) return 0;
    if (a->W % 4 || a->W < 8 || a->H < 8 || a->Cout % 8) return 0;
    const IgemmPlan q = plan_igemm(a->N, a->H, a->W, a->Cout, a->compute, true, a->norm_z == nullptr, a->Cin);
    return q.geo == 2 ? 1 : q.tiles_x * q.tiles_y * (q.nw8 ? 8 : 4);
}

size_t mtbc_conv3x3_wgrad_workspace(const mtbc_conv3x3_args* a) {
    if (check_conv(a)) return 0;
    WgPlan w = plan_wgrad(a);
    return (w.partial_elems + w.dbias_elems) * sizeof(float);
}

size_t mtbc_conv3x3_wgrad_sync_bytes(const mtbc_conv3x3_args* a) {
    if (check_conv(a) || a->operand_layout != MTBC_LAYOUT_C8 || a->Cin == 1) return 0;      // (only the channel-blocked kernels reduce in-kernel)
    const WgPlan w = plan_wgrad(a);
    SplitKFix f{};
    return (size_t)splitk_fix_plan(w.nsplit, &f) * w.coblocks * w.ciblocks * sizeof(int);
}

int mtbc_conv3x3_wgrad(const mtbc_conv3x3_args* a, void* stream) {
    int rc = check_conv(a); if (rc) return rc;
    if (!a->dout || !a->dw) return MTBC_E_BADARG;
    SegTable in;
    rc = make_segtable(a->in, a->n_in, a->Cin, &in); if (rc) return rc;
    WgPlan w = plan_wgrad(a);
    if (!a->workspace || a->workspace_bytes < (w.partial_elems + w.dbias_elems) * sizeof(float)) return MTBC_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* partial = reinterpret_cast<float*>(a->workspace);
    const size_t wel = (size_t)a->Cout * a->Cin * 9;
    if (a->operand_layout == MTBC_LAYOUT_C8 && a->Cin == 1) {
        // the stem: dz channel-blocked 16-bit, the 1-channel input fp32 planar (it has no channel-blocked form)
        if ((a->compute != 1 && a->compute != 2) || a->n_in != 1 || a->Cout % 8 || a->W % 4 || a->dbias || a->in[0].batch_stride % 4 ||
            ((reinterpret_cast<uintptr_t>(a->in[0].ptr) | reinterpret_cast<uintptr_t>(a->dout)) & 15)) return MTBC_E_UNSUPPORTED;
        const int S = w.nsplit / a->N;
        const dim3 grid(a->N * (a->Cout / 8) * S);
        const unsigned short* dz8 = reinterpret_cast<const unsigned short*>(a->dout);
        if (a->compute == 2) hipLaunchKernelGGL(conv3x3_wgrad_stem_c8_kernel<true>, grid, dim3(256), 0, st, a->in[0].ptr, (long long)a->in[0].batch_stride, dz8, partial, a->N, a->H, a->W, a->Cout, S);
        else hipLaunchKernelGGL(conv3x3_wgrad_stem_c8_kernel<false>, grid, dim3(256), 0, st, a->in[0].ptr, (long long)a->in[0].batch_stride, dz8, partial, a->N, a->H, a->W, a->Cout, S);
        MTBC_CHECK_LAUNCH();
        return mtbc_i_splitk_reduce(partial, a->dw, w.nsplit, wel, a->accumulate_dw, st);
    }
    if (a->operand_layout == MTBC_LAYOUT_C8) {
        mtbc_seg g{const_cast<float*>(a->dout), (int64_t)a->Cout * a->H * a->W, a->Cout, 0};
        if ((a->compute != 1 && a->compute != 2) || !c8_segs_ok(a->in, a->n_in) || !c8_segs_ok(&g, 1)) return MTBC_E_BADARG;
        WgC8P p; p.N = a->N; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout; p.in = in;
        p.dz = reinterpret_cast<const unsigned short*>(a->dout); p.partial = partial;
        p.want_bias = a->dbias ? 1 : 0; p.prow = (long long)wel + (a->dbias ? a->Cout : 0);
        // ONE split (the 1152 -> 512 conv on 16 x 16 maps: 240 channel blocks fill the chip by themselves): the "partial" IS the gradient -- the
        // blocks store straight into dw instead of into a 21 MB workspace that a reduction launch then copies (32 us)
        const bool direct = w.nsplit == 1 && !a->dbias && !a->accumulate_dw;
        if (direct) p.partial = a->dw;
        // the split-K partials reduced inside this launch by the last arriver of each group (splitk_fixup) when the caller hands in the
        // zeroed counter buffer; without it: one reduction launch behind the kernel
        p.fix = SplitKFix{};
        if (a->wgrad_sync && !direct) {
            const size_t need = (size_t)splitk_fix_plan(w.nsplit, &p.fix) * w.coblocks * w.ciblocks * sizeof(int);
            if (a->wgrad_sync_bytes < need || (reinterpret_cast<uintptr_t>(a->wgrad_sync) & 3)) return MTBC_E_WORKSPACE;
            p.fix.ctr = a->wgrad_sync; p.fix.dw = a->dw; p.fix.dbias = a->dbias; p.fix.accumulate = a->accumulate_dw;
        }
        p.tiles_x = w.tiles_x; p.tiles_y = w.tiles_y; p.total_tiles = w.total_tiles; p.tiles_per_split = w.tiles_per_split;
        p.ciblocks = w.ciblocks; p.coblocks = w.coblocks; p.cit = w.cit; p.segs = w.segs; p.seg_tiles = w.seg_tiles; p.depth = w.depth;
        { static const int hk = mtbc_probe_int("MTBC_C8W_HACK", 0); p.hack = hk; }
        const dim3 grid = (w.c8w || w.c8i) ? dim3(w.nsplit * w.coblocks * w.ciblocks) : dim3(w.nsplit, w.coblocks * w.ciblocks);
#ifdef MTBC_PROBES
        // MTBC_WG_TS=1: phase timestamps of every block (thread 0) of the 32 x 32 / wide-block weight-gradient kernels, printed after the launch
        static const int wts_env = mtbc_probe_int("MTBC_WG_TS", 0);
        static unsigned long long* wdts = nullptr;
        const size_t wnb = (size_t)grid.x * grid.y;
        p.ts = nullptr;
        if (wts_env && wnb <= 4096) {
            if (!wdts) (void)hipMalloc(&wdts, 4096 * 16 * sizeof(unsigned long long));
            (void)hipMemsetAsync(wdts, 0, wnb * 16 * sizeof(unsigned long long), st);
            p.ts = wdts;
        }
#endif
        if (w.c8i) {
            const size_t lds = c8i_lds_bytes(w.cit, w.cot);
            const dim3 gridi(w.nsplit * w.coblocks * w.ciblocks);
#define MTBC_C8I_LAUNCH(F16_, COT_, BIAS_)                                                                                             \
            do {                                                                                                                       \
                MTBC_ENSURE_DYN_LDS((&conv3x3_wgrad_c8i_kernel<F16_, COT_, BIAS_>), 160 * 1024);                                      \
                hipLaunchKernelGGL((conv3x3_wgrad_c8i_kernel<F16_, COT_, BIAS_>), gridi, dim3(512), lds, st, p);                      \
            } while (0)
            const int sel = (a->compute == 2 ? 4 : 0) + (w.cot == 3 ? 2 : 0) + (a->dbias ? 1 : 0);
            switch (sel) {
            case 0: MTBC_C8I_LAUNCH(false, 2, false); break;
            case 1: MTBC_C8I_LAUNCH(false, 2, true); break;
            case 2: MTBC_C8I_LAUNCH(false, 3, false); break;
            case 3: MTBC_C8I_LAUNCH(false, 3, true); break;
            case 4: MTBC_C8I_LAUNCH(true, 2, false); break;
            case 5: MTBC_C8I_LAUNCH(true, 2, true); break;
            case 6: MTBC_C8I_LAUNCH(true, 3, false); break;
            default: MTBC_C8I_LAUNCH(true, 3, true); break;
            }
#undef MTBC_C8I_LAUNCH
        } else if (w.c8w) {
            const size_t lds = c8w_lds_bytes(w.cit, w.cot, w.depth);
#define MTBC_C8W_LAUNCH(F16_, COT_, BIAS_)                                                                                             \
            do {                                                                                                                       \
                MTBC_ENSURE_DYN_LDS((&conv3x3_wgrad_c8w_kernel<F16_, COT_, BIAS_>), 160 * 1024);      /* up to 160 KB of dynamic LDS */   \
                hipLaunchKernelGGL((conv3x3_wgrad_c8w_kernel<F16_, COT_, BIAS_>), grid, dim3(512), lds, st, p);                       \
            } while (0)
            const int sel = (a->compute == 2 ? 4 : 0) + (w.cot == 3 ? 2 : 0) + (a->dbias ? 1 : 0);
            switch (sel) {
            case 0: MTBC_C8W_LAUNCH(false, 2, false); break;
            case 1: MTBC_C8W_LAUNCH(false, 2, true); break;
            case 2: MTBC_C8W_LAUNCH(false, 3, false); break;
            case 3: MTBC_C8W_LAUNCH(false, 3, true); break;
            case 4: MTBC_C8W_LAUNCH(true, 2, false); break;
            case 5: MTBC_C8W_LAUNCH(true, 2, true); break;
            case 6: MTBC_C8W_LAUNCH(true, 3, false); break;
            default: MTBC_C8W_LAUNCH(true, 3, true); break;
            }
#undef MTBC_C8W_LAUNCH
        } else if (w.geo == 1) {
            if (a->compute == 2) hipLaunchKernelGGL((conv3x3_wgrad_c8_kernel<true, 1>), grid, dim3(256), C8W_LDS, st, p);
            else hipLaunchKernelGGL((conv3x3_wgrad_c8_kernel<false, 1>), grid, dim3(256), C8W_LDS, st, p);
        } else {
            if (a->compute == 2) hipLaunchKernelGGL((conv3x3_wgrad_c8_kernel<true, 0>), grid, dim3(256), C8W_LDS, st, p);
            else hipLaunchKernelGGL((conv3x3_wgrad_c8_kernel<false, 0>), grid, dim3(256), C8W_LDS, st, p);
        }
        MTBC_CHECK_LAUNCH();
#ifdef MTBC_PROBES
        if (p.ts) {
            (void)hipStreamSynchronize(st);
            static unsigned long long hts[4096 * 16];
            (void)hipMemcpy(hts, wdts, wnb * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            unsigned long long t0 = ~0ull;
            for (size_t b = 0; b < wnb; ++b) if (hts[b * 16] && hts[b * 16] < t0) t0 = hts[b * 16];
            double mean[13] = {0}; int cnt[13] = {0};
            for (size_t b = 0; b < wnb; ++b)
                for (int k = 0; k < 13; ++k) if (hts[b * 16 + k]) { mean[k] += (double)(hts[b * 16 + k] - t0) * 0.01; ++cnt[k]; }
            for (int k = 0; k < 13; ++k) if (cnt[k]) mean[k] /= cnt[k];
            fprintf(stderr, "wg_ts %s %d->%d @%dx%d blocks %zu | mean us since the first block's entry: entry %.2f", w.c8i ? "c8i" : w.c8w ? "c8w" : "c8", p.Cin, p.Cout, p.H, p.W, wnb, mean[0]);
            if (w.c8i) {
                fprintf(stderr, " first image issued %.2f", mean[1]);
                for (int k = 0; k < 3; ++k) fprintf(stderr, " | image %d: landed %.2f next issued %.2f mfma done %.2f", k, mean[2 + 3 * k], mean[3 + 3 * k], mean[4 + 3 * k]);
                fprintf(stderr, " | loop done %.2f stored %.2f\n", mean[11], mean[12]);
            } else if (w.c8w) {
                fprintf(stderr, " first rows issued %.2f", mean[1]);
                for (int k = 0; k < 3; ++k) fprintf(stderr, " | step %d: landed %.2f next issued %.2f mfma done %.2f", k + 2, mean[2 + 3 * k], mean[3 + 3 * k], mean[4 + 3 * k]);
                fprintf(stderr, " | loop done %.2f stored %.2f\n", mean[11], mean[12]);
            } else {
                for (int k = 0; k < 3; ++k) fprintf(stderr, " | tile %d: start %.2f issued %.2f landed %.2f", k, mean[1 + 3 * k], mean[2 + 3 * k], mean[3 + 3 * k]);
                fprintf(stderr, " | loop done %.2f halves in LDS %.2f stored %.2f\n", mean[10], mean[11], mean[12]);
            }
        }
#endif
        if (direct || p.fix.ctr) return MTBC_OK;
        // one row per split = the weight-gradient partial followed by the bias-gradient partial: ONE reduction launch for both
        return mtbc_i_splitk_reduce2(partial, a->dw, a->dbias, w.nsplit, wel, a->dbias ? (size_t)a->Cout : 0, a->accumulate_dw, st);
    }
    if (a->operand_layout != MTBC_LAYOUT_PLANAR) return MTBC_E_BADARG;
    if (w.mfma) {
        WgP p; p.N = a->N; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout; p.in = in; p.dz = a->dout;
        p.partial = partial; p.tiles_x = w.tiles_x; p.tiles_y = w.tiles_y; p.total_tiles = w.total_tiles;
        p.tiles_per_split = w.tiles_per_split; p.ciblocks = w.ciblocks;
        static const int dbgw = mtbc_probe_int("MTBC_DBG", 0);
        p.dbg = dbgw;
        dim3 grid(w.nsplit, w.coblocks * w.ciblocks);
#ifdef MTBC_PROBES
        // MTBC_WG_TS=1: phase timestamps of every block (thread 0) of the fp32 weight-gradient kernel, printed after the launch
        static const int fts_env = mtbc_probe_int("MTBC_WG_TS", 0);
        static unsigned long long* fdts = nullptr;
        const size_t fnb = (size_t)grid.x * grid.y;
        p.ts = nullptr;
        if (fts_env && fnb <= 4096) {
            if (!fdts) (void)hipMalloc(&fdts, 4096 * 16 * sizeof(unsigned long long));
            (void)hipMemsetAsync(fdts, 0, fnb * 16 * sizeof(unsigned long long), st);
            p.ts = fdts;
        }
#endif
        const int zch = 16 * w.cot;
        const dim3 blk(128 * w.cot);
        static const int lowp_env = mtbc_probe_int("MTBC_LOWP", -1);
        const int lowp = lowp_env >= 0 ? lowp_env : a->compute;       // 0 fp32 (exact), 1 bf16, 2 fp16 MFMA operands
#define MTBC_WG_LAUNCH(GEO_, COT_)                                                                                         \
        do {                                                                                                               \
            if (lowp == 1) hipLaunchKernelGGL((conv3x3_wgrad_lp_kernel<GEO_, COT_, false>), grid, blk,                     \
                                              (32 * WGeoLP<GEO_>::PSX + zch * PSZ_LP) * sizeof(float), st, p);            \
            else if (lowp == 2) hipLaunchKernelGGL((conv3x3_wgrad_lp_kernel<GEO_, COT_, true>), grid, blk,                 \
                                                   (32 * WGeoLP<GEO_>::PSX + zch * PSZ_LP) * sizeof(float), st, p);       \
            else if (w.pack24) hipLaunchKernelGGL((conv3x3_wgrad_mfma_kernel<GEO_, COT_, true>), grid, blk,                \
                                                  (32 * WGeo<GEO_>::PSX + zch * PSZ) * sizeof(float), st, p);              \
            else hipLaunchKernelGGL((conv3x3_wgrad_mfma_kernel<GEO_, COT_, false>), grid, blk,                             \
                                    (32 * WGeo<GEO_>::PSX + zch * PSZ) * sizeof(float), st, p);                            \
        } while (0)
        static const bool lp1 = mtbc_probe_set("MTBC_WGRAD_LP1");      // A/B: first-generation 16-bit wgrad
        if (w.geo == 0 && w.cot == 2 && lowp != 0 && !lp1) {
            MTBC_ENSURE_DYN_LDS((&conv3x3_wgrad_lp2_kernel<false>), 64 * 1024);
            MTBC_ENSURE_DYN_LDS((&conv3x3_wgrad_lp2_kernel<true>), 64 * 1024);
            if (lowp == 2) hipLaunchKernelGGL((conv3x3_wgrad_lp2_kernel<true>), grid, dim3(256), W2_LDS, st, p);
            else hipLaunchKernelGGL((conv3x3_wgrad_lp2_kernel<false>), grid, dim3(256), W2_LDS, st, p);
        } else if (w.cot == 2) {
            if (w.geo == 0) MTBC_WG_LAUNCH(0, 2); else if (w.geo == 1) MTBC_WG_LAUNCH(1, 2); else MTBC_WG_LAUNCH(2, 2);
        } else {
            if (w.geo == 0) MTBC_WG_LAUNCH(0, 3); else if (w.geo == 1) MTBC_WG_LAUNCH(1, 3); else MTBC_WG_LAUNCH(2, 3);
        }
#undef MTBC_WG_LAUNCH
        MTBC_CHECK_LAUNCH();
#ifdef MTBC_PROBES
        if (p.ts) {
            (void)hipStreamSynchronize(st);
            static unsigned long long fhts[4096 * 16];
            (void)hipMemcpy(fhts, fdts, fnb * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            unsigned long long t0 = ~0ull;
            for (size_t b = 0; b < fnb; ++b) if (fhts[b * 16] && fhts[b * 16] < t0) t0 = fhts[b * 16];
            double mean[15] = {0}; int cnt[15] = {0};
            for (size_t b = 0; b < fnb; ++b)
                for (int k = 0; k < 15; ++k) if (fhts[b * 16 + k]) { mean[k] += (double)(fhts[b * 16 + k] - t0) * 0.01; ++cnt[k]; }
            for (int k = 0; k < 15; ++k) if (cnt[k]) mean[k] /= cnt[k];
            fprintf(stderr, "wg_ts f32 %d->%d @%dx%d cot %d blocks %zu tiles/block %d | mean us since the first block's entry: entry %.2f", p.Cin, p.Cout, p.H, p.W, w.cot, fnb, w.tiles_per_split, mean[0]);
            for (int k = 0; k < 3; ++k) fprintf(stderr, " | tile %d: all waves here %.2f own commit done %.2f published %.2f next loads issued %.2f", k, mean[1 + 4 * k], mean[2 + 4 * k], mean[3 + 4 * k], mean[4 + 4 * k]);
            fprintf(stderr, " | loop done %.2f stored %.2f\n", mean[13], mean[14]);
        }
#endif
    } else {
        DirWgP p; p.N = a->N; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout; p.nsplit = w.nsplit;
        p.in = in; p.dz = a->dout; p.partial = partial;
        if (w.smallcin)
            hipLaunchKernelGGL(conv3x3_wgrad_smallcin_kernel, dim3(w.nsplit * a->Cout), dim3(256), 0, st, p);
        else
            hipLaunchKernelGGL(conv3x3_wgrad_direct_kernel, dim3(a->Cout * a->Cin, w.nsplit), dim3(256), 0, st, p);
        MTBC_CHECK_LAUNCH();
    }
    rc = mtbc_i_splitk_reduce(partial, a->dw, w.nsplit, wel, a->accumulate_dw, st); if (rc) return rc;
    if (a->dbias) {
        rc = mtbc_i_channel_sums(a->dout, partial + w.partial_elems, a->dbias, a->N, a->Cout, a->H * a->W, a->accumulate_dw, st);
        if (rc) return rc;
    }
    return MTBC_OK;
}

int mtbc_c8_pack(const float* src, int64_t src_batch_stride, void* dst, int32_t N, int32_t C, int32_t HW, int32_t compute, void* stream) {
    if (!src || !dst || (compute != 1 && compute != 2) || (reinterpret_cast<uintptr_t>(dst) & 15)) return MTBC_E_BADARG;
    if (N <= 0 || C <= 0 || HW <= 0 || C % 8) return MTBC_E_BADSHAPE;
    const long long total = (long long)N * (C / 8) * HW;
    const dim3 grid((unsigned)cdiv64(total, 256));
    if (compute == 2) hipLaunchKernelGGL((c8_pack_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, src, (long long)src_batch_stride, reinterpret_cast<unsigned short*>(dst), C, HW, total);
    else hipLaunchKernelGGL((c8_pack_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, src, (long long)src_batch_stride, reinterpret_cast<unsigned short*>(dst), C, HW, total);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}
int mtbc_c8_unpack(const void* src, float* dst, int32_t N, int32_t C, int32_t HW, int32_t compute, void* stream) {
    if (!src || !dst || (compute != 1 && compute != 2) || (reinterpret_cast<uintptr_t>(src) & 15)) return MTBC_E_BADARG;
    if (N <= 0 || C <= 0 || HW <= 0 || C % 8) return MTBC_E_BADSHAPE;
    const long long total = (long long)N * (C / 8) * HW;
    const dim3 grid((unsigned)cdiv64(total, 256));
    if (compute == 2) hipLaunchKernelGGL((c8_unpack_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const unsigned short*>(src), dst, C, HW, total);
    else hipLaunchKernelGGL((c8_unpack_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const unsigned short*>(src), dst, C, HW, total);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}

int mtbc_conv3x3_weight_view(const float* w, float* dst, int32_t Cout, int32_t Cin, int32_t ci_off, int32_t ci_cnt, int32_t mode,
                             int32_t k_off, int32_t K, void* stream) {
    if (!w || !dst || Cout <= 0 || Cin <= 0 || ci_off < 0 || ci_cnt <= 0 || ci_off + ci_cnt > Cin || (mode != 0 && mode != 1)) return MTBC_E_BADARG;
    if (mode == 1 && (k_off < 0 || k_off + Cout > K)) return MTBC_E_BADARG;
    const int total = Cout * ci_cnt * 9;
    hipLaunchKernelGGL(wview_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, dst, Cout, Cin, ci_off, ci_cnt, mode, k_off, K);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}

int mtbc_conv3x3_weight_view_many(const mtbc_wview_desc* descs, int32_t n, void* stream) {
    if (!descs || n < 0) return MTBC_E_BADARG;
    for (int32_t base = 0; base < n; base += WVIEW_MANY) {
        const int32_t m = n - base < WVIEW_MANY ? n - base : WVIEW_MANY;
        WViewManyP q;
        int blocks = 0;
        for (int32_t i = 0; i < m; ++i) {
            const mtbc_wview_desc& d = descs[base + i];
            if (!d.w || !d.dst || d.Cout <= 0 || d.Cin <= 0 || d.ci_off < 0 || d.ci_cnt <= 0 || d.ci_off + d.ci_cnt > d.Cin) return MTBC_E_BADSHAPE;
            if (d.mode && (d.k_off < 0 || d.k_off + d.Cout > d.K)) return MTBC_E_BADSHAPE;
            q.w[i] = d.w; q.dst[i] = d.dst; q.Cout[i] = d.Cout; q.Cin[i] = d.Cin; q.off[i] = d.ci_off; q.cnt[i] = d.ci_cnt;
            q.koff[i] = d.k_off; q.K[i] = d.K; q.mode[i] = d.mode ? 1 : 0;
            q.first_block[i] = blocks;
            blocks += cdiv(d.Cout * d.ci_cnt * 9, 256);
        }
        q.first_block[m] = blocks;
        q.n = m;
        hipLaunchKernelGGL(wview_many_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, q);
        MTBC_CHECK_LAUNCH();
    }
    return MTBC_OK;
}

int mtbc_c8_pack16(const void* src, int64_t src_batch_stride, void* dst, int32_t N, int32_t C, int32_t HW, void* stream) {
    if (!src || !dst || (reinterpret_cast<uintptr_t>(dst) & 15) || (reinterpret_cast<uintptr_t>(src) & 7) || src_batch_stride % 4) return MTBC_E_BADARG;
    if (N <= 0 || C <= 0 || HW <= 0 || C % 8 || HW % 4) return MTBC_E_BADSHAPE;
    const long long total = (long long)N * (C / 8) * (HW / 4);
    hipLaunchKernelGGL(c8_pack16_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned short*>(src), (long long)src_batch_stride, reinterpret_cast<unsigned short*>(dst), C, HW, total);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}

}  // extern "C"
