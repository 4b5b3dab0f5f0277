// ConvTranspose2d with kernel == stride == 2: dgrad and wgrad straight from HBM into MFMA fragments (gfx950).
//
// Replaces the backward of the up-convolutions (MONAI UpSample "deconv" in MTUNetPlusPlus.py:24-76 via
// basic_unetplusplus UpCat; nn.ConvTranspose2d in MTnnUNet.py:96-100).  With k == s == 2 every output pixel has exactly
// one contributing input pixel, so both gradients are plain GEMMs whose reduction index is CONTIGUOUS in memory:
//
//   dgrad  dX[ci][p]        = sum_k W[ci][k] * dY[k @ p]      k = (co, a, b): W rows are k-contiguous; for one input
//                                                             pixel p = (i, jx) the four (a, b) values of a channel
//                                                             are dY[co][2i+a][2jx+b]: two float2, and TWO neighbouring
//                                                             pixels are one float4 per output row
//   wgrad  dW[ci][(co,a,b)] = sum_p X[ci][p] * dY[(co,a,b)@p]  p-contiguous in X; in dY eight consecutive p of one
//                                                             (co, a) are 16 consecutive floats with b interleaved
//
// v_mfma_f32_16x16x32_{bf16,f16} wants 8 consecutive reduction indices per lane, v_mfma_f32_16x16x4_f32 one -- and a
// sum does not care how the 32 indices of a step are dealt to the 4 lane groups.  So ONE register layout serves both:
// every lane loads 8 consecutive k (two float4), the 16-bit modes convert and issue one MFMA, the fp32 mode issues
// eight 16x16x4 MFMAs (the i-th uses element i of every lane).  Nothing is staged in LDS and there is no barrier: a
// wave is an independent task, its loads are whole 64..128-byte runs, and the next step's registers are in flight
// while the current step is multiplied.  (The generic 64x64 LDS GEMM in pool_up.hip gathered dY 8 bytes at a time:
// 26 TF on wgrad, 41 TF on dgrad.)  That kernel remains the fallback for k = 4, 8 and for odd shapes.
#include "common.h"
#include <utility>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// LP: 0 = fp32 (exact), 1 = bf16, 2 = fp16 operands; fp32 accumulation everywhere
template <int LP> struct Frag;
template <> struct Frag<0> {
    float v[8];
    __device__ __forceinline__ void set(const float* f) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = f[i];
    }
    static __device__ __forceinline__ f32x4 mma(const Frag& a, const Frag& b, f32x4 c) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[i], b.v[i], c, 0, 0, 0);
        return c;
    }
};
template <> struct Frag<1> {
    bf16x8 v;
    __device__ __forceinline__ void set(const float* f) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (__bf16)f[i];
    }
    static __device__ __forceinline__ f32x4 mma(const Frag& a, const Frag& b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
    }
};
template <> struct Frag<2> {
    f16x8 v;
    __device__ __forceinline__ void set(const float* f) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (_Float16)f[i];
    }
    static __device__ __forceinline__ f32x4 mma(const Frag& a, const Frag& b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v, b.v, c, 0, 0, 0);
    }
};

typedef unsigned ct_u32x4 __attribute__((ext_vector_type(4)));
template <int LP> __device__ __forceinline__ void frag_from_words(Frag<LP>& f, unsigned w0, unsigned w1, unsigned w2, unsigned w3) {
    static_assert(LP != 0, "16-bit operands only");
    f.v = __builtin_bit_cast(decltype(f.v), (ct_u32x4){w0, w1, w2, w3});
}
// sum of the two 16-bit values of a dword, in fp32
template <int LP> __device__ __forceinline__ float sum2_16(unsigned w) {
    if constexpr (LP == 2) { typedef _Float16 h2 __attribute__((ext_vector_type(2))); const h2 h = __builtin_bit_cast(h2, w); return (float)h[0] + (float)h[1]; }
    else return __uint_as_float(w << 16) + __uint_as_float(w & 0xffff0000u);
}

struct Ct2P {
    int N, H, W, Cin, Cout;            // input H x W, output 2H x 2W
    const float* x; long long xbs;
    const float* w;                    // [Cin][Cout][2][2]
    const float* dy; long long dybs;
    float* dx; long long dxbs; int acc_dx;
    float* partial;                    // wgrad: [nsplit][Cin][Cout][2][2]
    float* dbias_part;                 // wgrad: [nsplit][Cout] per-split sums of dY (bias gradient), or nullptr
    int mblocks;                       // blocks of 48 input channels
    int ctiles;                        // wgrad: tiles of 8 output channels
    int steps_per_split, nsplit;       // wgrad: 32-pixel steps per split (over the flattened (n, step) list)
    long long ntasks;
    int xcd_tasks;                     // convT2_wgrad16_kernel: wave tasks per XCD (> 0: all tasks of a split run on XCD split % 8), 0 = plain task order
};

struct S0 { static constexpr int value = 0; };
struct S1 { static constexpr int value = 1; };
// compile-time loop: f(slot constant) for 0 .. N-1 (register-set indices must be constants, see below)
template <int I> struct SC { static constexpr int value = I; };
template <typename F, int... Is> __device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, Is...>) { (f(SC<Is>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }
constexpr int MT = 3;                  // 16-row tiles of input channels per wave (48 = the U-Net++ channel quantum)

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ------------------------------------------------------------------ dgrad
// wave task = (48 input channels) x (32 consecutive input pixels); MFMA column j of tile E is pixel 2j, of tile O pixel
// 2j+1, so one float4 of an output row feeds both.  Step = 8 output channels (32 k); lane group kg owns channels
// 8s + 2kg, 8s + 2kg + 1.
// DY16: dY is a 16-bit planar tensor of the MFMA's own type (written by the 3x3 conv's dgrad, mtbc_seg.accumulate = 2): the
// four values of an output row that the fp32 form loads as one float4 and converts are ONE 8-byte load whose two dwords
// are the fragment words of the even / odd pixel as they stand -- half the bytes, no conversion, same MFMA operands.
// D register sets = D - 1 steps of loads in flight: the deep levels have a few waves per SIMD, each with a long chain of steps
// (Cout / 8 = 12 .. 24), and one step of prefetch leaves every step waiting for the L2.
template <int LP, bool DY16 = false, int D = 2>
__global__ __launch_bounds__(256) void convT2_dgrad_kernel(const Ct2P p) {
    static_assert(!DY16 || LP != 0, "a 16-bit dY feeds the 16-bit MFMA");
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (scalar: the task's indices and branches are wave-uniform)
    const int j = lane & 15, kg = lane >> 4;
    const long long task = (long long)blockIdx.x * 4 + wv;
    if (task >= p.ntasks) return;
    const int HW = p.H * p.W, oW = 2 * p.W, M4 = 4 * p.Cout;
    const int groups = HW / 32;
    const int mb = (int)(task % p.mblocks);
    const long long t = task / p.mblocks;
    const int g = (int)(t % groups), n = (int)(t / groups);
    const int pix = g * 32 + 2 * j;
    const int i = pix / p.W, jx = pix % p.W;
    const float* dyn = p.dy + (size_t)n * p.dybs + (size_t)(2 * i) * oW + 2 * jx;      // + co*4HW + a*oW
    const unsigned short* dyn16 = reinterpret_cast<const unsigned short*>(p.dy) + (size_t)n * p.dybs + (size_t)(2 * i) * oW + 2 * jx;
    const float* wrow[MT];
    bool rok[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int ci = mb * 48 + m * 16 + j;
        rok[m] = ci < p.Cin;
        wrow[m] = p.w + (size_t)(rok[m] ? ci : 0) * M4;
    }
    f32x4 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m) { acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    const int nsteps = (p.Cout + 7) / 8;
    // two register sets with COMPILE-TIME slot numbers (a runtime slot index would put the arrays in scratch)
    float4 ra[D][MT][2], rb[DY16 ? 1 : D][4];
    uint2 rh[DY16 ? D : 1][4];               // DY16: (row a, channel c) -> {even pixel (b0,b1), odd pixel (b0,b1)}
    auto load = [&](int s, auto SL) {       // issue only; masking happens at use
        constexpr int slot = decltype(SL)::value;
        const int c0 = 8 * s + 2 * kg;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float* q = c0 < p.Cout ? wrow[m] + 32 * s + 8 * kg : p.w;
            ra[slot][m][0] = ld4(q); ra[slot][m][1] = ld4(q + 4);      // channels c0, c0 + 1 (Cout is even)
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int co = c0 + c < p.Cout ? c0 + c : 0;
            if constexpr (DY16) {
                const unsigned short* q = dyn16 + (size_t)co * 4 * HW;
                rh[slot][2 * c] = *reinterpret_cast<const uint2*>(q); rh[slot][2 * c + 1] = *reinterpret_cast<const uint2*>(q + oW);
            } else {
                const float* q = dyn + (size_t)co * 4 * HW;
                rb[slot][2 * c] = ld4(q); rb[slot][2 * c + 1] = ld4(q + oW);
            }
        }
    };
    auto compute = [&](int s, auto SL) {
        constexpr int cur = decltype(SL)::value;
        const int c0 = 8 * s + 2 * kg;
        const bool ok0 = c0 < p.Cout, ok1 = c0 + 1 < p.Cout;
        constexpr int cb = DY16 ? 0 : cur;       // (the fp32 registers do not exist with a 16-bit dY)
        float e[8], o[8];
        e[0] = ok0 ? rb[cb][0].x : 0.f; e[1] = ok0 ? rb[cb][0].y : 0.f; e[2] = ok0 ? rb[cb][1].x : 0.f; e[3] = ok0 ? rb[cb][1].y : 0.f;
        e[4] = ok1 ? rb[cb][2].x : 0.f; e[5] = ok1 ? rb[cb][2].y : 0.f; e[6] = ok1 ? rb[cb][3].x : 0.f; e[7] = ok1 ? rb[cb][3].y : 0.f;
        o[0] = ok0 ? rb[cb][0].z : 0.f; o[1] = ok0 ? rb[cb][0].w : 0.f; o[2] = ok0 ? rb[cb][1].z : 0.f; o[3] = ok0 ? rb[cb][1].w : 0.f;
        o[4] = ok1 ? rb[cb][2].z : 0.f; o[5] = ok1 ? rb[cb][2].w : 0.f; o[6] = ok1 ? rb[cb][3].z : 0.f; o[7] = ok1 ? rb[cb][3].w : 0.f;
        Frag<LP> fe, fo;
        if constexpr (DY16) {
            frag_from_words<LP>(fe, ok0 ? rh[cur][0].x : 0u, ok0 ? rh[cur][1].x : 0u, ok1 ? rh[cur][2].x : 0u, ok1 ? rh[cur][3].x : 0u);
            frag_from_words<LP>(fo, ok0 ? rh[cur][0].y : 0u, ok0 ? rh[cur][1].y : 0u, ok1 ? rh[cur][2].y : 0u, ok1 ? rh[cur][3].y : 0u);
        } else { fe.set(e); fo.set(o); }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float a[8];
            const bool k0 = rok[m] && ok0, k1 = rok[m] && ok1;
            a[0] = k0 ? ra[cur][m][0].x : 0.f; a[1] = k0 ? ra[cur][m][0].y : 0.f; a[2] = k0 ? ra[cur][m][0].z : 0.f; a[3] = k0 ? ra[cur][m][0].w : 0.f;
            a[4] = k1 ? ra[cur][m][1].x : 0.f; a[5] = k1 ? ra[cur][m][1].y : 0.f; a[6] = k1 ? ra[cur][m][1].z : 0.f; a[7] = k1 ? ra[cur][m][1].w : 0.f;
            Frag<LP> fa;
            fa.set(a);
            acc[m][0] = Frag<LP>::mma(fa, fe, acc[m][0]);
            acc[m][1] = Frag<LP>::mma(fa, fo, acc[m][1]);
        }
    };
    // Step t lives in register set t % D.  The main loop has NO branch around a load (past the end it re-loads the last step into a
    // set nobody reads): with conditional loads the compiler's wait-count pass gave up and waited for ALL loads, the ones just issued
    // included, before the first MFMA of every round -- there was no prefetch at all.
    const int last = nsteps - 1;
    sfor<D - 1>([&](auto U) { load(min((int)U.value, last), U); });
    int s = 0;
    for (; s + D <= nsteps; s += D)
        sfor<D>([&](auto U) {
            constexpr int u = decltype(U)::value;
            load(min(s + u + D - 1, last), SC<(u + D - 1) % D>{});
            __builtin_amdgcn_sched_barrier(0);        // (the scheduler otherwise sinks the loads below the MFMAs that do not depend on them)
            compute(s + u, U);
            __builtin_amdgcn_sched_barrier(0);
        });
    sfor<D - 1>([&](auto U) { if (s + U.value < nsteps) compute(s + U.value, U); });      // the loads of the last < D steps are in flight
    // rows kg*4 + r of tile m, column j -> pixels (2j, 2j+1): one float2 per row
    float* dxn = p.dx + (size_t)n * p.dxbs + g * 32 + 2 * j;
    float2 old[MT][4];
    if (p.acc_dx) {         // all twelve loads of the read-modify-write in flight at once (rows past Cin re-read the last row)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) old[m][r] = *reinterpret_cast<const float2*>(dxn + (size_t)min(mb * 48 + m * 16 + kg * 4 + r, p.Cin - 1) * HW);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = mb * 48 + m * 16 + kg * 4 + r;
            if (ci >= p.Cin) continue;
            float2* d = reinterpret_cast<float2*>(dxn + (size_t)ci * HW);
            float2 v = make_float2(acc[m][0][r], acc[m][1][r]);
            if (p.acc_dx) { v.x += old[m][r].x; v.y += old[m][r].y; }
            *d = v;
        }
}

// ------------------------------------------------------------------ wgrad
// wave task = (48 input channels) x (8 output channels) x (a run of 32-pixel steps of the flattened (image, step)
// list).  MFMA column j is (co = 8*ct + j/2, a = j%2); b = 0 and b = 1 are two accumulator tiles fed by the even /
// odd elements of the same 64 bytes.  Lane group kg owns pixels 8kg .. 8kg+7 of the step.
// DY16: the 16 consecutive values of an output row (8 pixels x b) are 32 bytes; fragment word m of the b = 0 tile is
// (low half of dword 2m) | (low half of dword 2m+1) << 16, of the b = 1 tile the two high halves: two v_perm per word.
// (16-bit planes of x as well: convT2_wgrad16_kernel below.)
template <int LP, bool DY16 = false, int D = 2>
__global__ __launch_bounds__(256) void convT2_wgrad_kernel(const Ct2P p) {
    static_assert(!DY16 || LP != 0, "a 16-bit dY feeds the 16-bit MFMA");
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (scalar: the task's indices and branches are wave-uniform)
    const int j = lane & 15, kg = lane >> 4;
    const long long task = (long long)blockIdx.x * 4 + wv;
    if (task >= p.ntasks) return;
    const int HW = p.H * p.W, oW = 2 * p.W;
    const int steps_per_img = HW / 32;
    const int ct = (int)(task % p.ctiles);
    const long long t = task / p.ctiles;
    const int mb = (int)(t % p.mblocks), split = (int)(t / p.mblocks);
    const int total_steps = p.N * steps_per_img;
    const int g0 = split * p.steps_per_split, g1 = min(total_steps, g0 + p.steps_per_split);

    const float* xrow[MT];
    bool rok[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int ci = mb * 48 + m * 16 + j;
        rok[m] = ci < p.Cin;
        xrow[m] = p.x + (size_t)(rok[m] ? ci : 0) * HW + 8 * kg;
    }
    const int co = ct * 8 + (j >> 1);
    const bool cok = co < p.Cout;
    const float* dcol = p.dy + (size_t)(cok ? co : 0) * 4 * HW + (size_t)(j & 1) * oW;
    const unsigned short* dcol16 = reinterpret_cast<const unsigned short*>(p.dy) + (size_t)(cok ? co : 0) * 4 * HW + (size_t)(j & 1) * oW;

    f32x4 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m) { acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    const bool want_bias = p.dbias_part != nullptr && mb == 0;      // uniform
    float bsum = 0.f;
    float4 ra[D][MT][2], rb[DY16 ? 1 : D][4];
    ct_u32x4 rh[DY16 ? D : 1][2];            // DY16: 8 dwords = 8 pixels, each {b0, b1}
    auto load = [&](int g, auto SL) {
        constexpr int slot = decltype(SL)::value;
        const int n = g / steps_per_img, st = g % steps_per_img;
        const int pix = st * 32 + 8 * kg;
        const int i = pix / p.W, jx = pix % p.W;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float* q = xrow[m] + (size_t)n * p.xbs + st * 32;
            ra[slot][m][0] = ld4(q); ra[slot][m][1] = ld4(q + 4);
        }
        if constexpr (DY16) {
            const unsigned short* q = dcol16 + (size_t)n * p.dybs + (size_t)(2 * i) * oW + 2 * jx;
            rh[slot][0] = *reinterpret_cast<const ct_u32x4*>(q); rh[slot][1] = *reinterpret_cast<const ct_u32x4*>(q + 8);
        } else {
            const float* q = dcol + (size_t)n * p.dybs + (size_t)(2 * i) * oW + 2 * jx;
#pragma unroll
            for (int k = 0; k < 4; ++k) rb[slot][k] = ld4(q + 4 * k);
        }
    };
    auto compute = [&](auto SL) {
        constexpr int cur = decltype(SL)::value;
        float b0[8], b1[8];
        Frag<LP> f0, f1;
        if constexpr (DY16) {
            unsigned d[8], w0[4], w1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { d[k] = cok ? rh[cur][0][k] : 0u; d[4 + k] = cok ? rh[cur][1][k] : 0u; }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                w0[m] = (d[2 * m] & 0xffffu) | (d[2 * m + 1] << 16);
                w1[m] = (d[2 * m] >> 16) | (d[2 * m + 1] & 0xffff0000u);
            }
            if (want_bias) {               // the bias gradient is the plain sum of the stored dY values
#pragma unroll
                for (int k = 0; k < 8; ++k) bsum += sum2_16<LP>(d[k]);
            }
            frag_from_words<LP>(f0, w0[0], w0[1], w0[2], w0[3]);
            frag_from_words<LP>(f1, w1[0], w1[1], w1[2], w1[3]);
        } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            b0[2 * k] = cok ? rb[cur][k].x : 0.f; b1[2 * k] = cok ? rb[cur][k].y : 0.f;
            b0[2 * k + 1] = cok ? rb[cur][k].z : 0.f; b1[2 * k + 1] = cok ? rb[cur][k].w : 0.f;
        }
        if (want_bias) {                   // the bias gradient is the plain sum of dY: it rides on the wave that reads it
#pragma unroll
            for (int k = 0; k < 8; ++k) bsum += b0[k] + b1[k];
        }
        f0.set(b0); f1.set(b1);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float a[8];
            a[0] = rok[m] ? ra[cur][m][0].x : 0.f; a[1] = rok[m] ? ra[cur][m][0].y : 0.f; a[2] = rok[m] ? ra[cur][m][0].z : 0.f; a[3] = rok[m] ? ra[cur][m][0].w : 0.f;
            a[4] = rok[m] ? ra[cur][m][1].x : 0.f; a[5] = rok[m] ? ra[cur][m][1].y : 0.f; a[6] = rok[m] ? ra[cur][m][1].z : 0.f; a[7] = rok[m] ? ra[cur][m][1].w : 0.f;
            Frag<LP> fa;
            fa.set(a);
            acc[m][0] = Frag<LP>::mma(fa, f0, acc[m][0]);
            acc[m][1] = Frag<LP>::mma(fa, f1, acc[m][1]);
        }
    };
    if (g0 < g1) {          // (as in the dgrad: no branch around a load; step g0 + t lives in register set t % D)
        const int last = g1 - 1;
        sfor<D - 1>([&](auto U) { load(min(g0 + (int)U.value, last), U); });
        int g = g0;
        for (; g + D <= g1; g += D)
            sfor<D>([&](auto U) {
                constexpr int u = decltype(U)::value;
                load(min(g + u + D - 1, last), SC<(u + D - 1) % D>{});
                __builtin_amdgcn_sched_barrier(0);
                compute(U);
                __builtin_amdgcn_sched_barrier(0);
            });
        sfor<D - 1>([&](auto U) { if (g + U.value < g1) compute(U); });
    }
    if (want_bias) {                       // lanes (co, a) x 4 pixel groups -> one value per channel, fixed shuffle tree
        bsum += __shfl_xor(bsum, 16);
        bsum += __shfl_xor(bsum, 32);
        bsum += __shfl_xor(bsum, 1);
        if (kg == 0 && (j & 1) == 0 && cok) p.dbias_part[(size_t)split * p.Cout + co] = bsum;
    }
    if (!cok) return;
    float* part = p.partial + (size_t)split * p.Cin * p.Cout * 4;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = mb * 48 + m * 16 + kg * 4 + r;
            if (ci >= p.Cin) continue;
            *reinterpret_cast<float2*>(part + ((size_t)ci * p.Cout + co) * 4 + 2 * (j & 1)) = make_float2(acc[m][0][r], acc[m][1][r]);
        }
}

// The weight gradient with BOTH tensors as 16-bit planes (the 16-bit modes' product path), CT tiles of 8 output channels per wave:
// every x fragment a wave loads is multiplied with 2 CT dY fragments instead of 2, i.e. x travels L2 -> registers Cout / (8 CT)
// times instead of Cout / 8 (with CT = 1 this is convT2_wgrad_kernel<LP, true> fed the same values, and the results are the same bit for
// bit: the same products are summed in the same order, only on another wave).  No branch around a load, D register sets.
template <int LP, int CT, int D>
__global__ __launch_bounds__(256) void convT2_wgrad16_kernel(const Ct2P p) {
    static_assert(LP != 0, "16-bit operands");
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, kg = lane >> 4;
    const int HW = p.H * p.W, oW = 2 * p.W;
    const int steps_per_img = HW / 32;
    const int ctw = (p.ctiles + CT - 1) / CT;                 // wave tiles along the output channels
    // The (channel-tile, input-block) tasks of one split read the same pixels: x once per output-channel tile, dy once per block of 48 input
    // channels.  Workgroups go to the XCDs round-robin, so with xcd_tasks the tasks of split s are consecutive waves of XCD s % 8 and the
    // re-reads hit that XCD's L2 (in plain task order a split's tasks straddle neighbouring workgroups = different XCDs).
    int ct, mb, split;
    if (p.xcd_tasks > 0) {
        const int k = (int)(blockIdx.x >> 3) * 4 + wv, pst = ctw * p.mblocks;
        if (k >= p.xcd_tasks) return;
        split = 8 * (k / pst) + (int)(blockIdx.x & 7);
        if (split >= p.nsplit) return;
        const int within = k % pst;
        ct = within % ctw; mb = within / ctw;
    } else {
        const long long task = (long long)blockIdx.x * 4 + wv;
        if (task >= p.ntasks) return;
        ct = (int)(task % ctw);
        const long long t = task / ctw;
        mb = (int)(t % p.mblocks); split = (int)(t / p.mblocks);
    }
    const int total_steps = p.N * steps_per_img;
    const int g0 = split * p.steps_per_split, g1 = min(total_steps, g0 + p.steps_per_split);
    bool rok[MT];
    const unsigned short* xrow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int ci = mb * 48 + m * 16 + j;
        rok[m] = ci < p.Cin;
        xrow[m] = reinterpret_cast<const unsigned short*>(p.x) + (size_t)(rok[m] ? ci : 0) * HW + 8 * kg;
    }
    int co[CT]; bool cok[CT];
    const unsigned short* dcol[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        co[c] = (ct * CT + c) * 8 + (j >> 1);
        cok[c] = co[c] < p.Cout;
        dcol[c] = reinterpret_cast<const unsigned short*>(p.dy) + (size_t)(cok[c] ? co[c] : 0) * 4 * HW + (size_t)(j & 1) * oW;
    }
    f32x4 acc[CT][MT][2];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int m = 0; m < MT; ++m) { acc[c][m][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[c][m][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const bool want_bias = p.dbias_part != nullptr && mb == 0;      // uniform
    float bsum[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) bsum[c] = 0.f;
    ct_u32x4 rx[D][MT], rh[D][CT][2];
    auto load = [&](int g, auto SL) {
        constexpr int slot = decltype(SL)::value;
        const int n = g / steps_per_img, st = g % steps_per_img;
        const int pix = st * 32 + 8 * kg;
        const int i = pix / p.W, jx = pix % p.W;
#pragma unroll
        for (int m = 0; m < MT; ++m) rx[slot][m] = *reinterpret_cast<const ct_u32x4*>(xrow[m] + (size_t)n * p.xbs + st * 32);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const unsigned short* q = dcol[c] + (size_t)n * p.dybs + (size_t)(2 * i) * oW + 2 * jx;
            rh[slot][c][0] = *reinterpret_cast<const ct_u32x4*>(q); rh[slot][c][1] = *reinterpret_cast<const ct_u32x4*>(q + 8);
        }
    };
    auto compute = [&](auto SL) {
        constexpr int cur = decltype(SL)::value;
        Frag<LP> fa[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const ct_u32x4 w = rx[cur][m];
            frag_from_words<LP>(fa[m], rok[m] ? w[0] : 0u, rok[m] ? w[1] : 0u, rok[m] ? w[2] : 0u, rok[m] ? w[3] : 0u);
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            unsigned d[8], w0[4], w1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { d[k] = cok[c] ? rh[cur][c][0][k] : 0u; d[4 + k] = cok[c] ? rh[cur][c][1][k] : 0u; }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                w0[m] = (d[2 * m] & 0xffffu) | (d[2 * m + 1] << 16);
                w1[m] = (d[2 * m] >> 16) | (d[2 * m + 1] & 0xffff0000u);
            }
            if (want_bias) {
#pragma unroll
                for (int k = 0; k < 8; ++k) bsum[c] += sum2_16<LP>(d[k]);
            }
            Frag<LP> f0, f1;
            frag_from_words<LP>(f0, w0[0], w0[1], w0[2], w0[3]);
            frag_from_words<LP>(f1, w1[0], w1[1], w1[2], w1[3]);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[c][m][0] = Frag<LP>::mma(fa[m], f0, acc[c][m][0]);
                acc[c][m][1] = Frag<LP>::mma(fa[m], f1, acc[c][m][1]);
            }
        }
    };
    if (g0 < g1) {
        const int last = g1 - 1;
        sfor<D - 1>([&](auto U) { load(min(g0 + (int)U.value, last), U); });
        int g = g0;
        for (; g + D <= g1; g += D)
            sfor<D>([&](auto U) {
                constexpr int u = decltype(U)::value;
                load(min(g + u + D - 1, last), SC<(u + D - 1) % D>{});
                __builtin_amdgcn_sched_barrier(0);
                compute(U);
                __builtin_amdgcn_sched_barrier(0);
            });
        sfor<D - 1>([&](auto U) { if (g + U.value < g1) compute(U); });
    }
    float* part = p.partial + (size_t)split * p.Cin * p.Cout * 4;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        if (want_bias) {                   // lanes (co, a) x 4 pixel groups -> one value per channel, fixed shuffle tree
            float b = bsum[c];
            b += __shfl_xor(b, 16);
            b += __shfl_xor(b, 32);
            b += __shfl_xor(b, 1);
            if (kg == 0 && (j & 1) == 0 && cok[c]) p.dbias_part[(size_t)split * p.Cout + co[c]] = b;
        }
        if (!cok[c]) continue;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = mb * 48 + m * 16 + kg * 4 + r;
                if (ci >= p.Cin) continue;
                *reinterpret_cast<float2*>(part + ((size_t)ci * p.Cout + co[c]) * 4 + 2 * (j & 1)) = make_float2(acc[c][m][0][r], acc[c][m][1][r]);
            }
    }
}

// ------------------------------------------------------------------ forward, weights resident in LDS (small Cin)
// Y[(co,a,b)][p] = bias[co] + sum_ci W[ci][(co,a,b)] X[ci][p]: the reduction index is the strided one for both
// operands.  A first attempt in the style of the kernels above (8 strided weight loads per fragment from global) was
// latency-bound; here the whole weight matrix (Cin x 4*Cout <= 192 fp32) sits in LDS for the life of a persistent
// block, rows padded to a stride == 16 mod 32 (conflict-free ds_read_b32 fragments), and X arrives as one float2
// per (lane, k): pixels 2j / 2j+1 feed the two column tiles E / O.  Exact fp32: MFMA i of a group contracts
// ci = 4i .. 4i+3 (lane group kg holds ci = 4i + kg), so any Cin % 4 == 0 runs without padding K.  MFMA rows are
// (co, a, b) with 4 channels per tile: lane (j, kg) ends up with the four (a, b) of channel 4*tile + kg at pixels
// 2j, 2j+1 -> one float4 per output row, 16 lanes = 256 contiguous bytes.
constexpr int FMT = 12;                // row tiles: 4*Cout <= 192
constexpr int FKI = 16;                // Cin <= 64 -> at most 16 MFMA k-groups
template <int KI>
__global__ __launch_bounds__(256, 2) void convT2_fwd_lds_kernel(const Ct2P p, const float* __restrict__ bias, float* __restrict__ y,
                                                                long long ybs, int wstride) {
    extern __shared__ float Wsm[];                  // [Cin][wstride]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 15, kg = lane >> 4;
    const int HW = p.H * p.W, oW = 2 * p.W, M4 = 4 * p.Cout;
    for (int idx = tid; idx < p.Cin * (M4 / 4); idx += 256) {          // 16-byte copies, rows re-strided
        const int ci = idx / (M4 / 4), c4 = idx % (M4 / 4);
        *reinterpret_cast<float4*>(Wsm + ci * wstride + 4 * c4) = *reinterpret_cast<const float4*>(p.w + (size_t)ci * M4 + 4 * c4);
    }
    __syncthreads();
    const int nmt = (M4 + 15) / 16;                                     // uniform, <= FMT
    const int groups = HW / 32;
    const long long ngroups = (long long)p.N * groups;
    const long long gstride = (long long)gridDim.x * 4;
    long long gi = (long long)blockIdx.x * 4 + wv;
    float2 xv[2][KI];
    auto loadx = [&](long long g, auto SL) {
        constexpr int sl = decltype(SL)::value;
        const int n = (int)(g / groups), gg = (int)(g % groups);
        const float* xn = p.x + (size_t)n * p.xbs + gg * 32 + 2 * j + (size_t)kg * HW;
#pragma unroll
        for (int i = 0; i < KI; ++i) xv[sl][i] = *reinterpret_cast<const float2*>(xn + (size_t)(4 * i < p.Cin ? 4 * i : 0) * HW);
    };
    auto compute = [&](long long g, auto SL) {
        constexpr int sl = decltype(SL)::value;
        f32x4 acc[FMT][2];
#pragma unroll
        for (int m = 0; m < FMT; ++m) { acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            if (4 * i < p.Cin) {                                         // uniform
                const float* wr = Wsm + (4 * i + kg) * wstride + j;
#pragma unroll
                for (int m = 0; m < FMT; ++m) {
                    if (m < nmt) {
                        const float a = (16 * m + j < M4) ? wr[16 * m] : 0.f;
                        acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv[sl][i].x, acc[m][0], 0, 0, 0);
                        acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv[sl][i].y, acc[m][1], 0, 0, 0);
                    }
                }
            }
        }
        const int n = (int)(g / groups), gg = (int)(g % groups);
        const int pix = gg * 32 + 2 * j;
        const int iy = pix / p.W, jx = pix % p.W;
        float* yn = y + (size_t)n * ybs + (size_t)(2 * iy) * oW + 2 * jx;
#pragma unroll
        for (int m = 0; m < FMT; ++m) {
            const int co = 4 * m + kg;
            if (m >= nmt || co >= p.Cout) continue;
            const float bv = bias ? bias[co] : 0.f;
            float* d = yn + (size_t)co * 4 * HW;
            *reinterpret_cast<float4*>(d) = make_float4(acc[m][0][0] + bv, acc[m][0][1] + bv, acc[m][1][0] + bv, acc[m][1][1] + bv);
            *reinterpret_cast<float4*>(d + oW) = make_float4(acc[m][0][2] + bv, acc[m][0][3] + bv, acc[m][1][2] + bv, acc[m][1][3] + bv);
        }
    };
    if (gi < ngroups) loadx(gi, S0{});
    for (; gi < ngroups; gi += 2 * gstride) {
        if (gi + gstride < ngroups) loadx(gi + gstride, S1{});
        compute(gi, S0{});
        if (gi + gstride < ngroups) {
            if (gi + 2 * gstride < ngroups) loadx(gi + 2 * gstride, S0{});
            compute(gi + gstride, S1{});
        }
    }
}

// The same forward with the OUTPUT written straight into the 16-bit channel-blocked layout the 3x3 convs read
// (MTBC_LAYOUT_C8: [n][Cout/8][4*H*W][8]) instead of fp32 planes that a pack pass would convert: the up-sampled tensor is
// 4x the input and is consumed by 3x3 convs only, so the fp32 copy (write 4 B + read 4 B per element) disappears.
// Arithmetic is unchanged (fp32 MFMA over the same k order, bias, then ONE round-to-nearest-even) -> bit-identical to
// convT2_fwd_lds_kernel followed by mtbc_c8_pack.  Only the weight image differs: columns are (position, channel)
// instead of (channel, position), so a lane's four accumulator rows are four consecutive CHANNELS of one output
// pixel = one 8-byte store (the lane pair kg, kg^1 completes the 16-byte piece).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
template <bool F16> __device__ __forceinline__ unsigned cvt_pk16(float a, float b) {
    if constexpr (F16) return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a, b}, f16x2_t));
    else return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));
}
template <int KI, bool F16>
__global__ __launch_bounds__(256, 2) void convT2_fwd_c8_kernel(const Ct2P p, const float* __restrict__ bias, unsigned short* __restrict__ y8,
                                                               long long ybs, int wstride, int cp) {
    extern __shared__ float Wsm[];                  // [Cin][wstride], column = pos*cp + co ; then bias[cp]
    float* bias_s = Wsm + p.Cin * wstride;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 15, kg = lane >> 4;
    const int HW = p.H * p.W, oW = 2 * p.W, oHW = 4 * HW;
    for (int idx = tid; idx < p.Cin * cp * 4; idx += 256) {
        const int pos = idx & 3, co = (idx >> 2) % cp, ci = (idx >> 2) / cp;
        Wsm[ci * wstride + pos * cp + co] = co < p.Cout ? p.w[((size_t)ci * p.Cout + co) * 4 + pos] : 0.f;
    }
    for (int c = tid; c < cp; c += 256) bias_s[c] = (bias && c < p.Cout) ? bias[c] : 0.f;
    __syncthreads();
    const int ct = cp / 16, nmt = 4 * ct;                               // uniform, nmt <= FMT
    const int groups = HW / 32;
    const long long ngroups = (long long)p.N * groups;
    const long long gstride = (long long)gridDim.x * 4;
    long long gi = (long long)blockIdx.x * 4 + wv;
    float xv[2][KI][2];
    auto loadx = [&](long long g, auto SL) {
        constexpr int sl = decltype(SL)::value;
        const int n = (int)(g / groups), gg = (int)(g % groups);
        const float* xn = p.x + (size_t)n * p.xbs + gg * 32 + j + (size_t)kg * HW;
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            const float* q = xn + (size_t)(4 * i < p.Cin ? 4 * i : 0) * HW;
            xv[sl][i][0] = q[0]; xv[sl][i][1] = q[16];
        }
    };
    auto compute = [&](long long g, auto SL) {
        constexpr int sl = decltype(SL)::value;
        f32x4 acc[FMT][2];
#pragma unroll
        for (int m = 0; m < FMT; ++m) { acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            if (4 * i < p.Cin) {                                         // uniform
                const float* wr = Wsm + (4 * i + kg) * wstride + j;
#pragma unroll
                for (int m = 0; m < FMT; ++m) {
                    if (m < nmt) {
                        const float a = wr[16 * m];
                        acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv[sl][i][0], acc[m][0], 0, 0, 0);
                        acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv[sl][i][1], acc[m][1], 0, 0, 0);
                    }
                }
            }
        }
        const int n = (int)(g / groups), gg = (int)(g % groups);
        unsigned short* yn = y8 + (size_t)n * ybs + 4 * (kg & 1);
        int obase[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pix = gg * 32 + 16 * h + j;
            obase[h] = (2 * (pix / p.W)) * oW + 2 * (pix % p.W);
        }
#pragma unroll
        for (int m = 0; m < FMT; ++m) {
            if (m >= nmt) continue;
            const int pos = m / ct, cb = m - pos * ct;                   // uniform
            const int c4 = 16 * cb + 4 * kg;
            if (c4 >= p.Cout) continue;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + c4);
            unsigned short* d = yn + ((size_t)(c4 >> 3) * oHW + (pos >> 1) * oW + (pos & 1)) * 8;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 v = acc[m][h] + bv;
                *reinterpret_cast<uint2*>(d + (size_t)obase[h] * 8) = make_uint2(cvt_pk16<F16>(v[0], v[1]), cvt_pk16<F16>(v[2], v[3]));
            }
        }
    };
    if (gi < ngroups) loadx(gi, S0{});
    for (; gi < ngroups; gi += 2 * gstride) {
        if (gi + gstride < ngroups) loadx(gi + gstride, S1{});
        compute(gi, S0{});
        if (gi + gstride < ngroups) {
            if (gi + 2 * gstride < ngroups) loadx(gi + 2 * gstride, S0{});
            compute(gi + gstride, S1{});
        }
    }
}

// Forward on the 16-bit MFMA with BOTH tensors channel-blocked (x_layout = y_layout = MTBC_LAYOUT_C8): a stored piece of x
// (8 channels of one pixel) IS a lane's share of the B fragment of v_mfma_f32_16x16x32, so x goes straight from HBM into
// fragments (16-byte loads, 256 contiguous bytes per 16 lanes); the weights of the block's channel slice sit in LDS as
// 16-bit rows [(position, channel)][ci] (row stride = 2 mod 4 pieces: conflict-free), read as A fragments with ds_read_b128.  A wave =
// 32 input pixels x (4 positions x 16*MTC channels); fp32 accumulate, bias, one RNE, 8-byte stores into the output pieces.
// The level-0 up-convolution (48 -> 48 @128x128, 9.7 GFLOP) was fp32-MFMA-bound at 0.12 ms; wider ones took the generic
// fp32 GEMM + a pack pass.
template <int MTC, bool F16>
__global__ __launch_bounds__(256, 2) void convT2_fwd_lp_c8_kernel(const Ct2P p, const float* __restrict__ bias, const unsigned short* __restrict__ x8,
                                                                  long long xbs, unsigned short* __restrict__ y8, long long ybs, int kpad, int wrow) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned short Wl[];          // [4 * 16*MTC rows][wrow], then bias[16*MTC] (fp32)
    constexpr int CB = 16 * MTC;
    float* bias_s = reinterpret_cast<float*>(Wl + 4 * CB * wrow);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 15, kg = lane >> 4;
    const int HW = p.H * p.W, oW = 2 * p.W, oHW = 4 * HW;
    const int cb0 = blockIdx.y * CB;
    // the block's weight slice is 4 * CB CONTIGUOUS floats per input channel (w[ci][cb0 .. cb0 + CB)[pos]): read it in memory
    // order (consecutive threads = consecutive floats), scatter the 16-bit values into the LDS rows.  (Reading it in LDS
    // order -- ci fastest -- made every load a 4-byte access at a stride of 4 * Cout floats: 40 us of the 65 us the
    // 384 -> 192 up-convolution took.)
    // One (ci, channel) pair = the 4 positions = one 16-byte load and four 16-bit LDS stores, 8 loads in flight per thread: a block
    // has only a handful of pixel tasks on the deep levels, so a prologue of 96 dependent {4-byte load, convert, store} rounds WAS the
    // launch (43 us for 2 us of MFMA work).
    const int npairs = CB * kpad;
    for (int base = tid; base < npairs; base += 256 * 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + 256 * u, c = idx % CB, ci = idx / CB, co = cb0 + c;
            v[u] = (idx < npairs && ci < p.Cin && co < p.Cout) ? ld4(p.w + ((size_t)ci * p.Cout + co) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + 256 * u, c = idx % CB, ci = idx / CB;
            if (idx >= npairs) continue;
            const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int pos = 0; pos < 4; ++pos) {
                unsigned short h;
                if constexpr (F16) h = __builtin_bit_cast(unsigned short, (_Float16)e[pos]); else h = __builtin_bit_cast(unsigned short, (__bf16)e[pos]);
                Wl[(pos * CB + c) * wrow + ci] = h;
            }
        }
    }
    for (int c = tid; c < CB; c += 256) bias_s[c] = (bias && cb0 + c < p.Cout) ? bias[cb0 + c] : 0.f;
    __syncthreads();
    const int nchunks = kpad / 32, groups8 = p.Cin / 8;
    const int pgroups = HW / 32;
    const long long ntasks = (long long)p.N * pgroups;
    for (long long gi = (long long)blockIdx.x * 4 + wv; gi < ntasks; gi += (long long)gridDim.x * 4) {
        const int n = (int)(gi / pgroups), gg = (int)(gi % pgroups);
        const unsigned short* xn = x8 + (size_t)n * xbs + (size_t)(gg * 32 + j) * 8;
        f32x4 acc[4][MTC][2];
#pragma unroll
        for (int pos = 0; pos < 4; ++pos)
#pragma unroll
            for (int m = 0; m < MTC; ++m) { acc[pos][m][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[pos][m][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        u32x4_t b[2][2];
        auto loadb = [&](int ch, auto SL) {          // register-set index is a compile-time constant (S0 / S1)
            constexpr int slot = decltype(SL)::value;
            const int grp = ch * 4 + kg;
            const unsigned short* q = xn + (size_t)(grp < groups8 ? grp : 0) * HW * 8;
#pragma unroll
            for (int g = 0; g < 2; ++g) b[slot][g] = *reinterpret_cast<const u32x4_t*>(q + g * 16 * 8);
        };
        auto compute = [&](int ch, auto SL) {
            constexpr int cur = decltype(SL)::value;
            const bool live = ch * 4 + kg < groups8;                    // a ragged last chunk: absent groups contribute zeros
            u32x4_t bz[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) bz[g] = live ? b[cur][g] : (u32x4_t){0u, 0u, 0u, 0u};
#pragma unroll
            for (int pos = 0; pos < 4; ++pos)
#pragma unroll
                for (int m = 0; m < MTC; ++m) {
                    const u32x4_t a = *reinterpret_cast<const u32x4_t*>(Wl + ((pos * CB + m * 16 + j) * wrow + ch * 32 + 8 * kg));
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        if constexpr (F16) acc[pos][m][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, bz[g]), acc[pos][m][g], 0, 0, 0);
                        else acc[pos][m][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, bz[g]), acc[pos][m][g], 0, 0, 0);
                    }
                }
        };
        loadb(0, S0{});
        for (int ch = 0; ch < nchunks; ch += 2) {
            if (ch + 1 < nchunks) loadb(ch + 1, S1{});
            compute(ch, S0{});
            if (ch + 1 < nchunks) {
                if (ch + 2 < nchunks) loadb(ch + 2, S0{});
                compute(ch + 1, S1{});
            }
        }
        unsigned short* yn = y8 + (size_t)n * ybs + 4 * (kg & 1);
        int obase[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int pix = gg * 32 + 16 * g + j;
            obase[g] = (2 * (pix / p.W)) * oW + 2 * (pix % p.W);
        }
#pragma unroll
        for (int m = 0; m < MTC; ++m) {
            const int c4 = cb0 + 16 * m + 4 * kg;
            if (c4 >= p.Cout) continue;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + 16 * m + 4 * kg);
#pragma unroll
            for (int pos = 0; pos < 4; ++pos) {
                unsigned short* d = yn + ((size_t)(c4 >> 3) * oHW + (pos >> 1) * oW + (pos & 1)) * 8;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const f32x4 v = acc[pos][m][g] + bv;
                    *reinterpret_cast<uint2*>(d + (size_t)obase[g] * 8) = make_uint2(cvt_pk16<F16>(v[0], v[1]), cvt_pk16<F16>(v[2], v[3]));
                }
            }
        }
    }
}

// Input gradient with the weights of the block's 48 input channels resident in LDS (16-bit dY only).  convT2_dgrad_kernel above
// re-loads every weight it multiplies from L2 -- 48 x 4Cout fp32 values per 32-pixel task, three times the bytes of the dY the task
// reads, as fragment-shaped loads (16 rows x 128 bytes per instruction) -- and on the deep levels (a few thousand pixels, hundreds of
// channels) that load path was the whole launch: 48 us for 4 us of MFMA work, the same at every level because pixels x Cin x Cout
// is.  Here a block converts its slice [48][4Cout] once (memory order = LDS row order: two 16-byte loads, one ds_write_b128 per
// piece; row stride = 2 mod 4 pieces, conflict-free ds_read_b128), then its waves walk pixel tasks: A fragments from LDS, dY
// straight into B fragments as before, D register sets of dY in flight.  Same MFMA operands in the same order as the kernel above:
// bit-identical results.
template <int LP, int D>
__global__ __launch_bounds__(256, 2) void convT2_dgrad_lds_kernel(const Ct2P p, const int wrow) {
    static_assert(LP != 0, "16-bit operands");
    extern __shared__ __attribute__((aligned(16))) unsigned short Wd[];          // [48][wrow]
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kg = lane >> 4;
    const int mb = blockIdx.y;
    const int HW = p.H * p.W, oW = 2 * p.W, M4 = 4 * p.Cout;
    const int ppr = M4 / 8, npieces = 48 * ppr;
    for (int base = tid; base < npieces; base += 256 * 4) {
        float4 v[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = min(base + 256 * u, npieces - 1), r = idx / ppr, q = idx - r * ppr;
            const float* src = p.w + (size_t)min(mb * 48 + r, p.Cin - 1) * M4 + 8 * q;
            v[u][0] = ld4(src); v[u][1] = ld4(src + 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + 256 * u;
            if (idx >= npieces) continue;
            const int r = idx / ppr, q = idx - r * ppr;
            ct_u32x4 o = {cvt_pk16<LP == 2>(v[u][0].x, v[u][0].y), cvt_pk16<LP == 2>(v[u][0].z, v[u][0].w),
                          cvt_pk16<LP == 2>(v[u][1].x, v[u][1].y), cvt_pk16<LP == 2>(v[u][1].z, v[u][1].w)};
            if (mb * 48 + r >= p.Cin) o = (ct_u32x4){0u, 0u, 0u, 0u};
            *reinterpret_cast<ct_u32x4*>(Wd + (size_t)r * wrow + 8 * q) = o;
        }
    }
    __syncthreads();
    const int groups = HW / 32, nsteps = p.Cout / 8, last = nsteps - 1;
    const long long ntasks = (long long)p.N * groups;
    const unsigned short* arow = Wd + (size_t)j * wrow + 8 * kg;
    for (long long gi = (long long)blockIdx.x * 4 + wv; gi < ntasks; gi += (long long)gridDim.x * 4) {
        const int n = (int)(gi / groups), g = (int)(gi % groups);
        const int pix = g * 32 + 2 * j;
        const int i = pix / p.W, jx = pix % p.W;
        const unsigned short* dyn16 = reinterpret_cast<const unsigned short*>(p.dy) + (size_t)n * p.dybs + (size_t)(2 * i) * oW + 2 * jx +
                                      (size_t)(2 * kg) * 4 * HW;
        f32x4 acc[MT][2];
#pragma unroll
        for (int m = 0; m < MT; ++m) { acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        uint2 rh[D][4];                       // (channel c, row a) -> {even pixel (b0,b1), odd pixel (b0,b1)}
        auto load = [&](int s, auto SL) {
            constexpr int slot = decltype(SL)::value;
            const unsigned short* q = dyn16 + (size_t)(8 * s) * 4 * HW;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                rh[slot][2 * c] = *reinterpret_cast<const uint2*>(q + (size_t)c * 4 * HW);
                rh[slot][2 * c + 1] = *reinterpret_cast<const uint2*>(q + (size_t)c * 4 * HW + oW);
            }
        };
        auto compute = [&](int s, auto SL) {
            constexpr int cur = decltype(SL)::value;
            Frag<LP> fe, fo;
            frag_from_words<LP>(fe, rh[cur][0].x, rh[cur][1].x, rh[cur][2].x, rh[cur][3].x);
            frag_from_words<LP>(fo, rh[cur][0].y, rh[cur][1].y, rh[cur][2].y, rh[cur][3].y);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                Frag<LP> fa;
                fa.v = __builtin_bit_cast(decltype(fa.v), *reinterpret_cast<const ct_u32x4*>(arow + (size_t)(m * 16) * wrow + 32 * s));
                acc[m][0] = Frag<LP>::mma(fa, fe, acc[m][0]);
                acc[m][1] = Frag<LP>::mma(fa, fo, acc[m][1]);
            }
        };
        sfor<D - 1>([&](auto U) { load(min((int)U.value, last), U); });
        int s = 0;
        for (; s + D <= nsteps; s += D)
            sfor<D>([&](auto U) {
                constexpr int u = decltype(U)::value;
                load(min(s + u + D - 1, last), SC<(u + D - 1) % D>{});
                __builtin_amdgcn_sched_barrier(0);
                compute(s + u, U);
                __builtin_amdgcn_sched_barrier(0);
            });
        sfor<D - 1>([&](auto U) { if (s + U.value < nsteps) compute(s + U.value, U); });
        float* dxn = p.dx + (size_t)n * p.dxbs + g * 32 + 2 * j;
        float2 old[MT][4];
        if (p.acc_dx) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) old[m][r] = *reinterpret_cast<const float2*>(dxn + (size_t)min(mb * 48 + m * 16 + kg * 4 + r, p.Cin - 1) * HW);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = mb * 48 + m * 16 + kg * 4 + r;
                if (ci >= p.Cin) continue;
                float2 v = make_float2(acc[m][0][r], acc[m][1][r]);
                if (p.acc_dx) { v.x += old[m][r].x; v.y += old[m][r].y; }
                *reinterpret_cast<float2*>(dxn + (size_t)ci * HW) = v;
            }
    }
}

bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

void fill(const mtbc_convT_args* a, Ct2P* p) {
    p->N = a->N; p->H = a->H; p->W = a->W; p->Cin = a->Cin; p->Cout = a->Cout;
    p->x = a->x; p->xbs = a->x_batch_stride; p->w = a->w; p->dy = a->dy; p->dybs = a->dy_batch_stride;
    p->dx = a->dx; p->dxbs = a->dx_batch_stride; p->acc_dx = a->accumulate_dx; p->partial = nullptr; p->dbias_part = nullptr;
    p->mblocks = cdiv(a->Cin, 48); p->ctiles = cdiv(a->Cout, 8); p->steps_per_split = 1; p->nsplit = 1; p->ntasks = 0; p->xcd_tasks = 0;
}

}  // namespace

// eligibility of the direct-to-fragment kernels (k == 2 only); everything else takes the generic GEMM in pool_up.hip
bool mtbc_i_convT2_dgrad_ok(const mtbc_convT_args* a) {
    const int HW = a->H * a->W;
    if (a->dy_type16 && (a->dy_type16 != a->compute || a->dy_batch_stride % 8 != 0)) return false;
    return a->k == 2 && HW % 32 == 0 && a->W % 2 == 0 && a->Cout % 2 == 0 && al16(a->dy) && al16(a->w) && al16(a->dx) &&
           a->dy_batch_stride % 4 == 0 && a->dx_batch_stride % 2 == 0;
}
bool mtbc_i_convT2_wgrad_ok(const mtbc_convT_args* a) {
    const int HW = a->H * a->W;
    if (a->dy_type16 && (a->dy_type16 != a->compute || a->dy_batch_stride % 8 != 0)) return false;
    if (a->x_type16 && (a->x_type16 != a->compute || !a->dy_type16 || a->x_batch_stride % 8 != 0)) return false;
    return a->k == 2 && HW % 32 == 0 && a->W % 8 == 0 && al16(a->dy) && al16(a->x) && a->dy_batch_stride % 4 == 0 &&
           a->x_batch_stride % 4 == 0;
}
// 32-pixel steps per split: ~4096 wave tasks in flight, at least 8 steps each (the partial sums are real traffic).  With both tensors
// as 16-bit planes (convT2_wgrad16_kernel) the inputs are half the bytes and the partials weigh twice as much: 1536 tasks
// (sweep 512 .. 16384 per step of the bench: 1.16 / 0.65 / 0.50 / 0.52 / 0.55 / 0.67 / 0.75 ms at 512 / 1024 / 1536 / 2048 / 4096 / 8192 / 16384).
// (A workspace sized before x_type16 was set is sized for 4096: never too small.)
void mtbc_i_convT2_wgrad_plan(const mtbc_convT_args* a, int* steps_per_split, int* nsplit) {
    const long long total_steps = (long long)a->N * (a->H * a->W / 32);
    const long long per_split_tasks = (long long)cdiv(a->Cin, 48) * cdiv(a->Cout, 8);
    static const int tasks_env = mtbc_probe_int("MTBC_CT_WG_TASKS", 0);      // A/B
    long long want = cdiv64(tasks_env > 0 ? tasks_env : (a->x_type16 ? 1536 : 4096), per_split_tasks);
    if (want < 1) want = 1;
    long long sps = cdiv64(total_steps, want);
    if (sps < 8) sps = 8;
    if (sps > total_steps) sps = total_steps;
    *steps_per_split = (int)sps;
    *nsplit = (int)cdiv64(total_steps, sps);
}

// forward fast path: k == 2, the whole weight matrix fits LDS (Cin <= 64, 4*Cout <= 192)
bool mtbc_i_convT2_fwd_ok(const mtbc_convT_args* a) {
    const int HW = a->H * a->W;
    return a->k == 2 && HW % 32 == 0 && a->W % 2 == 0 && a->Cin % 4 == 0 && a->Cin <= 4 * FKI && 4 * a->Cout <= 16 * FMT &&
           al16(a->x) && al16(a->y) && al16(a->w) && a->x_batch_stride % 2 == 0 && a->y_batch_stride % 4 == 0;
}
int mtbc_i_convT2_fwd(const mtbc_convT_args* a, hipStream_t st) {
    Ct2P p; fill(a, &p);
    const int wstride = (4 * a->Cout + 31) / 32 * 32 + 16;             // == 16 mod 32
    const size_t lds = (size_t)a->Cin * wstride * sizeof(float);
    const long long groups = (long long)a->N * (a->H * a->W / 32);
    int blocks = (int)(groups < 4 * 512 ? cdiv64(groups, 4) : 512);      // 2 resident blocks per CU (register-limited)
    const int ki = cdiv(a->Cin, 4);
#define MTBC_CT2F(KI_)                                                                                                     \
    do {                                                                                                                   \
        MTBC_ENSURE_DYN_LDS((&convT2_fwd_lds_kernel<KI_>), 64 * 1024);                                                     \
        hipLaunchKernelGGL(convT2_fwd_lds_kernel<KI_>, dim3(blocks), dim3(256), lds, st, p, a->bias, a->y,                 \
                           (long long)a->y_batch_stride, wstride);                                                         \
    } while (0)
    if (ki <= 8) MTBC_CT2F(8); else if (ki <= 12) MTBC_CT2F(12); else MTBC_CT2F(16);
#undef MTBC_CT2F
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}

// forward into a 16-bit channel-blocked output (y_layout = MTBC_LAYOUT_C8)
bool mtbc_i_convT2_fwd_c8_ok(const mtbc_convT_args* a) {
    const int HW = a->H * a->W, cp = (a->Cout + 15) / 16 * 16;
    return a->k == 2 && HW % 32 == 0 && a->Cin % 4 == 0 && a->Cin <= 4 * FKI && 4 * cp <= 16 * FMT && a->Cout % 8 == 0 &&
           (a->y_type == 1 || a->y_type == 2) && al16(a->y) && a->y_batch_stride % 8 == 0;
}
int mtbc_i_convT2_fwd_c8(const mtbc_convT_args* a, hipStream_t st) {
    Ct2P p; fill(a, &p);
    const int cp = (a->Cout + 15) / 16 * 16;
    const int wstride = (4 * cp + 31) / 32 * 32 + 16;                  // == 16 mod 32
    const size_t lds = ((size_t)a->Cin * wstride + cp) * sizeof(float);
    const long long groups = (long long)a->N * (a->H * a->W / 32);
    int blocks = (int)(groups < 4 * 512 ? cdiv64(groups, 4) : 512);      // 2 resident blocks per CU (register-limited)
    const int ki = cdiv(a->Cin, 4);
    unsigned short* y8 = reinterpret_cast<unsigned short*>(a->y);
#define MTBC_CT2F8(KI_, F16_)                                                                                              \
    do {                                                                                                                   \
        MTBC_ENSURE_DYN_LDS((&convT2_fwd_c8_kernel<KI_, F16_>), 64 * 1024);                                                \
        hipLaunchKernelGGL((convT2_fwd_c8_kernel<KI_, F16_>), dim3(blocks), dim3(256), lds, st, p, a->bias, y8,            \
                           (long long)a->y_batch_stride, wstride, cp);                                                     \
    } while (0)
    if (a->y_type == 2) { if (ki <= 8) MTBC_CT2F8(8, true); else if (ki <= 12) MTBC_CT2F8(12, true); else MTBC_CT2F8(16, true); }
    else { if (ki <= 8) MTBC_CT2F8(8, false); else if (ki <= 12) MTBC_CT2F8(12, false); else MTBC_CT2F8(16, false); }
#undef MTBC_CT2F8
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}

// forward with channel-blocked 16-bit input AND output (x_layout = y_layout = MTBC_LAYOUT_C8): the 16-bit MFMA kernel
static int lp_c8_mtc(const mtbc_convT_args* a, int* kpad, int* wrow, size_t* lds) {
    *kpad = (a->Cin + 31) / 32 * 32; *wrow = *kpad + 16;      // row stride = 2 (mod 4) 16-byte pieces: conflict-free ds_read_b128 A fragments
    const int cp16 = (a->Cout + 15) / 16;
    for (int mtc = cp16 < 3 ? cp16 : 3; mtc >= 1; --mtc) {
        *lds = (size_t)4 * 16 * mtc * *wrow * 2 + 16 * mtc * sizeof(float);
        if (*lds <= 72 * 1024) return mtc;
    }
    return 0;
}
bool mtbc_i_convT2_fwd_lp_c8_ok(const mtbc_convT_args* a) {
    int kpad, wrow; size_t lds;
    return a->k == 2 && (a->H * a->W) % 32 == 0 && a->Cin % 8 == 0 && a->Cout % 8 == 0 && (a->y_type == 1 || a->y_type == 2) &&
           al16(a->y) && al16(a->x) && al16(a->w) && a->y_batch_stride % 8 == 0 && a->x_batch_stride % 8 == 0 && lp_c8_mtc(a, &kpad, &wrow, &lds) > 0;
}
int mtbc_i_convT2_fwd_lp_c8(const mtbc_convT_args* a, hipStream_t st) {
    Ct2P p; fill(a, &p);
    int kpad, wrow; size_t lds;
    const int mtc = lp_c8_mtc(a, &kpad, &wrow, &lds);
    const long long tasks = (long long)a->N * (a->H * a->W / 32);
    const int yb = cdiv(a->Cout, 16 * mtc);
    int gx = 512 / yb; if (gx < 1) gx = 1;
    if ((long long)gx * 4 > tasks) gx = (int)cdiv64(tasks, 4);
    const dim3 grid(gx, yb);
    const unsigned short* x8 = reinterpret_cast<const unsigned short*>(a->x);
    unsigned short* y8 = reinterpret_cast<unsigned short*>(a->y);
#define MTBC_CT2LP(MTC_, F16_)                                                                                             \
    do {                                                                                                                   \
        MTBC_ENSURE_DYN_LDS((&convT2_fwd_lp_c8_kernel<MTC_, F16_>), 80 * 1024);                                            \
        hipLaunchKernelGGL((convT2_fwd_lp_c8_kernel<MTC_, F16_>), grid, dim3(256), lds, st, p, a->bias, x8,                \
                           (long long)a->x_batch_stride, y8, (long long)a->y_batch_stride, kpad, wrow);                    \
    } while (0)
    if (a->y_type == 2) { if (mtc == 3) MTBC_CT2LP(3, true); else if (mtc == 2) MTBC_CT2LP(2, true); else MTBC_CT2LP(1, true); }
    else { if (mtc == 3) MTBC_CT2LP(3, false); else if (mtc == 2) MTBC_CT2LP(2, false); else MTBC_CT2LP(1, false); }
#undef MTBC_CT2LP
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}

int mtbc_i_convT2_dgrad(const mtbc_convT_args* a, int compute, hipStream_t st) {
    Ct2P p; fill(a, &p);
    p.ntasks = (long long)p.mblocks * a->N * (a->H * a->W / 32);
    const unsigned blocks = (unsigned)cdiv64(p.ntasks, 4);
    static const bool no_lds = mtbc_probe_set("MTBC_CT_DGRAD_DIRECT");      // A/B: the direct kernel for every shape
    const int wrow = 4 * a->Cout + 16;
    const size_t lds = (size_t)48 * wrow * 2;
    if (a->dy_type16 && !no_lds && a->Cout % 8 == 0 && lds <= 104 * 1024) {      // (Cout <= 256; one block per CU above 192)      // weights of the block's 48 input channels in LDS
        const long long tasks = (long long)a->N * (a->H * a->W / 32);
        int gx = 512 / p.mblocks; if (gx < 1) gx = 1;
        if ((long long)gx * 4 > tasks) gx = (int)cdiv64(tasks, 4);
        const dim3 grid(gx, p.mblocks);
#define MTBC_CT2DG(LP_, D_)                                                                                              \
        do {                                                                                                             \
            MTBC_ENSURE_DYN_LDS((&convT2_dgrad_lds_kernel<LP_, D_>), 104 * 1024);                                        \
            hipLaunchKernelGGL((convT2_dgrad_lds_kernel<LP_, D_>), grid, dim3(256), lds, st, p, wrow);                   \
        } while (0)
        if (compute == 1) MTBC_CT2DG(1, 4); else MTBC_CT2DG(2, 4);
#undef MTBC_CT2DG
        MTBC_CHECK_LAUNCH();
        return MTBC_OK;
    }
    static const int depth_env = mtbc_probe_int("MTBC_CT_DEPTH", 0);      // A/B: register sets (2 | 4), 0 = by shape
    const bool deep = depth_env ? depth_env >= 4 : a->Cout >= 96;          // long step chains, few waves: three steps of loads in flight
    if (a->dy_type16) {
        if (compute == 1) { if (deep) hipLaunchKernelGGL((convT2_dgrad_kernel<1, true, 4>), dim3(blocks), dim3(256), 0, st, p); else hipLaunchKernelGGL((convT2_dgrad_kernel<1, true>), dim3(blocks), dim3(256), 0, st, p); }
        else { if (deep) hipLaunchKernelGGL((convT2_dgrad_kernel<2, true, 4>), dim3(blocks), dim3(256), 0, st, p); else hipLaunchKernelGGL((convT2_dgrad_kernel<2, true>), dim3(blocks), dim3(256), 0, st, p); }
    } else if (compute == 1) hipLaunchKernelGGL(convT2_dgrad_kernel<1>, dim3(blocks), dim3(256), 0, st, p);
    else if (compute == 2) hipLaunchKernelGGL(convT2_dgrad_kernel<2>, dim3(blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(convT2_dgrad_kernel<0>, dim3(blocks), dim3(256), 0, st, p);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}

int mtbc_i_convT2_wgrad(const mtbc_convT_args* a, int compute, float* partial, float* dbias_part, int steps_per_split, int nsplit, hipStream_t st) {
    Ct2P p; fill(a, &p);
    p.partial = partial; p.dbias_part = dbias_part; p.steps_per_split = steps_per_split; p.nsplit = nsplit;
    p.ntasks = (long long)p.mblocks * p.ctiles * nsplit;
    const unsigned blocks = (unsigned)cdiv64(p.ntasks, 4);
    if (a->x_type16) {          // 16-bit planar x (and dy): both operands arrive as fragments
        if (!a->dy_type16 || a->x_type16 != compute) return MTBC_E_UNSUPPORTED;
        static const int ct_env = mtbc_probe_int("MTBC_CT_WG_CT", 0);      // A/B: tiles of 8 output channels per wave (1 | 2), 0 = by shape
        const int ctn = ct_env ? ct_env : (p.ctiles % 2 == 0 ? 2 : 1);
        static const int xcd_env = mtbc_probe_int("MTBC_CT_WG_XCD", 1);      // A/B: a split's tasks on one XCD (1) | plain task order (0)
        const long long pst = (long long)p.mblocks * cdiv(p.ctiles, ctn);
        const long long per_xcd = (long long)cdiv(nsplit, 8) * pst;
        unsigned xblocks = 0;
        if (xcd_env && nsplit >= 8 && per_xcd < (1ll << 30)) { p.xcd_tasks = (int)per_xcd; xblocks = 8u * (unsigned)cdiv64(per_xcd, 4); }
        if (ctn == 2) {
            p.ntasks = (long long)p.mblocks * ((p.ctiles + 1) / 2) * nsplit;
            const unsigned b2 = xblocks ? xblocks : (unsigned)cdiv64(p.ntasks, 4);
            static const int depth_env = mtbc_probe_int("MTBC_CT_DEPTH", 2);      // A/B: register sets (2 | 4); measured 0.59 vs 0.62 ms per step
            if (depth_env == 2) {
                if (compute == 1) hipLaunchKernelGGL((convT2_wgrad16_kernel<1, 2, 2>), dim3(b2), dim3(256), 0, st, p);
                else hipLaunchKernelGGL((convT2_wgrad16_kernel<2, 2, 2>), dim3(b2), dim3(256), 0, st, p);
            } else if (compute == 1) hipLaunchKernelGGL((convT2_wgrad16_kernel<1, 2, 4>), dim3(b2), dim3(256), 0, st, p);
            else hipLaunchKernelGGL((convT2_wgrad16_kernel<2, 2, 4>), dim3(b2), dim3(256), 0, st, p);
        } else {
            const unsigned b1 = xblocks ? xblocks : blocks;
            if (compute == 1) hipLaunchKernelGGL((convT2_wgrad16_kernel<1, 1, 4>), dim3(b1), dim3(256), 0, st, p);
            else hipLaunchKernelGGL((convT2_wgrad16_kernel<2, 1, 4>), dim3(b1), dim3(256), 0, st, p);
        }
    } else if (a->dy_type16) {
        if (compute == 1) hipLaunchKernelGGL((convT2_wgrad_kernel<1, true>), dim3(blocks), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((convT2_wgrad_kernel<2, true>), dim3(blocks), dim3(256), 0, st, p);
    } else if (compute == 1) hipLaunchKernelGGL(convT2_wgrad_kernel<1>, dim3(blocks), dim3(256), 0, st, p);
    else if (compute == 2) hipLaunchKernelGGL(convT2_wgrad_kernel<2>, dim3(blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(convT2_wgrad_kernel<0>, dim3(blocks), dim3(256), 0, st, p);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}
