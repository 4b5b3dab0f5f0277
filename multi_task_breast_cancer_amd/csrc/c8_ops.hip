// The small consumers of a conv-cell activation -- MaxPool2d(2,2) and the 1x1 deep-supervision heads -- on the 16-bit
// channel-blocked tensor the 3x3 convs read (MTBC_LAYOUT_C8, [n][C/8][H*W][8]), so that in the 16-bit compute modes no
// conv-cell output needs fp32 planes at all: InstanceNorm then writes 2 instead of 2 + 4 bytes per element and these
// kernels read 2 instead of 4.
//   max-pool  replaces nn.MaxPool2d(2,2):  MTnnUNet.py:103, MONAI Down in MTUNetPlusPlus.py:48-51
//   conv 1x1  replaces nn.Conv2d(k=1):     MTnnUNet.py:118, MTUNetPlusPlus.py:73-76
// All HBM-bound streaming kernels: one lane = one pixel (or one output window) x one 16-byte piece.
#include "common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float c8_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 c8_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 c8_f16x2 __attribute__((ext_vector_type(2)));

// the two 16-bit values of one dword -> fp32 (exact)
template <bool F16> __device__ __forceinline__ void unpack2(unsigned w, float& lo, float& hi) {
    if constexpr (F16) {
        const c8_f16x2 h = __builtin_bit_cast(c8_f16x2, w);
        lo = (float)h[0]; hi = (float)h[1];
    } else {
        lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xffff0000u);
    }
}
template <bool F16> __device__ __forceinline__ unsigned pack2(float lo, float hi) {      // RNE (exact for representable values)
    if constexpr (F16) return __builtin_bit_cast(unsigned, __builtin_convertvector((c8_f32x2){lo, hi}, c8_f16x2));
    else return __builtin_bit_cast(unsigned, __builtin_convertvector((c8_f32x2){lo, hi}, c8_bf16x2));
}
template <bool F16> __device__ __forceinline__ void unpack8(const u32x4 p, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) unpack2<F16>(p[i], v[2 * i], v[2 * i + 1]);
}

// ------------------------------------------------------------------ MaxPool 2x2 on channel-blocked tensors
struct MpC8P {
    int N, G8, H, W;                       // input spatial size; G8 = C / 8
    const unsigned short* x; long long xbs;          // batch strides in 16-bit elements
    unsigned short* y; long long ybs;
    const float* dy; long long dybs; float* dx; long long dxbs; int acc;      // backward: fp32 planar gradients
    unsigned short* arg;                   // forward, optional: per (n, group, output pixel) the 8 two-bit positions of the maxima
};
// forward: one lane = one output pixel of one channel group: reads 2 rows x 32 contiguous bytes, writes one piece.
// max commutes with the (monotonic) rounding, so this IS the fp32 pool followed by the pack.
template <bool F16>
__global__ void maxpool_c8_fwd_kernel(const MpC8P p) {
    const int oW = p.W >> 1, oH = p.H >> 1;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)p.N * p.G8 * oH * oW;
    if (idx >= total) return;
    const int ox = idx % oW; size_t t = idx / oW;
    const int oy = t % oH; t /= oH;
    const int g = t % p.G8, n = t / p.G8;
    const unsigned short* src = p.x + (size_t)n * p.xbs + ((size_t)g * p.H * p.W + (size_t)(2 * oy) * p.W + 2 * ox) * 8;
    const u32x4 a0 = *reinterpret_cast<const u32x4*>(src), a1 = *reinterpret_cast<const u32x4*>(src + 8);
    const u32x4 b0 = *reinterpret_cast<const u32x4*>(src + (size_t)p.W * 8), b1 = *reinterpret_cast<const u32x4*>(src + (size_t)p.W * 8 + 8);
    float va[8], vb[8], vc[8], vd[8];
    unpack8<F16>(a0, va); unpack8<F16>(a1, vb); unpack8<F16>(b0, vc); unpack8<F16>(b1, vd);
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        o[i] = pack2<F16>(fmaxf(fmaxf(va[2 * i], vb[2 * i]), fmaxf(vc[2 * i], vd[2 * i])),
                          fmaxf(fmaxf(va[2 * i + 1], vb[2 * i + 1]), fmaxf(vc[2 * i + 1], vd[2 * i + 1])));
    *reinterpret_cast<u32x4*>(p.y + (size_t)n * p.ybs + ((size_t)g * oH * oW + (size_t)oy * oW + ox) * 8) = o;
    if (p.arg) {        // where the backward sends a window's gradient: its first maximal element in (0,0),(0,1),(1,0),(1,1) order (ATen)
        unsigned code = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float m = va[c]; unsigned arg = 0;
            if (vb[c] > m || vb[c] != vb[c]) { m = vb[c]; arg = 1; }
            if (vc[c] > m || vc[c] != vc[c]) { m = vc[c]; arg = 2; }
            if (vd[c] > m || vd[c] != vd[c]) { m = vd[c]; arg = 3; }
            code |= arg << (2 * c);
        }
        p.arg[((size_t)n * p.G8 + g) * oH * oW + (size_t)oy * oW + ox] = (unsigned short)code;
    }
}
// backward: the gradient of a window goes to its first maximal element in (0,0),(0,1),(1,0),(1,1) order (ATen), judged
// on the stored (rounded) values -- the tensor the forward pooled.  dy / dx: fp32 planar.
template <bool F16>
__global__ void maxpool_c8_bwd_kernel(const MpC8P p) {
    const int oW = p.W >> 1, oH = p.H >> 1;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)p.N * p.G8 * oH * oW;
    if (idx >= total) return;
    const int ox = idx % oW; size_t t = idx / oW;
    const int oy = t % oH; t /= oH;
    const int g = t % p.G8, n = t / p.G8;
    const unsigned short* src = p.x + (size_t)n * p.xbs + ((size_t)g * p.H * p.W + (size_t)(2 * oy) * p.W + 2 * ox) * 8;
    const u32x4 a0 = *reinterpret_cast<const u32x4*>(src), a1 = *reinterpret_cast<const u32x4*>(src + 8);
    const u32x4 b0 = *reinterpret_cast<const u32x4*>(src + (size_t)p.W * 8), b1 = *reinterpret_cast<const u32x4*>(src + (size_t)p.W * 8 + 8);
    float va[8], vb[8], vc[8], vd[8];
    unpack8<F16>(a0, va); unpack8<F16>(a1, vb); unpack8<F16>(b0, vc); unpack8<F16>(b1, vd);
    const size_t oHW = (size_t)oH * oW, HW = (size_t)p.H * p.W;
    const float* gy = p.dy + (size_t)n * p.dybs + (size_t)(8 * g) * oHW + (size_t)oy * oW + ox;
    float* o = p.dx + (size_t)n * p.dxbs + (size_t)(8 * g) * HW + (size_t)(2 * oy) * p.W + 2 * ox;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float gv = gy[(size_t)c * oHW];
        float m = va[c]; int arg = 0;
        if (vb[c] > m || vb[c] != vb[c]) { m = vb[c]; arg = 1; }
        if (vc[c] > m || vc[c] != vc[c]) { m = vc[c]; arg = 2; }
        if (vd[c] > m || vd[c] != vd[c]) { m = vd[c]; arg = 3; }
        float2 d0 = make_float2(arg == 0 ? gv : 0.f, arg == 1 ? gv : 0.f);
        float2 d1 = make_float2(arg == 2 ? gv : 0.f, arg == 3 ? gv : 0.f);
        float* oc = o + (size_t)c * HW;
        if (p.acc) {
            const float2 e0 = *reinterpret_cast<float2*>(oc), e1 = *reinterpret_cast<float2*>(oc + p.W);
            d0.x += e0.x; d0.y += e0.y; d1.x += e1.x; d1.y += e1.y;
        }
        *reinterpret_cast<float2*>(oc) = d0;
        *reinterpret_cast<float2*>(oc + p.W) = d1;
    }
}

// ------------------------------------------------------------------ conv1x1 with a channel-blocked input (Cout <= 8)
struct C1C8P {
    int N, HW, G8, Cin, Cout;
    const unsigned short* x; long long xbs;          // 16-bit elements
    const float* w; const float* bias; float* y;
    const float* dy; float* partial; int nblk, chunk;
};
// forward: one lane = one pixel, all output channels; fmaf chain over ci in ascending order starting from the bias -- the
// arithmetic of conv1x1_fwd_kernel (pool_up.hip) on the unpacked tensor, bit for bit.
template <bool F16>
__global__ void conv1x1_c8_fwd_kernel(const C1C8P p) {
    const int px = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
    if (px >= p.HW) return;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (p.bias && i < p.Cout) ? p.bias[i] : 0.f;
    const unsigned short* xs = p.x + (size_t)n * p.xbs + (size_t)px * 8;
    for (int g = 0; g < p.G8; ++g) {
        float v[8];
        unpack8<F16>(*reinterpret_cast<const u32x4*>(xs + (size_t)g * p.HW * 8), v);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i >= p.Cout) break;
            const float* wr = p.w + (size_t)i * p.Cin + 8 * g;          // uniform: scalar loads
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[i] = fmaf(wr[e], v[e], acc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i >= p.Cout) break;
        p.y[((size_t)n * p.Cout + i) * p.HW + px] = acc[i];
    }
}
// weight / bias gradient: block (b, n) sums its pixel range; per channel group 8 x Cout products per pixel, reduced over
// the block in a fixed order -> partial[(n * nblk + b)][co][ci] and, behind all of them, [..][Cout*Cin + co] = sum dy.
template <bool F16>
__global__ void conv1x1_c8_wgrad_kernel(const C1C8P p) {
    __shared__ float red[32];
    const int b = blockIdx.x, n = blockIdx.y;
    const int lo = b * p.chunk, hi = min(p.HW, lo + p.chunk);
    const unsigned short* xs = p.x + (size_t)n * p.xbs;
    float* out = p.partial + (size_t)(n * p.nblk + b) * ((size_t)p.Cout * p.Cin + p.Cout);
    for (int co = 0; co < p.Cout; ++co) {
        const float* gs = p.dy + ((size_t)n * p.Cout + co) * p.HW;
        for (int g = 0; g < p.G8; ++g) {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            float sdy = 0.f;
            for (int px = lo + (int)threadIdx.x; px < hi; px += blockDim.x) {
                float v[8];
                unpack8<F16>(*reinterpret_cast<const u32x4*>(xs + ((size_t)g * p.HW + px) * 8), v);
                const float gv = gs[px];
                sdy += gv;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(v[e], gv, acc[e]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float s = block_sum(acc[e], red);
                if (threadIdx.x == 0) out[(size_t)co * p.Cin + 8 * g + e] = s;
            }
            if (g == 0) {
                const float s = block_sum(sdy, red);
                if (threadIdx.x == 0) out[(size_t)p.Cout * p.Cin + co] = s;
            }
        }
    }
}
// out[i] (+)= sum_k partial[k][off + i], i < cnt, rows of `row` floats: one wave per element, lane l sums rows l, l + 64, ...
// and the 64 lane sums are combined by the fixed butterfly of wave_sum -> deterministic
__global__ void c8_rows_sum_kernel(const float* __restrict__ partial, int nrows, int row, int off, int cnt, float* __restrict__ out, int accumulate) {
    const int i = blockIdx.x;
    float s = 0.f;
    for (int k = threadIdx.x; k < nrows; k += 64) s += partial[(size_t)k * row + off + i];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[i] = accumulate ? out[i] + s : s;
}

inline bool al16p(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
inline int c1_blocks(int HW) { int nb = HW / 4096; if (nb < 1) nb = 1; if (nb > 32) nb = 32; return nb; }

}  // namespace

// ---------------------------------------------------------------- internal entry points (called from pool_up.hip)
int mtbc_i_maxpool_c8_fwd(const mtbc_maxpool_args* a, hipStream_t st) {
    if (a->C % 8 || (a->type16 != 1 && a->type16 != 2)) return MTBC_E_BADARG;
    if (!al16p(a->x) || !al16p(a->y) || (a->x_batch_stride & 7) || (a->y_batch_stride & 7)) return MTBC_E_UNSUPPORTED;
    MpC8P p{a->N, a->C / 8, a->H, a->W, reinterpret_cast<const unsigned short*>(a->x), a->x_batch_stride,
            reinterpret_cast<unsigned short*>(a->y), a->y_batch_stride, nullptr, 0, nullptr, 0, 0, reinterpret_cast<unsigned short*>(a->argmax)};
    const size_t total = (size_t)a->N * p.G8 * (a->H / 2) * (a->W / 2);
    if (a->type16 == 2) hipLaunchKernelGGL(maxpool_c8_fwd_kernel<true>, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(maxpool_c8_fwd_kernel<false>, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, p);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}
int mtbc_i_maxpool_c8_bwd(const mtbc_maxpool_args* a, hipStream_t st) {
    if (a->C % 8 || (a->type16 != 1 && a->type16 != 2)) return MTBC_E_BADARG;
    if (!al16p(a->x) || (a->x_batch_stride & 7) || (reinterpret_cast<uintptr_t>(a->dx) & 7) || (a->dx_batch_stride & 1)) return MTBC_E_UNSUPPORTED;
    MpC8P p{a->N, a->C / 8, a->H, a->W, reinterpret_cast<const unsigned short*>(a->x), a->x_batch_stride, nullptr, 0,
            a->dy, a->dy_batch_stride, a->dx, a->dx_batch_stride, a->accumulate_dx, nullptr};
    const size_t total = (size_t)a->N * p.G8 * (a->H / 2) * (a->W / 2);
    if (a->type16 == 2) hipLaunchKernelGGL(maxpool_c8_bwd_kernel<true>, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(maxpool_c8_bwd_kernel<false>, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, p);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}
static int fill_c1c8(const mtbc_conv1x1_args* a, C1C8P* p) {
    if (a->Cin % 8 || a->Cout > 8 || (a->x_type != 1 && a->x_type != 2)) return MTBC_E_UNSUPPORTED;
    if (!al16p(a->x) || (a->x_batch_stride & 7)) return MTBC_E_UNSUPPORTED;
    p->N = a->N; p->HW = a->H * a->W; p->G8 = a->Cin / 8; p->Cin = a->Cin; p->Cout = a->Cout;
    p->x = reinterpret_cast<const unsigned short*>(a->x); p->xbs = a->x_batch_stride; p->w = a->w; p->bias = a->bias; p->y = a->y;
    p->dy = a->dy; p->partial = nullptr; p->nblk = c1_blocks(p->HW); p->chunk = cdiv(p->HW, p->nblk);
    return MTBC_OK;
}
int mtbc_i_conv1x1_c8_fwd(const mtbc_conv1x1_args* a, hipStream_t st) {
    C1C8P p; int rc = fill_c1c8(a, &p); if (rc) return rc;
    const dim3 grid(cdiv(p.HW, 256), a->N);
    if (a->x_type == 2) hipLaunchKernelGGL(conv1x1_c8_fwd_kernel<true>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(conv1x1_c8_fwd_kernel<false>, grid, dim3(256), 0, st, p);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}
size_t mtbc_i_conv1x1_c8_wgrad_workspace(const mtbc_conv1x1_args* a) {
    return (size_t)a->N * c1_blocks(a->H * a->W) * ((size_t)a->Cout * a->Cin + a->Cout) * sizeof(float);
}
int mtbc_i_conv1x1_c8_wgrad(const mtbc_conv1x1_args* a, hipStream_t st) {
    C1C8P p; int rc = fill_c1c8(a, &p); if (rc) return rc;
    if (!a->workspace || a->workspace_bytes < mtbc_i_conv1x1_c8_wgrad_workspace(a)) return MTBC_E_WORKSPACE;
    p.partial = reinterpret_cast<float*>(a->workspace);
    const dim3 grid(p.nblk, a->N);
    if (a->x_type == 2) hipLaunchKernelGGL(conv1x1_c8_wgrad_kernel<true>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(conv1x1_c8_wgrad_kernel<false>, grid, dim3(256), 0, st, p);
    MTBC_CHECK_LAUNCH();
    const int rows = a->N * p.nblk, row = a->Cout * a->Cin + a->Cout, wel = a->Cout * a->Cin;
    hipLaunchKernelGGL(c8_rows_sum_kernel, dim3(wel), dim3(64), 0, st, p.partial, rows, row, 0, wel, a->dw, a->accumulate_dw);
    MTBC_CHECK_LAUNCH();
    if (a->dbias) {
        hipLaunchKernelGGL(c8_rows_sum_kernel, dim3(a->Cout), dim3(64), 0, st, p.partial, rows, row, wel, a->Cout, a->dbias, a->accumulate_dw);
        MTBC_CHECK_LAUNCH();
    }
    return MTBC_OK;
}
