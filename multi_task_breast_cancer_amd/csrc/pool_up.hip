// MaxPool2d(2,2), ConvTranspose2d(kernel == stride == k) and conv1x1 for gfx950.
//
// ConvTranspose with kernel == stride is a 1x1 GEMM plus a pixel-shuffle scatter:
//   fwd  : Y[(co,a,b)][pix] = sum_ci W[ci][(co,a,b)] X[ci][pix]          M = Cout*k*k, K = Cin
//   dgrad: dX[ci][pix]      = sum_(co,a,b) W[ci][(co,a,b)] dY[(co,a,b)@pix]   M = Cin,  K = Cout*k*k
//   wgrad: dW[ci][(co,a,b)] = sum_(n,pix) X[ci][pix] dY[(co,a,b)@pix]     M = Cin, N = Cout*k*k, split-K
// all three run on one LDS-tiled 64x64x16 block GEMM built on v_mfma_f32_16x16x4_f32 (exact fp32).
// Replaces nn.ConvTranspose2d (MTnnUNet.py:96-100,106-116; MONAI UpSample "deconv"), nn.MaxPool2d(2,2)
// (MTnnUNet.py:103; MONAI Down) and the 1x1 output convs (MTnnUNet.py:106-118; MTUNetPlusPlus.py:73-76).
#include "common.h"
#include <cstdlib>

namespace {

// ------------------------------------------------------------------ generic 64x64 block GEMM
constexpr int GRS = 81;   // LDS row stride (floats): conflict-free for both fill patterns

// AL: float operator()(int m, int k) ; BL: float operator()(int k, int col) ;
// CS: void operator()(int m4, int col, f32x4 v)  -- rows m4..m4+3 (m4 % 4 == 0) of column col
template <bool A_KCONTIG, bool B_KCONTIG, class AL, class BL, class CS>
__device__ __forceinline__ void gemm_block_64x64(int m0, int c0, int kbeg, int kend, AL al, BL bl, CS cs) {
    constexpr int BK = 32;                      // K per LDS stage; the next stage is prefetched into registers
    __shared__ float As[BK * GRS];
    __shared__ float Bs[BK * GRS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int mrow = 32 * (wv >> 1), ccol = 32 * (wv & 1);
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float ra[8], rb[8];
    auto fetch = [&](int k0) {                  // 64 x BK elements of A and of B: 8 + 8 per thread
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int m, k;
            if (A_KCONTIG) { k = tid & 31; m = (tid >> 5) + 8 * i; } else { m = tid & 63; k = (tid >> 6) + 4 * i; }
            ra[i] = (k0 + k < kend) ? al(m0 + m, k0 + k) : 0.f;
            int c, k2;
            if (B_KCONTIG) { k2 = tid & 31; c = (tid >> 5) + 8 * i; } else { c = tid & 63; k2 = (tid >> 6) + 4 * i; }
            rb[i] = (k0 + k2 < kend) ? bl(k0 + k2, c0 + c) : 0.f;
        }
    };
    fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int m, k;
            if (A_KCONTIG) { k = tid & 31; m = (tid >> 5) + 8 * i; } else { m = tid & 63; k = (tid >> 6) + 4 * i; }
            As[k * GRS + m] = ra[i];
            int c, k2;
            if (B_KCONTIG) { k2 = tid & 31; c = (tid >> 5) + 8 * i; } else { c = tid & 63; k2 = (tid >> 6) + 4 * i; }
            Bs[k2 * GRS + c] = rb[i];
        }
        __syncthreads();
        if (k0 + BK < kend) fetch(k0 + BK);     // in flight under the MFMAs below
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            float a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = As[(ks * 4 + kk) * GRS + mrow + t * 16 + j];
                b[t] = Bs[(ks * 4 + kk) * GRS + ccol + t * 16 + j];
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[t], acc[s][t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t) cs(m0 + mrow + s * 16 + kk * 4, c0 + ccol + t * 16 + j, acc[s][t]);
}

// ------------------------------------------------------------------ ConvTranspose k == stride
struct CtP {
    int N, H, W, Cin, Cout, k, M;      // M = Cout*k*k ; input H x W
    const float* x; long long xbs;
    const float* w; const float* bias;
    float* y; long long ybs;
    const float* dy; long long dybs;
    float* dx; long long dxbs; int acc_dx;
    float* partial; int S, cols_per_split;
};

template <int KS>
__global__ __launch_bounds__(256) void convT_fwd_kernel(const CtP p) {
    const int n = blockIdx.z, HW = p.H * p.W, oW = p.W * KS;
    const float* xn = p.x + (size_t)n * p.xbs;
    float* yn = p.y + (size_t)n * p.ybs;
    auto al = [&](int m, int k) { return m < p.M ? p.w[(size_t)k * p.M + m] : 0.f; };
    auto bl = [&](int k, int col) { return col < HW ? xn[(size_t)k * HW + col] : 0.f; };
    auto cs = [&](int m4, int col, f32x4 v) {
        if (col >= HW || m4 >= p.M) return;
        const int i = col / p.W, jx = col % p.W;
        const int co = m4 / (KS * KS), r = m4 % (KS * KS);
        const float bv = p.bias ? p.bias[co] : 0.f;
        float* base = yn + (size_t)co * HW * KS * KS;
        if (KS == 2) {   // m4..m4+3 = (a,b) in {00,01,10,11}
            *reinterpret_cast<float2*>(base + (size_t)(2 * i) * oW + 2 * jx) = make_float2(v[0] + bv, v[1] + bv);
            *reinterpret_cast<float2*>(base + (size_t)(2 * i + 1) * oW + 2 * jx) = make_float2(v[2] + bv, v[3] + bv);
        } else {         // four consecutive b at one a
            const int a = r / KS, b0 = r % KS;
            *reinterpret_cast<float4*>(base + (size_t)(KS * i + a) * oW + KS * jx + b0) =
                make_float4(v[0] + bv, v[1] + bv, v[2] + bv, v[3] + bv);
        }
    };
    gemm_block_64x64<false, false>(blockIdx.y * 64, blockIdx.x * 64, 0, p.Cin, al, bl, cs);
}

template <int KS>
__global__ __launch_bounds__(256) void convT_dgrad_kernel(const CtP p) {
    const int n = blockIdx.z, HW = p.H * p.W, oW = p.W * KS;
    const float* gn = p.dy + (size_t)n * p.dybs;
    float* dn = p.dx + (size_t)n * p.dxbs;
    auto al = [&](int m, int k) { return m < p.Cin ? p.w[(size_t)m * p.M + k] : 0.f; };
    auto bl = [&](int k, int col) {
        if (col >= HW) return 0.f;
        const int co = k / (KS * KS), r = k % (KS * KS), a = r / KS, b = r % KS;
        const int i = col / p.W, jx = col % p.W;
        return gn[(size_t)co * HW * KS * KS + (size_t)(KS * i + a) * oW + KS * jx + b];
    };
    auto cs = [&](int m4, int col, f32x4 v) {
        if (col >= HW) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = m4 + r;
            if (ci >= p.Cin) break;
            float* d = dn + (size_t)ci * HW + col;
            *d = p.acc_dx ? *d + v[r] : v[r];
        }
    };
    gemm_block_64x64<true, true>(blockIdx.y * 64, blockIdx.x * 64, 0, p.M, al, bl, cs);
}

// wgrad: block z = split index over (n, column range); K index kk -> column col_begin + kk of image n
template <int KS>
__global__ __launch_bounds__(256) void convT_wgrad_kernel(const CtP p) {
    const int split = blockIdx.z, HW = p.H * p.W, oW = p.W * KS;
    const int n = split / p.S, c_begin = (split % p.S) * p.cols_per_split;
    const int c_end = min(HW, c_begin + p.cols_per_split);
    const float* xn = p.x + (size_t)n * p.xbs;
    const float* gn = p.dy + (size_t)n * p.dybs;
    float* part = p.partial + (size_t)split * p.Cin * p.M;
    auto al = [&](int m, int k) { return m < p.Cin ? xn[(size_t)m * HW + k] : 0.f; };
    auto bl = [&](int k, int col) {
        if (col >= p.M) return 0.f;
        const int co = col / (KS * KS), r = col % (KS * KS), a = r / KS, b = r % KS;
        const int i = k / p.W, jx = k % p.W;
        return gn[(size_t)co * HW * KS * KS + (size_t)(KS * i + a) * oW + KS * jx + b];
    };
    auto cs = [&](int m4, int col, f32x4 v) {
        if (col >= p.M) return;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (m4 + r < p.Cin) part[(size_t)(m4 + r) * p.M + col] = v[r];
    };
    gemm_block_64x64<true, false>(blockIdx.y * 64, blockIdx.x * 64, c_begin, c_end, al, bl, cs);
}

int fill_ct(const mtbc_convT_args* a, CtP* p) {
    if (!a || a->N <= 0 || a->H <= 0 || a->W <= 0 || a->Cin <= 0 || a->Cout <= 0) return MTBC_E_BADSHAPE;
    if (a->k != 2 && a->k != 4 && a->k != 8) return MTBC_E_UNSUPPORTED;
    p->N = a->N; p->H = a->H; p->W = a->W; p->Cin = a->Cin; p->Cout = a->Cout; p->k = a->k; p->M = a->Cout * a->k * a->k;
    p->x = a->x; p->xbs = a->x_batch_stride; p->w = a->w; p->bias = a->bias; p->y = a->y; p->ybs = a->y_batch_stride;
    p->dy = a->dy; p->dybs = a->dy_batch_stride; p->dx = a->dx; p->dxbs = a->dx_batch_stride; p->acc_dx = a->accumulate_dx;
    p->partial = nullptr; p->S = 1; p->cols_per_split = a->H * a->W;
    return MTBC_OK;
}
void plan_ct_wgrad(const mtbc_convT_args* a, int* S, int* cols) {
    const int HW = a->H * a->W, M = a->Cout * a->k * a->k;
    const long long blocks = (long long)cdiv(M, 64) * cdiv(a->Cin, 64) * a->N;
    int s = (int)cdiv64(1024, blocks);
    const int smax = HW / 256 > 0 ? HW / 256 : 1;
    if (s > smax) s = smax;
    if (s < 1) s = 1;
    int c = cdiv(HW, s);
    c = cdiv(c, 32) * 32;          // K stages of 32
    *cols = c; *S = cdiv(HW, c);
}

// ------------------------------------------------------------------ MaxPool 2x2
struct MpP {
    int N, C, H, W;
    const float* x; long long xbs; float* y; long long ybs;
    const float* dy; long long dybs; float* dx; long long dxbs; int acc;
};
// one thread = 2 adjacent outputs (one float4 per input row)
__global__ void maxpool_fwd_kernel(const MpP p) {
    const int oW = p.W >> 1, oH = p.H >> 1, w2 = oW >> 1;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)p.N * p.C * oH * w2;
    if (idx >= total) return;
    const int q = idx % w2; size_t t = idx / w2;
    const int oy = t % oH; t /= oH;
    const int c = t % p.C, n = t / p.C;
    const float* r0 = p.x + (size_t)n * p.xbs + ((size_t)c * p.H + 2 * oy) * p.W + 4 * q;
    const float4 a = *reinterpret_cast<const float4*>(r0), b = *reinterpret_cast<const float4*>(r0 + p.W);
    float2 o;
    o.x = fmaxf(fmaxf(a.x, a.y), fmaxf(b.x, b.y));
    o.y = fmaxf(fmaxf(a.z, a.w), fmaxf(b.z, b.w));
    *reinterpret_cast<float2*>(p.y + (size_t)n * p.ybs + ((size_t)c * oH + oy) * oW + 2 * q) = o;
}
__global__ void maxpool_fwd_scalar_kernel(const MpP p) {
    const int oW = p.W >> 1, oH = p.H >> 1;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)p.N * p.C * oH * oW;
    if (idx >= total) return;
    const int ox = idx % oW; size_t t = idx / oW;
    const int oy = t % oH; t /= oH;
    const int c = t % p.C, n = t / p.C;
    const float* r0 = p.x + (size_t)n * p.xbs + ((size_t)c * p.H + 2 * oy) * p.W + 2 * ox;
    p.y[(size_t)n * p.ybs + ((size_t)c * oH + oy) * oW + ox] = fmaxf(fmaxf(r0[0], r0[1]), fmaxf(r0[p.W], r0[p.W + 1]));
}
// bwd: one thread = one output window; grad goes to the first maximal element (ATen order)
__global__ void maxpool_bwd_kernel(const MpP p) {
    const int oW = p.W >> 1, oH = p.H >> 1;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)p.N * p.C * oH * oW;
    if (idx >= total) return;
    const int ox = idx % oW; size_t t = idx / oW;
    const int oy = t % oH; t /= oH;
    const int c = t % p.C, n = t / p.C;
    const size_t off = ((size_t)c * p.H + 2 * oy) * p.W + 2 * ox;
    const float2 a = *reinterpret_cast<const float2*>(p.x + (size_t)n * p.xbs + off);
    const float2 b = *reinterpret_cast<const float2*>(p.x + (size_t)n * p.xbs + off + p.W);
    const float g = p.dy[(size_t)n * p.dybs + ((size_t)c * oH + oy) * oW + ox];
    float m = a.x; int arg = 0;
    if (a.y > m || a.y != a.y) { m = a.y; arg = 1; }
    if (b.x > m || b.x != b.x) { m = b.x; arg = 2; }
    if (b.y > m || b.y != b.y) { m = b.y; arg = 3; }
    float2 d0 = make_float2(arg == 0 ? g : 0.f, arg == 1 ? g : 0.f);
    float2 d1 = make_float2(arg == 2 ? g : 0.f, arg == 3 ? g : 0.f);
    float* o = p.dx + (size_t)n * p.dxbs + off;
    if (p.acc) {
        const float2 e0 = *reinterpret_cast<float2*>(o), e1 = *reinterpret_cast<float2*>(o + p.W);
        d0.x += e0.x; d0.y += e0.y; d1.x += e1.x; d1.y += e1.y;
    }
    *reinterpret_cast<float2*>(o) = d0;
    *reinterpret_cast<float2*>(o + p.W) = d1;
}

// ------------------------------------------------------------------ conv1x1 (few output channels)
struct C1P {
    int N, HW, Cin, Cout;
    const float* x; long long xbs; const float* w; const float* bias; float* y;
    const float* dy; float* dx; long long dxbs; int acc_dx;
    float* partial;
};
// thread = 4 pixels x up to 8 output channels (grid.y = channel group)
__global__ void conv1x1_fwd_kernel(const C1P p) {
    const int n4 = p.HW >> 2;
    const int q = blockIdx.x * blockDim.x + threadIdx.x, cog = blockIdx.y * 8, n = blockIdx.z;
    if (q >= n4) return;
    float4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float b = (p.bias && cog + i < p.Cout) ? p.bias[cog + i] : 0.f; acc[i] = make_float4(b, b, b, b); }
    const float4* xs = reinterpret_cast<const float4*>(p.x + (size_t)n * p.xbs) + q;
    for (int ci = 0; ci < p.Cin; ++ci) {
        const float4 v = xs[(size_t)ci * n4];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (cog + i >= p.Cout) break;
            const float wv = p.w[(size_t)(cog + i) * p.Cin + ci];
            acc[i].x = fmaf(wv, v.x, acc[i].x); acc[i].y = fmaf(wv, v.y, acc[i].y);
            acc[i].z = fmaf(wv, v.z, acc[i].z); acc[i].w = fmaf(wv, v.w, acc[i].w);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (cog + i >= p.Cout) break;
        reinterpret_cast<float4*>(p.y + ((size_t)n * p.Cout + cog + i) * p.HW)[q] = acc[i];
    }
}
// dx[n][ci][p] (+)= sum_co w[co][ci] dy[n][co][p]; thread = 4 pixels, loops over ci (Cout <= 8 per pass)
__global__ void conv1x1_dgrad_kernel(const C1P p) {
    const int n4 = p.HW >> 2;
    const int q = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.z;
    if (q >= n4) return;
    float4* dst = reinterpret_cast<float4*>(p.dx + (size_t)n * p.dxbs) + q;
    for (int cog = 0; cog < p.Cout; cog += 8) {
        float4 g[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            g[i] = (cog + i < p.Cout) ? reinterpret_cast<const float4*>(p.dy + ((size_t)n * p.Cout + cog + i) * p.HW)[q]
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ci = 0; ci < p.Cin; ++ci) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (cog + i >= p.Cout) break;
                const float wv = p.w[(size_t)(cog + i) * p.Cin + ci];
                s.x = fmaf(wv, g[i].x, s.x); s.y = fmaf(wv, g[i].y, s.y); s.z = fmaf(wv, g[i].z, s.z); s.w = fmaf(wv, g[i].w, s.w);
            }
            if (p.acc_dx || cog > 0) { const float4 e = dst[(size_t)ci * n4]; s.x += e.x; s.y += e.y; s.z += e.z; s.w += e.w; }
            dst[(size_t)ci * n4] = s;
        }
    }
}
// partial[n][co][ci] = sum_p dy[n][co][p] x[n][ci][p]; block = (ci, n)
__global__ void conv1x1_wgrad_kernel(const C1P p) {
    __shared__ float red[32];
    const int ci = blockIdx.x, n = blockIdx.y, n4 = p.HW >> 2;
    const float4* xs = reinterpret_cast<const float4*>(p.x + (size_t)n * p.xbs + (size_t)ci * p.HW);
    for (int co = 0; co < p.Cout; ++co) {
        const float4* gs = reinterpret_cast<const float4*>(p.dy + ((size_t)n * p.Cout + co) * p.HW);
        float s = 0.f;
        for (int i = threadIdx.x; i < n4; i += blockDim.x) {
            const float4 a = xs[i], b = gs[i];
            s += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
        }
        s = block_sum(s, red);
        if (threadIdx.x == 0) p.partial[((size_t)n * p.Cout + co) * p.Cin + ci] = s;
    }
}

bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

extern "C" {

// ---------------------------------------------------------------- ConvTranspose
int mtbc_convT_fwd(const mtbc_convT_args* a, void* stream) {
    CtP p; int rc = fill_ct(a, &p); if (rc) return rc;
    if (!p.x || !p.w || !p.y) return MTBC_E_BADARG;
    if (a->x_layout == MTBC_LAYOUT_C8) {        // both tensors 16-bit channel-blocked: the 16-bit MFMA kernel or nothing
        if (a->y_layout != MTBC_LAYOUT_C8) return MTBC_E_BADARG;
        return mtbc_i_convT2_fwd_lp_c8_ok(a) ? mtbc_i_convT2_fwd_lp_c8(a, (hipStream_t)stream) : MTBC_E_UNSUPPORTED;
    }
    if (a->x_layout != MTBC_LAYOUT_PLANAR) return MTBC_E_BADARG;
    if (a->y_layout == MTBC_LAYOUT_C8)          // 16-bit channel-blocked output: the direct-to-fragment kernel or nothing
        return mtbc_i_convT2_fwd_c8_ok(a) ? mtbc_i_convT2_fwd_c8(a, (hipStream_t)stream) : MTBC_E_UNSUPPORTED;
    if (a->y_layout != MTBC_LAYOUT_PLANAR) return MTBC_E_BADARG;
    if (!al16(p.y) || (p.ybs & 3) || ((a->W * a->k) & 3)) return MTBC_E_UNSUPPORTED;
    static const bool generic_f = mtbc_probe_set("MTBC_CONVT_GENERIC");      // A/B switch
    if (!generic_f && mtbc_i_convT2_fwd_ok(a)) return mtbc_i_convT2_fwd(a, (hipStream_t)stream);
    dim3 grid(cdiv(a->H * a->W, 64), cdiv(p.M, 64), a->N);
    hipStream_t st = (hipStream_t)stream;
    if (a->k == 2) hipLaunchKernelGGL(convT_fwd_kernel<2>, grid, dim3(256), 0, st, p);
    else if (a->k == 4) hipLaunchKernelGGL(convT_fwd_kernel<4>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(convT_fwd_kernel<8>, grid, dim3(256), 0, st, p);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}
int mtbc_convT_fwd_c8_supported(const mtbc_convT_args* a) {
    CtP p;
    if (!a || fill_ct(a, &p)) return 0;
    if (a->x_layout == MTBC_LAYOUT_C8) return a->y_layout == MTBC_LAYOUT_C8 && mtbc_i_convT2_fwd_lp_c8_ok(a) ? 1 : 0;
    return mtbc_i_convT2_fwd_c8_ok(a) ? 1 : 0;
}
int mtbc_convT_dgrad(const mtbc_convT_args* a, void* stream) {
    CtP p; int rc = fill_ct(a, &p); if (rc) return rc;
    if (!p.dy || !p.w || !p.dx) return MTBC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    static const bool generic = mtbc_probe_set("MTBC_CONVT_GENERIC");      // A/B switch
    if (a->dy_type16 && !((a->compute == 1 || a->compute == 2) && mtbc_i_convT2_dgrad_ok(a))) return MTBC_E_UNSUPPORTED;      // 16-bit dY: the direct kernel or nothing
    if ((!generic || a->dy_type16) && mtbc_i_convT2_dgrad_ok(a)) return mtbc_i_convT2_dgrad(a, a->compute, st);
    dim3 grid(cdiv(a->H * a->W, 64), cdiv(a->Cin, 64), a->N);
    if (a->k == 2) hipLaunchKernelGGL(convT_dgrad_kernel<2>, grid, dim3(256), 0, st, p);
    else if (a->k == 4) hipLaunchKernelGGL(convT_dgrad_kernel<4>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(convT_dgrad_kernel<8>, grid, dim3(256), 0, st, p);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}
size_t mtbc_convT_wgrad_workspace(const mtbc_convT_args* a) {
    CtP p; if (fill_ct(a, &p)) return 0;
    int S, cols; plan_ct_wgrad(a, &S, &cols);
    size_t splits = (size_t)a->N * S;
    if (mtbc_i_convT2_wgrad_ok(a)) {          // pointer alignment may still send the call to the generic kernel: cover both
        int sps, ns; mtbc_i_convT2_wgrad_plan(a, &sps, &ns);
        if ((size_t)ns > splits) splits = ns;
    }
    const size_t brows = splits > (size_t)a->N ? splits : (size_t)a->N;      // per-split or per-image partial bias sums
    return (splits * a->Cin * p.M + (a->dbias ? brows * a->Cout : 0)) * sizeof(float);
}
int mtbc_convT_wgrad(const mtbc_convT_args* a, void* stream) {
    CtP p; int rc = fill_ct(a, &p); if (rc) return rc;
    if (!p.x || !p.dy || !a->dw) return MTBC_E_BADARG;
    if (!a->workspace || a->workspace_bytes < mtbc_convT_wgrad_workspace(a)) return MTBC_E_WORKSPACE;
    int S, cols; plan_ct_wgrad(a, &S, &cols);
    p.partial = reinterpret_cast<float*>(a->workspace); p.S = S; p.cols_per_split = cols;
    int nsplit = a->N * S;
    hipStream_t st = (hipStream_t)stream;
    static const bool generic = mtbc_probe_set("MTBC_CONVT_GENERIC");      // A/B switch
    bool bias_done = false;
    if ((a->dy_type16 || a->x_type16) && !((a->compute == 1 || a->compute == 2) && mtbc_i_convT2_wgrad_ok(a))) return MTBC_E_UNSUPPORTED;      // 16-bit dY / x: the direct kernel or nothing
    if ((!generic || a->dy_type16) && mtbc_i_convT2_wgrad_ok(a)) {
        int sps; mtbc_i_convT2_wgrad_plan(a, &sps, &nsplit);
        float* bpart = a->dbias ? p.partial + (size_t)nsplit * a->Cin * p.M : nullptr;
        rc = mtbc_i_convT2_wgrad(a, a->compute, p.partial, bpart, sps, nsplit, st); if (rc) return rc;
        if (bpart) {       // the wgrad waves summed dY on the way: no second pass over the gradient
            rc = mtbc_i_splitk_reduce(bpart, a->dbias, nsplit, (size_t)a->Cout, a->accumulate_dw, st); if (rc) return rc;
            bias_done = true;
        }
    } else {
        dim3 grid(cdiv(p.M, 64), cdiv(a->Cin, 64), nsplit);
        if (a->k == 2) hipLaunchKernelGGL(convT_wgrad_kernel<2>, grid, dim3(256), 0, st, p);
        else if (a->k == 4) hipLaunchKernelGGL(convT_wgrad_kernel<4>, grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(convT_wgrad_kernel<8>, grid, dim3(256), 0, st, p);
        MTBC_CHECK_LAUNCH();
    }
    const size_t wel = (size_t)a->Cin * p.M;
    rc = mtbc_i_splitk_reduce(p.partial, a->dw, nsplit, wel, a->accumulate_dw, st); if (rc) return rc;
    if (a->dbias && !bias_done) {
        if (p.dybs != (long long)a->Cout * a->H * a->W * a->k * a->k) return MTBC_E_UNSUPPORTED;
        rc = mtbc_i_channel_sums(p.dy, p.partial + (size_t)nsplit * wel, a->dbias, a->N, a->Cout, a->H * a->W * a->k * a->k,
                                 a->accumulate_dw, st);
        if (rc) return rc;
    }
    return MTBC_OK;
}

// ---------------------------------------------------------------- MaxPool
static int fill_mp(const mtbc_maxpool_args* a, MpP* p) {
    if (!a || a->N <= 0 || a->C <= 0 || a->H < 2 || a->W < 2 || (a->H & 1) || (a->W & 1)) return MTBC_E_BADSHAPE;
    p->N = a->N; p->C = a->C; p->H = a->H; p->W = a->W; p->x = a->x; p->xbs = a->x_batch_stride; p->y = a->y;
    p->ybs = a->y_batch_stride; p->dy = a->dy; p->dybs = a->dy_batch_stride; p->dx = a->dx; p->dxbs = a->dx_batch_stride;
    p->acc = a->accumulate_dx;
    return MTBC_OK;
}
int mtbc_maxpool2_fwd(const mtbc_maxpool_args* a, void* stream) {
    MpP p; int rc = fill_mp(a, &p); if (rc) return rc;
    if (!p.x || !p.y) return MTBC_E_BADARG;
    if (a->layout == MTBC_LAYOUT_C8) return mtbc_i_maxpool_c8_fwd(a, (hipStream_t)stream);
    if (a->layout != MTBC_LAYOUT_PLANAR) return MTBC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if ((a->W & 3) == 0 && al16(p.x) && (p.xbs & 3) == 0 && (reinterpret_cast<uintptr_t>(p.y) & 7) == 0 && (p.ybs & 1) == 0) {
        const size_t total = (size_t)a->N * a->C * (a->H / 2) * (a->W / 4);
        hipLaunchKernelGGL(maxpool_fwd_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, p);
    } else {
        const size_t total = (size_t)a->N * a->C * (a->H / 2) * (a->W / 2);
        hipLaunchKernelGGL(maxpool_fwd_scalar_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, p);
    }
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}
int mtbc_maxpool2_bwd(const mtbc_maxpool_args* a, void* stream) {
    MpP p; int rc = fill_mp(a, &p); if (rc) return rc;
    if (!p.x || !p.dy || !p.dx) return MTBC_E_BADARG;
    if (a->layout == MTBC_LAYOUT_C8) return mtbc_i_maxpool_c8_bwd(a, (hipStream_t)stream);
    if (a->layout != MTBC_LAYOUT_PLANAR) return MTBC_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(p.x) & 7) || (reinterpret_cast<uintptr_t>(p.dx) & 7) || (p.xbs & 1) || (p.dxbs & 1))
        return MTBC_E_UNSUPPORTED;
    const size_t total = (size_t)a->N * a->C * (a->H / 2) * (a->W / 2);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, p);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}

// ---------------------------------------------------------------- conv1x1
static int fill_c1(const mtbc_conv1x1_args* a, C1P* p) {
    if (!a || a->N <= 0 || a->H <= 0 || a->W <= 0 || a->Cin <= 0 || a->Cout <= 0) return MTBC_E_BADSHAPE;
    if ((a->H * a->W) & 3) return MTBC_E_UNSUPPORTED;
    p->N = a->N; p->HW = a->H * a->W; p->Cin = a->Cin; p->Cout = a->Cout; p->x = a->x; p->xbs = a->x_batch_stride;
    p->w = a->w; p->bias = a->bias; p->y = a->y; p->dy = a->dy; p->dx = a->dx; p->dxbs = a->dx_batch_stride;
    p->acc_dx = a->accumulate_dx; p->partial = nullptr;
    return MTBC_OK;
}
int mtbc_conv1x1_fwd(const mtbc_conv1x1_args* a, void* stream) {
    C1P p; int rc = fill_c1(a, &p); if (rc) return rc;
    if (!p.x || !p.w || !p.y) return MTBC_E_BADARG;
    if (a->x_layout == MTBC_LAYOUT_C8) return mtbc_i_conv1x1_c8_fwd(a, (hipStream_t)stream);
    if (a->x_layout != MTBC_LAYOUT_PLANAR) return MTBC_E_BADARG;
    if (!al16(p.x) || !al16(p.y) || (p.xbs & 3)) return MTBC_E_UNSUPPORTED;
    dim3 grid(cdiv(p.HW / 4, 256), cdiv(a->Cout, 8), a->N);
    hipLaunchKernelGGL(conv1x1_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}
int mtbc_conv1x1_dgrad(const mtbc_conv1x1_args* a, void* stream) {
    C1P p; int rc = fill_c1(a, &p); if (rc) return rc;
    if (!p.dy || !p.w || !p.dx) return MTBC_E_BADARG;
    if (!al16(p.dy) || !al16(p.dx) || (p.dxbs & 3)) return MTBC_E_UNSUPPORTED;
    dim3 grid(cdiv(p.HW / 4, 256), 1, a->N);
    hipLaunchKernelGGL(conv1x1_dgrad_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}
size_t mtbc_conv1x1_wgrad_workspace(const mtbc_conv1x1_args* a) {
    if (!a) return 0;
    if (a->x_layout == MTBC_LAYOUT_C8) return mtbc_i_conv1x1_c8_wgrad_workspace(a);
    return ((size_t)a->N * a->Cout * a->Cin + (size_t)a->N * a->Cout) * sizeof(float);
}
int mtbc_conv1x1_wgrad(const mtbc_conv1x1_args* a, void* stream) {
    C1P p; int rc = fill_c1(a, &p); if (rc) return rc;
    if (!p.x || !p.dy || !a->dw) return MTBC_E_BADARG;
    if (a->x_layout == MTBC_LAYOUT_C8) return mtbc_i_conv1x1_c8_wgrad(a, (hipStream_t)stream);
    if (a->x_layout != MTBC_LAYOUT_PLANAR) return MTBC_E_BADARG;
    if (!al16(p.x) || !al16(p.dy) || (p.xbs & 3)) return MTBC_E_UNSUPPORTED;
    if (!a->workspace || a->workspace_bytes < mtbc_conv1x1_wgrad_workspace(a)) return MTBC_E_WORKSPACE;
    p.partial = reinterpret_cast<float*>(a->workspace);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(conv1x1_wgrad_kernel, dim3(a->Cin, a->N), dim3(256), 0, st, p);
    MTBC_CHECK_LAUNCH();
    const size_t wel = (size_t)a->Cout * a->Cin;
    rc = mtbc_i_splitk_reduce(p.partial, a->dw, a->N, wel, a->accumulate_dw, st); if (rc) return rc;
    if (a->dbias) {
        rc = mtbc_i_channel_sums(p.dy, p.partial + (size_t)a->N * wel, a->dbias, a->N, a->Cout, p.HW, a->accumulate_dw, st);
        if (rc) return rc;
    }
    return MTBC_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- fused ConvT + 1x1 head (MTnnUNet deep supervision)
namespace {
__global__ void head_combine_kernel(const mtbc_head_fuse_args a) {
    const int kk = a.k * a.k;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = a.Cin * a.R * kk;
    if (idx < total) {
        const int ab = idx % kk, r = (idx / kk) % a.R, ci = idx / (kk * a.R);
        float s = 0.f;
        for (int co = 0; co < a.Cmid; ++co) s = fmaf(a.wT[((size_t)ci * a.Cmid + co) * kk + ab], a.w1[(size_t)r * a.Cmid + co], s);
        a.Wc[idx] = s;
    }
    if (idx < a.R) {
        float s = a.b1 ? a.b1[idx] : 0.f;
        if (a.bT) for (int co = 0; co < a.Cmid; ++co) s = fmaf(a.bT[co], a.w1[(size_t)idx * a.Cmid + co], s);
        a.bc[idx] = s;
    }
}
// block (r, co): dw1[r][co]; thread-strided part: dwT
__global__ void head_expand_dwT_kernel(const mtbc_head_fuse_args a) {
    const int kk = a.k * a.k;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = a.Cin * a.Cmid * kk;
    if (idx >= total) return;
    const int ab = idx % kk, co = (idx / kk) % a.Cmid, ci = idx / (kk * a.Cmid);
    float s = 0.f;
    for (int r = 0; r < a.R; ++r) s = fmaf(a.G[((size_t)ci * a.R + r) * kk + ab], a.w1[(size_t)r * a.Cmid + co], s);
    a.dwT[idx] = a.acc_wT ? a.dwT[idx] + s : s;
}
__global__ void head_expand_small_kernel(const mtbc_head_fuse_args a) {
    __shared__ float red[32];
    const int r = blockIdx.x / a.Cmid, co = blockIdx.x % a.Cmid, kk = a.k * a.k;
    float s = 0.f;
    for (int i = threadIdx.x; i < a.Cin * kk; i += blockDim.x) {
        const int ci = i / kk, ab = i % kk;
        s = fmaf(a.G[((size_t)ci * a.R + r) * kk + ab], a.wT[((size_t)ci * a.Cmid + co) * kk + ab], s);
    }
    s = block_sum(s, red);
    if (threadIdx.x != 0) return;
    if (a.bT) s = fmaf(a.gb[r], a.bT[co], s);
    a.dw1[(size_t)r * a.Cmid + co] = a.acc_w1 ? a.dw1[(size_t)r * a.Cmid + co] + s : s;
    if (r == 0 && a.dbT) {
        float t = 0.f;
        for (int q = 0; q < a.R; ++q) t = fmaf(a.gb[q], a.w1[(size_t)q * a.Cmid + co], t);
        a.dbT[co] = a.acc_bT ? a.dbT[co] + t : t;
    }
    if (co == 0 && a.db1) a.db1[r] = a.acc_b1 ? a.db1[r] + a.gb[r] : a.gb[r];
}
}  // namespace

extern "C" int mtbc_convT_head_combine(const mtbc_head_fuse_args* a, void* stream) {
    if (!a || !a->wT || !a->w1 || !a->Wc || !a->bc) return MTBC_E_BADARG;
    if (a->Cin <= 0 || a->Cmid <= 0 || a->R <= 0 || a->k <= 0) return MTBC_E_BADSHAPE;
    const int total = a->Cin * a->R * a->k * a->k;
    hipLaunchKernelGGL(head_combine_kernel, dim3(cdiv(total > a->R ? total : a->R, 128)), dim3(128), 0, (hipStream_t)stream, *a);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}
extern "C" int mtbc_convT_head_expand(const mtbc_head_fuse_args* a, void* stream) {
    if (!a || !a->wT || !a->w1 || !a->G || !a->gb || !a->dwT || !a->dw1) return MTBC_E_BADARG;
    if (a->Cin <= 0 || a->Cmid <= 0 || a->R <= 0 || a->k <= 0) return MTBC_E_BADSHAPE;
    const int total = a->Cin * a->Cmid * a->k * a->k;
    hipLaunchKernelGGL(head_expand_dwT_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, *a);
    MTBC_CHECK_LAUNCH();
    hipLaunchKernelGGL(head_expand_small_kernel, dim3(a->R * a->Cmid), dim3(256), 0, (hipStream_t)stream, *a);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}
