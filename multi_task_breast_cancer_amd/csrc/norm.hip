// InstanceNorm2d(eps, optional affine) + LeakyReLU, forward and backward, HBM-bound.
// One workgroup per (n, c) plane; the plane is held in registers between the statistics and the
// normalisation so HBM sees one read + one write (forward).  Per-image statistics are reduced with
// wavefront shuffles (64 lanes) then across the waves through LDS -- deterministic.
// Replaces nn.InstanceNorm2d + nn.LeakyReLU (MTnnUNet.py:35-36) and MONAI ADN "NDA" (MTUNetPlusPlus.py:20-22).
#include "common.h"
#include <cstdlib>

namespace {

struct InP {
    int N, C, HW;
    float eps, slope;
    const float* z; const float* gamma; const float* beta;
    float* y; long long ybs;
    float* mean; float* rstd;
    const float* dy; long long dybs;
    const float* dyx[4]; int nx;     // extra gradient contributions (fan-in), summed on the fly
    float* dz;
    float* part;       // bwd: [N*C][3] = {sum g, sum g*xh, sum dz}
    unsigned short* y16; unsigned short* dz16; int f16;      // 16-bit planar outputs instead of y / dz
};
typedef float in_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 in_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 in_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 in_cvt4(const float4& o, int f16) {      // 4 floats -> 4 x 16 bit, RNE
    if (f16) return make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector((in_f32x2){o.x, o.y}, in_f16x2)),
                               __builtin_bit_cast(unsigned, __builtin_convertvector((in_f32x2){o.z, o.w}, in_f16x2)));
    return make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector((in_f32x2){o.x, o.y}, in_bf16x2)),
                      __builtin_bit_cast(unsigned, __builtin_convertvector((in_f32x2){o.z, o.w}, in_bf16x2)));
}

// ---- forward, register-resident plane: HW % 4 == 0 and HW/4 <= VPT * blockDim
template <int VPT>
__global__ void in_fwd_reg_kernel(const InP p) {
    __shared__ float red[32];
    const int plane = blockIdx.x, n = plane / p.C, c = plane % p.C;
    const float4* src = reinterpret_cast<const float4*>(p.z + (size_t)plane * p.HW);
    const int n4 = p.HW >> 2;
    float4 v[VPT];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int idx = threadIdx.x + i * blockDim.x;
        v[i] = idx < n4 ? src[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = block_sum(s, red) / (float)p.HW;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int idx = threadIdx.x + i * blockDim.x;
        if (idx < n4) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float var = block_sum(q, red) / (float)p.HW;
    const float rstd = 1.0f / sqrtf(var + p.eps);
    if (threadIdx.x == 0) { p.mean[plane] = mean; p.rstd[plane] = rstd; }
    const float g = p.gamma ? p.gamma[c] : 1.f, b = p.beta ? p.beta[c] : 0.f;
    float4* dst = reinterpret_cast<float4*>(p.y + (size_t)n * p.ybs + (size_t)c * p.HW);
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int idx = threadIdx.x + i * blockDim.x;
        if (idx < n4) {
            float4 o;
            float t;
            t = (v[i].x - mean) * rstd * g + b; o.x = t > 0.f ? t : t * p.slope;
            t = (v[i].y - mean) * rstd * g + b; o.y = t > 0.f ? t : t * p.slope;
            t = (v[i].z - mean) * rstd * g + b; o.z = t > 0.f ? t : t * p.slope;
            t = (v[i].w - mean) * rstd * g + b; o.w = t > 0.f ? t : t * p.slope;
            if (p.y16) reinterpret_cast<uint2*>(p.y16 + (size_t)plane * p.HW)[idx] = in_cvt4(o, p.f16);
            else dst[idx] = o;
        }
    }
}

// ---- forward, planes too large for one workgroup's registers (512x512 and up): chunks of <= 64K elements.
//      Kernel 1: one workgroup per (plane, chunk) holds its chunk in registers and writes the chunk's exact
//      (count, mean, M2); kernel 2 combines the chunks of its plane with Chan's formula in a fixed order (every
//      workgroup of the plane redundantly -- S <= 64 triples) and normalises its chunk: 2 reads + 1 write of the
//      tensor instead of the streaming kernel's 3 + 1.
constexpr int IN_CHUNK4 = 16 * 1024;           // float4 per chunk = 16 per thread x 1024 threads
__global__ void in_fwd_chunk_stats_kernel(const InP p, float* __restrict__ part, int S) {
    __shared__ float red[32];
    const int plane = blockIdx.x / S, chunk = blockIdx.x % S;
    const int n4 = p.HW >> 2, lo = chunk * IN_CHUNK4, cnt4 = min(IN_CHUNK4, n4 - lo);
    const float4* src = reinterpret_cast<const float4*>(p.z + (size_t)plane * p.HW) + lo;
    float4 v[16];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int idx = threadIdx.x + i * 1024;
        v[i] = idx < cnt4 ? src[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float n = 4.f * (float)cnt4;
    const float mean = block_sum(s, red) / n;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int idx = threadIdx.x + i * 1024;
        if (idx < cnt4) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    }
    q = block_sum(q, red);
    if (threadIdx.x == 0) { float* o = part + 3 * (size_t)blockIdx.x; o[0] = n; o[1] = mean; o[2] = q; }
}
__global__ void in_fwd_chunk_apply_kernel(const InP p, const float* __restrict__ part, int S) {
    const int plane = blockIdx.x / S, chunk = blockIdx.x % S, n = plane / p.C, c = plane % p.C;
    // Chan et al.: combine (n, mean, M2) left to right -- same order in every workgroup of the plane
    const float* q = part + 3 * (size_t)plane * S;
    float cn = q[0], cm = q[1], cM2 = q[2];
    for (int k = 1; k < S; ++k) {
        const float nb = q[3 * k], mb = q[3 * k + 1], Mb = q[3 * k + 2];
        const float tot = cn + nb, delta = mb - cm;
        cm += delta * (nb / tot);
        cM2 += Mb + delta * delta * (cn * nb / tot);
        cn = tot;
    }
    const float mean = cm, rstd = 1.0f / sqrtf(cM2 / cn + p.eps);
    if (chunk == 0 && threadIdx.x == 0) { p.mean[plane] = mean; p.rstd[plane] = rstd; }
    const float g = p.gamma ? p.gamma[c] : 1.f, b = p.beta ? p.beta[c] : 0.f;
    const int n4 = p.HW >> 2, lo = chunk * IN_CHUNK4, cnt4 = min(IN_CHUNK4, n4 - lo);
    const float4* src = reinterpret_cast<const float4*>(p.z + (size_t)plane * p.HW) + lo;
    float4* dst = reinterpret_cast<float4*>(p.y + (size_t)n * p.ybs + (size_t)c * p.HW) + lo;
    for (int idx = threadIdx.x; idx < cnt4; idx += blockDim.x) {
        const float4 v = src[idx];
        float4 o; float t;
        t = (v.x - mean) * rstd * g + b; o.x = t > 0.f ? t : t * p.slope;
        t = (v.y - mean) * rstd * g + b; o.y = t > 0.f ? t : t * p.slope;
        t = (v.z - mean) * rstd * g + b; o.z = t > 0.f ? t : t * p.slope;
        t = (v.w - mean) * rstd * g + b; o.w = t > 0.f ? t : t * p.slope;
        dst[idx] = o;
    }
}

// ---- forward, streaming (any HW): three passes, the last two hit L2
__global__ void in_fwd_stream_kernel(const InP p) {
    __shared__ float red[32];
    const int plane = blockIdx.x, n = plane / p.C, c = plane % p.C;
    const float* src = p.z + (size_t)plane * p.HW;
    float s = 0.f;
    for (int i = threadIdx.x; i < p.HW; i += blockDim.x) s += src[i];
    const float mean = block_sum(s, red) / (float)p.HW;
    float q = 0.f;
    for (int i = threadIdx.x; i < p.HW; i += blockDim.x) { const float d = src[i] - mean; q += d * d; }
    const float var = block_sum(q, red) / (float)p.HW;
    const float rstd = 1.0f / sqrtf(var + p.eps);
    if (threadIdx.x == 0) { p.mean[plane] = mean; p.rstd[plane] = rstd; }
    const float g = p.gamma ? p.gamma[c] : 1.f, b = p.beta ? p.beta[c] : 0.f;
    float* dst = p.y + (size_t)n * p.ybs + (size_t)c * p.HW;
    for (int i = threadIdx.x; i < p.HW; i += blockDim.x) {
        const float t = (src[i] - mean) * rstd * g + b;
        dst[i] = t > 0.f ? t : t * p.slope;
    }
}

// ---- backward.  xh = (z-mean)*rstd ; yh = g*xh + b ; gy = dy * (yh > 0 ? 1 : slope)
//      dz = rstd * g * (gy - mean(gy) - xh * mean(gy*xh))
template <bool VEC>
__global__ void in_bwd_kernel(const InP p) {
    __shared__ float red[32];
    const int plane = blockIdx.x, n = plane / p.C, c = plane % p.C;
    const float* zs = p.z + (size_t)plane * p.HW;
    const size_t goff = (size_t)n * p.dybs + (size_t)c * p.HW;
    const float* gs = p.dy + goff;
    float* ds = p.dz + (size_t)plane * p.HW;
    // dy + extras, always added in the same order (both passes see bit-identical sums)
    auto g4sum = [&](int i) {
        float4 v = reinterpret_cast<const float4*>(gs)[i];
        for (int k = 0; k < p.nx; ++k) {
            const float4 e = reinterpret_cast<const float4*>(p.dyx[k] + goff)[i];
            v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
        }
        return v;
    };
    auto g1sum = [&](int i) {
        float v = gs[i];
        for (int k = 0; k < p.nx; ++k) v += (p.dyx[k] + goff)[i];
        return v;
    };
    const float mean = p.mean[plane], rstd = p.rstd[plane];
    const float g = p.gamma ? p.gamma[c] : 1.f, b = p.beta ? p.beta[c] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    if (VEC) {
        const int n4 = p.HW >> 2;
        const float4* z4 = reinterpret_cast<const float4*>(zs);
        for (int i = threadIdx.x; i < n4; i += blockDim.x) {
            const float4 zv = z4[i], gv = g4sum(i);
            float xh, gy;
            xh = (zv.x - mean) * rstd; gy = gv.x * ((xh * g + b) > 0.f ? 1.f : p.slope); s1 += gy; s2 += gy * xh;
            xh = (zv.y - mean) * rstd; gy = gv.y * ((xh * g + b) > 0.f ? 1.f : p.slope); s1 += gy; s2 += gy * xh;
            xh = (zv.z - mean) * rstd; gy = gv.z * ((xh * g + b) > 0.f ? 1.f : p.slope); s1 += gy; s2 += gy * xh;
            xh = (zv.w - mean) * rstd; gy = gv.w * ((xh * g + b) > 0.f ? 1.f : p.slope); s1 += gy; s2 += gy * xh;
        }
    } else {
        for (int i = threadIdx.x; i < p.HW; i += blockDim.x) {
            const float xh = (zs[i] - mean) * rstd;
            const float gy = g1sum(i) * ((xh * g + b) > 0.f ? 1.f : p.slope);
            s1 += gy; s2 += gy * xh;
        }
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0 && p.part) { p.part[3 * plane] = s1; p.part[3 * plane + 1] = s2; }
    const float m1 = s1 / (float)p.HW, m2 = s2 / (float)p.HW, k = rstd * g;
    float s3 = 0.f;                                    // sum of dz (gradient of a bias in front of the norm)
    if (VEC) {
        const int n4 = p.HW >> 2;
        const float4* z4 = reinterpret_cast<const float4*>(zs);
        float4* d4 = reinterpret_cast<float4*>(ds);
        for (int i = threadIdx.x; i < n4; i += blockDim.x) {
            const float4 zv = z4[i], gv = g4sum(i);
            float4 o; float xh, gy;
            xh = (zv.x - mean) * rstd; gy = gv.x * ((xh * g + b) > 0.f ? 1.f : p.slope); o.x = k * (gy - m1 - xh * m2);
            xh = (zv.y - mean) * rstd; gy = gv.y * ((xh * g + b) > 0.f ? 1.f : p.slope); o.y = k * (gy - m1 - xh * m2);
            xh = (zv.z - mean) * rstd; gy = gv.z * ((xh * g + b) > 0.f ? 1.f : p.slope); o.z = k * (gy - m1 - xh * m2);
            xh = (zv.w - mean) * rstd; gy = gv.w * ((xh * g + b) > 0.f ? 1.f : p.slope); o.w = k * (gy - m1 - xh * m2);
            d4[i] = o;
            s3 += (o.x + o.y) + (o.z + o.w);
        }
    } else {
        for (int i = threadIdx.x; i < p.HW; i += blockDim.x) {
            const float xh = (zs[i] - mean) * rstd;
            const float gy = g1sum(i) * ((xh * g + b) > 0.f ? 1.f : p.slope);
            const float o = k * (gy - m1 - xh * m2);
            ds[i] = o;
            s3 += o;
        }
    }
    if (p.part) {
        s3 = block_sum(s3, red);
        if (threadIdx.x == 0) p.part[3 * plane + 2] = s3;
    }
}

// ---- backward, register-resident plane (HW % 4 == 0, HW/4 == VPT * blockDim exactly or less): xh stays in registers
//      between the reduction pass and the output pass, and for planes <= 128x128 (VPT <= 4) so does gy -- the second
//      pass then reads nothing (3 tensor passes = the algorithmic minimum; larger planes re-read only dy: 4 instead of
//      the streaming kernel's 5).  Same per-thread summation order as in_bwd_kernel<true> -> bit-identical results.
template <int VPT, bool BOTH>
__global__ void in_bwd_reg_kernel(const InP p) {
    __shared__ float red[32];
    const int plane = blockIdx.x, n = plane / p.C, c = plane % p.C;
    const float4* z4 = reinterpret_cast<const float4*>(p.z + (size_t)plane * p.HW);
    const size_t goff = (size_t)n * p.dybs + (size_t)c * p.HW;
    const float4* g4 = reinterpret_cast<const float4*>(p.dy + goff);
    float4* d4 = reinterpret_cast<float4*>(p.dz + (size_t)plane * p.HW);
    const int n4 = p.HW >> 2;
    auto g4sum = [&](int i) {                // dy + extras, always in the same order
        float4 v = g4[i];
        for (int k = 0; k < p.nx; ++k) {
            const float4 e = reinterpret_cast<const float4*>(p.dyx[k] + goff)[i];
            v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
        }
        return v;
    };
    const float mean = p.mean[plane], rstd = p.rstd[plane];
    const float g = p.gamma ? p.gamma[c] : 1.f, b = p.beta ? p.beta[c] : 0.f;
    float4 xh[VPT], gy[BOTH ? VPT : 1];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int idx = threadIdx.x + i * blockDim.x;
        if (idx < n4) {
            const float4 zv = z4[idx], gv = g4sum(idx);
            float4 x, y;
            x.x = (zv.x - mean) * rstd; y.x = gv.x * ((x.x * g + b) > 0.f ? 1.f : p.slope); s1 += y.x; s2 += y.x * x.x;
            x.y = (zv.y - mean) * rstd; y.y = gv.y * ((x.y * g + b) > 0.f ? 1.f : p.slope); s1 += y.y; s2 += y.y * x.y;
            x.z = (zv.z - mean) * rstd; y.z = gv.z * ((x.z * g + b) > 0.f ? 1.f : p.slope); s1 += y.z; s2 += y.z * x.z;
            x.w = (zv.w - mean) * rstd; y.w = gv.w * ((x.w * g + b) > 0.f ? 1.f : p.slope); s1 += y.w; s2 += y.w * x.w;
            xh[i] = x;
            if (BOTH) gy[i] = y;
        }
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0 && p.part) { p.part[3 * plane] = s1; p.part[3 * plane + 1] = s2; }
    const float m1 = s1 / (float)p.HW, m2 = s2 / (float)p.HW, k = rstd * g;
    float s3 = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int idx = threadIdx.x + i * blockDim.x;
        if (idx < n4) {
            const float4 x = xh[i];
            float4 y;
            if (BOTH) y = gy[i];
            else {
                const float4 gv = g4sum(idx);
                y.x = gv.x * ((x.x * g + b) > 0.f ? 1.f : p.slope); y.y = gv.y * ((x.y * g + b) > 0.f ? 1.f : p.slope);
                y.z = gv.z * ((x.z * g + b) > 0.f ? 1.f : p.slope); y.w = gv.w * ((x.w * g + b) > 0.f ? 1.f : p.slope);
            }
            float4 o;
            o.x = k * (y.x - m1 - x.x * m2); o.y = k * (y.y - m1 - x.y * m2);
            o.z = k * (y.z - m1 - x.z * m2); o.w = k * (y.w - m1 - x.w * m2);
            if (p.dz16) reinterpret_cast<uint2*>(p.dz16 + (size_t)plane * p.HW)[idx] = in_cvt4(o, p.f16);
            else d4[idx] = o;
            s3 += (o.x + o.y) + (o.z + o.w);
        }
    }
    if (p.part) {
        s3 = block_sum(s3, red);
        if (threadIdx.x == 0) p.part[3 * plane + 2] = s3;
    }
}

// dgamma[c] = sum_n part[n,c,1] ; dbeta[c] = sum_n part[n,c,0] ; dbias_pre[c] = sum_n part[n,c,2]
// one wave per channel, lanes over the images (fixed butterfly -> deterministic); a serial loop over N in one thread
// per channel was pure latency: 9 us per launch, 36 launches per step
__global__ void in_dparam_kernel(const float* __restrict__ part, float* dgamma, float* dbeta, float* dbias_pre, int N,
                                 int C, int accumulate, const float* __restrict__ part3 = nullptr, int T = 0) {
    const int c = blockIdx.x, lane = threadIdx.x;
    float sb = 0.f, sg = 0.f, sz = 0.f;
    for (int n = lane; n < N; n += 64) {
        const float* q = part + 3 * ((size_t)n * C + c);
        sb += q[0]; sg += q[1]; sz += q[2];
        if (part3) {                                  // cooperative backward: every team member's share of sum dz
            const float* r = part3 + ((size_t)n * C + c) * T;
            for (int m = 0; m < T; ++m) sz += r[m];
        }
    }
    sb = wave_sum(sb); sg = wave_sum(sg); sz = wave_sum(sz);
    if (lane != 0) return;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + sg : sg;
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + sb : sb;
    if (dbias_pre) dbias_pre[c] = accumulate ? dbias_pre[c] + sz : sz;
}

// the same reduction for many cells in one launch (descriptors by value): block -> (descriptor, channel) by a scalar scan
constexpr int DPARAM_MANY = 40;
struct DparamManyP {
    const float* part[DPARAM_MANY];
    float* dgamma[DPARAM_MANY]; float* dbeta[DPARAM_MANY]; float* dbias[DPARAM_MANY];
    int N[DPARAM_MANY], C[DPARAM_MANY], T[DPARAM_MANY];
    unsigned char acc[DPARAM_MANY];
    int first_block[DPARAM_MANY + 1];
    int n;
};
__global__ void in_dparam_many_kernel(const DparamManyP q) {
    int d = 0;
    while (d + 1 < q.n && (int)blockIdx.x >= q.first_block[d + 1]) ++d;
    const int c = (int)blockIdx.x - q.first_block[d], lane = threadIdx.x;
    const int N = q.N[d], C = q.C[d], T = q.T[d];
    const float* part = q.part[d];
    const float* part3 = part + (size_t)3 * N * C;
    float sb = 0.f, sg = 0.f, sz = 0.f;
    for (int n = lane; n < N; n += 64) {
        const float* p3 = part + 3 * ((size_t)n * C + c);
        sb += p3[0]; sg += p3[1]; sz += p3[2];
        const float* r = part3 + ((size_t)n * C + c) * T;
        for (int m = 0; m < T; ++m) sz += r[m];
    }
    sb = wave_sum(sb); sg = wave_sum(sg); sz = wave_sum(sz);
    if (lane != 0) return;
    const bool acc = q.acc[d] != 0;
    if (q.dgamma[d]) q.dgamma[d][c] = acc ? q.dgamma[d][c] + sg : sg;
    if (q.dbeta[d]) q.dbeta[d][c] = acc ? q.dbeta[d][c] + sb : sb;
    if (q.dbias[d]) q.dbias[d][c] = acc ? q.dbias[d][c] + sz : sz;
}

int fill(const mtbc_instnorm_args* a, InP* p) {
    if (!a || a->N <= 0 || a->C <= 0 || a->H <= 0 || a->W <= 0) return MTBC_E_BADSHAPE;
    p->N = a->N; p->C = a->C; p->HW = a->H * a->W; p->eps = a->eps; p->slope = a->slope;
    p->z = a->z; p->gamma = a->gamma; p->beta = a->beta; p->y = a->y; p->ybs = a->y_batch_stride;
    p->mean = a->mean; p->rstd = a->rstd; p->dy = a->dy; p->dybs = a->dy_batch_stride; p->dz = a->dz; p->part = nullptr;
    p->nx = a->n_dy_extra;
    for (int k = 0; k < 4; ++k) p->dyx[k] = a->dy_extra[k];
    p->y16 = reinterpret_cast<unsigned short*>(a->y16); p->dz16 = reinterpret_cast<unsigned short*>(a->dz16);
    p->f16 = a->out16_type == 2;
    if ((a->y16 || a->dz16) && a->out16_type != 1 && a->out16_type != 2) return MTBC_E_BADARG;
    return MTBC_OK;
}
bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

extern "C" {

// bytes of workspace the forward wants for planes larger than 64K elements (0 otherwise; without it the streaming kernel runs)
size_t mtbc_instnorm_fwd_workspace(const mtbc_instnorm_args* a) {
    if (!a || a->N <= 0 || a->C <= 0 || a->H <= 0 || a->W <= 0) return 0;
    const long long n4 = (long long)a->H * a->W / 4;
    if (((long long)a->H * a->W) % 4 != 0 || n4 <= IN_CHUNK4) return 0;
    return (size_t)a->N * a->C * cdiv64(n4, IN_CHUNK4) * 3 * sizeof(float);
}

int mtbc_instnorm_lrelu_fwd(const mtbc_instnorm_args* a, void* stream) {
    if (a && a->y8) return mtbc_i_instnorm_fwd_c8(a, (hipStream_t)stream);
    InP p; int rc = fill(a, &p); if (rc) return rc;
    if (!p.z || (!p.y && !p.y16) || !p.mean || !p.rstd) return MTBC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int planes = a->N * a->C, HW = p.HW, n4 = HW / 4;
    const bool vec = HW % 4 == 0 && al16(p.z) && (p.y16 ? (reinterpret_cast<uintptr_t>(p.y16) & 7) == 0 : al16(p.y) && p.ybs % 4 == 0);
    if (p.y16 && !(vec && n4 <= 16 * 1024)) return MTBC_E_UNSUPPORTED;
    if (vec && n4 <= 16 * 1024) {
        if (n4 <= 64) hipLaunchKernelGGL(in_fwd_reg_kernel<1>, dim3(planes), dim3(64), 0, st, p);
        else if (n4 <= 4 * 64) hipLaunchKernelGGL(in_fwd_reg_kernel<4>, dim3(planes), dim3(64), 0, st, p);
        else if (n4 <= 4 * 256) hipLaunchKernelGGL(in_fwd_reg_kernel<4>, dim3(planes), dim3(256), 0, st, p);
        else if (n4 <= 16 * 256) hipLaunchKernelGGL(in_fwd_reg_kernel<16>, dim3(planes), dim3(256), 0, st, p);
        else hipLaunchKernelGGL(in_fwd_reg_kernel<16>, dim3(planes), dim3(1024), 0, st, p);
    } else if (vec && a->workspace && a->workspace_bytes >= mtbc_instnorm_fwd_workspace(a)) {
        const int S = cdiv(n4, IN_CHUNK4);
        float* part = reinterpret_cast<float*>(a->workspace);
        hipLaunchKernelGGL(in_fwd_chunk_stats_kernel, dim3(planes * S), dim3(1024), 0, st, p, part, S);
        MTBC_CHECK_LAUNCH();
        hipLaunchKernelGGL(in_fwd_chunk_apply_kernel, dim3(planes * S), dim3(1024), 0, st, p, part, S);
    } else {
        hipLaunchKernelGGL(in_fwd_stream_kernel, dim3(planes), dim3(HW >= 4096 ? 1024 : 256), 0, st, p);
    }
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}

int32_t mtbc_instnorm_bwd_team(const mtbc_instnorm_args* a) { return mtbc_i_instnorm_bwd_c8_team(a); }

int mtbc_instnorm_dparam_many(const mtbc_dparam_desc* descs, int32_t n, void* stream) {
    if (n < 0 || (n > 0 && !descs)) return MTBC_E_BADARG;
    int done = 0;
    while (done < n) {
        DparamManyP q;
        q.n = 0;
        int blocks = 0;
        while (done < n && q.n < DPARAM_MANY) {
            const mtbc_dparam_desc& d = descs[done];
            if (!d.part || d.N < 1 || d.C < 1 || d.T < 0 || (!d.dgamma && !d.dbeta && !d.dbias_pre)) return MTBC_E_BADARG;
            const int i = q.n++;
            q.part[i] = d.part; q.dgamma[i] = d.dgamma; q.dbeta[i] = d.dbeta; q.dbias[i] = d.dbias_pre;
            q.N[i] = d.N; q.C[i] = d.C; q.T[i] = d.T; q.acc[i] = d.accumulate ? 1 : 0;
            q.first_block[i] = blocks;
            blocks += d.C;
            ++done;
        }
        q.first_block[q.n] = blocks;
        hipLaunchKernelGGL(in_dparam_many_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, q);
        MTBC_CHECK_LAUNCH();
    }
    return MTBC_OK;
}

int mtbc_instnorm_lrelu_bwd(const mtbc_instnorm_args* a, void* stream) {
    InP p; int rc = fill(a, &p); if (rc) return rc;
    if (!p.z || (!p.dy && !(a->dz8 && (a->dy_rank1 || a->dy_pool))) || (!p.dz && !p.dz16 && !a->dz8) || !p.mean || !p.rstd) return MTBC_E_BADARG;
    if ((a->dy_rank1 || a->dy_pool) && !a->dz8) return MTBC_E_UNSUPPORTED;      // the rank-1 head / pooled terms: channel-group kernels only
    if (p.nx < 0 || p.nx > 4) return MTBC_E_BADARG;
    for (int k = 0; k < p.nx; ++k) if (!p.dyx[k] || !al16(p.dyx[k])) return MTBC_E_BADARG;
    const bool want = a->dgamma || a->dbeta || a->dbias_pre;
    const int planes = a->N * a->C;
    if (want) {
        if (!a->workspace || a->workspace_bytes < (size_t)planes * 3 * sizeof(float)) return MTBC_E_WORKSPACE;
        p.part = reinterpret_cast<float*>(a->workspace);
    }
    hipStream_t st = (hipStream_t)stream;
    if (a->dz8) {
        const bool epi = a->stats_partial != nullptr;      // reductions from the gathered dgrad's epilogue: no team, no per-member partials
        const int T = epi ? 2 : mtbc_i_instnorm_bwd_c8_team(a);
        if (T < 1) return MTBC_E_UNSUPPORTED;
        if ((want || epi) && (!a->workspace || a->workspace_bytes < (size_t)planes * (3 + T) * sizeof(float))) return MTBC_E_WORKSPACE;
        if (epi && !want) p.part = reinterpret_cast<float*>(a->workspace);
        rc = mtbc_i_instnorm_bwd_c8(a, want ? p.part : nullptr, st); if (rc) return rc;
        if (a->defer_dparams && (epi || !want)) return MTBC_E_UNSUPPORTED;
        if (want && !a->defer_dparams) {
            hipLaunchKernelGGL(in_dparam_kernel, dim3(a->C), dim3(64), 0, st, p.part, a->dgamma, a->dbeta,
                               a->dbias_pre, a->N, a->C, a->accumulate_dparams, epi ? nullptr : p.part + (size_t)3 * planes, epi ? 0 : T);
            MTBC_CHECK_LAUNCH();
        }
        return MTBC_OK;
    }
    if (a->defer_dparams) return MTBC_E_UNSUPPORTED;
    const bool vec = p.HW % 4 == 0 && al16(p.z) && al16(p.dy) && p.dybs % 4 == 0 &&
                     (p.dz16 ? (reinterpret_cast<uintptr_t>(p.dz16) & 7) == 0 : al16(p.dz));
    const int threads = p.HW >= 16384 ? 1024 : (p.HW >= 1024 ? 256 : 64);
    static const bool stream_only = mtbc_probe_set("MTBC_IN_BWD_STREAM");      // A/B switch
    const int vpt = vec ? cdiv(p.HW / 4, threads) : 0;
    if (p.dz16 && !(vec && !stream_only && vpt <= 16)) return MTBC_E_UNSUPPORTED;
    if (vec && !stream_only && vpt <= 16) {
        const dim3 g(planes), b(threads);
        if (vpt <= 1) hipLaunchKernelGGL((in_bwd_reg_kernel<1, true>), g, b, 0, st, p);
        else if (vpt <= 2) hipLaunchKernelGGL((in_bwd_reg_kernel<2, true>), g, b, 0, st, p);
        else if (vpt <= 4) hipLaunchKernelGGL((in_bwd_reg_kernel<4, true>), g, b, 0, st, p);
        else if (vpt <= 8) hipLaunchKernelGGL((in_bwd_reg_kernel<8, true>), g, b, 0, st, p);
        else hipLaunchKernelGGL((in_bwd_reg_kernel<16, false>), g, b, 0, st, p);
    } else if (vec) hipLaunchKernelGGL(in_bwd_kernel<true>, dim3(planes), dim3(threads), 0, st, p);
    else hipLaunchKernelGGL(in_bwd_kernel<false>, dim3(planes), dim3(threads), 0, st, p);
    MTBC_CHECK_LAUNCH();
    if (want) {
        hipLaunchKernelGGL(in_dparam_kernel, dim3(a->C), dim3(64), 0, st, p.part, a->dgamma, a->dbeta,
                           a->dbias_pre, a->N, a->C, a->accumulate_dparams);
        MTBC_CHECK_LAUNCH();
    }
    return MTBC_OK;
}

}  // extern "C"
