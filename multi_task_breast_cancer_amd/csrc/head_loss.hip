// Pooled classification head (GAP + Linear), fused Dice / Focal losses, loss mix + NaN flag,
// fused Adam and the train-loop Dice counters.  All HBM-/latency-bound; reductions use wavefront
// shuffles (64 lanes) then LDS across waves -- deterministic, no float atomics.
#include "common.h"

namespace {

// ------------------------------------------------------------------ GAP (one wave per plane)
__global__ void gap_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int planes, int HW) {
    const int plane = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (plane >= planes) return;
    const float* src = x + (size_t)plane * HW;
    float s = 0.f;
    for (int i = lane; i < HW; i += 64) s += src[i];
    s = wave_sum(s);
    if (lane == 0) y[plane] = s / (float)HW;
}
__global__ void gap_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, size_t total, int HW) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) dx[i] = dy[i / HW] / (float)HW;
}

// ------------------------------------------------------------------ Linear (one wave per output)
__global__ void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                  float* __restrict__ y, int N, int In, int Out, int relu) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= N * Out) return;
    const int n = o / Out, oc = o % Out;
    float s = 0.f;
    for (int i = lane; i < In; i += 64) s = fmaf(x[(size_t)n * In + i], w[(size_t)oc * In + i], s);
    s = wave_sum(s);
    if (lane == 0) {
        s += b ? b[oc] : 0.f;
        y[o] = (relu && s < 0.f) ? 0.f : s;
    }
}
// g = dy * (relu ? y > 0 : 1)
__global__ void linear_mask_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ g, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) g[i] = y[i] > 0.f ? dy[i] : 0.f;
}
__global__ void linear_dx_kernel(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ dx, int N,
                                 int In, int Out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * In) return;
    const int n = idx / In, i = idx % In;
    float s = 0.f;
    int o = 0;
    for (; o + 8 <= Out; o += 8) {           // sixteen loads in flight, the fma chain in the plain loop's order (one round trip per o otherwise)
        float gv[8], wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { gv[u] = g[(size_t)n * Out + o + u]; wv[u] = w[(size_t)(o + u) * In + i]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) s = fmaf(gv[u], wv[u], s);
    }
    for (; o < Out; ++o) s = fmaf(g[(size_t)n * Out + o], w[(size_t)o * In + i], s);
    dx[idx] = s;
}
__global__ void linear_dw_kernel(const float* __restrict__ g, const float* __restrict__ x, float* __restrict__ dw,
                                 float* __restrict__ db, int N, int In, int Out, int accumulate) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < Out * In) {
        const int o = idx / In, i = idx % In;
        float s = 0.f;
        int n = 0;
        for (; n + 8 <= N; n += 8) {
            float gv[8], xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { gv[u] = g[(size_t)(n + u) * Out + o]; xv[u] = x[(size_t)(n + u) * In + i]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) s = fmaf(gv[u], xv[u], s);
        }
        for (; n < N; ++n) s = fmaf(g[(size_t)n * Out + o], x[(size_t)n * In + i], s);
        dw[idx] = accumulate ? dw[idx] + s : s;
    }
    if (db && idx < Out) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += g[(size_t)n * Out + idx];
        db[idx] = accumulate ? db[idx] + s : s;
    }
}

// ------------------------------------------------------------------ Dice
struct DiceP {
    int n_heads, planes, HW;
    float nr, dr;
    const float* x[4]; const float* target;
    float hw[4];
    float* stats; float* loss; float* dx[4];
    float gscale; const float* gscale_dev;
};
__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// block = (plane, head): I = sum p t, P2 = sum p^2, T2 = sum t^2
__global__ void dice_stats_kernel(const DiceP p) {
    __shared__ float red[32];
    const int plane = blockIdx.x, h = blockIdx.y;
    const float* xs = p.x[h] + (size_t)plane * p.HW;
    const float* ts = p.target + (size_t)plane * p.HW;
    float si = 0.f, sp = 0.f, stt = 0.f;
    if ((p.HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(xs) | reinterpret_cast<uintptr_t>(ts)) & 15) == 0) {
        // 16-byte loads, four rounds in flight per thread (the scalar loop below was one dependent 4-byte round trip per element pair:
        // 40 us for the four 256 x 256 heads of a step -- 128 blocks pulling 512 KB each)
        const float4* x4 = reinterpret_cast<const float4*>(xs);
        const float4* t4 = reinterpret_cast<const float4*>(ts);
        const int n4 = p.HW >> 2, B = blockDim.x;
        int i = threadIdx.x;
        for (; i + 3 * B < n4; i += 4 * B) {
            float4 xv[4], tv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { xv[u] = x4[i + u * B]; tv[u] = t4[i + u * B]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float px = sigmoidf_(xv[u].x), py = sigmoidf_(xv[u].y), pz = sigmoidf_(xv[u].z), pw = sigmoidf_(xv[u].w);
                si = fmaf(px, tv[u].x, si); si = fmaf(py, tv[u].y, si); si = fmaf(pz, tv[u].z, si); si = fmaf(pw, tv[u].w, si);
                sp = fmaf(px, px, sp); sp = fmaf(py, py, sp); sp = fmaf(pz, pz, sp); sp = fmaf(pw, pw, sp);
                stt = fmaf(tv[u].x, tv[u].x, stt); stt = fmaf(tv[u].y, tv[u].y, stt); stt = fmaf(tv[u].z, tv[u].z, stt); stt = fmaf(tv[u].w, tv[u].w, stt);
            }
        }
        for (; i < n4; i += B) {
            const float4 xv = x4[i], tv = t4[i];
            const float px = sigmoidf_(xv.x), py = sigmoidf_(xv.y), pz = sigmoidf_(xv.z), pw = sigmoidf_(xv.w);
            si = fmaf(px, tv.x, si); si = fmaf(py, tv.y, si); si = fmaf(pz, tv.z, si); si = fmaf(pw, tv.w, si);
            sp = fmaf(px, px, sp); sp = fmaf(py, py, sp); sp = fmaf(pz, pz, sp); sp = fmaf(pw, pw, sp);
            stt = fmaf(tv.x, tv.x, stt); stt = fmaf(tv.y, tv.y, stt); stt = fmaf(tv.z, tv.z, stt); stt = fmaf(tv.w, tv.w, stt);
        }
    } else {
        for (int i = threadIdx.x; i < p.HW; i += blockDim.x) {
            const float pr = sigmoidf_(xs[i]), t = ts[i];
            si = fmaf(pr, t, si); sp = fmaf(pr, pr, sp); stt = fmaf(t, t, stt);
        }
    }
    si = block_sum(si, red); sp = block_sum(sp, red); stt = block_sum(stt, red);
    if (threadIdx.x == 0) {
        float* s = p.stats + ((size_t)h * p.planes + plane) * 3;
        s[0] = si; s[1] = sp; s[2] = stt;
    }
}
// one block: loss[h] = mean_plane(1 - (2I+nr)/(P2+T2+dr)); loss[n_heads] = sum_h hw[h] loss[h]
__global__ void dice_finalize_kernel(const DiceP p) {
    __shared__ float red[32];
    float total = 0.f;
    for (int h = 0; h < p.n_heads; ++h) {
        float s = 0.f;
        for (int i = threadIdx.x; i < p.planes; i += blockDim.x) {
            const float* st = p.stats + ((size_t)h * p.planes + i) * 3;
            s += 1.0f - (2.0f * st[0] + p.nr) / (st[1] + st[2] + p.dr);
        }
        s = block_sum(s, red) / (float)p.planes;
        if (threadIdx.x == 0) p.loss[h] = s;
        total += p.hw[h] * s;
    }
    if (threadIdx.x == 0) p.loss[p.n_heads] = total;
}
// dx = scale * d f / d p * p (1 - p),  d f / d p = -(2 t (D+dr) - (2I+nr) 2 p) / (D+dr)^2
__global__ void dice_bwd_kernel(const DiceP p) {
    const int h = blockIdx.y;
    const size_t total = (size_t)p.planes * p.HW;
    const float scale = p.gscale * (p.gscale_dev ? *p.gscale_dev : 1.f) * p.hw[h] / (float)p.planes;
    if ((p.HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(p.x[h]) | reinterpret_cast<uintptr_t>(p.target) | reinterpret_cast<uintptr_t>(p.dx[h])) & 15) == 0) {
        // four pixels of one plane per thread: 16-byte loads / stores, the plane's constants once per four elements (same formula per element)
        const float4* x4 = reinterpret_cast<const float4*>(p.x[h]);
        const float4* t4 = reinterpret_cast<const float4*>(p.target);
        float4* d4 = reinterpret_cast<float4*>(p.dx[h]);
        const int hw4 = p.HW >> 2;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (total >> 2); i += (size_t)gridDim.x * blockDim.x) {
            const int plane = (int)(i / hw4);
            const float* st = p.stats + ((size_t)h * p.planes + plane) * 3;
            const float den = st[1] + st[2] + p.dr, num = 2.0f * st[0] + p.nr;
            const float4 xv = x4[i], tv = t4[i];
            const float xe[4] = {xv.x, xv.y, xv.z, xv.w}, te[4] = {tv.x, tv.y, tv.z, tv.w};
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pr = sigmoidf_(xe[e]);
                const float dfdp = -(2.0f * te[e] * den - num * 2.0f * pr) / (den * den);
                o[e] = scale * dfdp * pr * (1.0f - pr);
            }
            d4[i] = make_float4(o[0], o[1], o[2], o[3]);
        }
        return;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int plane = i / p.HW;
        const float* st = p.stats + ((size_t)h * p.planes + plane) * 3;
        const float den = st[1] + st[2] + p.dr, num = 2.0f * st[0] + p.nr;
        const float pr = sigmoidf_(p.x[h][i]), t = p.target[i];
        const float dfdp = -(2.0f * t * den - num * 2.0f * pr) / (den * den);
        p.dx[h][i] = scale * dfdp * pr * (1.0f - pr);
    }
}

// ------------------------------------------------------------------ Focal (one block)
struct FocalP {
    int N, C; float alpha, gamma;
    const float* x; const float* t; const float* w; float* loss; float* dx; float gscale; const float* gscale_dev;
};
__global__ void focal_kernel(const FocalP p) {
    __shared__ float red[32];
    float acc = 0.f;
    const float gs = p.gscale * (p.gscale_dev ? *p.gscale_dev : 1.f) / (float)p.N;
    for (int n = threadIdx.x; n < p.N; n += blockDim.x) {
        const float* xs = p.x + (size_t)n * p.C;
        const float* ts = p.t + (size_t)n * p.C;
        if (p.C == 1) {
            // ONE logit = the reference's binary head (n_classes == 2, MTUNetPlusPlus.py:39-41) with torch.nn.BCEWithLogitsLoss
            // (experiment_init.py:241-242): ce = (1 - t) x + softplus(-x), the stable form torch evaluates; alpha = 1, gamma = 0 give
            // exactly its mean; d ce / d x = sigmoid(x) - t.  (`weight`, if given, multiplies the sample's term by w[0].)
            const float x = xs[0], t = ts[0], wc = p.w ? p.w[0] : 1.f;
            const float ce = wc * ((1.0f - t) * x + (fmaxf(-x, 0.f) + log1pf(expf(-fabsf(x)))));
            const float pt = expf(-ce), om = 1.0f - pt;
            const float mod = p.gamma == 0.f ? 1.0f : powf(om, p.gamma);
            acc += p.alpha * mod * ce;
            if (p.dx) {
                const float dmod = p.gamma == 0.f ? 0.f : ((om > 0.f || p.gamma >= 1.f) ? p.gamma * powf(om, p.gamma - 1.0f) : 0.f);
                const float dfdce = p.alpha * (dmod * pt * ce + mod);
                p.dx[n] = gs * dfdce * wc * (1.0f / (1.0f + expf(-x)) - t);
            }
            continue;
        }
        float m = xs[0];
        for (int c = 1; c < p.C; ++c) m = fmaxf(m, xs[c]);
        float se = 0.f;
        for (int c = 0; c < p.C; ++c) se += expf(xs[c] - m);
        const float lse = m + logf(se);
        float ce = 0.f, wt = 0.f;
        for (int c = 0; c < p.C; ++c) {
            const float wc = p.w ? p.w[c] : 1.f;
            ce -= wc * ts[c] * (xs[c] - lse);
            wt += wc * ts[c];
        }
        const float pt = expf(-ce), om = 1.0f - pt;
        const float mod = powf(om, p.gamma);
        acc += p.alpha * mod * ce;
        if (p.dx) {
            // d/dce [alpha (1-pt)^g ce] = alpha ( g (1-pt)^(g-1) pt ce + (1-pt)^g )
            const float dmod = (om > 0.f || p.gamma >= 1.f) ? p.gamma * powf(om, p.gamma - 1.0f) : 0.f;
            const float dfdce = p.alpha * (dmod * pt * ce + mod);
            for (int c = 0; c < p.C; ++c) {
                const float wc = p.w ? p.w[c] : 1.f;
                const float sm = expf(xs[c] - lse);
                p.dx[(size_t)n * p.C + c] = gs * dfdce * (sm * wt - wc * ts[c]);
            }
        }
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) p.loss[0] = acc / (float)p.N;
}

__global__ void loss_mix_kernel(const float* seg, const float* cls, float alpha, float* out4) {
    const float s = *seg, c = *cls;
    out4[0] = alpha * s + (1.0f - alpha) * c;
    out4[1] = s; out4[2] = c;
    out4[3] = (s != s || c != c) ? 1.f : 0.f;
}

// ------------------------------------------------------------------ Adam
struct AdamP { long long n; float* p; float* g; float* m; float* v; float gs, b1, b2, eps, step_size, inv_bc2_sqrt; int zero; const float* dyn; };
__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamP& a) {
    g *= a.gs;
    m = m + (g - m) * (1.0f - a.b1);                     // lerp, as torch
    v = v * a.b2 + (1.0f - a.b2) * g * g;
    const float denom = sqrtf(v) * a.inv_bc2_sqrt + a.eps;
    p = p - a.step_size * (m / denom);
}
__global__ void adam_kernel(AdamP a) {
    if (a.dyn) { a.gs = a.dyn[0]; a.step_size = a.dyn[1]; a.inv_bc2_sqrt = a.dyn[2]; }      // the per-step scalars from memory (graph replay), uniform loads
    const long long n4 = a.n >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 p = reinterpret_cast<float4*>(a.p)[i], g = reinterpret_cast<float4*>(a.g)[i];
        float4 m = reinterpret_cast<float4*>(a.m)[i], v = reinterpret_cast<float4*>(a.v)[i];
        adam1(p.x, g.x, m.x, v.x, a); adam1(p.y, g.y, m.y, v.y, a); adam1(p.z, g.z, m.z, v.z, a); adam1(p.w, g.w, m.w, v.w, a);
        reinterpret_cast<float4*>(a.p)[i] = p; reinterpret_cast<float4*>(a.m)[i] = m; reinterpret_cast<float4*>(a.v)[i] = v;
        if (a.zero) reinterpret_cast<float4*>(a.g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        adam1(a.p[i], a.g[i], a.m[i], a.v[i], a);
        if (a.zero) a.g[i] = 0.f;
    }
}

// ------------------------------------------------------------------ Dice metric counters (integer, exact)
__global__ void dice_counts_kernel(const float* __restrict__ x, const float* __restrict__ t, long long n,
                                   unsigned long long* cnt) {
    unsigned int tp = 0, fp = 0, fn = 0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const bool s = sigmoidf_(x[i]) > 0.5f, g = t[i] != 0.f;
        tp += (s && g); fp += (s && !g); fn += (!s && g);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { tp += __shfl_xor(tp, o, 64); fp += __shfl_xor(fp, o, 64); fn += __shfl_xor(fn, o, 64); }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&cnt[0], (unsigned long long)tp); atomicAdd(&cnt[1], (unsigned long long)fp); atomicAdd(&cnt[2], (unsigned long long)fn);
    }
}
__global__ void counts_to_double_kernel(double* out3) {
    if (threadIdx.x < 3) {
        const unsigned long long v = reinterpret_cast<unsigned long long*>(out3)[threadIdx.x];
        out3[threadIdx.x] = (double)v;
    }
}

}  // namespace

extern "C" {

int mtbc_gap_fwd(const mtbc_gap_args* a, void* stream) {
    if (!a || a->N <= 0 || a->C <= 0 || a->H <= 0 || a->W <= 0) return MTBC_E_BADSHAPE;
    if (!a->x || !a->y) return MTBC_E_BADARG;
    const int planes = a->N * a->C;
    hipLaunchKernelGGL(gap_fwd_kernel, dim3(cdiv(planes, 4)), dim3(256), 0, (hipStream_t)stream, a->x, a->y, planes, a->H * a->W);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}
int mtbc_gap_bwd(const mtbc_gap_args* a, void* stream) {
    if (!a || a->N <= 0 || a->C <= 0 || a->H <= 0 || a->W <= 0) return MTBC_E_BADSHAPE;
    if (!a->dy || !a->dx) return MTBC_E_BADARG;
    const size_t total = (size_t)a->N * a->C * a->H * a->W;
    hipLaunchKernelGGL(gap_bwd_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, a->dy, a->dx, total, a->H * a->W);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}

int mtbc_linear_fwd(const mtbc_linear_args* a, void* stream) {
    if (!a || a->N <= 0 || a->In <= 0 || a->Out <= 0) return MTBC_E_BADSHAPE;
    if (!a->x || !a->w || !a->y) return MTBC_E_BADARG;
    hipLaunchKernelGGL(linear_fwd_kernel, dim3(cdiv(a->N * a->Out, 4)), dim3(256), 0, (hipStream_t)stream, a->x, a->w, a->bias,
                       a->y, a->N, a->In, a->Out, a->relu);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}
int mtbc_linear_bwd(const mtbc_linear_args* a, void* stream) {
    if (!a || a->N <= 0 || a->In <= 0 || a->Out <= 0) return MTBC_E_BADSHAPE;
    if (!a->x || !a->w || !a->dy || !a->dw) return MTBC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const float* g = a->dy;
    if (a->relu) {
        if (!a->y) return MTBC_E_BADARG;
        if (!a->workspace || a->workspace_bytes < (size_t)a->N * a->Out * sizeof(float)) return MTBC_E_WORKSPACE;
        float* gm = reinterpret_cast<float*>(a->workspace);
        hipLaunchKernelGGL(linear_mask_kernel, dim3(cdiv(a->N * a->Out, 256)), dim3(256), 0, st, a->dy, a->y, gm, a->N * a->Out);
        MTBC_CHECK_LAUNCH();
        g = gm;
    }
    if (a->dx) {
        hipLaunchKernelGGL(linear_dx_kernel, dim3(cdiv(a->N * a->In, 256)), dim3(256), 0, st, g, a->w, a->dx, a->N, a->In, a->Out);
        MTBC_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(linear_dw_kernel, dim3(cdiv(a->Out * a->In, 256)), dim3(256), 0, st, g, a->x, a->dw, a->dbias, a->N, a->In,
                       a->Out, a->accumulate_dw);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}

static int fill_dice(const mtbc_dice_args* a, DiceP* p) {
    if (!a || a->n_heads < 1 || a->n_heads > 4 || a->N <= 0 || a->C <= 0 || a->H <= 0 || a->W <= 0) return MTBC_E_BADSHAPE;
    if (!a->target || !a->stats) return MTBC_E_BADARG;
    p->n_heads = a->n_heads; p->planes = a->N * a->C; p->HW = a->H * a->W; p->nr = a->smooth_nr; p->dr = a->smooth_dr;
    p->target = a->target; p->stats = a->stats; p->loss = a->loss; p->gscale = a->gscale; p->gscale_dev = a->gscale_dev;
    for (int h = 0; h < 4; ++h) { p->x[h] = a->x[h]; p->dx[h] = a->dx[h]; p->hw[h] = a->head_weight[h]; }
    for (int h = 0; h < a->n_heads; ++h) if (!a->x[h]) return MTBC_E_BADARG;
    return MTBC_OK;
}
int mtbc_dice_fwd(const mtbc_dice_args* a, void* stream) {
    DiceP p; int rc = fill_dice(a, &p); if (rc) return rc;
    if (!p.loss) return MTBC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dice_stats_kernel, dim3(p.planes, p.n_heads), dim3(p.HW >= 16384 ? 1024 : 256), 0, st, p);
    MTBC_CHECK_LAUNCH();
    hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(256), 0, st, p);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}
int mtbc_dice_bwd(const mtbc_dice_args* a, void* stream) {
    DiceP p; int rc = fill_dice(a, &p); if (rc) return rc;
    for (int h = 0; h < a->n_heads; ++h) if (!a->dx[h]) return MTBC_E_BADARG;
    const size_t total = (size_t)p.planes * p.HW;
    size_t blocks = cdiv64(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dice_bwd_kernel, dim3((unsigned)blocks, p.n_heads), dim3(256), 0, (hipStream_t)stream, p);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}

int mtbc_focal_fwd_bwd(const mtbc_focal_args* a, void* stream) {
    if (!a || a->N <= 0 || a->C <= 0) return MTBC_E_BADSHAPE;
    if (!a->x || !a->target || !a->loss) return MTBC_E_BADARG;
    FocalP p{a->N, a->C, a->alpha, a->gamma, a->x, a->target, a->weight, a->loss, a->dx, a->gscale, a->gscale_dev};
    hipLaunchKernelGGL(focal_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, p);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}

int mtbc_loss_mix(const float* seg, const float* cls, float alpha, float* out4, void* stream) {
    if (!seg || !cls || !out4) return MTBC_E_BADARG;
    hipLaunchKernelGGL(loss_mix_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, seg, cls, alpha, out4);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}

// bias corrections in double on the host, exactly as torch.optim.Adam's scalar path
static void adam_scalars(const mtbc_adam_args* a, float out3[3]) {
    const double bc1 = 1.0 - pow((double)a->beta1, (double)a->step);
    const double bc2 = 1.0 - pow((double)a->beta2, (double)a->step);
    out3[0] = a->grad_scale;
    out3[1] = (float)((double)a->lr / bc1);
    out3[2] = (float)(1.0 / sqrt(bc2));
}
int mtbc_adam_dynamic(const mtbc_adam_args* a, float out3[3]) {
    if (!a || !out3 || a->step < 1) return MTBC_E_BADARG;
    adam_scalars(a, out3);
    return MTBC_OK;
}
int mtbc_adam_step(const mtbc_adam_args* a, void* stream) {
    if (!a || a->n <= 0 || a->step < 1) return MTBC_E_BADSHAPE;
    if (!a->p || !a->g || !a->m || !a->v) return MTBC_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(a->p) | reinterpret_cast<uintptr_t>(a->g) | reinterpret_cast<uintptr_t>(a->m) |
         reinterpret_cast<uintptr_t>(a->v)) & 15)
        return MTBC_E_UNSUPPORTED;
    if (a->dynamic && (reinterpret_cast<uintptr_t>(a->dynamic) & 3)) return MTBC_E_BADARG;
    float dyn[3];
    adam_scalars(a, dyn);
    AdamP p;
    p.n = a->n; p.p = a->p; p.g = const_cast<float*>(a->g); p.m = a->m; p.v = a->v; p.gs = dyn[0];
    p.b1 = a->beta1; p.b2 = a->beta2; p.eps = a->eps;
    p.step_size = dyn[1];
    p.inv_bc2_sqrt = dyn[2];
    p.zero = a->zero_grad;
    p.dyn = a->dynamic;
    long long blocks = cdiv64(a->n / 4 + 1, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}

int mtbc_dice_counts(const float* logits, const float* target, int64_t n, double* out3, void* stream) {
    if (!logits || !target || !out3 || n <= 0) return MTBC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out3, 0, 3 * sizeof(double), st) != hipSuccess) return MTBC_E_LAUNCH;
    long long blocks = cdiv64(n, 256 * 8);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(dice_counts_kernel, dim3((unsigned)blocks), dim3(256), 0, st, logits, target, (long long)n,
                       reinterpret_cast<unsigned long long*>(out3));
    MTBC_CHECK_LAUNCH();
    hipLaunchKernelGGL(counts_to_double_kernel, dim3(1), dim3(64), 0, st, out3);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}

}  // extern "C"
