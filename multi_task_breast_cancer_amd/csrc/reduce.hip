// Small deterministic reductions shared by the wgrad paths (split-K partial sums, per-channel bias grads).
#include "common.h"

namespace {
// out[i] (+)= sum_k partial[k][i].  Block = 64 consecutive elements x G split lanes: lane group g sums splits
// g, g+G, ... (coalesced 256-B rows), the G partial sums are combined by a fixed pairwise tree -> deterministic.
// G = 16 for the many-split partials of the wide maps (a level-0 wgrad has 512 splits of only ~5k elements: with 4
// lanes each thread walked 128 dependent rows and the launch took 18 us for 10 MB).
// E = elements per block: 64, or 16 for the SMALL many-split partials (a 24 -> 24 level-0 wgrad: 1024 splits of 5184 elements = 81 blocks of
// 64 elements on 256 CUs, each pulling 260 KB through one CU; with 16 elements x 64 split lanes it is 324 blocks and 16 rows per thread).
// Two destinations (round 3): elements [0, elems1) of a row go to out, [elems1, elems) to out2 -- a weight gradient's split-K partials
// and its bias-gradient partials are ONE row per split, reduced by one launch (the separate bias launch took 4.8 us for a few
// hundred floats, 35 times per step).
template <int G, int E = 64>
__global__ void splitk_reduce_k(const float* __restrict__ partial, float* __restrict__ out, int nsplit, size_t elems,
                                int accumulate, float* __restrict__ out2, size_t elems1) {
    __shared__ float red[G][E];
    const int e = threadIdx.x % E, g = threadIdx.x / E;
    const size_t i = (size_t)blockIdx.x * E + e;
    float s0 = 0.f, s1 = 0.f;               // two independent chains per thread, combined in a fixed order
    if (i < elems) {
        int k = g;
        // eight rows in flight per round (the two-row loop below waited for its loads every round: a memory round trip per two rows);
        // the adds keep the order of that loop -- chain 0 takes rows g, g + 2G, ..., chain 1 rows g + G, g + 3G, ... -- bit-identical sums
        for (; k + 7 * G < nsplit; k += 8 * G) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(k + u * G) * elems + i];
#pragma unroll
            for (int u = 0; u < 8; u += 2) { s0 += v[u]; s1 += v[u + 1]; }
        }
        for (; k + G < nsplit; k += 2 * G) { s0 += partial[(size_t)k * elems + i]; s1 += partial[(size_t)(k + G) * elems + i]; }
        if (k < nsplit) s0 += partial[(size_t)k * elems + i];
    }
    red[g][e] = s0 + s1;
    __syncthreads();
#pragma unroll
    for (int w = G / 2; w >= 1; w >>= 1) {
        if (g < w) red[g][e] += red[g + w][e];
        __syncthreads();
    }
    if (g == 0 && i < elems) {
        float* dst = i < elems1 ? out + i : out2 + (i - elems1);
        *dst = accumulate ? *dst + red[0][e] : red[0][e];
    }
}
__global__ void plane_sum_k(const float* __restrict__ x, float* __restrict__ out, int HW) {
    __shared__ float red[32];
    const float* src = x + (size_t)blockIdx.x * HW;
    float s = 0.f;
    if ((HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        for (int i = threadIdx.x; i < (HW >> 2); i += blockDim.x) { const float4 v = s4[i]; s += (v.x + v.y) + (v.z + v.w); }
    } else {
        for (int i = threadIdx.x; i < HW; i += blockDim.x) s += src[i];
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}
__global__ void sum_over_n_k(const float* __restrict__ planes, float* __restrict__ out, int N, int C, int accumulate) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += planes[(size_t)n * C + c];
    out[c] = accumulate ? out[c] + s : s;
}
}  // namespace

int mtbc_i_splitk_reduce2(const float* partial, float* out, float* out2, int nsplit, size_t elems1, size_t elems2, int accumulate, hipStream_t st) {
    const size_t elems = elems1 + elems2;
    if (nsplit >= 256 && elems <= 16384) hipLaunchKernelGGL((splitk_reduce_k<64, 16>), dim3((unsigned)cdiv64(elems, 16)), dim3(1024), 0, st, partial, out, nsplit, elems, accumulate, out2, elems1);
    else if (nsplit > 32) hipLaunchKernelGGL(splitk_reduce_k<16>, dim3((unsigned)cdiv64(elems, 64)), dim3(1024), 0, st, partial, out, nsplit, elems, accumulate, out2, elems1);
    else hipLaunchKernelGGL(splitk_reduce_k<4>, dim3((unsigned)cdiv64(elems, 64)), dim3(256), 0, st, partial, out, nsplit, elems, accumulate, out2, elems1);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}
int mtbc_i_splitk_reduce(const float* partial, float* out, int nsplit, size_t elems, int accumulate, hipStream_t st) {
    return mtbc_i_splitk_reduce2(partial, out, nullptr, nsplit, elems, 0, accumulate, st);
}
// out[c] (+)= sum over n and the HW plane of x[n][c][:]; planes_ws holds N*C floats
int mtbc_i_channel_sums(const float* x, float* planes_ws, float* out, int N, int C, int HW, int accumulate, hipStream_t st) {
    hipLaunchKernelGGL(plane_sum_k, dim3(N * C), dim3(HW >= 4096 ? 256 : 64), 0, st, x, planes_ws, HW);
    MTBC_CHECK_LAUNCH();
    hipLaunchKernelGGL(sum_over_n_k, dim3(cdiv(C, 128)), dim3(128), 0, st, planes_ws, out, N, C, accumulate);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}
