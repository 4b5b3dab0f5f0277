// Shared device/host helpers for the gfx950 kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mtbc.h"

#define MTBC_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MTBC_CHECK_LAUNCH()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return MTBC_E_LAUNCH;          \
    } while (0)

// Timing / A-B probes (environment switches, load- / store-skipping bits inside the kernels) exist only in the `make probes`
// build (-DMTBC_PROBES -> libmtbc_hip_probes.so).  The shipped library reads no environment variable: every switch below
// folds to its default at compile time and the probe branches are dead code.
#ifdef MTBC_PROBES
#include <stdlib.h>
static inline int mtbc_probe_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
static inline bool mtbc_probe_set(const char* name) { return getenv(name) != nullptr; }
#define MTBC_DBG_BIT(p, bit) ((p).dbg & (bit))
#else
#define mtbc_probe_int(name, dflt) (dflt)
#define mtbc_probe_set(name) (false)
#define MTBC_DBG_BIT(p, bit) (0)
#endif

// Per-DEVICE facts (a kernel's opt-in to > 64 KB of dynamic LDS, the CU count, a kernel's resident-block capacity) are remembered per
// device, not per process: one bit / one slot per device ordinal in an atomic owned by the call site.  A process that drives several
// devices sets / queries each of them once; a race repeats an idempotent call.  (Round 3 kept them in `static bool` guards: right for one
// process per GPU, wrong the day one process drives two.)
#include <atomic>
constexpr int MTBC_MAX_DEVICES = 64;
struct DevOnce { std::atomic<unsigned long long> mask{0}; };
static inline int mtbc_current_device() { int dev = 0; return hipGetDevice(&dev) == hipSuccess ? (dev & (MTBC_MAX_DEVICES - 1)) : 0; }
template <typename K> static inline void mtbc_ensure_dyn_lds(DevOnce& once, K kernel, int bytes) {
    const unsigned long long bit = 1ull << mtbc_current_device();
    if (once.mask.load(std::memory_order_relaxed) & bit) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    once.mask.fetch_or(bit, std::memory_order_relaxed);
}
#define MTBC_ENSURE_DYN_LDS(kernel, bytes) do { static DevOnce once__; mtbc_ensure_dyn_lds(once__, (kernel), (bytes)); } while (0)
struct DevInts { std::atomic<int> v[MTBC_MAX_DEVICES]; DevInts() { for (auto& x : v) x.store(-1, std::memory_order_relaxed); } };
// value of `query()` on the current device, asked once per device (-1 = not asked yet; query() >= 0)
template <typename F> static inline int mtbc_per_device(DevInts& cache, F&& query) {
    std::atomic<int>& slot = cache.v[mtbc_current_device()];
    int c = slot.load(std::memory_order_relaxed);
    if (c < 0) { c = query(); if (c < 0) c = 0; slot.store(c, std::memory_order_relaxed); }
    return c;
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Segment table passed by value to kernels (virtual channel concat).
// Pointers that reach a kernel through LDS (segment tables) lose their address space and would compile to flat_*
// loads / stores (slower, and they tie vmcnt to lgkmcnt); these types put them back into global memory.
#ifdef MTBC_FLAT_AB   // A/B probe only
typedef float gfloat;
typedef char gchar;
#else
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) char gchar;
#endif
typedef __attribute__((address_space(1))) f32x4 gf32x4;

struct SegTable {
    float* ptr[MTBC_MAX_SEGS];
    long long bstride[MTBC_MAX_SEGS];
    int cbegin[MTBC_MAX_SEGS + 1];   // prefix sums of channels; cbegin[n] = total
    int accumulate[MTBC_MAX_SEGS];
    int n;
};

static inline int make_segtable(const mtbc_seg* segs, int n, int expect_channels, SegTable* t) {
    if (n < 1 || n > MTBC_MAX_SEGS) return MTBC_E_BADARG;
    int c = 0;
    for (int i = 0; i < MTBC_MAX_SEGS; ++i) {
        t->ptr[i] = nullptr; t->bstride[i] = 0; t->accumulate[i] = 0; t->cbegin[i] = c;
        if (i < n) {
            if (!segs[i].ptr || segs[i].channels <= 0) return MTBC_E_BADARG;
            t->ptr[i] = segs[i].ptr; t->bstride[i] = segs[i].batch_stride;
            t->accumulate[i] = segs[i].accumulate;
            c += segs[i].channels;
        }
    }
    t->cbegin[MTBC_MAX_SEGS] = c;
    for (int i = n; i <= MTBC_MAX_SEGS; ++i) t->cbegin[i] = c;
    t->n = n;
    return c == expect_channels ? MTBC_OK : MTBC_E_BADSHAPE;
}

// channel c -> its segment (select chain over constant indices: no dynamic kernarg indexing)
struct SegRef { float* ptr; long long bs; int cb; int acc; };
__device__ __forceinline__ SegRef seg_ref(const SegTable& t, int c) {
    SegRef r{t.ptr[0], t.bstride[0], 0, t.accumulate[0]};
#pragma unroll
    for (int i = 1; i < MTBC_MAX_SEGS; ++i) {
        const bool hit = i < t.n && c >= t.cbegin[i];
        r.ptr = hit ? t.ptr[i] : r.ptr;
        r.bs = hit ? t.bstride[i] : r.bs;
        r.cb = hit ? t.cbegin[i] : r.cb;
        r.acc = hit ? t.accumulate[i] : r.acc;
    }
    return r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum; every thread gets the result.  `red` = shared float[17+].  Deterministic.
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();                       // protect `red` from a previous use
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}

// norm_coop.hip: InstanceNorm straight into the 16-bit channel-blocked layout
extern "C" int mtbc_i_instnorm_fwd_c8(const mtbc_instnorm_args* a, hipStream_t st);
extern "C" int mtbc_i_instnorm_bwd_c8(const mtbc_instnorm_args* a, float* part, hipStream_t st);
extern "C" int mtbc_i_instnorm_bwd_c8_team(const mtbc_instnorm_args* a);
// c8_ops.hip: max-pool and 1x1 conv on 16-bit channel-blocked tensors
int mtbc_i_maxpool_c8_fwd(const mtbc_maxpool_args* a, hipStream_t st);
int mtbc_i_maxpool_c8_bwd(const mtbc_maxpool_args* a, hipStream_t st);
int mtbc_i_conv1x1_c8_fwd(const mtbc_conv1x1_args* a, hipStream_t st);
size_t mtbc_i_conv1x1_c8_wgrad_workspace(const mtbc_conv1x1_args* a);
int mtbc_i_conv1x1_c8_wgrad(const mtbc_conv1x1_args* a, hipStream_t st);
// internal (C++) helpers implemented in reduce.hip
int mtbc_i_splitk_reduce(const float* partial, float* out, int nsplit, size_t elems, int accumulate, hipStream_t st);
// rows of elems1 + elems2 floats per split: [0, elems1) summed into out, the rest into out2
int mtbc_i_splitk_reduce2(const float* partial, float* out, float* out2, int nsplit, size_t elems1, size_t elems2, int accumulate, hipStream_t st);
// convt2.hip: ConvTranspose k == s == 2 backward, operands straight from HBM into MFMA fragments
bool mtbc_i_convT2_fwd_ok(const mtbc_convT_args* a);
int mtbc_i_convT2_fwd(const mtbc_convT_args* a, hipStream_t st);
bool mtbc_i_convT2_fwd_c8_ok(const mtbc_convT_args* a);
int mtbc_i_convT2_fwd_c8(const mtbc_convT_args* a, hipStream_t st);
bool mtbc_i_convT2_fwd_lp_c8_ok(const mtbc_convT_args* a);
int mtbc_i_convT2_fwd_lp_c8(const mtbc_convT_args* a, hipStream_t st);
bool mtbc_i_convT2_dgrad_ok(const mtbc_convT_args* a);
bool mtbc_i_convT2_wgrad_ok(const mtbc_convT_args* a);
void mtbc_i_convT2_wgrad_plan(const mtbc_convT_args* a, int* steps_per_split, int* nsplit);
int mtbc_i_convT2_dgrad(const mtbc_convT_args* a, int compute, hipStream_t st);
int mtbc_i_convT2_wgrad(const mtbc_convT_args* a, int compute, float* partial, float* dbias_part, int steps_per_split, int nsplit, hipStream_t st);
int mtbc_i_channel_sums(const float* x, float* planes_ws, float* out, int N, int C, int HW, int accumulate, hipStream_t st);
