// InstanceNorm + LeakyReLU whose output IS the 16-bit channel-blocked layout the 3x3 convs read (MTBC_LAYOUT_C8):
// forward  z (fp32 planes, or 16-bit channel-blocked: z_layout) -> a8  [n][C/8][H*W][8]   (+ fp32 planes y when something else reads them)
// backward z, dy (fp32 planes, or 16-bit channel-blocked [+ an fp32 planar partial]: z_layout / dy_layout) -> dz8 [n][C/8][H*W][8]
// replaces nn.InstanceNorm2d + nn.LeakyReLU (MTnnUNet.py:35-36, MTUNetPlusPlus.py:20-22) in the 16-bit modes.
//
// A 16-byte piece holds 8 channels of one pixel, so a workgroup must own all 8 planes of a channel group -- 2 MB at
// 256x256, four times a CU's register file.  Split-plane cooperative kernel: a persistent grid of resident workgroups,
// TEAMS of T workgroups per (n, channel group); each member keeps a pixel slab of all 8 planes in registers (4-byte
// loads coalesced per plane, one 16-byte store per pixel), reduces its slab, and the members exchange the per-channel
// partials through a mailbox in global memory.  One HBM pass: forward 6 (10 with fp32 y) B/element instead of 8 + 6
// (InstanceNorm + pack), backward 10 instead of 14..16 + 4.
//
// Mailbox protocol (no fences, no L2 write-back across XCDs): a word is {fp32 value, 32-bit tag} written and polled
// with relaxed agent-scope 64-bit atomics -- single-copy atomic, so the value is valid whenever the tag matches.
// tag = (epoch mod (2^22 - 1) + 1) << 10 | exchange number; `epoch` lives in the caller's persistent state block, is read by every
// workgroup when it starts and advanced by the LAST workgroup to finish (which thereby knows everybody has read it).
// Two mailbox slots by exchange parity: a member can post exchange s+2 only after all members posted s+1, i.e. after
// they finished reading s.  Polls are bounded (CO_SPIN_MAX) and give up together: a protocol failure sets state->err (sticky;
// the HOST must read it -- trainer.FusedTrainStep.check_nan does -- because the outputs are then garbage), never a hang.
// Requires all workgroups of a team to be resident: the grid is sized from the occupancy query, capped at 2 per CU, minus
// the CUs the caller reserves per call for kernels of other streams (mtbc_instnorm_args.coop_reserve_cus).
#include "common.h"
#include <stddef.h>
#include <type_traits>

namespace {

constexpr int CO_THREADS = 512, CO_WAVES = CO_THREADS / 64;       // (256-thread workgroups in teams of up to 64: forward 18 % slower)
constexpr int CO_MAXT = 128, CO_NV = 16;     // 128 members = a 512x512 plane group at 4 pixels per thread
constexpr unsigned CO_SPIN_MAX = 1u << 26;      // about a minute of polling: a member that waits for a CU held by a collective of another stream is late, not lost
constexpr size_t CO_MAILBOX_OFF = 256;
constexpr int CO_MAX_TEAMS = 1024;
constexpr size_t CO_TEAM_WORDS = 2 * CO_MAXT * CO_NV;              // u64 words per team
struct CoopHdr { unsigned epoch, done, err, pad; };

struct CoP {
    int N, C, HW, G8, T, items, nteams, f16;
    float eps, slope;
    const float* z; const float* gamma; const float* beta;
    float* y; long long ybs;                 // optional fp32 planes
    unsigned short* y8;                      // forward output
    float* mean; float* rstd;
    const float* dy; long long dybs;
    unsigned short* dz8;                     // backward output
    float* part;                             // backward: [N*C][3] = {sum g, sum g*xh, 0} or nullptr
    float* part3;                            // backward: [N*C][T] per-member sums of dz
    void* state;
    const unsigned short* z8;                // z as 16-bit channel-blocked [n][C/8][HW][8] (else nullptr: fp32 planes `z`)
    const unsigned short* dy8; long long dy8bs;   // dy as 16-bit channel-blocked (else nullptr: fp32 planes `dy`)
    const float* dyx;                        // with dy8: optional fp32 planar partial gradient (N,C,H,W), added while loading
    int zf16;                                // z8 holds fp16 values (whatever the output type)
    const float* r1; const float* r1w;       // backward: rank-1 gradient term w[c] * r1[n][pixel] (a one-output 1x1 head), or nullptr
    float* r1dw; float* r1db;                // ... and the head's own weight / bias gradient partials: [N*C][T], [N][T], or nullptr
    const float* pg; const unsigned short* pa; int W;      // backward: gradient of a 2 x 2 max-pool of this activation (pooled fp32 planes + argmax codes), or nullptr
    unsigned short* py8; unsigned short* parg;             // forward (streaming pass): the 2 x 2 max-pool of the activation + its argmax codes, or nullptr
    unsigned short* y16;                     // forward (streaming pass): the planar copy as 16-bit planes (N,C,H,W) of the output type instead of `y`, or nullptr
#ifdef MTBC_PROBES
    unsigned long long* ts;                  // phase timestamps of the channel-group backward (MTBC_INB_TS=1): [block][16] ticks of the 100 MHz clock
#endif
};
#ifdef MTBC_PROBES
#define MTBC_NTS(p, k) do { if ((p).ts && threadIdx.x == 0) (p).ts[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#else
#define MTBC_NTS(p, k) do { } while (0)
#endif

typedef float co_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 co_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 co_f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned co_u32x4 __attribute__((ext_vector_type(4)));
template <bool F16> __device__ __forceinline__ unsigned co_pk(float a, float b) {
    if constexpr (F16) return __builtin_bit_cast(unsigned, __builtin_convertvector((co_f32x2){a, b}, co_f16x2));
    else return __builtin_bit_cast(unsigned, __builtin_convertvector((co_f32x2){a, b}, co_bf16x2));
}

// the 8 channels of a stored piece as fp32 (exact)
typedef _Float16 co_f16x8 __attribute__((ext_vector_type(8)));
template <bool F16> __device__ __forceinline__ void co_unpk(const co_u32x4 w, float (&o)[8]) {
    if constexpr (F16) {          // (bit-cast the WHOLE vector: __builtin_bit_cast of a subscripted element reads element 0)
        const co_f16x8 t = __builtin_bit_cast(co_f16x8, w);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (float)t[i];
    } else {
#pragma unroll
        for (int h = 0; h < 4; ++h) { const unsigned u = w[h]; o[2 * h] = __uint_as_float(u << 16); o[2 * h + 1] = __uint_as_float(u & 0xffff0000u); }
    }
}

__device__ __forceinline__ void mb_post(unsigned long long* slot, float v, unsigned tag) {
    __hip_atomic_store(slot, ((unsigned long long)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// A poll gives up after CO_SPIN_MAX tries (sets hdr->err) -- and as soon as ANY wave of ANY workgroup has given up: err
// is re-read every CO_ERR_EVERY failed tries, so that one timeout ends the whole launch within
// microseconds instead of every later wait of every exchange spinning its own full budget.  The host reads the word
// (FusedTrainStep.check_nan / FusedEvalStep.result) and raises.
constexpr unsigned CO_ERR_EVERY = 256;
__device__ __forceinline__ float mb_wait(unsigned long long* slot, unsigned tag, CoopHdr* hdr) {
    unsigned long long w = 0;
    unsigned spins = 0;
    for (;;) {
        w = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(w >> 32) == tag) break;
        ++spins;        // (the error word is looked at only after CO_ERR_EVERY failed polls: never on the fast path of an exchange)
        if ((spins & (CO_ERR_EVERY - 1)) == 0 && __hip_atomic_load(&hdr->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        if (spins >= CO_SPIN_MAX) { __hip_atomic_store(&hdr->err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        __builtin_amdgcn_s_sleep(2);
    }
    return __uint_as_float((unsigned)w);
}
// Every member posts its NV values (thread i < NV holds value i in `mine`) and collects everybody's into xch[m][i].
template <int NV, int THREADS>
__device__ __forceinline__ void team_exchange(float mine, unsigned long long* mb, int member, int T, unsigned& seq, unsigned epoch,
                                              CoopHdr* hdr, float (*xch)[CO_NV]) {
    const int tid = threadIdx.x;
    // 22 bits of launch epoch (never 0: zeroed / never-written mailbox words cannot match) + 10 bits of exchange number; a word
    // is rewritten at least once per step program, so a repeat after 4M launches cannot meet a stale twin
    const unsigned tag = ((epoch % 4194303u + 1u) << 10) | (seq & 1023u);
    unsigned long long* base = mb + (size_t)(seq & 1u) * CO_MAXT * CO_NV;
    if (tid < NV) mb_post(base + member * CO_NV + tid, mine, tag);
    for (int idx = tid; idx < T * CO_NV; idx += THREADS) {
        const int m = idx >> 4, i = idx & 15;
        if (i < NV) xch[m][i] = mb_wait(base + m * CO_NV + i, tag, hdr);
    }
    ++seq;
    __syncthreads();
}
// Sums of NV per-thread values over the workgroup -> tot[0..NV) in LDS (fixed order: deterministic).  The per-channel
// quantities stay in LDS and are read back (broadcast) where they are used: 8 channels x {pivot, sums, mean, scale,
// shift, ...} as registers cost 60+ VGPRs (or as many spilled SGPRs) and with them the second resident workgroup.
// Wave sum on the VALU: four DPP row-shift adds leave each 16-lane row's sum in its last lane, four lane reads add the rows (fixed order).
// The ds_bpermute butterfly of wave_sum is six dependent LDS round trips per value -- with 16 .. 25 values per reduction that was
// ~200 of them per item in the backward kernel.
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));      // row_shr:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));      // row_shr:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));      // row_shr:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));      // row_shr:1
    const int r = __builtin_bit_cast(int, v);
    return ((__builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 15)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 31))) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 47))) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 63));
}
template <int NV, int THREADS>
__device__ __forceinline__ void block_reduce_lds(float (&a)[NV], float (*red)[16], float* tot) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) a[i] = wave_sum_dpp(a[i]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[wv][i] = a[i];
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < THREADS / 64; ++w) t += red[w][threadIdx.x];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
}
// tot[0..NV) of every member -> their sum in tot[0..NV) (member order: deterministic)
template <int NV, int THREADS>
__device__ __forceinline__ void team_sum(float* tot, unsigned long long* mb, int member, int T, unsigned& seq, unsigned epoch,
                                         CoopHdr* hdr, float (*xch)[CO_NV]) {
    if (T == 1) return;
    const float mine = threadIdx.x < NV ? tot[threadIdx.x] : 0.f;
    team_exchange<NV, THREADS>(mine, mb, member, T, seq, epoch, hdr, xch);
    if (threadIdx.x < NV) { float t = 0.f; for (int m = 0; m < T; ++m) t += xch[m][threadIdx.x]; tot[threadIdx.x] = t; }
    __syncthreads();
}
__device__ __forceinline__ void coop_finish(CoopHdr* hdr, unsigned epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned d = __hip_atomic_fetch_add(&hdr->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d == gridDim.x - 1) {             // everybody else has finished, hence has read `epoch`
            __hip_atomic_store(&hdr->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&hdr->epoch, epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// COOP = teams of resident workgroups per (n, channel group) (planes of 16K pixels and more); !COOP = ONE workgroup owns the
// whole group of 8 planes (up to 64 x 64: THREADS x PPT >= H*W), grid = items, no mailbox, no state.
// Inputs per launch (uniform branches): z / dy as fp32 planes (raw buffer loads, plane stride in the scalar offset) or as
// 16-bit channel-blocked pieces (ONE 16-byte load per pixel = its 8 channels).
template <int THREADS, int PPT, bool F16, bool COOP, int ZC8>      // ZC8: 0 = fp32 planar z, 1 = 16-bit channel-blocked of the output's type, 2 = channel-blocked fp16 (bf16 output)
__global__ __launch_bounds__(THREADS, THREADS >= 512 ? 4 : 2) void in_fwd_c8_kernel(const CoP p) {
    constexpr bool ZF16 = F16 || ZC8 == 2;
    __shared__ float red[THREADS / 64][16];
    __shared__ float xch[COOP ? CO_MAXT : 1][CO_NV];
    __shared__ float tot[16];
    __shared__ float cst[4][8];              // mean, scale = gamma * rstd, shift = beta, pivot
    const int tid = threadIdx.x;
    CoopHdr* hdr = COOP ? reinterpret_cast<CoopHdr*>(p.state) : nullptr;
    unsigned epoch = 0;
    if constexpr (COOP) epoch = __hip_atomic_load(&hdr->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int team = COOP ? blockIdx.x / p.T : blockIdx.x, member = COOP ? blockIdx.x % p.T : 0;
    unsigned long long* mb = COOP ? reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(p.state) + CO_MAILBOX_OFF) + (size_t)team * CO_TEAM_WORDS : nullptr;
    const int slab = p.HW / p.T;
    const float inv = 1.0f / (float)p.HW;
    unsigned seq = 0;
    for (int item = team; item < p.items; item += p.nteams) {
        const int n = item / p.G8, g = item % p.G8;
        float v[8][PPT];
        if constexpr (ZC8 != 0) {
            // one 16-byte piece per pixel; padding lanes read the plane's first pixel (= the pivot)
            const unsigned short* zg = p.z8 + ((size_t)n * p.G8 + g) * p.HW * 8;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int px = tid + THREADS * k;
                const size_t idx = px < slab ? (size_t)member * slab + px : 0;
                float o[8];
                co_unpk<ZF16>(*reinterpret_cast<const co_u32x4*>(zg + idx * 8), o);
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c][k] = o[c];
            }
            if (tid < 8) cst[3][tid] = ZF16 ? (float)reinterpret_cast<const _Float16*>(zg)[tid] : __uint_as_float((unsigned)zg[tid] << 16);
        } else {
            const float* zp = p.z + ((size_t)n * p.C + 8 * g) * p.HW;         // the 8 planes of the group
            // raw buffer loads over the group's 8 planes: the plane stride rides in the scalar offset, so a load costs one
            // 32-bit VGPR offset instead of a 64-bit address (the 32 addresses of a slab were half of the kernel's VGPRs)
            const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(zp), 0, 8 * p.HW * 4, 0x00020000);
            const int plane_b = p.HW * 4;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int px = tid + THREADS * k;
                const int off = px < slab ? (member * slab + px) * 4 : 0;       // padding lanes read the pivot
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(zr, off, c * plane_b, 0));
            }
            if (tid < 8) cst[3][tid] = zp[(size_t)tid * p.HW];
        }
        __syncthreads();
        // One pass: sums of (x - pivot) and (x - pivot)^2 with the plane's first pixel as the pivot (the same for every
        // member; a sample of the plane, so var = Q/HW - (S/HW)^2 cancels a few bits at most; padding lanes hold the pivot)
        float sq[16];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float pv = cst[3][c];
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int k = 0; k < PPT; ++k) { const float d = v[c][k] - pv; s += d; q += d * d; }
            sq[c] = s; sq[8 + c] = q;
        }
        block_reduce_lds<16, THREADS>(sq, red, tot);
        if constexpr (COOP) team_sum<16, THREADS>(tot, mb, member, p.T, seq, epoch, hdr, xch);
        if (tid < 8) {
            const float ms = tot[tid] * inv;
            const float mean = cst[3][tid] + ms;
            const float rstd = 1.0f / sqrtf(fmaxf(tot[8 + tid] * inv - ms * ms, 0.f) + p.eps);
            cst[0][tid] = mean; cst[1][tid] = (p.gamma ? p.gamma[8 * g + tid] : 1.f) * rstd; cst[2][tid] = p.beta ? p.beta[8 * g + tid] : 0.f;
            if (member == 0) { p.mean[(size_t)n * p.C + 8 * g + tid] = mean; p.rstd[(size_t)n * p.C + 8 * g + tid] = rstd; }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float mean = cst[0][c], ga = cst[1][c], be = cst[2][c];
#pragma unroll
            for (int k = 0; k < PPT; ++k) { const float t = (v[c][k] - mean) * ga + be; v[c][k] = t > 0.f ? t : t * p.slope; }
        }
        unsigned short* ob = p.y8 + (((size_t)n * p.G8 + g) * p.HW + (size_t)member * slab) * 8;
        float* yb = p.y ? p.y + (size_t)n * p.ybs + (size_t)(8 * g) * p.HW + (size_t)member * slab : nullptr;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int px = tid + THREADS * k;
            if (px < slab) {
                co_u32x4 w;
#pragma unroll
                for (int h = 0; h < 4; ++h) w[h] = co_pk<F16>(v[2 * h][k], v[2 * h + 1][k]);
                *reinterpret_cast<co_u32x4*>(ob + (size_t)px * 8) = w;
                if (yb) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) yb[(size_t)c * p.HW + px] = v[c][k];
                }
            }
        }
        __syncthreads();                     // cst / tot are rewritten by the next item
    }
    if constexpr (COOP) coop_finish(hdr, epoch);
}

template <int THREADS, int PPT, bool F16, bool COOP, int ZC8, int DY8>      // ZC8 as above; DY8: 0 = fp32 planar dy, 1 = 16-bit channel-blocked, 2 = that + an fp32 planar partial
__global__ __launch_bounds__(THREADS, THREADS >= 512 ? 4 : 2) void in_bwd_c8_kernel(const CoP p) {
    constexpr bool ZF16 = F16 || ZC8 == 2;
    __shared__ float red[THREADS / 64][16];
    __shared__ float xch[COOP ? CO_MAXT : 1][CO_NV];
    __shared__ float tot[16];
    __shared__ float cst[3][8];              // k = rstd * gamma, m1 = S1 / HW, m2 = S2 / HW
    __shared__ float4 cq[8];                 // {mean, rstd, gamma, beta} of the item's 8 channels
    const int tid = threadIdx.x;
    CoopHdr* hdr = COOP ? reinterpret_cast<CoopHdr*>(p.state) : nullptr;
    unsigned epoch = 0;
    if constexpr (COOP) epoch = __hip_atomic_load(&hdr->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (round 4: teams pinned to ONE XCD -- team t = the workgroups 8 (T (t / 8) + m) + t % 8 -- were built and measured: 11.33 / 11.34 / 11.37 ms
    //  per step against 11.36 / 11.36 / 11.33, InstanceNorm backward 1.809 against 1.812 ms: nothing.  The mailbox words are agent-scope
    //  atomics; they are served by the memory side wherever the members sit.  profiles/r04_ab.txt)
    const int team = COOP ? blockIdx.x / p.T : blockIdx.x, member = COOP ? blockIdx.x % p.T : 0;
    unsigned long long* mb = COOP ? reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(p.state) + CO_MAILBOX_OFF) + (size_t)team * CO_TEAM_WORDS : nullptr;
    const int slab = p.HW / p.T;
    const float inv = 1.0f / (float)p.HW;
    unsigned seq = 0;
    int itk = 0;          // (probes: item counter of the phase stamps)
    for (int item = team; item < p.items; item += p.nteams, ++itk) {
        const int n = item / p.G8, g = item % p.G8;
        const size_t plane0 = (size_t)n * p.C + 8 * g;
        const int plane_b = p.HW * 4;
        if (itk < 2) MTBC_NTS(p, 7 * itk);
        // (the pixel offsets are recomputed per item from an opaque copy of the thread index: hoisted out of this loop they lived across the
        //  team exchange, were spilled, and every reload -- a scratch load, counted by vmcnt -- sat between the z loads of the next item, which
        //  then went out one memory round trip at a time)
        int tl = tid;
        asm volatile("" : "+v"(tl));
        // The per-channel constants: thread c < 8 requests channel c's NOW, in front of the tensor loads, and publishes them through LDS.
        // (Read where they are used -- `p.mean[plane0 + c]` inside the channel loop below -- they were 8 x 4 vector loads with a wait per
        //  channel: eight dependent memory round trips per item AFTER the tensor data had arrived, and one more for rstd * gamma behind
        //  the team exchange.  The compiler cannot use scalar loads here: the kernel stores through other pointers.)
        float4 myc = make_float4(0.f, 1.f, 1.f, 0.f);
        if (tid < 8) myc = make_float4(p.mean[plane0 + tid], p.rstd[plane0 + tid], p.gamma ? p.gamma[8 * g + tid] : 1.f, p.beta ? p.beta[8 * g + tid] : 0.f);
        float xh[8][PPT], gy[8][PPT];
        // out-of-slab lanes: offsets past the buffer, the bounds check returns 0
        if constexpr (ZC8 != 0) {
            const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.z8 + ((size_t)n * p.G8 + g) * p.HW * 8), 0, p.HW * 16, 0x00020000);
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int px = tl + THREADS * k;
                const int off = px < slab ? (member * slab + px) * 16 : 0x7ffffff0;
                float o[8];
                co_unpk<ZF16>(__builtin_bit_cast(co_u32x4, __builtin_amdgcn_raw_buffer_load_b128(zr, off, 0, 0)), o);
#pragma unroll
                for (int c = 0; c < 8; ++c) xh[c][k] = o[c];
            }
        } else {
            const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.z + plane0 * p.HW), 0, 8 * p.HW * 4, 0x00020000);
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int px = tl + THREADS * k;
                const int off = px < slab ? (member * slab + px) * 4 : 0x7ffffff0;
#pragma unroll
                for (int c = 0; c < 8; ++c) xh[c][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(zr, off, c * plane_b, 0));
            }
        }
        if constexpr (DY8 != 0) {
            const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.dy8 + (size_t)n * p.dy8bs + (size_t)g * p.HW * 8), 0, p.HW * 16, 0x00020000);
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int px = tl + THREADS * k;
                const int off = px < slab ? (member * slab + px) * 16 : 0x7ffffff0;
                float o[8];
                co_unpk<F16>(__builtin_bit_cast(co_u32x4, __builtin_amdgcn_raw_buffer_load_b128(gr, off, 0, 0)), o);
#pragma unroll
                for (int c = 0; c < 8; ++c) gy[c][k] = o[c];
            }
            if constexpr (DY8 == 2) {        // the fp32 planar partial from the tensor's other readers (pool / ConvT / 1x1 backward)
                const __amdgpu_buffer_rsrc_t er = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dyx + plane0 * p.HW), 0, 8 * p.HW * 4, 0x00020000);
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const int px = tl + THREADS * k;
                    const int off = px < slab ? (member * slab + px) * 4 : 0x7ffffff0;
#pragma unroll
                    for (int c = 0; c < 8; ++c) gy[c][k] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(er, off, c * plane_b, 0));
                }
            }
        } else if (p.dy) {
            const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy + (size_t)n * p.dybs + (size_t)(8 * g) * p.HW), 0, 8 * p.HW * 4, 0x00020000);
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int px = tl + THREADS * k;
                const int off = px < slab ? (member * slab + px) * 4 : 0x7ffffff0;
#pragma unroll
                for (int c = 0; c < 8; ++c) gy[c][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gr, off, c * plane_b, 0));
            }
            if (p.dyx) {        // a second fp32 planar contribution (the gathered dgrad wrote its own buffer instead of read-modify-writing dy)
                const __amdgpu_buffer_rsrc_t er = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dyx + plane0 * p.HW), 0, 8 * p.HW * 4, 0x00020000);
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const int px = tl + THREADS * k;
                    const int off = px < slab ? (member * slab + px) * 4 : 0x7ffffff0;
#pragma unroll
                    for (int c = 0; c < 8; ++c) gy[c][k] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(er, off, c * plane_b, 0));
                }
            }
        } else {        // no gradient tensor at all: the rank-1 term below is everything (a tensor that only a 1x1 head reads)
#pragma unroll
            for (int k = 0; k < PPT; ++k)
#pragma unroll
                for (int c = 0; c < 8; ++c) gy[c][k] = 0.f;
        }
        float hv[PPT];          // the head's gradient at this thread's pixels (0 without a head, and outside the slab)
#pragma unroll
        for (int k = 0; k < PPT; ++k) hv[k] = 0.f;
        if (p.r1) {
            // the input gradient of a ONE-output 1x1 conv head reading this activation is rank 1, w[c] * dyhead[n, pixel]: formed
            // here from the head's 4-byte-per-pixel gradient instead of being written (4 B x C per pixel) by the head's dgrad
            // and read back -- same products, added in the order the fan-in would have added them last
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.r1 + (size_t)n * p.HW), 0, p.HW * 4, 0x00020000);
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int px = tl + THREADS * k;
                hv[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, px < slab ? (member * slab + px) * 4 : 0x7ffffff0, 0, 0));
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float wc = p.r1w[8 * g + c];
#pragma unroll
                for (int k = 0; k < PPT; ++k) gy[c][k] += wc * hv[k];
            }
        }
        if (p.pg) {
            // the gradient coming back through a 2 x 2 max-pool of this activation: the pooled gradient goes to the window position the
            // forward recorded (2 bits per channel); formed here instead of being written as a 4x larger fp32 tensor and read back
            const int oHW = p.HW >> 2, oW = p.W >> 1;
            const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.pg + plane0 * oHW), 0, 8 * oHW * 4, 0x00020000);
            const unsigned short* pa = p.pa + ((size_t)n * p.G8 + g) * oHW;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int px = tl + THREADS * k;
                const bool ok = px < slab;
                const int q = ok ? member * slab + px : 0;
                const int yy = q / p.W, xx = q - yy * p.W;
                const int pq = (yy >> 1) * oW + (xx >> 1);
                const unsigned pos = ((yy & 1) << 1) | (xx & 1);
                const unsigned code = ok ? pa[pq] : 0u;
                const int off = ok ? pq * 4 : 0x7ffffff0;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float gv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pr, off, c * oHW * 4, 0));
                    gy[c][k] += ((code >> (2 * c)) & 3u) == pos ? gv : 0.f;
                }
            }
        }
        if (itk < 2) MTBC_NTS(p, 7 * itk + 1);          // loads issued
        if (tid < 8) cq[tid] = myc;
        __syncthreads();
        if (itk < 2) MTBC_NTS(p, 7 * itk + 2);          // constants published (their loads have landed)
        float ss[16], sw[9];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float4 qc = cq[c];
            const float mean = qc.x, rstd = qc.y, ga = qc.z, be = qc.w;
            float s1 = 0.f, s2 = 0.f, s4 = 0.f;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const bool ok = tl + THREADS * k < slab;
                const float x = ok ? (xh[c][k] - mean) * rstd : 0.f;
                const float pre = x * ga + be;
                const float y = gy[c][k] * (pre > 0.f ? 1.f : p.slope);
                xh[c][k] = x; gy[c][k] = y;
                s1 += y; s2 += y * x;
                if (p.r1dw) {           // the head's weight gradient: sum over pixels of the STORED (rounded) activation times the head's gradient
                    const float act = pre > 0.f ? pre : pre * p.slope;
                    const unsigned u = co_pk<F16>(act, 0.f);
                    float ar;
                    if constexpr (F16) { const co_f16x2 t2 = __builtin_bit_cast(co_f16x2, u); ar = (float)t2[0]; } else ar = __uint_as_float(u << 16);
                    s4 += ar * hv[k];
                }
            }
            ss[c] = s1; ss[8 + c] = s2; sw[c] = s4;
        }
        if (p.r1dw) {
            float sb = 0.f;
#pragma unroll
            for (int k = 0; k < PPT; ++k) sb += hv[k];
            sw[8] = sb;
            block_reduce_lds<9, THREADS>(sw, red, tot);
            if (tid < 8) p.r1dw[(plane0 + tid) * p.T + member] = tot[tid];
            if (tid == 8 && g == 0) p.r1db[(size_t)n * p.T + member] = tot[8];
            __syncthreads();
        }
        if (itk < 2) MTBC_NTS(p, 7 * itk + 3);          // tensor data landed, per-thread sums formed
        block_reduce_lds<16, THREADS>(ss, red, tot);
        if (itk < 2) MTBC_NTS(p, 7 * itk + 4);          // block sums
        if constexpr (COOP) team_sum<16, THREADS>(tot, mb, member, p.T, seq, epoch, hdr, xch);
        if (itk < 2) MTBC_NTS(p, 7 * itk + 5);          // team exchange done
        if (tid < 8) {
            cst[0][tid] = myc.y * myc.z;
            cst[1][tid] = tot[tid] * inv; cst[2][tid] = tot[8 + tid] * inv;
            if (p.part && member == 0) { float* q = p.part + 3 * (plane0 + tid); q[0] = tot[tid]; q[1] = tot[8 + tid]; q[2] = 0.f; }
        }
        __syncthreads();
        float s3[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float kk = cst[0][c], m1 = cst[1][c], m2 = cst[2][c];
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const float o = (tl + THREADS * k < slab) ? kk * (gy[c][k] - m1 - xh[c][k] * m2) : 0.f;
                gy[c][k] = o; t += o;
            }
            s3[c] = t;
        }
        unsigned short* ob = p.dz8 + (((size_t)n * p.G8 + g) * p.HW + (size_t)member * slab) * 8;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int px = tl + THREADS * k;
            if (px < slab) {
                co_u32x4 w;
#pragma unroll
                for (int h = 0; h < 4; ++h) w[h] = co_pk<F16>(gy[2 * h][k], gy[2 * h + 1][k]);
                *reinterpret_cast<co_u32x4*>(ob + (size_t)px * 8) = w;
            }
        }
        if (p.part) {       // every member's share of sum dz; in_dparam_kernel adds them up (no second exchange)
            block_reduce_lds<8, THREADS>(s3, red, tot);
            if (tid < 8) p.part3[(plane0 + tid) * p.T + member] = tot[tid];
        }
        if (itk < 2) MTBC_NTS(p, 7 * itk + 6);          // dz stored (issued), parameter partials
        __syncthreads();                     // cst / tot are rewritten by the next item
    }
    MTBC_NTS(p, 14);
    if constexpr (COOP) coop_finish(hdr, epoch);
    MTBC_NTS(p, 15);
}

// ---------------------------------------------------------------- statistics from the conv epilogue + streaming normalisation
// The producing convolution left {sum, sum of squares} of the stored z per (image, pixel subset, channel)
// (mtbc_conv3x3_args.stats_partial): one wave per (n, c) adds the subsets up in a fixed order, in double (raw moments of
// 16-bit values: the cancellation in E[z^2] - E[z]^2 is bounded by what the 16-bit storage of z leaves of a plane whose mean
// dwarfs its spread), and the normalisation itself is ONE streaming pass -- no reduction, no team exchange, any grid.
__global__ void in_stats_finalize_kernel(const float* __restrict__ part, int slots, int C, int HW, float eps, float* __restrict__ mean,
                                         float* __restrict__ rstd, int planes) {
    const int plane = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (plane >= planes) return;
    const int n = plane / C, c = plane % C;
    const float* q = part + ((size_t)n * slots * C + c) * 2;
    double s = 0.0, qq = 0.0;
    int t = lane;
    for (; t + 192 < slots; t += 256) {          // four loads in flight, added in the order of the plain loop below (one load per round = a memory round trip per 64 subsets)
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float2*>(q + (size_t)(t + 64 * u) * C * 2);
#pragma unroll
        for (int u = 0; u < 4; ++u) { s += (double)v[u].x; qq += (double)v[u].y; }
    }
    for (; t < slots; t += 64) { const float2 v = *reinterpret_cast<const float2*>(q + (size_t)t * C * 2); s += (double)v.x; qq += (double)v.y; }
    s = wave_sum_d(s); qq = wave_sum_d(qq);
    if (lane == 0) {
        const double m = s / (double)HW, var = fmax(qq / (double)HW - m * m, 0.0);
        mean[plane] = (float)m; rstd[plane] = (float)(1.0 / sqrt(var + (double)eps));
    }
}
constexpr int AP_THREADS = 256, AP_PPT = 2;
// FIN: the workgroup first adds up the conv epilogue's partials of its 8 channels itself (few pixel subsets: planes up to 64 x 64)
// -- no separate finalize launch; thread t sums subsets t/8, t/8 + 32, ... of channel t % 8 in double, the 32 partial sums are
// added in a fixed order, and the workgroup of the plane group's first pixels writes mean / rstd for the backward pass.
// POOL: the activation is max-pooled (2 x 2) by its next reader: a thread normalises one WINDOW (2 rows x 2 adjacent pixels = 4 pieces) and
// also writes the window's maxima of the STORED values as one piece of the pooled tensor (+ the 2-bit positions of the maxima for the
// pool's backward) -- what maxpool_c8_fwd_kernel would compute from the tensor this kernel has just written, without reading it back.
template <bool F16, bool FIN, bool ZF16, bool POOL = false>
__global__ __launch_bounds__(AP_THREADS) void in_apply_fwd_c8_kernel(const CoP p, const float* __restrict__ part, const int slots) {
    const int item = blockIdx.y, n = item / p.G8, g = item % p.G8;
    float mu[8], ga[8], be[8];
    if constexpr (FIN) {
        __shared__ double acc_s[32][8], acc_q[32][8];
        __shared__ float st[2][8];
        const int c = threadIdx.x & 7, sub = threadIdx.x >> 3;
        const float* q = part + (((size_t)n * slots) * p.C + 8 * g + c) * 2;
        double s = 0.0, qq = 0.0;
        for (int t = sub; t < slots; t += 32) { const float2 v = *reinterpret_cast<const float2*>(q + (size_t)t * p.C * 2); s += (double)v.x; qq += (double)v.y; }
        acc_s[sub][c] = s; acc_q[sub][c] = qq;
        __syncthreads();
        if (threadIdx.x < 8) {
            double ts = 0.0, tq = 0.0;
            for (int k = 0; k < 32; ++k) { ts += acc_s[k][threadIdx.x]; tq += acc_q[k][threadIdx.x]; }
            const double m = ts / (double)p.HW, var = fmax(tq / (double)p.HW - m * m, 0.0);
            const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)p.eps));
            st[0][threadIdx.x] = mean; st[1][threadIdx.x] = rstd;
            if (blockIdx.x == 0) { p.mean[(size_t)n * p.C + 8 * g + threadIdx.x] = mean; p.rstd[(size_t)n * p.C + 8 * g + threadIdx.x] = rstd; }
        }
        __syncthreads();
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) { mu[c2] = st[0][c2]; ga[c2] = (p.gamma ? p.gamma[8 * g + c2] : 1.f) * st[1][c2]; be[c2] = p.beta ? p.beta[8 * g + c2] : 0.f; }
    } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const size_t pl = (size_t)n * p.C + 8 * g + c;
            mu[c] = p.mean[pl]; ga[c] = (p.gamma ? p.gamma[8 * g + c] : 1.f) * p.rstd[pl]; be[c] = p.beta ? p.beta[8 * g + c] : 0.f;
        }
    }
    const unsigned short* zg = p.z8 + ((size_t)n * p.G8 + g) * p.HW * 8;
    unsigned short* ob = p.y8 + ((size_t)n * p.G8 + g) * p.HW * 8;
    float* yb = p.y ? p.y + (size_t)n * p.ybs + (size_t)(8 * g) * p.HW : nullptr;
    // 16-bit planes: a thread owns two ADJACENT pixels, so that a channel's two values are one 4-byte store (uniform branch)
    unsigned short* yh = p.y16 ? p.y16 + (size_t)n * p.ybs + (size_t)(8 * g) * p.HW : nullptr;
    if constexpr (POOL) {
        const int oW = p.W >> 1, oHW = p.HW >> 2;
        unsigned short* pb = p.py8 + ((size_t)n * p.G8 + g) * oHW * 8;
        for (int q = blockIdx.x * AP_THREADS + threadIdx.x; q < oHW; q += gridDim.x * AP_THREADS) {
            const int oy = q / oW, ox = q - oy * oW;
            const size_t p00 = (size_t)(2 * oy) * p.W + 2 * ox;
            const size_t pos[4] = {p00, p00 + 1, p00 + p.W, p00 + p.W + 1};
            co_u32x4 wz[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) wz[e] = *reinterpret_cast<const co_u32x4*>(zg + pos[e] * 8);
            float best[8];
            unsigned code = 0;
            co_u32x4 oprev = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v[8];
                co_unpk<ZF16>(wz[e], v);
#pragma unroll
                for (int c = 0; c < 8; ++c) { const float t = (v[c] - mu[c]) * ga[c] + be[c]; v[c] = t > 0.f ? t : t * p.slope; }
                co_u32x4 o;
#pragma unroll
                for (int h = 0; h < 4; ++h) o[h] = co_pk<F16>(v[2 * h], v[2 * h + 1]);
                *reinterpret_cast<co_u32x4*>(ob + pos[e] * 8) = o;
                if (yb) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) yb[(size_t)c * p.HW + pos[e]] = v[c];
                }
                if (yh) {           // pieces 0|1 and 2|3 are neighbours in a row: halves c of the two pieces -> one dword
                    if (e & 1) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const unsigned lo = (oprev[c >> 1] >> (16 * (c & 1))) & 0xffffu, hi = (o[c >> 1] >> (16 * (c & 1))) & 0xffffu;
                            *reinterpret_cast<unsigned*>(yh + (size_t)c * p.HW + pos[e - 1]) = lo | (hi << 16);
                        }
                    } else oprev = o;
                }
                float r[8];
                co_unpk<F16>(o, r);              // the stored values: what the pool compares
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (e == 0) best[c] = r[c];
                    else if (r[c] > best[c] || r[c] != r[c]) { best[c] = r[c]; code = (code & ~(3u << (2 * c))) | ((unsigned)e << (2 * c)); }
                }
            }
            co_u32x4 o;
#pragma unroll
            for (int h = 0; h < 4; ++h) o[h] = co_pk<F16>(best[2 * h], best[2 * h + 1]);
            *reinterpret_cast<co_u32x4*>(pb + (size_t)q * 8) = o;
            if (p.parg) p.parg[((size_t)n * p.G8 + g) * oHW + q] = (unsigned short)code;
        }
        return;
    }
    static_assert(AP_PPT == 2, "the 16-bit planes pair a thread's two pixels");
    const int pxb = blockIdx.x * (AP_THREADS * AP_PPT), pstep = yh ? 1 : AP_THREADS;
    const int px0 = pxb + (yh ? 2 * (int)threadIdx.x : (int)threadIdx.x);
    co_u32x4 w[AP_PPT], okeep = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < AP_PPT; ++k) {
        const int px = px0 + pstep * k;
        if (px < p.HW) w[k] = *reinterpret_cast<const co_u32x4*>(zg + (size_t)px * 8);
    }
#pragma unroll
    for (int k = 0; k < AP_PPT; ++k) {
        const int px = px0 + pstep * k;
        if (px >= p.HW) continue;
        float v[8];
        co_unpk<ZF16>(w[k], v);
#pragma unroll
        for (int c = 0; c < 8; ++c) { const float t = (v[c] - mu[c]) * ga[c] + be[c]; v[c] = t > 0.f ? t : t * p.slope; }
        co_u32x4 o;
#pragma unroll
        for (int h = 0; h < 4; ++h) o[h] = co_pk<F16>(v[2 * h], v[2 * h + 1]);
        *reinterpret_cast<co_u32x4*>(ob + (size_t)px * 8) = o;
        if (yb) {
#pragma unroll
            for (int c = 0; c < 8; ++c) yb[(size_t)c * p.HW + px] = v[c];
        }
        if (yh) {           // H*W is even: both pixels of the pair are inside the plane
            if (k == 1) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const unsigned lo = (okeep[c >> 1] >> (16 * (c & 1))) & 0xffffu, hi = (o[c >> 1] >> (16 * (c & 1))) & 0xffffu;
                    *reinterpret_cast<unsigned*>(yh + (size_t)c * p.HW + px - 1) = lo | (hi << 16);
                }
            } else okeep = o;
        }
    }
}

// Backward twin: the gathered dgrad that wrote dy left {sum g, sum g * xhat} per (image, pixel subset, channel)
// (mtbc_conv3x3_args.norm_z).  One wave per (n, c) adds them up (double, fixed order) -> m1 = S1 / HW, m2 = S2 / HW and the
// parameter-gradient partials {S1, S2, 0}; then dz = rstd * gamma * (g - m1 - xhat * m2) is one streaming pass.
__global__ void in_bstats_finalize_kernel(const float* __restrict__ part, int slots, int C, int HW, float* __restrict__ m12,
                                          float* __restrict__ dparam_part, int planes) {
    const int plane = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (plane >= planes) return;
    const int n = plane / C, c = plane % C;
    const float* q = part + ((size_t)n * slots * C + c) * 2;
    double s1 = 0.0, s2 = 0.0;
    for (int t = lane; t < slots; t += 64) { const float2 v = *reinterpret_cast<const float2*>(q + (size_t)t * C * 2); s1 += (double)v.x; s2 += (double)v.y; }
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
    if (lane == 0) {
        m12[2 * plane] = (float)(s1 / (double)HW); m12[2 * plane + 1] = (float)(s2 / (double)HW);
        if (dparam_part) { dparam_part[3 * plane] = (float)s1; dparam_part[3 * plane + 1] = (float)s2; dparam_part[3 * plane + 2] = 0.f; }
    }
}
template <bool F16, bool ZF16>
__global__ __launch_bounds__(AP_THREADS) void in_apply_bwd_c8_kernel(const CoP p, const float* __restrict__ m12) {
    const int item = blockIdx.y, n = item / p.G8, g = item % p.G8;
    float mu[8], rs[8], ga[8], be[8], m1[8], m2[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const size_t pl = (size_t)n * p.C + 8 * g + c;
        mu[c] = p.mean[pl]; rs[c] = p.rstd[pl]; ga[c] = p.gamma ? p.gamma[8 * g + c] : 1.f; be[c] = p.beta ? p.beta[8 * g + c] : 0.f;
        m1[c] = m12[2 * pl]; m2[c] = m12[2 * pl + 1];
    }
    const unsigned short* zg = p.z8 + ((size_t)n * p.G8 + g) * p.HW * 8;
    const unsigned short* dg = p.dy8 + (size_t)n * p.dy8bs + (size_t)g * p.HW * 8;
    unsigned short* ob = p.dz8 + ((size_t)n * p.G8 + g) * p.HW * 8;
    const int px0 = blockIdx.x * (AP_THREADS * AP_PPT) + threadIdx.x;
    co_u32x4 wz[AP_PPT], wd[AP_PPT];
#pragma unroll
    for (int k = 0; k < AP_PPT; ++k) {
        const int px = px0 + AP_THREADS * k;
        if (px < p.HW) { wz[k] = *reinterpret_cast<const co_u32x4*>(zg + (size_t)px * 8); wd[k] = *reinterpret_cast<const co_u32x4*>(dg + (size_t)px * 8); }
    }
#pragma unroll
    for (int k = 0; k < AP_PPT; ++k) {
        const int px = px0 + AP_THREADS * k;
        if (px >= p.HW) continue;
        float z[8], d[8];
        co_unpk<ZF16>(wz[k], z);
        co_unpk<F16>(wd[k], d);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float xh = (z[c] - mu[c]) * rs[c];
            const float gg = d[c] * ((xh * ga[c] + be[c]) > 0.f ? 1.f : p.slope);
            d[c] = rs[c] * ga[c] * (gg - m1[c] - xh * m2[c]);
        }
        co_u32x4 o;
#pragma unroll
        for (int h = 0; h < 4; ++h) o[h] = co_pk<F16>(d[2 * h], d[2 * h + 1]);
        *reinterpret_cast<co_u32x4*>(ob + (size_t)px * 8) = o;
    }
}

// the one-output 1x1 head's weight / bias gradient from the per-member partials the InstanceNorm backward left: one wave per
// channel (the last wave: the bias), lanes over (image, member), fixed butterfly
__global__ void in_r1_finalize_kernel(const float* __restrict__ dwp, const float* __restrict__ dbp, float* __restrict__ dw, float* __restrict__ db,
                                      int N, int C, int T, int accumulate) {
    const int c = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    if (c < C) { for (int i = lane; i < N * T; i += 64) s += dwp[((size_t)(i / T) * C + c) * T + i % T]; }
    else { for (int i = lane; i < N * T; i += 64) s += dbp[i]; }
    s = wave_sum(s);
    if (lane != 0) return;
    if (c < C) dw[c] = accumulate ? dw[c] + s : s;
    else if (db) db[0] = accumulate ? db[0] + s : s;
}

struct CoPlan { bool ok; int ppt, T, grid, nteams; };
template <typename K> int resident_blocks(K kernel) {
    int per_cu = 0, dev = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, CO_THREADS, 0) != hipSuccess || per_cu < 1) return 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    // The occupancy query is advisory: for SGPR-heavy kernels it has been seen one block per CU high where it answers the
    // wave-slot maximum (MI355X_MICROARCH.md, Residency; 512-thread workgroups: 4 per CU).  A VGPR-limited answer (2 or 3
    // per CU) is exact, so only the maximum is lowered by one.
    if (per_cu > 3) per_cu = 3;
    return per_cu * prop.multiProcessorCount;
}
int device_cus() {           // of the CURRENT device, asked once per device (common.h: per-device caches, not per-process ones)
    static DevInts cache;
    return mtbc_per_device(cache, [] { int dev = 0; hipDeviceProp_t prop; return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 0; });
}
// capacity with `reserve` of the device's CUs left to kernels of other streams (a per-call argument: no process state)
int usable(int cap_all_cus, int reserve) {
    const int cus = device_cus();
    if (reserve <= 0 || cus <= 0) return cap_all_cus;
    const int keep = cus - reserve;
    return keep < 8 ? cap_all_cus / cus * 8 : cap_all_cus / cus * keep;
}
// team size / pixels per thread: the most even split of `items` over the resident teams, larger slabs on ties
CoPlan plan_coop(int items, int HW, int max_ppt, const int* cap_by_ppt, int reserve) {
    CoPlan best{false, 0, 0, 0, 0};
    double best_eff = -1.0;
    for (int ppt = max_ppt; ppt >= 1; ppt >>= 1) {
        int T;
        if (HW <= CO_THREADS * ppt) { if (ppt > 1 && HW <= CO_THREADS * (ppt / 2)) continue; T = 1; }
        else { if (HW % (CO_THREADS * ppt)) continue; T = HW / (CO_THREADS * ppt); }
        if (T > CO_MAXT) continue;
        const int cap = usable(cap_by_ppt[ppt], reserve);
        if (cap < T) continue;
        int teams = cap / T;
        if (teams > items) teams = items;
        if (teams > CO_MAX_TEAMS) teams = CO_MAX_TEAMS;
        const int rounds = (items + teams - 1) / teams;
        if (2 * rounds >= 1000) continue;                 // exchange numbers live in 10 bits
        const double eff = (double)items / ((double)teams * rounds);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = CoPlan{true, ppt, T, teams * T, teams}; }
    }
    return best;
}
int fill_coop(const mtbc_instnorm_args* a, CoP* p) {
    if (!a || a->N <= 0 || a->C <= 0 || a->H <= 0 || a->W <= 0) return MTBC_E_BADSHAPE;
    if (a->C % 8 || (a->out16_type != 1 && a->out16_type != 2)) return MTBC_E_BADARG;
    if ((a->z_layout != MTBC_LAYOUT_PLANAR && a->z_layout != MTBC_LAYOUT_C8) || (a->dy_layout != MTBC_LAYOUT_PLANAR && a->dy_layout != MTBC_LAYOUT_C8)) return MTBC_E_BADARG;
    p->N = a->N; p->C = a->C; p->HW = a->H * a->W; p->G8 = a->C / 8; p->items = a->N * p->G8; p->f16 = a->out16_type == 2;
    p->eps = a->eps; p->slope = a->slope; p->z = a->z; p->gamma = a->gamma; p->beta = a->beta; p->y = a->y; p->ybs = a->y_batch_stride;
    p->y8 = reinterpret_cast<unsigned short*>(a->y8); p->mean = a->mean; p->rstd = a->rstd; p->dy = a->dy; p->dybs = a->dy_batch_stride;
    p->dz8 = reinterpret_cast<unsigned short*>(a->dz8); p->part = nullptr; p->part3 = nullptr; p->state = a->coop_state;
    p->z8 = nullptr; p->dy8 = nullptr; p->dy8bs = 0; p->dyx = nullptr;
    if (a->z_type != 0 && a->z_type != a->out16_type && !(a->z_type == 2 && a->z_layout == MTBC_LAYOUT_C8)) return MTBC_E_BADARG;
    p->zf16 = (a->out16_type == 2 || a->z_type == 2) ? 1 : 0;
    p->r1 = a->dy_rank1; p->r1w = a->dy_rank1_w; p->r1dw = nullptr; p->r1db = nullptr;
    if ((p->r1 == nullptr) != (p->r1w == nullptr)) return MTBC_E_BADARG;
    p->pg = a->dy_pool; p->pa = reinterpret_cast<const unsigned short*>(a->dy_pool_arg); p->W = a->W;
    p->py8 = reinterpret_cast<unsigned short*>(a->pool_y8); p->parg = reinterpret_cast<unsigned short*>(a->pool_arg);
    p->y16 = reinterpret_cast<unsigned short*>(a->y16);
    if (p->y16) { p->y = nullptr; if ((a->H * a->W) % 8 || a->y_batch_stride % 8 || (reinterpret_cast<uintptr_t>(a->y16) & 15)) return MTBC_E_BADARG; }
    if ((p->pg == nullptr) != (p->pa == nullptr) || (p->pg && ((a->H | a->W) & 1))) return MTBC_E_BADARG;
    if (a->z_layout == MTBC_LAYOUT_C8) {
        if (reinterpret_cast<uintptr_t>(a->z) & 15) return MTBC_E_BADARG;
        p->z8 = reinterpret_cast<const unsigned short*>(a->z); p->z = nullptr;
    }
    return MTBC_OK;
}
// One workgroup per (n, channel group): planes up to 64 x 64
constexpr int SOLO_MAX_HW = 4096;
// Resident capacity per kernel instantiation (the bf16 and fp16 variants have their own register counts): one occupancy
// query each PER DEVICE, cached in a per-device slot (immutable once asked).
template <auto K> int cap_of() { static DevInts cache; return mtbc_per_device(cache, [] { return resident_blocks(K); }); }
// the variant a launch runs: (output type, z layout, dy layout)
struct Var { bool f16; int zc8, dy8; };
template <bool BWD, int THREADS, int PPT, bool COOP, bool F16, int ZC8, int DY8> constexpr auto kernel_of() {
    if constexpr (BWD) return &in_bwd_c8_kernel<THREADS, PPT, F16, COOP, ZC8, DY8>;
    else return &in_fwd_c8_kernel<THREADS, PPT, F16, COOP, ZC8>;
}
// calls f(kernel pointer as a compile-time constant) for the variant `v`
template <bool BWD, int THREADS, int PPT, bool COOP, typename F> auto with_kernel(const Var& v, F&& f) {
#define MTBC_V(F16_, Z_, D_) f(std::integral_constant<decltype(kernel_of<BWD, THREADS, PPT, COOP, F16_, Z_, D_>()), kernel_of<BWD, THREADS, PPT, COOP, F16_, Z_, D_>()>{})
    const int d = BWD ? v.dy8 : 0;
#define MTBC_VD(F16_, Z_) (d == 2 ? MTBC_V(F16_, Z_, 2) : d == 1 ? MTBC_V(F16_, Z_, 1) : MTBC_V(F16_, Z_, 0))
    if (v.f16) return v.zc8 ? MTBC_VD(true, 1) : MTBC_VD(true, 0);
    return v.zc8 == 2 ? MTBC_VD(false, 2) : v.zc8 == 1 ? MTBC_VD(false, 1) : MTBC_VD(false, 0);
#undef MTBC_VD
#undef MTBC_V
}
template <bool BWD, int PPT> int cap_ppt(const Var& v) { return with_kernel<BWD, CO_THREADS, PPT, true>(v, [](auto k) { return cap_of<decltype(k)::value>(); }); }
template <bool BWD> CoPlan plan_team(int items, int HW, const Var& v, int reserve) {
    int cap[9] = {0};
    cap[1] = cap_ppt<BWD, 1>(v); cap[2] = cap_ppt<BWD, 2>(v); cap[4] = cap_ppt<BWD, 4>(v);
    // (a backward that keeps its slab packed as stored -- 8 pixels per thread, half the team size and rounds -- was built: the
    //  compiler spills ~150 VGPRs at 4 waves/SIMD; with 4 pixels per thread it is no faster than this one, nor is starting the
    //  odd teams late so that their loads fall into the even teams' exchanges: tools/experiments/in_bwd_probe.sh)
    return plan_coop(items, HW, 4, cap, reserve);
}
Var var_of(const mtbc_instnorm_args* a) { return Var{a->out16_type == 2, a->z_layout == MTBC_LAYOUT_C8 ? ((a->z_type == 2 && a->out16_type == 1) ? 2 : 1) : 0, a->dy_layout == MTBC_LAYOUT_C8 ? (a->n_dy_extra ? 2 : 1) : 0}; }

template <bool BWD, int THREADS, int PPT, bool COOP>
void launch_c8(const CoP& p0, int grid, hipStream_t st) {
    CoP p = p0;
    const Var v{p.f16 != 0, p.z8 ? (p.zf16 && !p.f16 ? 2 : 1) : 0, p.dy8 ? (p.dyx ? 2 : 1) : 0};      // (planar dy: a second planar contribution is a runtime branch)
#ifdef MTBC_PROBES
    // MTBC_INB_TS=1: phase timestamps of every workgroup (thread 0) of the channel-group backward, printed after the launch
    static const int ts_env = mtbc_probe_int("MTBC_INB_TS", 0);
    static unsigned long long* dts = nullptr;
    p.ts = nullptr;
    if (BWD && ts_env && grid <= 4096) {
        if (!dts) (void)hipMalloc(&dts, 4096 * 16 * sizeof(unsigned long long));
        (void)hipMemsetAsync(dts, 0, (size_t)grid * 16 * sizeof(unsigned long long), st);
        p.ts = dts;
    }
#endif
    with_kernel<BWD, THREADS, PPT, COOP>(v, [&](auto k) { hipLaunchKernelGGL(decltype(k)::value, dim3(grid), dim3(THREADS), 0, st, p); return 0; });
#ifdef MTBC_PROBES
    if (p.ts) {
        (void)hipStreamSynchronize(st);
        static unsigned long long hts[4096 * 16];
        (void)hipMemcpy(hts, dts, (size_t)grid * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < grid; ++b) if (hts[b * 16] && hts[b * 16] < t0) t0 = hts[b * 16];
        double mean[16] = {0}; int cnt[16] = {0};
        for (int b = 0; b < grid; ++b)
            for (int k = 0; k < 16; ++k) if (hts[b * 16 + k]) { mean[k] += (double)(hts[b * 16 + k] - t0) * 0.01; ++cnt[k]; }
        for (int k = 0; k < 16; ++k) if (cnt[k]) mean[k] /= cnt[k];
        fprintf(stderr, "inb_ts C%d HW%d T%d coop%d threads%d ppt%d grid %d items %d | mean us since the first workgroup's start:", p.C, p.HW, p.T, (int)COOP, THREADS, PPT, grid, p.items);
        for (int k = 0; k < 2; ++k)
            fprintf(stderr, " item%d: start %.2f loads issued %.2f constants %.2f data + thread sums %.2f block sums %.2f exchange %.2f stored %.2f |", k,
                    mean[7 * k], mean[7 * k + 1], mean[7 * k + 2], mean[7 * k + 3], mean[7 * k + 4], mean[7 * k + 5], mean[7 * k + 6]);
        fprintf(stderr, " loop done %.2f finished %.2f\n", mean[14], mean[15]);
    }
#endif
}
// one workgroup per item, sized to the plane
template <bool BWD> void launch_solo(CoP& p, hipStream_t st) {
    p.T = 1; p.nteams = p.items;
    if (p.HW <= 64) launch_c8<BWD, 64, 1, false>(p, p.items, st);
    else if (p.HW <= 256) launch_c8<BWD, 256, 1, false>(p, p.items, st);
    else if (p.HW <= 1024) launch_c8<BWD, 256, 4, false>(p, p.items, st);
    else launch_c8<BWD, 1024, 4, false>(p, p.items, st);
}
template <bool BWD> void launch_team(CoP& p, const CoPlan& pl, hipStream_t st) {
    p.T = pl.T; p.nteams = pl.nteams;
    if (pl.ppt == 4) launch_c8<BWD, CO_THREADS, 4, true>(p, pl.grid, st);
    else if (pl.ppt == 2) launch_c8<BWD, CO_THREADS, 2, true>(p, pl.grid, st);
    else launch_c8<BWD, CO_THREADS, 1, true>(p, pl.grid, st);
}

}  // namespace

extern "C" {

size_t mtbc_instnorm_coop_error_offset(void) { return offsetof(CoopHdr, err); }

size_t mtbc_instnorm_coop_state_bytes(void) { return CO_MAILBOX_OFF + (size_t)CO_MAX_TEAMS * CO_TEAM_WORDS * sizeof(unsigned long long); }

int mtbc_instnorm_c8_supported(const mtbc_instnorm_args* a, int32_t backward) {
    if (!a || a->N <= 0 || a->C <= 0 || a->H <= 0 || a->W <= 0 || a->C % 8) return 0;
    if (backward && a->n_dy_extra != 0 && a->n_dy_extra != 1) return 0;
    if (a->H * a->W <= SOLO_MAX_HW) return 1;
    const CoPlan pl = backward ? plan_team<true>(a->N * (a->C / 8), a->H * a->W, var_of(a), a->coop_reserve_cus)
                               : plan_team<false>(a->N * (a->C / 8), a->H * a->W, var_of(a), a->coop_reserve_cus);
    return pl.ok ? 1 : 0;
}

int mtbc_i_instnorm_fwd_c8(const mtbc_instnorm_args* a, hipStream_t st) {
    CoP p; int rc = fill_coop(a, &p); if (rc) return rc;
    if ((!p.z && !p.z8) || !p.y8 || !p.mean || !p.rstd || (reinterpret_cast<uintptr_t>(p.y8) & 15)) return MTBC_E_BADARG;
    if (p.y16 && !a->stats_partial) return MTBC_E_UNSUPPORTED;       // 16-bit planes beside y8: the streaming pass only
    if (a->stats_partial) {         // statistics from the conv epilogue: finalize (one wave per plane) + one streaming pass
        if (!p.z8 || a->stats_slots <= 0) return MTBC_E_BADARG;
        const int planes = a->N * a->C;
        if (p.py8) {           // the activation's 2 x 2 max-pool written by the same pass (one thread per window)
            if (((a->H | a->W) & 1) || (reinterpret_cast<uintptr_t>(p.py8) & 15)) return MTBC_E_BADARG;
            int gx = cdiv(p.HW / 4, AP_THREADS * 2); if (gx < 1) gx = 1;
            const dim3 gp(gx, p.items);
            const bool fin = a->stats_slots <= 64;
            if (!fin) {
                hipLaunchKernelGGL(in_stats_finalize_kernel, dim3(cdiv(planes, 4)), dim3(256), 0, st, a->stats_partial, a->stats_slots, a->C, p.HW, a->eps, a->mean, a->rstd, planes);
                MTBC_CHECK_LAUNCH();
            }
#define MTBC_AP(F16_, FIN_, ZF_) hipLaunchKernelGGL((in_apply_fwd_c8_kernel<F16_, FIN_, ZF_, true>), gp, dim3(AP_THREADS), 0, st, p, a->stats_partial, a->stats_slots)
            if (p.f16) { if (fin) MTBC_AP(true, true, true); else MTBC_AP(true, false, true); }
            else if (p.zf16) { if (fin) MTBC_AP(false, true, true); else MTBC_AP(false, false, true); }
            else { if (fin) MTBC_AP(false, true, false); else MTBC_AP(false, false, false); }
#undef MTBC_AP
            MTBC_CHECK_LAUNCH();
            return MTBC_OK;
        }
        const dim3 g(cdiv(p.HW, AP_THREADS * AP_PPT), p.items);
        if (a->stats_slots <= 64) {          // few subsets per plane: every workgroup finalizes its own 8 channels (one launch)
            if (p.f16) hipLaunchKernelGGL((in_apply_fwd_c8_kernel<true, true, true>), g, dim3(AP_THREADS), 0, st, p, a->stats_partial, a->stats_slots);
            else if (p.zf16) hipLaunchKernelGGL((in_apply_fwd_c8_kernel<false, true, true>), g, dim3(AP_THREADS), 0, st, p, a->stats_partial, a->stats_slots);
            else hipLaunchKernelGGL((in_apply_fwd_c8_kernel<false, true, false>), g, dim3(AP_THREADS), 0, st, p, a->stats_partial, a->stats_slots);
            MTBC_CHECK_LAUNCH();
            return MTBC_OK;
        }
        hipLaunchKernelGGL(in_stats_finalize_kernel, dim3(cdiv(planes, 4)), dim3(256), 0, st, a->stats_partial, a->stats_slots, a->C, p.HW, a->eps, a->mean, a->rstd, planes);
        MTBC_CHECK_LAUNCH();
        if (p.f16) hipLaunchKernelGGL((in_apply_fwd_c8_kernel<true, false, true>), g, dim3(AP_THREADS), 0, st, p, a->stats_partial, a->stats_slots);
        else if (p.zf16) hipLaunchKernelGGL((in_apply_fwd_c8_kernel<false, false, true>), g, dim3(AP_THREADS), 0, st, p, a->stats_partial, a->stats_slots);
        else hipLaunchKernelGGL((in_apply_fwd_c8_kernel<false, false, false>), g, dim3(AP_THREADS), 0, st, p, a->stats_partial, a->stats_slots);
        MTBC_CHECK_LAUNCH();
        return MTBC_OK;
    }
    if (p.HW <= SOLO_MAX_HW) { launch_solo<false>(p, st); MTBC_CHECK_LAUNCH(); return MTBC_OK; }
    if (!p.state) return MTBC_E_BADARG;
    const CoPlan pl = plan_team<false>(p.items, p.HW, var_of(a), a->coop_reserve_cus);
    if (!pl.ok) return MTBC_E_UNSUPPORTED;
    launch_team<false>(p, pl, st);
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}

// members per team of the backward launch (the parameter-gradient partials are [N*C][3] + [N*C][T] floats)
int mtbc_i_instnorm_bwd_c8_team(const mtbc_instnorm_args* a) {
    if (!a || a->N <= 0 || a->C <= 0 || a->C % 8) return 0;
    if (a->H * a->W <= SOLO_MAX_HW) return 1;
    const CoPlan pl = plan_team<true>(a->N * (a->C / 8), a->H * a->W, var_of(a), a->coop_reserve_cus);
    return pl.ok ? pl.T : 0;
}
int mtbc_i_instnorm_bwd_c8(const mtbc_instnorm_args* a, float* part, hipStream_t st) {
    CoP p; int rc = fill_coop(a, &p); if (rc) return rc;
    if ((!p.z && !p.z8) || (!p.dy && !p.r1 && !p.pg) || !p.dz8 || !p.mean || !p.rstd || (reinterpret_cast<uintptr_t>(p.dz8) & 15)) return MTBC_E_BADARG;
    if (!p.dy && a->dy_layout == MTBC_LAYOUT_C8) return MTBC_E_BADARG;
    if ((p.r1 || p.pg) && a->stats_partial) return MTBC_E_UNSUPPORTED;
    if (a->dy_layout == MTBC_LAYOUT_C8) {
        if ((reinterpret_cast<uintptr_t>(a->dy) & 15) || a->dy_batch_stride % 8 || a->n_dy_extra < 0 || a->n_dy_extra > 1) return MTBC_E_BADARG;
        p.dy8 = reinterpret_cast<const unsigned short*>(a->dy); p.dy8bs = a->dy_batch_stride; p.dy = nullptr;
        if (a->n_dy_extra == 1) { if (!a->dy_extra[0]) return MTBC_E_BADARG; p.dyx = a->dy_extra[0]; }
    } else if (a->n_dy_extra == 1 && p.dy) {        // fp32 planar dy + one more fp32 planar contribution (batch stride C*H*W)
        if (!a->dy_extra[0]) return MTBC_E_BADARG;
        p.dyx = a->dy_extra[0];
    } else if (a->n_dy_extra != 0) return MTBC_E_BADARG;
    p.part = part;
    const bool head_dw = a->dy_rank1 && a->dy_rank1_dw;
    const int planes_ = a->N * a->C;
    int T_ = 1;
    if (p.HW > SOLO_MAX_HW) { const CoPlan pl0 = plan_team<true>(p.items, p.HW, var_of(a), a->coop_reserve_cus); if (!pl0.ok) return MTBC_E_UNSUPPORTED; T_ = pl0.T; }
    if (head_dw) {       // partials behind the parameter-gradient regions of the workspace: [3 * planes][planes * T] | [planes * T][N * T]
        const size_t need = ((size_t)planes_ * (3 + 2 * T_) + (size_t)a->N * T_) * sizeof(float);
        if (!a->workspace || a->workspace_bytes < need || a->stats_partial) return MTBC_E_WORKSPACE;
        p.r1dw = reinterpret_cast<float*>(a->workspace) + (size_t)planes_ * (3 + T_);
        p.r1db = p.r1dw + (size_t)planes_ * T_;
    }
    if (a->stats_partial) {         // {sum g, sum g * xhat} from the gathered dgrad's epilogue: finalize + one streaming pass
        if (!p.z8 || !p.dy8 || p.dyx || a->stats_slots <= 0 || !a->workspace || a->workspace_bytes < (size_t)a->N * a->C * 5 * sizeof(float)) return MTBC_E_BADARG;
        const int planes = a->N * a->C;
        float* m12 = reinterpret_cast<float*>(a->workspace) + (size_t)3 * planes;
        hipLaunchKernelGGL(in_bstats_finalize_kernel, dim3(cdiv(planes, 4)), dim3(256), 0, st, a->stats_partial, a->stats_slots, a->C, p.HW, m12, part, planes);
        MTBC_CHECK_LAUNCH();
        const dim3 g(cdiv(p.HW, AP_THREADS * AP_PPT), p.items);
        if (p.f16) hipLaunchKernelGGL((in_apply_bwd_c8_kernel<true, true>), g, dim3(AP_THREADS), 0, st, p, m12);
        else if (p.zf16) hipLaunchKernelGGL((in_apply_bwd_c8_kernel<false, true>), g, dim3(AP_THREADS), 0, st, p, m12);
        else hipLaunchKernelGGL((in_apply_bwd_c8_kernel<false, false>), g, dim3(AP_THREADS), 0, st, p, m12);
        MTBC_CHECK_LAUNCH();
        return MTBC_OK;
    }
    if (p.HW <= SOLO_MAX_HW) {
        p.part3 = part ? part + (size_t)3 * a->N * a->C : nullptr;
        launch_solo<true>(p, st); MTBC_CHECK_LAUNCH();
    } else {
        if (!p.state) return MTBC_E_BADARG;
        const CoPlan pl = plan_team<true>(p.items, p.HW, var_of(a), a->coop_reserve_cus);
        if (!pl.ok) return MTBC_E_UNSUPPORTED;
        p.part3 = part ? part + (size_t)3 * a->N * a->C : nullptr;
        launch_team<true>(p, pl, st);
        MTBC_CHECK_LAUNCH();
    }
    if (head_dw) {
        hipLaunchKernelGGL(in_r1_finalize_kernel, dim3(a->C + 1), dim3(64), 0, st, p.r1dw, p.r1db, a->dy_rank1_dw, a->dy_rank1_db, a->N, a->C, T_, a->dy_rank1_accumulate);
        MTBC_CHECK_LAUNCH();
    }
    return MTBC_OK;
}

}  // extern "C"
