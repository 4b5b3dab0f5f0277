// InstanceNorm + LeakyReLU whose output IS the 16-bit channel-blocked layout the 3x3 convs read (MTBC_LAYOUT_C8):
// forward  z (fp32 planes) -> a8  [n][C/8][H*W][8]   (+ fp32 planes y when something else reads them)
// backward z, dy (fp32 planes) -> dz8 [n][C/8][H*W][8]
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
// tag = (epoch + 1) << 10 | exchange number; `epoch` lives in the caller's persistent state block, is read by every
// workgroup when it starts and advanced by the LAST workgroup to finish (which thereby knows everybody has read it).
// Two mailbox slots by exchange parity: a member can post exchange s+2 only after all members posted s+1, i.e. after
// they finished reading s.  Polls are bounded (CO_SPIN_MAX): a protocol failure sets state->err and produces garbage,
// never a hang.  Requires all workgroups of a team to be resident: the grid is sized from the occupancy query.
#include "common.h"

namespace {

constexpr int CO_THREADS = 512, CO_WAVES = CO_THREADS / 64;
constexpr int CO_MAXT = 32, CO_NV = 16;
constexpr unsigned CO_SPIN_MAX = 1u << 22;
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
    float* part;                             // backward: [N*C][3] = {sum g, sum g*xh, sum dz} or nullptr
    void* state;
};

typedef float co_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 co_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 co_f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned co_u32x4 __attribute__((ext_vector_type(4)));
template <bool F16> __device__ __forceinline__ unsigned co_pk(float a, float b) {
    if constexpr (F16) return __builtin_bit_cast(unsigned, __builtin_convertvector((co_f32x2){a, b}, co_f16x2));
    else return __builtin_bit_cast(unsigned, __builtin_convertvector((co_f32x2){a, b}, co_bf16x2));
}

__device__ __forceinline__ void mb_post(unsigned long long* slot, float v, unsigned tag) {
    __hip_atomic_store(slot, ((unsigned long long)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float mb_wait(unsigned long long* slot, unsigned tag, CoopHdr* hdr) {
    unsigned long long w = 0;
    unsigned spins = 0;
    for (;;) {
        w = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(w >> 32) == tag) break;
        if (++spins >= CO_SPIN_MAX) { __hip_atomic_store(&hdr->err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        __builtin_amdgcn_s_sleep(2);
    }
    return __uint_as_float((unsigned)w);
}
// Every member posts its NV values (thread i < NV holds value i in `mine`) and collects everybody's into xch[m][i].
template <int NV>
__device__ __forceinline__ void team_exchange(float mine, unsigned long long* mb, int member, int T, unsigned& seq, unsigned epoch,
                                              CoopHdr* hdr, float (*xch)[CO_NV]) {
    const int tid = threadIdx.x;
    const unsigned tag = ((epoch + 1u) << 10) | (seq & 1023u);
    unsigned long long* base = mb + (size_t)(seq & 1u) * CO_MAXT * CO_NV;
    if (tid < NV) mb_post(base + member * CO_NV + tid, mine, tag);
    const int m = tid >> 4, i = tid & 15;
    if (m < T && i < NV) xch[m][i] = mb_wait(base + m * CO_NV + i, tag, hdr);
    ++seq;
    __syncthreads();
}
// sums of 8 per-thread values over the workgroup, result in every thread (fixed order: deterministic)
__device__ __forceinline__ void block_reduce8(float (&s)[8], float (*red)[8]) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 8; ++c) s[c] = wave_sum(s[c]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c) red[wv][c] = s[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < CO_WAVES; ++w) t += red[w][c];
        s[c] = t;
    }
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

template <int PPT, bool F16>
__global__ __launch_bounds__(CO_THREADS, 2) void in_fwd_coop_kernel(const CoP p) {
    __shared__ float red[CO_WAVES][8];
    __shared__ float xch[CO_MAXT][CO_NV];
    __shared__ float stat[16];
    const int tid = threadIdx.x;
    CoopHdr* hdr = reinterpret_cast<CoopHdr*>(p.state);
    const unsigned epoch = __hip_atomic_load(&hdr->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int team = blockIdx.x / p.T, member = blockIdx.x % p.T;
    unsigned long long* mb = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(p.state) + CO_MAILBOX_OFF) + (size_t)team * CO_TEAM_WORDS;
    const int slab = p.HW / p.T;
    unsigned seq = 0;
    for (int item = team; item < p.items; item += p.nteams) {
        const int n = item / p.G8, g = item % p.G8;
        const float* zb = p.z + ((size_t)n * p.C + 8 * g) * p.HW + (size_t)member * slab;
        float v[8][PPT];
        float s[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) s[c] = 0.f;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int px = tid + CO_THREADS * k;
            const bool ok = px < slab;
#pragma unroll
            for (int c = 0; c < 8; ++c) { v[c][k] = ok ? zb[(size_t)c * p.HW + px] : 0.f; s[c] += v[c][k]; }
        }
        block_reduce8(s, red);
        float lm[8], q[8];
        const float cnt = (float)slab;
#pragma unroll
        for (int c = 0; c < 8; ++c) { lm[c] = s[c] / cnt; q[c] = 0.f; }
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const bool ok = tid + CO_THREADS * k < slab;
#pragma unroll
            for (int c = 0; c < 8; ++c) { const float d = v[c][k] - lm[c]; q[c] += ok ? d * d : 0.f; }
        }
        block_reduce8(q, red);
        float mean[8], rstd[8];
        if (p.T == 1) {
#pragma unroll
            for (int c = 0; c < 8; ++c) { mean[c] = lm[c]; rstd[c] = 1.0f / sqrtf(q[c] / (float)p.HW + p.eps); }
        } else {
            float mine = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) { if (tid == c) mine = lm[c]; if (tid == 8 + c) mine = q[c]; }
            team_exchange<16>(mine, mb, member, p.T, seq, epoch, hdr, xch);
            if (tid < 8) {                       // Chan's combination of the members' (count, mean, M2), fixed order
                float nt = 0.f, mu = 0.f, m2 = 0.f;
                for (int m = 0; m < p.T; ++m) {
                    const float d = xch[m][tid] - mu, nn = nt + cnt;
                    mu += d * (cnt / nn);
                    m2 += xch[m][8 + tid] + d * d * (nt * cnt / nn);
                    nt = nn;
                }
                stat[tid] = mu; stat[8 + tid] = 1.0f / sqrtf(m2 / (float)p.HW + p.eps);
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 8; ++c) { mean[c] = stat[c]; rstd[c] = stat[8 + c]; }
        }
        if (member == 0 && tid < 8) {
#pragma unroll
            for (int c = 0; c < 8; ++c) if (tid == c) { p.mean[(size_t)n * p.C + 8 * g + c] = mean[c]; p.rstd[(size_t)n * p.C + 8 * g + c] = rstd[c]; }
        }
        float ga[8], be[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) { ga[c] = (p.gamma ? p.gamma[8 * g + c] : 1.f) * rstd[c]; be[c] = p.beta ? p.beta[8 * g + c] : 0.f; }
        unsigned short* ob = p.y8 + (((size_t)n * p.G8 + g) * p.HW + (size_t)member * slab) * 8;
        float* yb = p.y ? p.y + (size_t)n * p.ybs + (size_t)(8 * g) * p.HW + (size_t)member * slab : nullptr;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int px = tid + CO_THREADS * k;
            if (px < slab) {
                float o[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) { const float t = (v[c][k] - mean[c]) * ga[c] + be[c]; o[c] = t > 0.f ? t : t * p.slope; }
                co_u32x4 w;
#pragma unroll
                for (int h = 0; h < 4; ++h) w[h] = co_pk<F16>(o[2 * h], o[2 * h + 1]);
                *reinterpret_cast<co_u32x4*>(ob + (size_t)px * 8) = w;
                if (yb) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) yb[(size_t)c * p.HW + px] = o[c];
                }
            }
        }
    }
    coop_finish(hdr, epoch);
}

template <int PPT, bool F16>
__global__ __launch_bounds__(CO_THREADS, 2) void in_bwd_coop_kernel(const CoP p) {
    __shared__ float red[CO_WAVES][8];
    __shared__ float xch[CO_MAXT][CO_NV];
    __shared__ float stat[16];
    const int tid = threadIdx.x;
    CoopHdr* hdr = reinterpret_cast<CoopHdr*>(p.state);
    const unsigned epoch = __hip_atomic_load(&hdr->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int team = blockIdx.x / p.T, member = blockIdx.x % p.T;
    unsigned long long* mb = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(p.state) + CO_MAILBOX_OFF) + (size_t)team * CO_TEAM_WORDS;
    const int slab = p.HW / p.T;
    unsigned seq = 0;
    for (int item = team; item < p.items; item += p.nteams) {
        const int n = item / p.G8, g = item % p.G8;
        const size_t plane0 = (size_t)n * p.C + 8 * g;
        const float* zb = p.z + plane0 * p.HW + (size_t)member * slab;
        const float* gb = p.dy + (size_t)n * p.dybs + (size_t)(8 * g) * p.HW + (size_t)member * slab;
        float mean[8], rstd[8], ga[8], be[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            mean[c] = p.mean[plane0 + c]; rstd[c] = p.rstd[plane0 + c];
            ga[c] = p.gamma ? p.gamma[8 * g + c] : 1.f; be[c] = p.beta ? p.beta[8 * g + c] : 0.f;
        }
        float xh[8][PPT], gy[8][PPT];
        float s1[8], s2[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) { s1[c] = 0.f; s2[c] = 0.f; }
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int px = tid + CO_THREADS * k;
            const bool ok = px < slab;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float zv = ok ? zb[(size_t)c * p.HW + px] : mean[c], dv = ok ? gb[(size_t)c * p.HW + px] : 0.f;
                const float x = (zv - mean[c]) * rstd[c];
                const float y = dv * ((x * ga[c] + be[c]) > 0.f ? 1.f : p.slope);
                xh[c][k] = x; gy[c][k] = y;
                s1[c] += y; s2[c] += y * x;
            }
        }
        block_reduce8(s1, red);
        block_reduce8(s2, red);
        float S1[8], S2[8];
        if (p.T == 1) {
#pragma unroll
            for (int c = 0; c < 8; ++c) { S1[c] = s1[c]; S2[c] = s2[c]; }
        } else {
            float mine = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) { if (tid == c) mine = s1[c]; if (tid == 8 + c) mine = s2[c]; }
            team_exchange<16>(mine, mb, member, p.T, seq, epoch, hdr, xch);
            if (tid < 16) { float t = 0.f; for (int m = 0; m < p.T; ++m) t += xch[m][tid]; stat[tid] = t; }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 8; ++c) { S1[c] = stat[c]; S2[c] = stat[8 + c]; }
        }
        unsigned short* ob = p.dz8 + (((size_t)n * p.G8 + g) * p.HW + (size_t)member * slab) * 8;
        float s3[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) s3[c] = 0.f;
        const float inv = 1.0f / (float)p.HW;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int px = tid + CO_THREADS * k;
            if (px < slab) {
                float o[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    o[c] = rstd[c] * ga[c] * (gy[c][k] - S1[c] * inv - xh[c][k] * (S2[c] * inv));
                    s3[c] += o[c];
                }
                co_u32x4 w;
#pragma unroll
                for (int h = 0; h < 4; ++h) w[h] = co_pk<F16>(o[2 * h], o[2 * h + 1]);
                *reinterpret_cast<co_u32x4*>(ob + (size_t)px * 8) = w;
            }
        }
        if (p.part) {                            // {sum g, sum g*xh, sum dz} per plane for the parameter gradients
            block_reduce8(s3, red);
            float S3[8];
            if (p.T == 1) {
#pragma unroll
                for (int c = 0; c < 8; ++c) S3[c] = s3[c];
            } else {
                float mine = 0.f;
#pragma unroll
                for (int c = 0; c < 8; ++c) if (tid == c) mine = s3[c];
                team_exchange<8>(mine, mb, member, p.T, seq, epoch, hdr, xch);
                if (tid < 8) { float t = 0.f; for (int m = 0; m < p.T; ++m) t += xch[m][tid]; stat[tid] = t; }
                __syncthreads();
#pragma unroll
                for (int c = 0; c < 8; ++c) S3[c] = stat[c];
            }
            if (member == 0 && tid < 8) {
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (tid == c) { float* q = p.part + 3 * (plane0 + c); q[0] = S1[c]; q[1] = S2[c]; q[2] = S3[c]; }
            }
        }
    }
    coop_finish(hdr, epoch);
}

struct CoPlan { bool ok; int ppt, T, grid, nteams; };
template <typename K> int resident_blocks(K kernel) {
    int per_cu = 0, dev = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, CO_THREADS, 0) != hipSuccess || per_cu < 1) return 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    return per_cu * prop.multiProcessorCount;
}
// team size / pixels per thread: the most even split of `items` over the resident teams, larger slabs on ties
CoPlan plan_coop(int items, int HW, int max_ppt, const int* cap_by_ppt) {
    CoPlan best{false, 0, 0, 0, 0};
    double best_eff = -1.0;
    for (int ppt = max_ppt; ppt >= 1; ppt >>= 1) {
        int T;
        if (HW <= CO_THREADS * ppt) { if (ppt > 1 && HW <= CO_THREADS * (ppt / 2)) continue; T = 1; }
        else { if (HW % (CO_THREADS * ppt)) continue; T = HW / (CO_THREADS * ppt); }
        if (T > CO_MAXT) continue;
        const int cap = cap_by_ppt[ppt];
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
    if (a->C % 8 || (a->out16_type != 1 && a->out16_type != 2) || !a->coop_state) return MTBC_E_BADARG;
    p->N = a->N; p->C = a->C; p->HW = a->H * a->W; p->G8 = a->C / 8; p->items = a->N * p->G8; p->f16 = a->out16_type == 2;
    p->eps = a->eps; p->slope = a->slope; p->z = a->z; p->gamma = a->gamma; p->beta = a->beta; p->y = a->y; p->ybs = a->y_batch_stride;
    p->y8 = reinterpret_cast<unsigned short*>(a->y8); p->mean = a->mean; p->rstd = a->rstd; p->dy = a->dy; p->dybs = a->dy_batch_stride;
    p->dz8 = reinterpret_cast<unsigned short*>(a->dz8); p->part = nullptr; p->state = a->coop_state;
    return MTBC_OK;
}
CoPlan plan_fwd(int items, int HW) {
    static int cap[9] = {-1};
    if (cap[0] < 0) {
        cap[1] = resident_blocks(in_fwd_coop_kernel<1, false>); cap[2] = resident_blocks(in_fwd_coop_kernel<2, false>);
        cap[4] = resident_blocks(in_fwd_coop_kernel<4, false>); cap[8] = resident_blocks(in_fwd_coop_kernel<8, false>);
        cap[0] = 0;
    }
    return plan_coop(items, HW, 8, cap);
}
CoPlan plan_bwd(int items, int HW) {
    static int cap[9] = {-1};
    if (cap[0] < 0) {
        cap[1] = resident_blocks(in_bwd_coop_kernel<1, false>); cap[2] = resident_blocks(in_bwd_coop_kernel<2, false>);
        cap[4] = resident_blocks(in_bwd_coop_kernel<4, false>); cap[8] = 0;
        cap[0] = 0;
    }
    return plan_coop(items, HW, 4, cap);
}

}  // namespace

extern "C" {

size_t mtbc_instnorm_coop_state_bytes(void) { return CO_MAILBOX_OFF + (size_t)CO_MAX_TEAMS * CO_TEAM_WORDS * sizeof(unsigned long long); }

int mtbc_instnorm_c8_supported(const mtbc_instnorm_args* a, int32_t backward) {
    if (!a || a->N <= 0 || a->C <= 0 || a->H <= 0 || a->W <= 0 || a->C % 8) return 0;
    if (backward && a->n_dy_extra != 0) return 0;
    const CoPlan pl = backward ? plan_bwd(a->N * (a->C / 8), a->H * a->W) : plan_fwd(a->N * (a->C / 8), a->H * a->W);
    return pl.ok ? 1 : 0;
}

int mtbc_i_instnorm_fwd_c8(const mtbc_instnorm_args* a, hipStream_t st) {
    CoP p; int rc = fill_coop(a, &p); if (rc) return rc;
    if (!p.z || !p.y8 || !p.mean || !p.rstd || (reinterpret_cast<uintptr_t>(p.y8) & 15)) return MTBC_E_BADARG;
    const CoPlan pl = plan_fwd(p.items, p.HW);
    if (!pl.ok) return MTBC_E_UNSUPPORTED;
    p.T = pl.T; p.nteams = pl.nteams;
    const dim3 g(pl.grid), b(CO_THREADS);
#define MTBC_CO_F(PPT_)                                                                                  \
    do { if (p.f16) hipLaunchKernelGGL((in_fwd_coop_kernel<PPT_, true>), g, b, 0, st, p);                \
         else hipLaunchKernelGGL((in_fwd_coop_kernel<PPT_, false>), g, b, 0, st, p); } while (0)
    if (pl.ppt == 8) MTBC_CO_F(8); else if (pl.ppt == 4) MTBC_CO_F(4); else if (pl.ppt == 2) MTBC_CO_F(2); else MTBC_CO_F(1);
#undef MTBC_CO_F
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}

int mtbc_i_instnorm_bwd_c8(const mtbc_instnorm_args* a, float* part, hipStream_t st) {
    CoP p; int rc = fill_coop(a, &p); if (rc) return rc;
    if (!p.z || !p.dy || !p.dz8 || !p.mean || !p.rstd || (reinterpret_cast<uintptr_t>(p.dz8) & 15) || a->n_dy_extra != 0) return MTBC_E_BADARG;
    const CoPlan pl = plan_bwd(p.items, p.HW);
    if (!pl.ok) return MTBC_E_UNSUPPORTED;
    p.T = pl.T; p.nteams = pl.nteams; p.part = part;
    const dim3 g(pl.grid), b(CO_THREADS);
#define MTBC_CO_B(PPT_)                                                                                  \
    do { if (p.f16) hipLaunchKernelGGL((in_bwd_coop_kernel<PPT_, true>), g, b, 0, st, p);                \
         else hipLaunchKernelGGL((in_bwd_coop_kernel<PPT_, false>), g, b, 0, st, p); } while (0)
    if (pl.ppt == 4) MTBC_CO_B(4); else if (pl.ppt == 2) MTBC_CO_B(2); else MTBC_CO_B(1);
#undef MTBC_CO_B
    MTBC_CHECK_LAUNCH();
    return MTBC_OK;
}

}  // extern "C"
