/*
 * mtbc.h -- C-ABI of libmtbc_hip.so: the MI355X (gfx950) hot path of the multi-task
 * conv encoder-decoder training step of caumente/multi_task_breast_cancer.
 *
 * The reference has no FFI of its own (it is pure Python on torch.nn); each
 * entry point below names the reference call it replaces (file:line, relative to the
 * upstream repository).  All tensors are fp32, NCHW, plane-contiguous (channel stride =
 * H*W) device pointers BORROWED from the caller until the stream operation completes; the
 * 16-bit compute modes may additionally hand the MFMA kernels their operand tensors -- and take
 * the conv outputs between a convolution and its norm -- in the matrix pipe's own format
 * (MTBC_LAYOUT_C8: bf16 / fp16, [N][C/8][H*W][8]) -- every such field says so where it is declared, and what the reference would see (parameters, gradients of
 * parameters, losses, logits) is fp32 in every mode.
 *
 * Conventions: return 0 (MTBC_OK) or a negative MTBC_E_* code; never throws, never
 * allocates, never synchronises the device; deterministic (no float atomics); thread-safe
 * for distinct streams (no mutable process-wide state).  `stream` is a hipStream_t passed as void*.
 */
#ifndef MTBC_H
#define MTBC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTBC_VERSION 202            /* 0.2.2 (round 4: mtbc_adam_args.dynamic appended + mtbc_adam_dynamic); 0.2.1: mtbc_conv3x3_args.wgrad_sync appended; 0.2.0: the argument structs grew in round 2 (fields appended); a binding compiled against
                                       another version must refuse the library (mtbc_version()) -- layouts are not negotiated */
#define MTBC_MAX_SEGS 6

enum {
    MTBC_OK = 0,
    MTBC_E_BADSHAPE = -1,
    MTBC_E_BADARG = -2,
    MTBC_E_WORKSPACE = -3,
    MTBC_E_LAUNCH = -4,
    MTBC_E_UNSUPPORTED = -5
};

int mtbc_version(void);
const char* mtbc_strerror(int code);
/* name of the GPU architecture the device code was built for ("gfx950") */
const char* mtbc_arch(void);

/* A channel segment of a virtually concatenated NCHW tensor (replaces torch.cat(dim=1),
 * MTnnUNet.py:161-169,174; MTUNetPlusPlus.py:107-118,128).  Element (n, c, y, x) of the
 * segment lives at ptr[n*batch_stride + c*H*W + y*W + x].  `accumulate` is honoured when
 * the segment is an OUTPUT (gradient fan-in): 0 = overwrite, 1 = add to what is there;
 * 2 (mtbc_conv3x3_dgrad with operand_layout C8 only) = overwrite a 16-BIT planar (N,C,H,W)
 * segment of the type of `compute` (ptr 8-byte aligned, batch_stride in 16-bit elements): the
 * fp32 result rounded to nearest even -- for a gradient whose only reader rounds it to that
 * type anyway (the k = 2 transposed conv backward, mtbc_convT_args.dy_type16);
 * 3 (same call) = overwrite a 16-BIT CHANNEL-BLOCKED segment [N][channels/8][H*W][8] of that type (MTBC_LAYOUT_C8;
 * ptr 16-byte aligned, batch_stride in 16-bit elements, channels % 8 == 0) -- for a gradient with this one writer
 * whose reader takes the layout (mtbc_instnorm_args.dy_layout).  Either every segment of a launch has mode 3 or none. */
typedef struct {
    float* ptr;
    int64_t batch_stride;   /* elements */
    int32_t channels;
    int32_t accumulate;
} mtbc_seg;

/* ---------------------------------------------------------------- conv 3x3, stride 1, pad 1
 * replaces nn.Conv2d(k=3, padding=1): MTnnUNet.py:12-16 ; MONAI Convolution in
 * MTUNetPlusPlus.py:47-81.  Weight layout = torch (Cout, Cin, 3, 3).                       */
typedef struct {
    int32_t N, H, W, Cin, Cout;
    int32_t n_in;                    /* fwd/wgrad: input segments (sum channels == Cin)      */
    mtbc_seg in[MTBC_MAX_SEGS];      /* dgrad: OUTPUT dx segments (accumulate honoured)      */
    const float* w;                  /* (Cout,Cin,3,3)                                       */
    const float* w_packed;           /* fwd: mtbc_conv3x3_pack_fwd image, dgrad: _pack_dgrad;
                                        NULL selects the direct (non-MFMA) kernel           */
    const float* bias;               /* (Cout) or NULL                                       */
    float* out;                      /* fwd: z (N,Cout,H,W) ; dgrad: unused                  */
    const float* dout;               /* dgrad/wgrad: dz (N,Cout,H,W)                         */
    float* dw;                       /* wgrad: (Cout,Cin,3,3), overwritten or accumulated    */
    float* dbias;                    /* wgrad: (Cout) or NULL                                */
    int32_t accumulate_dw;           /* wgrad: 1 = add into dw/dbias (shared modules, F10)   */
    int32_t force_direct;            /* 1 = never use the MFMA kernels (debug / tests)       */
    void* workspace;                 /* wgrad split-K partials                               */
    size_t workspace_bytes;
    int32_t compute;                 /* MFMA operand type: 0 = fp32 (exact, the parity path), 1 = bf16,
                                        2 = fp16 -- storage and accumulation stay fp32; fwd/dgrad then need
                                        w_packed from mtbc_conv3x3_pack_lp                    */
    int32_t operand_layout;          /* how the tensors the MFMA READS are stored (fwd: `in`; dgrad: `dout`;
                                        wgrad: `in` and `dout`):
                                          0 = fp32 planar (N,C,H,W), converted while staging (default);
                                          1 = MTBC_LAYOUT_C8: already in the 16-bit type of `compute` (1 or 2),
                                              channel-blocked [N][C/8][H*W][8] (mtbc_c8_pack) -- pointers are
                                              passed through the float* / const float* fields, batch strides count
                                              16-bit elements, every segment holds a multiple of 8 channels and is
                                              16-byte aligned.  What the kernels WRITE (z, dx segments, dw, dbias)
                                              stays fp32 planar unless out_layout / a segment mode says otherwise.  No fallback: shapes the MFMA kernels do not take
                                              (W % 4 != 0, H or W < 8) return MTBC_E_UNSUPPORTED.                 */
    int32_t out_accumulate;          /* fwd with operand_layout C8 only: 1 = add the result to `out` instead of overwriting
                                        it.  A forward launch over the dz of ALL 3x3 consumers of a tensor, with
                                        mtbc_conv3x3_weight_view(mode 1) weights, is that tensor's gathered dgrad.  */
    int32_t out_layout;              /* fwd with operand_layout C8 only: MTBC_LAYOUT_C8 = `out` is written as a 16-bit
                                        channel-blocked tensor [N][Cout/8][H*W][8] of the type of `compute` (fp32 accumulate,
                                        + bias, ONE round-to-nearest-even) instead of fp32 planes: the conv output of the
                                        16-bit modes, read by mtbc_instnorm_args.z_layout = C8 (what torch.autocast stores
                                        between a convolution and its normalisation).  Cout % 8 == 0, out 16-byte aligned,
                                        out_accumulate = 0.                                                             */
    float* stats_partial;            /* fwd with out_layout C8 only, or NULL: InstanceNorm statistics from the conv EPILOGUE
                                        (MTnnUNet.py:30-38, MONAI ADN in MTUNetPlusPlus.py:20-22: the norm that follows every
                                        3x3 conv).  [N][slots][Cout][2] floats, slots = mtbc_conv3x3_stats_slots(args): per
                                        image, per wave-sized pixel subset and per channel {sum, sum of squares} of the STORED
                                        (rounded) outputs, fp32, no atomics; mtbc_instnorm_lrelu_fwd(stats_partial, stats_slots)
                                        combines them (fixed order, in double) and the normalisation becomes one streaming
                                        pass with no reduction of its own.                                                  */
    /* A gathered dgrad (forward-type launch over the dz of a tensor's 3x3 consumers, out_layout C8) can also prepare the
       InstanceNorm + LeakyReLU BACKWARD of the tensor it differentiates (all optional, norm_z selects it):
         out_partial : fp32 planar (N,Cout,H,W) partial gradient written by the tensor's other readers (pool / ConvT / 1x1
                       backward), added in fp32 BEFORE the one rounding of `out`;
         norm_z      : the tensor's own conv output z, 16-bit channel-blocked like `out`; norm_mean / norm_rstd (N*Cout),
                       norm_gamma / norm_beta (Cout or NULL), norm_slope: its InstanceNorm + LeakyReLU;
       stats_partial then receives, per image, pixel subset and channel, {sum g, sum g * xhat} with g = dy * lrelu'(.) of the
       STORED dy -- the two reductions of the norm's backward (mtbc_instnorm_lrelu_bwd(stats_partial, stats_slots) becomes
       one streaming pass).                                                                                               */
    const float* out_partial;
    const void* norm_z;
    const float* norm_mean;
    const float* norm_rstd;
    const float* norm_gamma;
    const float* norm_beta;
    float norm_slope;
    int32_t out_type;                /* fwd with out_layout C8: 0 = the type of `compute`; 2 with compute = 1 (bf16 operands): the
                                        output is stored as fp16 (saturated at +-65504) -- for conv outputs of O(1) that a norm
                                        follows: 11 significant bits in the same 2 bytes (mtbc_instnorm_args.z_type)              */
    int32_t* wgrad_sync;             /* wgrad with operand_layout C8 (Cin >= 8), or NULL: mtbc_conv3x3_wgrad_sync_bytes(args) bytes of
                                        ZEROED device memory (4-byte aligned) that stay the caller's between launches.  With it the
                                        split-K partials are reduced INSIDE the weight-gradient launch -- the block that stores the last
                                        partial of a group sums the group in row order (deterministic; no block waits for another, no
                                        co-residency requirement), a tree of 1 - 3 levels whose top writes dw / dbias -- instead of by a
                                        second launch.  Every counter is back at zero when the launch ends: one buffer serves all the
                                        launches of a stream, in order.  Launches on DIFFERENT streams need different buffers.  (backward
                                        of nn.Conv2d in MTUNetPlusPlus.py:47-81 / MTnnUNet.py:12-16, training_multitask.py:102)       */
    size_t wgrad_sync_bytes;
} mtbc_conv3x3_args;
#define MTBC_LAYOUT_PLANAR 0
#define MTBC_LAYOUT_C8 1

/* Views of a (Cout,Cin,3,3) weight for backward launches that cover only part of a conv's input channels:
 *   mode 0: dst (Cout, ci_cnt, 3, 3)  = w[:, ci_off : ci_off + ci_cnt]                 -- dgrad into a suffix of the segments
 *   mode 1: dst (ci_cnt, K, 3, 3)[:, k_off : k_off + Cout] = that slice transposed with flipped taps, i.e. the weight of
 *           the FORWARD conv over dz that computes the slice's input gradient; several convs fill one dst side by side
 *           (K = sum of their Cout): one launch then produces the gradient of a tensor from all its consumers
 *           (no read-modify-write fan-in).  Rebuilt every step like the packed images.                              */
int mtbc_conv3x3_weight_view(const float* w, float* dst, int32_t Cout, int32_t Cin, int32_t ci_off, int32_t ci_cnt, int32_t mode,
                             int32_t k_off, int32_t K, void* stream);
/* Batched form: every weight view a step needs in one launch (they are rebuilt after each optimizer update). */
typedef struct mtbc_wview_desc {
    const float* w;
    float* dst;
    int32_t Cout, Cin, ci_off, ci_cnt, mode, k_off, K;
} mtbc_wview_desc;
int mtbc_conv3x3_weight_view_many(const mtbc_wview_desc* descs, int32_t n, void* stream);

/* fp32 planar (N,C,H,W; batch stride in elements) -> 16-bit channel-blocked [N][C/8][H*W][8] in the type of
 * `compute` (1 = bf16, 2 = fp16; round to nearest even -- the same conversion the staging of operand_layout 0 does),
 * and back (exact).  C % 8 == 0, dst 16-byte aligned.  Producers that write this layout themselves (next step:
 * InstanceNorm+LeakyReLU, pooling, ConvTranspose) make the 3x3 convolutions' staging pure LDS-DMA.           */
int mtbc_c8_pack(const float* src, int64_t src_batch_stride, void* dst, int32_t N, int32_t C, int32_t HW, int32_t compute, void* stream);
int mtbc_c8_unpack(const void* src, float* dst, int32_t N, int32_t C, int32_t HW, int32_t compute, void* stream);
/* 16-bit planar (N,C,H,W; batch stride in 16-bit elements) -> the same values channel-blocked [N][C/8][H*W][8].
 * Type-agnostic (moves 16-bit words).  C % 8 == 0, HW % 4 == 0, src 8-byte and dst 16-byte aligned.             */
int mtbc_c8_pack16(const void* src, int64_t src_batch_stride, void* dst, int32_t N, int32_t C, int32_t HW, void* stream);

size_t mtbc_conv3x3_packed_elems(int32_t Cin, int32_t Cout);          /* fwd image size      */
size_t mtbc_conv3x3_packed_dgrad_elems(int32_t Cin, int32_t Cout);    /* dgrad image size    */
int mtbc_conv3x3_pack_fwd(const float* w, float* packed, int32_t Cin, int32_t Cout, void* stream);
int mtbc_conv3x3_pack_dgrad(const float* w, float* packed, int32_t Cin, int32_t Cout, void* stream);
/* 16-bit operand images for compute = 1 (bf16) / 2 (fp16): [mtile][chunk32][tap][16][32] 16-bit elements;
 * dgrad = 0 packs the forward image, 1 the flipped/transposed dgrad image.  Pass as w_packed. */
size_t mtbc_conv3x3_packed_lp_elems(int32_t Cin, int32_t Cout, int32_t dgrad);
int mtbc_conv3x3_pack_lp(const float* w, void* packed, int32_t Cin, int32_t Cout, int32_t dgrad, int32_t compute, void* stream);

/* Batched form of the four pack entry points above: every weight image a step needs in one launch (the images are
 * re-made after each optimizer update; ~70 tiny launches cost more than the copies).  kind: 0 = mtbc_conv3x3_pack_fwd,
 * 1 = _pack_dgrad, 2 = _pack_lp(dgrad=0), 3 = _pack_lp(dgrad=1); `compute` (1 bf16 | 2 fp16) is read for kinds 2,3. */
typedef struct mtbc_pack_desc {
    const float* w;
    void* packed;
    int32_t Cin, Cout;
    int32_t kind, compute;
} mtbc_pack_desc;
int mtbc_conv3x3_pack_many(const mtbc_pack_desc* descs, int32_t n, void* stream);
/* pixel subsets per image of the forward launch these arguments select (0: the launch does not produce statistics) */
int32_t mtbc_conv3x3_stats_slots(const mtbc_conv3x3_args* a);
size_t mtbc_conv3x3_wgrad_workspace(const mtbc_conv3x3_args* a);
/* bytes of zeroed counters mtbc_conv3x3_args.wgrad_sync needs for these arguments (0: this launch has no in-kernel reduction) */
size_t mtbc_conv3x3_wgrad_sync_bytes(const mtbc_conv3x3_args* a);
int mtbc_conv3x3_fwd(const mtbc_conv3x3_args* a, void* stream);
int mtbc_conv3x3_dgrad(const mtbc_conv3x3_args* a, void* stream);
int mtbc_conv3x3_wgrad(const mtbc_conv3x3_args* a, void* stream);

/* ------------------------------------------------- InstanceNorm2d(eps, affine?) + LeakyReLU
 * replaces nn.InstanceNorm2d + nn.LeakyReLU: MTnnUNet.py:35-36 (no affine, slope 0.01);
 * MONAI ADN "NDA" in MTUNetPlusPlus.py:20-22 (affine, slope 0.1).
 *   y = lrelu(gamma * (z - mean) * rstd + beta), biased variance over H*W per (n, c).
 * The output may be a channel slice of a wider buffer (out_batch_stride).                   */
typedef struct {
    int32_t N, C, H, W;
    float eps, slope;
    const float* z;          /* (N,C,H,W) conv output                                        */
    const float* gamma;      /* (C) or NULL                                                  */
    const float* beta;       /* (C) or NULL                                                  */
    float* y;                /* activation, element (n,c,p) at y[n*y_batch_stride + c*HW + p] */
    int64_t y_batch_stride;
    float* mean;             /* (N*C) saved for backward                                     */
    float* rstd;             /* (N*C)                                                        */
    /* backward */
    const float* dy;         /* grad wrt y, same addressing as y (dy_batch_stride)           */
    int64_t dy_batch_stride;
    int32_t n_dy_extra;      /* gradient fan-in: up to 4 more contributions, added to dy on the */
    const float* dy_extra[4];/* fly (same addressing) -- each consumer of y writes its own buffer
                                with plain stores instead of read-modify-write accumulation      */
    float* dz;               /* (N,C,H,W) grad wrt z                                         */
    float* dgamma;           /* (C) or NULL */
    float* dbeta;            /* (C) or NULL */
    float* dbias_pre;        /* (C) or NULL: gradient of a per-channel bias added in front of the
                                norm (the conv bias) = sum over n,h,w of dz, fused here so the
                                conv wgrad does not re-read dz                                 */
    int32_t accumulate_dparams;
    void* workspace;         /* bwd with any of the three: N*C*3 floats (N*C*131 with dz8)    */
    size_t workspace_bytes;
    /* 16-bit planar outputs (the 16-bit compute modes): when y16 / dz16 is set, the forward activation / the
       backward dz is written ONLY as a 16-bit (N,C,H,W) tensor of type out16_type (1 = bf16, 2 = fp16; round to
       nearest even of the fp32 value) and y / dz is not touched -- for tensors that nothing but the 3x3 convs' MFMAs
       read (mtbc_c8_pack16 then brings them into MTBC_LAYOUT_C8): 2 instead of 4 bytes written and re-read.
       Register-resident planes only (H*W % 4 == 0, H*W <= 65536), else MTBC_E_UNSUPPORTED.            */
    void* y16;
    void* dz16;
    int32_t out16_type;
    /* 16-bit CHANNEL-BLOCKED outputs (MTBC_LAYOUT_C8, [N][C/8][H*W][8] of type out16_type): the forward activation
       goes to y8 (and, if y != NULL, also to y as fp32 planes), the backward dz to dz8 only.  One HBM pass by a
       cooperative kernel (csrc/norm_coop.hip: teams of resident workgroups exchange per-channel partials through
       coop_state).  coop_state: mtbc_instnorm_coop_state_bytes() of device memory, ZEROED once by the caller, kept
       for the lifetime of the stream's launches and never used from two streams at once.  C % 8 == 0, no dy_extra;
       shapes: ask mtbc_instnorm_c8_supported.  Statistics are combined with Chan's formula (fixed order).
       y8 together with y16 (forward with stats_partial only, H*W % 8 == 0; otherwise MTBC_E_UNSUPPORTED): the PLANAR copy is
       the 16-bit one -- the same rounded values as y8, for a reader that wants planes of the MFMA type
       (mtbc_convT_args.x_type16) -- and y is not touched; y_batch_stride then counts 16-bit elements of y16 (% 8 == 0).  */
    void* y8;
    void* dz8;
    void* coop_state;
    /* CUs this launch leaves out of the cooperative grid (0 = none): the teams need every member resident at once, and
       kernels of OTHER streams that hold CUs while it runs (RCCL collectives overlapping the backward pass) would make
       members wait for a slot.  Per call -- the library keeps no process-wide setting.                              */
    int32_t coop_reserve_cus;
    /* 16-bit channel-blocked INPUTS (with y8 / dz8 outputs only; type out16_type, C % 8 == 0, 16-byte aligned):
       z_layout  = MTBC_LAYOUT_C8: z is [N][C/8][H*W][8] as written by mtbc_conv3x3_fwd(out_layout = C8); statistics are
                   those of the stored (rounded) values, accumulated in fp32;
       dy_layout = MTBC_LAYOUT_C8: dy is [N][C/8][H*W][8] as written by mtbc_conv3x3_dgrad (segment mode 3) or by a
                   gathered forward-type launch with out_layout = C8 (dy_batch_stride in 16-bit elements); with
                   n_dy_extra = 1, dy_extra[0] is an fp32 planar (N,C,H,W) partial gradient from the tensor's other
                   readers (batch stride C*H*W), added to it in fp32 while loading.                                     */
    int32_t z_layout;
    int32_t dy_layout;
    /* forward with z_layout C8 and y8: the statistics come from the producing convolution's epilogue
       (mtbc_conv3x3_args.stats_partial, [N][stats_slots][C][2]); mean / rstd are still written for the backward pass.
       backward with z_layout C8, dy_layout C8 and dz8: {sum g, sum g * xhat} come from the gathered dgrad that wrote dy
       (mtbc_conv3x3_args.norm_z); the conv-bias gradient dbias_pre (mathematically zero in front of a norm) is then written
       as exact zeros instead of the rounding noise of a sum.  workspace: N*C*5 floats.                                  */
    const float* stats_partial;
    int32_t stats_slots;
    int32_t z_type;                  /* with z_layout C8: 0 = z has the type of out16_type; 2 = z is fp16 although out16_type is bf16 */
    /* backward with dz8: the input gradient of a ONE-output 1x1 conv head that reads this activation (MTUNetPlusPlus.py:73-76,
       final_conv_0_j) is rank 1 -- dy[n,c,p] += dy_rank1_w[c] * dy_rank1[n*H*W + p] -- and is formed here from the head's own
       gradient (N,1,H,W) and weight (C) instead of being written by mtbc_conv1x1_dgrad and read back; dy may then be NULL
       (a tensor nothing else reads).  Both NULL or both set.                                                                */
    const float* dy_rank1;
    const float* dy_rank1_w;
    /* ... and, optionally, that head's OWN parameter gradients (what mtbc_conv1x1_wgrad computes from the stored activation):
       dy_rank1_dw[c] (+)= sum_{n,p} y_stored[n,c,p] * dy_rank1[n,p] with y re-formed from z and rounded to out16_type,
       dy_rank1_db[0] (+)= sum dy_rank1.  workspace: N*C*(3 + 2T) + N*T floats, T = team size (131 -> 260 per plane covers it). */
    float* dy_rank1_dw;
    float* dy_rank1_db;
    int32_t dy_rank1_accumulate;
    /* backward with dz8: the gradient coming back through a MaxPool2d(2,2) that reads this activation (MTnnUNet.py:103, MONAI Down):
       dy[n,c,y,x] += dy_pool[n,c,y/2,x/2] if the window's maximum sits at (y,x) according to dy_pool_arg (mtbc_maxpool_args.argmax),
       formed while loading.  dy_pool: fp32 planar (N,C,H/2,W/2).  Both NULL or both set; H and W even.                          */
    const float* dy_pool;
    const void* dy_pool_arg;
    /* forward with stats_partial (the streaming pass): also write the activation's MaxPool2d(2,2) -- pool_y8: channel-blocked 16-bit
       [N][C/8][H/2*W/2][8] of out16_type, the maxima of the STORED values (= mtbc_maxpool2_fwd on y8, bit for bit); pool_arg: optional
       argmax codes (mtbc_maxpool_args.argmax).  H and W even.                                                                     */
    void* pool_y8;
    void* pool_arg;
    int32_t defer_dparams;           /* backward with dz8 and any of dgamma / dbeta / dbias_pre: 1 = stop after the norm kernel: the
                                        per-plane partial sums stay in `workspace` (a buffer of the caller's that lives until
                                        mtbc_instnorm_dparam_many has reduced it) and dgamma / dbeta / dbias_pre are not touched --
                                        one launch for the parameter gradients of many cells instead of one 5 us launch each      */
} mtbc_instnorm_args;
size_t mtbc_instnorm_coop_state_bytes(void);
/* Byte offset, inside a coop_state block, of the 32-bit STICKY error word: non-zero once any cooperative launch on that
 * state gave up a bounded mailbox poll (a team member was not resident).  Every output of that launch and of the ones after
 * it is then garbage: callers read the word when they read their losses and stop (trainer.FusedTrainStep.check_nan).   */
size_t mtbc_instnorm_coop_error_offset(void);
int mtbc_instnorm_c8_supported(const mtbc_instnorm_args* a, int32_t backward);
/* team size of the channel-group backward these arguments select (1 for planes <= 64 x 64; 0: not supported) */
int32_t mtbc_instnorm_bwd_team(const mtbc_instnorm_args* a);
/* dgamma[c] (+)= sum_n part[n,c,1] ; dbeta[c] (+)= sum_n part[n,c,0] ; dbias_pre[c] (+)= sum_n (part[n,c,2] + sum_m part3[n,c,m]) for
 * every descriptor in ONE launch, from the partials a deferred backward left: part = workspace, part3 = workspace + 3*N*C (T floats per
 * plane).  Same summation order as the reduction inside mtbc_instnorm_lrelu_bwd.                                                   */
typedef struct mtbc_dparam_desc {
    const float* part;
    float* dgamma;                   /* (C) or NULL */
    float* dbeta;                    /* (C) or NULL */
    float* dbias_pre;                /* (C) or NULL */
    int32_t N, C, T, accumulate;
} mtbc_dparam_desc;
int mtbc_instnorm_dparam_many(const mtbc_dparam_desc* descs, int32_t n, void* stream);

/* workspace bytes the forward can use for planes > 64K elements (chunked statistics: 2 reads + 1 write instead of the
 * streaming kernel's 3 + 1); 0 for smaller planes.  Passing no workspace is valid (streaming kernel). */
size_t mtbc_instnorm_fwd_workspace(const mtbc_instnorm_args* a);
int mtbc_instnorm_lrelu_fwd(const mtbc_instnorm_args* a, void* stream);
int mtbc_instnorm_lrelu_bwd(const mtbc_instnorm_args* a, void* stream);

/* --------------------------------------------------------------------- MaxPool2d(2, 2)
 * replaces nn.MaxPool2d(2,2): MTnnUNet.py:103 ; MONAI Down in MTUNetPlusPlus.py:48-51.
 * bwd routes dy to the first maximal element in (0,0),(0,1),(1,0),(1,1) order, like ATen.  */
typedef struct {
    int32_t N, C, H, W;              /* INPUT spatial size; output is H/2 x W/2              */
    const float* x;  int64_t x_batch_stride;
    float* y;        int64_t y_batch_stride;
    const float* dy; int64_t dy_batch_stride;
    float* dx;       int64_t dx_batch_stride;
    int32_t accumulate_dx;
    int32_t layout;                  /* 0 = fp32 planar x / y.  MTBC_LAYOUT_C8 (the 16-bit compute modes): x and y are 16-bit
                                        channel-blocked [N][C/8][H*W][8] of type16 (1 = bf16, 2 = fp16), batch strides in
                                        16-bit elements, C % 8 == 0 -- forward = the fp32 pool of the stored values followed
                                        by mtbc_c8_pack, bit for bit (max commutes with rounding); backward routes on the
                                        stored values; dy / dx stay fp32 planar                                            */
    int32_t type16;
    void* argmax;                    /* fwd with layout C8, optional: (N, C/8, H/2 * W/2) 16-bit codes, bits [2c+1 : 2c] = which of the four
                                        window positions (0,0),(0,1),(1,0),(1,1) holds the first maximum of channel 8g + c -- where the
                                        backward routes the gradient.  With it the backward of the pool can be formed inside the
                                        InstanceNorm backward of the pooled tensor (mtbc_instnorm_args.dy_pool) instead of writing a
                                        4x larger, three-quarters-zero fp32 tensor that is read back                                   */
} mtbc_maxpool_args;

int mtbc_maxpool2_fwd(const mtbc_maxpool_args* a, void* stream);
int mtbc_maxpool2_bwd(const mtbc_maxpool_args* a, void* stream);

/* --------------------------------------------- ConvTranspose2d, kernel == stride == k
 * replaces nn.ConvTranspose2d(k, stride=k): MTnnUNet.py:96-100,106-116 (k = 2, 4, 8);
 * MONAI UpSample("deconv") in UpCat.  Weight layout = torch (Cin, Cout, k, k).
 *   y[n,co,k*i+a,k*j+b] = bias[co] + sum_ci x[n,ci,i,j] * W[ci,co,a,b]                      */
typedef struct {
    int32_t N, H, W, Cin, Cout, k;   /* INPUT spatial size H x W; output kH x kW             */
    const float* x;  int64_t x_batch_stride;
    const float* w;
    const float* bias;               /* (Cout) or NULL */
    float* y;        int64_t y_batch_stride;
    const float* dy; int64_t dy_batch_stride;
    float* dx;       int64_t dx_batch_stride;
    int32_t accumulate_dx;
    float* dw;       float* dbias;
    int32_t accumulate_dw;
    void* workspace; size_t workspace_bytes;
    int32_t compute;                 /* k == 2 backward only: 0 = fp32 MFMA (exact), 1 = bf16, 2 = fp16 operands */
    int32_t y_layout;                /* fwd: 0 = fp32 planar y; MTBC_LAYOUT_C8 = y is written as 16-bit channel-blocked
                                        [N][Cout/8][k*H*k*W][8] (what a 3x3 conv with operand_layout C8 reads; batch
                                        stride in 16-bit elements) -- same fp32 arithmetic, one RNE at the store, i.e.
                                        bit-identical to the planar forward followed by mtbc_c8_pack                */
    int32_t y_type;                  /* with y_layout C8: 1 = bf16, 2 = fp16                                        */
    int32_t x_layout;                /* fwd: MTBC_LAYOUT_C8 = x is ALSO 16-bit channel-blocked (type y_type, batch stride in
                                        16-bit elements; needs y_layout C8): the forward then runs on the 16-bit MFMA
                                        with rounded x and w (fp32 accumulate, bias, one RNE) -- k == 2, Cin % 8 == 0  */
    int32_t dy_type16;               /* dgrad / wgrad, k == 2 with compute = 1 | 2 only: non-zero (= compute) says dy is a 16-BIT
                                        planar (N,Cout,kH,kW) tensor of that type (batch stride in 16-bit elements), as
                                        written by mtbc_conv3x3_dgrad into a segment with accumulate = 2; the MFMAs read it
                                        as is (what they did to an fp32 dy while loading it), the bias gradient is the
                                        sum of the stored values.  Shapes the direct kernels do not take: MTBC_E_UNSUPPORTED */
    int32_t x_type16;                /* wgrad, with dy_type16 only: non-zero (= compute) says x is a 16-BIT planar (N,Cin,H,W) tensor of
                                        that type too (batch stride in 16-bit elements, % 8 == 0), as written by
                                        mtbc_instnorm_lrelu_fwd into y16 beside y8: 8 pixels of a channel are one 16-byte load that
                                        is the MFMA fragment -- the operands the fp32 x gave after rounding, bit for bit      */
} mtbc_convT_args;
/* 1 if mtbc_convT_fwd takes these arguments with y_layout = MTBC_LAYOUT_C8 (fp32 x: k == 2, Cin <= 64, Cout <= 48,
 * Cout % 8 == 0, H*W % 32 == 0; x_layout C8: k == 2, channel counts % 8 == 0, H*W % 32 == 0), else 0: the caller then
 * keeps the planar forward and converts with mtbc_c8_pack.                                                     */
int mtbc_convT_fwd_c8_supported(const mtbc_convT_args* a);

size_t mtbc_convT_wgrad_workspace(const mtbc_convT_args* a);
int mtbc_convT_fwd(const mtbc_convT_args* a, void* stream);
int mtbc_convT_dgrad(const mtbc_convT_args* a, void* stream);
int mtbc_convT_wgrad(const mtbc_convT_args* a, void* stream);

/* -------------------------------------------------------------------------- conv 1x1 + bias
 * replaces nn.Conv2d(k=1): MTnnUNet.py:6-9,106-118 ; MTUNetPlusPlus.py:73-76 (Cout = regions) */
typedef struct {
    int32_t N, H, W, Cin, Cout;
    const float* x;  int64_t x_batch_stride;
    const float* w;                  /* (Cout,Cin,1,1) */
    const float* bias;
    float* y;                        /* (N,Cout,H,W) */
    const float* dy;
    float* dx;       int64_t dx_batch_stride;
    int32_t accumulate_dx;
    float* dw;       float* dbias;
    int32_t accumulate_dw;
    void* workspace; size_t workspace_bytes;
    int32_t x_layout;                /* 0 = fp32 planar x.  MTBC_LAYOUT_C8: x is 16-bit channel-blocked [N][Cin/8][H*W][8] of
                                        x_type (1 = bf16, 2 = fp16; batch stride in 16-bit elements; Cin % 8 == 0, Cout <= 8):
                                        forward and weight gradient read the stored (rounded) activation with fp32 weights,
                                        products and sums (forward = mtbc_conv1x1_fwd on the unpacked tensor, bit for bit);
                                        y, dy, dx, dw stay fp32.  Ask mtbc_conv1x1_wgrad_workspace with the same fields set */
    int32_t x_type;
} mtbc_conv1x1_args;

size_t mtbc_conv1x1_wgrad_workspace(const mtbc_conv1x1_args* a);
int mtbc_conv1x1_fwd(const mtbc_conv1x1_args* a, void* stream);
int mtbc_conv1x1_dgrad(const mtbc_conv1x1_args* a, void* stream);
int mtbc_conv1x1_wgrad(const mtbc_conv1x1_args* a, void* stream);

/* ----------------------------------------------- AdaptiveAvgPool2d(1) + Flatten
 * replaces MTnnUNet.py:126-127 ; MTUNetPlusPlus.py:81-82                                    */
typedef struct {
    int32_t N, C, H, W;
    const float* x;  float* y;       /* y: (N,C) */
    const float* dy; float* dx;      /* dx: (N,C,H,W) overwritten */
} mtbc_gap_args;
int mtbc_gap_fwd(const mtbc_gap_args* a, void* stream);
int mtbc_gap_bwd(const mtbc_gap_args* a, void* stream);

/* ----------------------------------------------------------- Linear (+ optional ReLU)
 * replaces nn.Linear/nn.ReLU: MTnnUNet.py:128-131 ; MTUNetPlusPlus.py:83-86.
 * y = relu?(x W^T + b), W: (Out, In).  bwd needs y when relu != 0.                          */
typedef struct {
    int32_t N, In, Out, relu;
    const float* x; const float* w; const float* bias; float* y;
    const float* dy; float* dx;      /* dx (N,In) overwritten; may be NULL */
    float* dw; float* dbias; int32_t accumulate_dw;
    void* workspace; size_t workspace_bytes;   /* bwd with relu: N*Out floats */
} mtbc_linear_args;
int mtbc_linear_fwd(const mtbc_linear_args* a, void* stream);
int mtbc_linear_bwd(const mtbc_linear_args* a, void* stream);

/* ------------------------------------------------------------------------------ Dice loss
 * replaces monai.losses.DiceLoss(include_background=True, sigmoid=True, squared_pred=True,
 * smooth_nr=1, smooth_dr=1) built at experiment_init.py:210-211, fused over up to 4 deep
 * supervision heads with the 1/(j+1) weights of criterions.py:62.
 *   loss_h = mean_{n,c}( 1 - (2 I + nr) / (D + dr) ),  I = sum p t, D = sum p^2 + sum t^2
 * fwd : stats[h][n*C+c] = {I, P2, T2};  loss[h] = loss_h (unweighted);
 *       loss[n_heads] = sum_h head_weight[h] * loss_h
 * bwd : dx_h = gscale * head_weight[h] / (N*C) * d f / d x   (gscale: host scalar times an
 *       optional device scalar *gscale_dev, e.g. the upstream autograd gradient)            */
typedef struct {
    int32_t n_heads, N, C, H, W;
    float smooth_nr, smooth_dr;
    const float* x[4];               /* logits per head (N,C,H,W) */
    const float* target;             /* (N,C,H,W) */
    float head_weight[4];
    float* stats;                    /* n_heads * N*C * 3 floats */
    float* loss;                     /* n_heads + 1 floats */
    float* dx[4];
    float gscale;
    const float* gscale_dev;         /* may be NULL */
} mtbc_dice_args;
int mtbc_dice_fwd(const mtbc_dice_args* a, void* stream);
int mtbc_dice_bwd(const mtbc_dice_args* a, void* stream);

/* ----------------------------------------------------------------------------- Focal loss
 * replaces FocalLoss.forward (criterions.py:14-24, reduction='mean', soft/one-hot float
 * targets, optional class weight).  One launch computes the loss and d loss / d logits.
 *   ce_i = -sum_c w_c t_ic log_softmax(x_i)_c ; pt = exp(-ce_i) ;
 *   loss = mean_i alpha (1 - pt)^gamma ce_i ;  dx = gscale * (*gscale_dev) * d loss / d x
 * gamma = 0 is torch.nn.CrossEntropyLoss(reduction='mean') on soft targets (experiment_init.py:257-261, criterion "CE").
 * C == 1 is the binary head (n_classes == 2: ONE logit, MTUNetPlusPlus.py:39-41): ce_i = BCEWithLogits(x_i, t_i) =
 *   (1 - t) x + softplus(-x), target (N,1) in {0,1}; alpha = 1, gamma = 0 = torch.nn.BCEWithLogitsLoss() (experiment_init.py:241-242). */
typedef struct {
    int32_t N, C;
    float alpha, gamma;
    const float* x; const float* target; const float* weight;   /* weight may be NULL */
    float* loss;                     /* 1 float */
    float* dx;                       /* (N,C) or NULL */
    float gscale;
    const float* gscale_dev;
} mtbc_focal_args;
int mtbc_focal_fwd_bwd(const mtbc_focal_args* a, void* stream);

/* -------------------------------------------------------------- loss mix + NaN guard
 * replaces training_multitask.py:98 (alpha-mix) and the isnan guard of criterions.py:72-76:
 *   out[0] = alpha*seg + (1-alpha)*cls, out[1] = seg, out[2] = cls, out[3] = nan flag (0/1)  */
int mtbc_loss_mix(const float* seg, const float* cls, float alpha, float* out4, void* stream);

/* ------------------------------------------------------------------------------------ Adam
 * replaces torch.optim.Adam(lr, betas=(.9,.999), eps=1e-4).step(): experiment_init.py:187,
 * training_multitask.py:103 -- one fused launch over the flat parameter buffer.
 *   g' = grad_scale * g ; m = m + (1-b1)(g'-m) ; v = b2 v + (1-b2) g'^2
 *   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)                                 */
typedef struct {
    int64_t n;
    float* p; const float* g; float* m; float* v;
    float lr, beta1, beta2, eps, grad_scale;
    int32_t step;                    /* t >= 1 */
    int32_t zero_grad;               /* 1 = also clear g (optimizer.zero_grad) */
    const float* dynamic;            /* optional, DEVICE memory, 3 floats {grad_scale, lr / (1-b1^t), 1 / sqrt(1-b2^t)} as mtbc_adam_dynamic writes them: the kernel
                                        reads the three per-step scalars from here instead of taking them from lr / step / grad_scale as launch arguments --
                                        a step captured into a hipGraph is then replayed with the learning rate and step count of the DAY (the caller
                                        refreshes the 12 bytes in stream order before every replay).  NULL: the launch arguments.  Same arithmetic, same bits. */
} mtbc_adam_args;
int mtbc_adam_step(const mtbc_adam_args* a, void* stream);
/* host only, no GPU call: the three per-step scalars of `a` (lr, betas, step, grad_scale) exactly as mtbc_adam_step computes them -- bias corrections in double,
 * as torch.optim.Adam's scalar path -- for the caller to place in `dynamic`. */
int mtbc_adam_dynamic(const mtbc_adam_args* a, float out3[3]);

/* Whole-batch TP/FP/FN of (sigmoid(x) > .5) vs target, the train-loop Dice metric of
 * metrics.py:255-267 (training_multitask.py:66-71).  out3 = {tp, fp, fn} as float64.        */
int mtbc_dice_counts(const float* logits, const float* target, int64_t n, double* out3, void* stream);

/* ---------------------------------------------------------------------------- step program
 * A training step is a static list of the ops above with every pointer resolved at plan
 * time; mtbc_program_run issues them back-to-back on one stream (no host work in between). */
enum {
    MTBC_OP_CONV3_FWD = 1, MTBC_OP_CONV3_DGRAD, MTBC_OP_CONV3_WGRAD,
    MTBC_OP_CONV3_PACK_FWD, MTBC_OP_CONV3_PACK_DGRAD,
    MTBC_OP_IN_FWD, MTBC_OP_IN_BWD, MTBC_OP_POOL_FWD, MTBC_OP_POOL_BWD,
    MTBC_OP_CONVT_FWD, MTBC_OP_CONVT_DGRAD, MTBC_OP_CONVT_WGRAD,
    MTBC_OP_CONV1_FWD, MTBC_OP_CONV1_DGRAD, MTBC_OP_CONV1_WGRAD,
    MTBC_OP_GAP_FWD, MTBC_OP_GAP_BWD, MTBC_OP_LINEAR_FWD, MTBC_OP_LINEAR_BWD,
    MTBC_OP_DICE_FWD, MTBC_OP_DICE_BWD, MTBC_OP_FOCAL, MTBC_OP_LOSS_MIX, MTBC_OP_ADAM,
    MTBC_OP_MEMSET, MTBC_OP_DICE_COUNTS, MTBC_OP_CONV3_PACK_LP, MTBC_OP_HEAD_COMBINE, MTBC_OP_HEAD_EXPAND,
    MTBC_OP_C8_PACK, MTBC_OP_C8_PACK16, MTBC_OP_CONV3_WVIEW,
    MTBC_OP_SET_STREAM, MTBC_OP_EVENT_RECORD, MTBC_OP_EVENT_WAIT,
    MTBC_OP_IN_DPARAM
};
/* (the last three: stream control of mtbc_program_run_ms -- independent ops of a step, the weight gradient and the input
 * gradient of one layer, overlap on two HIP streams: the small latency-bound launches of the deep levels fill the chip together) */

/* ---- deep-supervision head of MTnnUNet: ConvTranspose2d(Cin->Cmid, k=s) followed by Conv2d(Cmid->R, 1x1)
 * (MTnnUNet.py:106-116: output4 k=8, output3 k=4, output2 k=2).  With no non-linearity in between the pair is ONE
 * transposed conv with Wc[ci][r][a][b] = sum_co wT[ci][co][a][b] * w1[r][co], bc[r] = sum_co bT[co] * w1[r][co] + b1[r]:
 * the Cmid x (kH x kW) intermediate (2.1 GB at B=64, k=8) and 128x the FLOPs disappear.  `combine` builds Wc / bc
 * (every step: the weights move), the ordinary mtbc_convT_{fwd,dgrad,wgrad} run with Cout = R, and `expand` maps the
 * gradient G of Wc and gb of bc back onto the four parameter tensors:
 *   dwT[ci][co][ab] = sum_r G[ci][r][ab] w1[r][co]        dbT[co] = sum_r gb[r] w1[r][co]
 *   dw1[r][co] = sum_{ci,ab} G[ci][r][ab] wT[ci][co][ab] + gb[r] bT[co]        db1[r] = gb[r]
 * Same function as the reference's two layers up to fp32 re-association (checked against the reference's goldens). */
typedef struct {
    int32_t Cin, Cmid, R, k;
    const float* wT; const float* bT;     /* (Cin,Cmid,k,k), (Cmid) */
    const float* w1; const float* b1;     /* (R,Cmid), (R)          */
    float* Wc; float* bc;                 /* (Cin,R,k,k), (R): written by combine */
    const float* G; const float* gb;      /* gradients of Wc / bc: read by expand  */
    float* dwT; float* dbT; float* dw1; float* db1;
    int32_t acc_wT, acc_bT, acc_w1, acc_b1;
} mtbc_head_fuse_args;
int mtbc_convT_head_combine(const mtbc_head_fuse_args* a, void* stream);
int mtbc_convT_head_expand(const mtbc_head_fuse_args* a, void* stream);

/* ---- training-time augmentation: RandomHorizontalFlip(.5) -> RandomVerticalFlip(.5) ->
 * RandomRotation(360) of training_multitask.py:193-197 / BUSI_dataset.py:142-147 on the joint (mask, image) stack,
 * torchvision semantics (nearest, zero fill, centre rotation).  src/dst (N,C,H,W), src != dst; params (N,4) =
 * {cos a, sin a, flip_h, flip_v} per sample, a = rotation angle (counter-clockwise, as torchvision's `angle`). */
int mtbc_augment_flip_rotate(const float* src, float* dst, const float* params, int32_t N, int32_t C, int32_t H, int32_t W,
                             void* stream);

typedef struct {
    int32_t kind;
    int32_t tag;                     /* free for the caller (layer id) */
    union {
        mtbc_conv3x3_args conv3;
        mtbc_instnorm_args inorm;
        mtbc_maxpool_args pool;
        mtbc_convT_args convT;
        mtbc_conv1x1_args conv1;
        mtbc_gap_args gap;
        mtbc_linear_args linear;
        mtbc_dice_args dice;
        mtbc_focal_args focal;
        mtbc_adam_args adam;
        struct { const float* w; float* packed; int32_t Cin, Cout; int32_t dgrad, compute; } pack;
        struct { const float* seg; const float* cls; float alpha; float* out4; } mix;
        struct { void* ptr; size_t bytes; } memset0;
        struct { const float* logits; const float* target; int64_t n; double* out3; } counts;
        mtbc_head_fuse_args head;
        struct { const float* w; float* dst; int32_t Cout, Cin, ci_off, ci_cnt, mode, k_off, K; } wview;
        struct { const float* src; int64_t src_batch_stride; void* dst; int32_t N, C, HW, compute; } c8pack;   /* C8_PACK and C8_PACK16 (src = 16-bit planar, `compute` unused) */
        mtbc_dparam_desc dparam;         /* MTBC_OP_IN_DPARAM: a run of them is issued as one mtbc_instnorm_dparam_many launch */
        struct { void* event; int32_t index; } sync;      /* SET_STREAM: following ops go to streams[index]; EVENT_RECORD / EVENT_WAIT: on the current stream */
    } u;
} mtbc_op;

/* run ops[first .. first+count) on `stream`; returns 0 or the first failing op's error code,
 * with *failed_index (may be NULL) set to its index. */
int mtbc_program_run(const mtbc_op* ops, int32_t first, int32_t count, void* stream, int32_t* failed_index);
/* The same with several streams: ops run on streams[0] until a MTBC_OP_SET_STREAM op selects another; MTBC_OP_EVENT_RECORD /
 * MTBC_OP_EVENT_WAIT order the streams (events from mtbc_event_create, reusable: a wait refers to the latest record issued
 * before it).  A range must begin and end on streams[0] with everything joined.  With n_streams == 1 every op runs on that
 * stream (indices are clamped): the same program, serialised.                                                            */
int mtbc_program_run_ms(const mtbc_op* ops, int32_t first, int32_t count, void* const* streams, int32_t n_streams, int32_t* failed_index);
/* timing-disabled HIP events for the two ops above; created when a step program is built, never inside a step */
int mtbc_event_create(void** event);
int mtbc_event_destroy(void* event);

#ifdef __cplusplus
}
#endif
#endif /* MTBC_H */
