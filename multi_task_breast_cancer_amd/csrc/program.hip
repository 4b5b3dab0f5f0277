// Step-program executor: issues a pre-resolved list of ops back-to-back on one stream, so the host does no
// per-op work inside a training step (the reference drives the same sequence op-by-op from Python autograd,
// training_multitask.py:87-103).  Also hosts version / error-string entry points.
#include "common.h"

extern "C" {

int mtbc_version(void) { return MTBC_VERSION; }
const char* mtbc_arch(void) { return "gfx950"; }

const char* mtbc_strerror(int code) {
    switch (code) {
        case MTBC_OK: return "ok";
        case MTBC_E_BADSHAPE: return "bad shape";
        case MTBC_E_BADARG: return "bad argument (null or misaligned pointer)";
        case MTBC_E_WORKSPACE: return "workspace missing or too small";
        case MTBC_E_LAUNCH: return "kernel launch failed";
        case MTBC_E_UNSUPPORTED: return "unsupported configuration";
        default: return "unknown error";
    }
}

static inline bool is_pack(int32_t k) { return k == MTBC_OP_CONV3_PACK_FWD || k == MTBC_OP_CONV3_PACK_DGRAD || k == MTBC_OP_CONV3_PACK_LP; }

int mtbc_event_create(void** event) {
    if (!event) return MTBC_E_BADARG;
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return MTBC_E_LAUNCH;
    *event = e;
    return MTBC_OK;
}
int mtbc_event_destroy(void* event) {
    if (!event) return MTBC_E_BADARG;
    return hipEventDestroy((hipEvent_t)event) == hipSuccess ? MTBC_OK : MTBC_E_LAUNCH;
}

int mtbc_program_run(const mtbc_op* ops, int32_t first, int32_t count, void* stream, int32_t* failed_index) {
    void* one[1] = {stream};
    return mtbc_program_run_ms(ops, first, count, one, 1, failed_index);
}

int mtbc_program_run_ms(const mtbc_op* ops, int32_t first, int32_t count, void* const* streams, int32_t n_streams, int32_t* failed_index) {
    if (!ops || first < 0 || count < 0 || !streams || n_streams < 1) return MTBC_E_BADARG;
    void* stream = streams[0];
    for (int32_t i = first; i < first + count; ++i) {
        const mtbc_op* o = &ops[i];
        int rc;
        if (o->kind == MTBC_OP_SET_STREAM) {
            if (o->u.sync.index < 0) { if (failed_index) *failed_index = i; return MTBC_E_BADARG; }
            stream = streams[o->u.sync.index < n_streams ? o->u.sync.index : 0];
            continue;
        }
        if (o->kind == MTBC_OP_EVENT_RECORD || o->kind == MTBC_OP_EVENT_WAIT) {
            hipError_t e = hipErrorInvalidValue;
            if (o->u.sync.event)
                e = o->kind == MTBC_OP_EVENT_RECORD ? hipEventRecord((hipEvent_t)o->u.sync.event, (hipStream_t)stream)
                                                    : hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)o->u.sync.event, 0);
            if (e != hipSuccess) { if (failed_index) *failed_index = i; return o->u.sync.event ? MTBC_E_LAUNCH : MTBC_E_BADARG; }
            continue;
        }
        if (is_pack(o->kind)) {               // a run of weight-image ops becomes one batched launch
            mtbc_pack_desc d[128];
            int32_t n = 0;
            while (n < 128 && i + n < first + count && is_pack(ops[i + n].kind)) {
                const mtbc_op* q = &ops[i + n];
                d[n].w = q->u.pack.w; d[n].packed = q->u.pack.packed; d[n].Cin = q->u.pack.Cin; d[n].Cout = q->u.pack.Cout;
                d[n].compute = q->u.pack.compute;
                d[n].kind = q->kind == MTBC_OP_CONV3_PACK_FWD ? 0 : q->kind == MTBC_OP_CONV3_PACK_DGRAD ? 1 : (q->u.pack.dgrad ? 3 : 2);
                ++n;
            }
            if (n > 1) {
                rc = mtbc_conv3x3_pack_many(d, n, stream);
                if (rc != MTBC_OK) { if (failed_index) *failed_index = i; return rc; }
                i += n - 1;
                continue;
            }
        }
        if (o->kind == MTBC_OP_CONV3_WVIEW) {          // ... and so does a run of weight views
            mtbc_wview_desc d[128];
            int32_t n = 0;
            while (n < 128 && i + n < first + count && ops[i + n].kind == MTBC_OP_CONV3_WVIEW) {
                const mtbc_op* q = &ops[i + n];
                d[n].w = q->u.wview.w; d[n].dst = q->u.wview.dst; d[n].Cout = q->u.wview.Cout; d[n].Cin = q->u.wview.Cin;
                d[n].ci_off = q->u.wview.ci_off; d[n].ci_cnt = q->u.wview.ci_cnt; d[n].mode = q->u.wview.mode;
                d[n].k_off = q->u.wview.k_off; d[n].K = q->u.wview.K;
                ++n;
            }
            if (n > 1) {
                rc = mtbc_conv3x3_weight_view_many(d, n, stream);
                if (rc != MTBC_OK) { if (failed_index) *failed_index = i; return rc; }
                i += n - 1;
                continue;
            }
        }
        if (o->kind == MTBC_OP_IN_DPARAM) {            // ... and a run of deferred InstanceNorm parameter-gradient reductions
            mtbc_dparam_desc d[64];
            int32_t n = 0;
            while (n < 64 && i + n < first + count && ops[i + n].kind == MTBC_OP_IN_DPARAM) { d[n] = ops[i + n].u.dparam; ++n; }
            rc = mtbc_instnorm_dparam_many(d, n, stream);
            if (rc != MTBC_OK) { if (failed_index) *failed_index = i; return rc; }
            i += n - 1;
            continue;
        }
        switch (o->kind) {
            case MTBC_OP_CONV3_FWD: rc = mtbc_conv3x3_fwd(&o->u.conv3, stream); break;
            case MTBC_OP_CONV3_DGRAD: rc = mtbc_conv3x3_dgrad(&o->u.conv3, stream); break;
            case MTBC_OP_CONV3_WGRAD: rc = mtbc_conv3x3_wgrad(&o->u.conv3, stream); break;
            case MTBC_OP_CONV3_PACK_FWD: rc = mtbc_conv3x3_pack_fwd(o->u.pack.w, o->u.pack.packed, o->u.pack.Cin, o->u.pack.Cout, stream); break;
            case MTBC_OP_CONV3_PACK_DGRAD: rc = mtbc_conv3x3_pack_dgrad(o->u.pack.w, o->u.pack.packed, o->u.pack.Cin, o->u.pack.Cout, stream); break;
            case MTBC_OP_HEAD_COMBINE: rc = mtbc_convT_head_combine(&o->u.head, stream); break;
            case MTBC_OP_HEAD_EXPAND: rc = mtbc_convT_head_expand(&o->u.head, stream); break;
            case MTBC_OP_CONV3_WVIEW: rc = mtbc_conv3x3_weight_view(o->u.wview.w, o->u.wview.dst, o->u.wview.Cout, o->u.wview.Cin, o->u.wview.ci_off, o->u.wview.ci_cnt, o->u.wview.mode, o->u.wview.k_off, o->u.wview.K, stream); break;
            case MTBC_OP_C8_PACK16: rc = mtbc_c8_pack16(o->u.c8pack.src, o->u.c8pack.src_batch_stride, o->u.c8pack.dst, o->u.c8pack.N, o->u.c8pack.C, o->u.c8pack.HW, stream); break;
            case MTBC_OP_C8_PACK: rc = mtbc_c8_pack(o->u.c8pack.src, o->u.c8pack.src_batch_stride, o->u.c8pack.dst, o->u.c8pack.N, o->u.c8pack.C, o->u.c8pack.HW, o->u.c8pack.compute, stream); break;
            case MTBC_OP_CONV3_PACK_LP: rc = mtbc_conv3x3_pack_lp(o->u.pack.w, o->u.pack.packed, o->u.pack.Cin, o->u.pack.Cout, o->u.pack.dgrad, o->u.pack.compute, stream); break;
            case MTBC_OP_IN_FWD: rc = mtbc_instnorm_lrelu_fwd(&o->u.inorm, stream); break;
            case MTBC_OP_IN_BWD: rc = mtbc_instnorm_lrelu_bwd(&o->u.inorm, stream); break;
            case MTBC_OP_POOL_FWD: rc = mtbc_maxpool2_fwd(&o->u.pool, stream); break;
            case MTBC_OP_POOL_BWD: rc = mtbc_maxpool2_bwd(&o->u.pool, stream); break;
            case MTBC_OP_CONVT_FWD: rc = mtbc_convT_fwd(&o->u.convT, stream); break;
            case MTBC_OP_CONVT_DGRAD: rc = mtbc_convT_dgrad(&o->u.convT, stream); break;
            case MTBC_OP_CONVT_WGRAD: rc = mtbc_convT_wgrad(&o->u.convT, stream); break;
            case MTBC_OP_CONV1_FWD: rc = mtbc_conv1x1_fwd(&o->u.conv1, stream); break;
            case MTBC_OP_CONV1_DGRAD: rc = mtbc_conv1x1_dgrad(&o->u.conv1, stream); break;
            case MTBC_OP_CONV1_WGRAD: rc = mtbc_conv1x1_wgrad(&o->u.conv1, stream); break;
            case MTBC_OP_GAP_FWD: rc = mtbc_gap_fwd(&o->u.gap, stream); break;
            case MTBC_OP_GAP_BWD: rc = mtbc_gap_bwd(&o->u.gap, stream); break;
            case MTBC_OP_LINEAR_FWD: rc = mtbc_linear_fwd(&o->u.linear, stream); break;
            case MTBC_OP_LINEAR_BWD: rc = mtbc_linear_bwd(&o->u.linear, stream); break;
            case MTBC_OP_DICE_FWD: rc = mtbc_dice_fwd(&o->u.dice, stream); break;
            case MTBC_OP_DICE_BWD: rc = mtbc_dice_bwd(&o->u.dice, stream); break;
            case MTBC_OP_FOCAL: rc = mtbc_focal_fwd_bwd(&o->u.focal, stream); break;
            case MTBC_OP_LOSS_MIX: rc = mtbc_loss_mix(o->u.mix.seg, o->u.mix.cls, o->u.mix.alpha, o->u.mix.out4, stream); break;
            case MTBC_OP_ADAM: rc = mtbc_adam_step(&o->u.adam, stream); break;
            case MTBC_OP_MEMSET:
                rc = hipMemsetAsync(o->u.memset0.ptr, 0, o->u.memset0.bytes, (hipStream_t)stream) == hipSuccess ? MTBC_OK : MTBC_E_LAUNCH;
                break;
            case MTBC_OP_DICE_COUNTS: rc = mtbc_dice_counts(o->u.counts.logits, o->u.counts.target, o->u.counts.n, o->u.counts.out3, stream); break;
            default: rc = MTBC_E_BADARG;
        }
        if (rc != MTBC_OK) {
            if (failed_index) *failed_index = i;
            return rc;
        }
    }
    return MTBC_OK;
}

}  // extern "C"
