"""ctypes binding of libmtbc_hip.so (the C-ABI declared in include/mtbc.h).

The product path has NO CPU fallback: if the shared library is missing or a call
returns an error code this module raises, loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MTBC_LIB") or os.path.join(_HERE, "libmtbc_hip.so")   # MTBC_LIB: A/B builds of the same ABI

MAX_SEGS = 6
c_float_p = C.POINTER(C.c_float)


class MtbcError(RuntimeError):
    pass


LAYOUT_PLANAR, LAYOUT_C8 = 0, 1


class Seg(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("batch_stride", C.c_int64), ("channels", C.c_int32), ("accumulate", C.c_int32)]


class Conv3x3Args(C.Structure):
    _fields_ = [("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32),
                ("n_in", C.c_int32), ("in_", Seg * MAX_SEGS),
                ("w", C.c_void_p), ("w_packed", C.c_void_p), ("bias", C.c_void_p),
                ("out", C.c_void_p), ("dout", C.c_void_p), ("dw", C.c_void_p), ("dbias", C.c_void_p),
                ("accumulate_dw", C.c_int32), ("force_direct", C.c_int32),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("compute", C.c_int32), ("operand_layout", C.c_int32), ("out_accumulate", C.c_int32), ("out_layout", C.c_int32),
                ("stats_partial", C.c_void_p), ("out_partial", C.c_void_p), ("norm_z", C.c_void_p), ("norm_mean", C.c_void_p),
                ("norm_rstd", C.c_void_p), ("norm_gamma", C.c_void_p), ("norm_beta", C.c_void_p), ("norm_slope", C.c_float),
                ("out_type", C.c_int32), ("wgrad_sync", C.c_void_p), ("wgrad_sync_bytes", C.c_size_t)]


class InstNormArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("eps", C.c_float), ("slope", C.c_float),
                ("z", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("y", C.c_void_p), ("y_batch_stride", C.c_int64),
                ("mean", C.c_void_p), ("rstd", C.c_void_p),
                ("dy", C.c_void_p), ("dy_batch_stride", C.c_int64),
                ("n_dy_extra", C.c_int32), ("dy_extra", C.c_void_p * 4),
                ("dz", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("dbias_pre", C.c_void_p),
                ("accumulate_dparams", C.c_int32),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("y16", C.c_void_p), ("dz16", C.c_void_p), ("out16_type", C.c_int32),
                ("y8", C.c_void_p), ("dz8", C.c_void_p), ("coop_state", C.c_void_p), ("coop_reserve_cus", C.c_int32),
                ("z_layout", C.c_int32), ("dy_layout", C.c_int32), ("stats_partial", C.c_void_p), ("stats_slots", C.c_int32),
                ("z_type", C.c_int32), ("dy_rank1", C.c_void_p), ("dy_rank1_w", C.c_void_p),
                ("dy_rank1_dw", C.c_void_p), ("dy_rank1_db", C.c_void_p), ("dy_rank1_accumulate", C.c_int32),
                ("dy_pool", C.c_void_p), ("dy_pool_arg", C.c_void_p), ("pool_y8", C.c_void_p), ("pool_arg", C.c_void_p),
                ("defer_dparams", C.c_int32)]


class MaxPoolArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("x", C.c_void_p), ("x_batch_stride", C.c_int64),
                ("y", C.c_void_p), ("y_batch_stride", C.c_int64),
                ("dy", C.c_void_p), ("dy_batch_stride", C.c_int64),
                ("dx", C.c_void_p), ("dx_batch_stride", C.c_int64),
                ("accumulate_dx", C.c_int32), ("layout", C.c_int32), ("type16", C.c_int32), ("argmax", C.c_void_p)]


class ConvTArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32),
                ("k", C.c_int32),
                ("x", C.c_void_p), ("x_batch_stride", C.c_int64),
                ("w", C.c_void_p), ("bias", C.c_void_p),
                ("y", C.c_void_p), ("y_batch_stride", C.c_int64),
                ("dy", C.c_void_p), ("dy_batch_stride", C.c_int64),
                ("dx", C.c_void_p), ("dx_batch_stride", C.c_int64),
                ("accumulate_dx", C.c_int32),
                ("dw", C.c_void_p), ("dbias", C.c_void_p), ("accumulate_dw", C.c_int32),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("compute", C.c_int32),
                ("y_layout", C.c_int32), ("y_type", C.c_int32), ("x_layout", C.c_int32), ("dy_type16", C.c_int32), ("x_type16", C.c_int32)]


class Conv1x1Args(C.Structure):
    _fields_ = [("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32),
                ("x", C.c_void_p), ("x_batch_stride", C.c_int64),
                ("w", C.c_void_p), ("bias", C.c_void_p), ("y", C.c_void_p), ("dy", C.c_void_p),
                ("dx", C.c_void_p), ("dx_batch_stride", C.c_int64), ("accumulate_dx", C.c_int32),
                ("dw", C.c_void_p), ("dbias", C.c_void_p), ("accumulate_dw", C.c_int32),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("x_layout", C.c_int32), ("x_type", C.c_int32)]


class GapArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("x", C.c_void_p), ("y", C.c_void_p), ("dy", C.c_void_p), ("dx", C.c_void_p)]


class LinearArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("In", C.c_int32), ("Out", C.c_int32), ("relu", C.c_int32),
                ("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("y", C.c_void_p),
                ("dy", C.c_void_p), ("dx", C.c_void_p), ("dw", C.c_void_p), ("dbias", C.c_void_p),
                ("accumulate_dw", C.c_int32), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class DiceArgs(C.Structure):
    _fields_ = [("n_heads", C.c_int32), ("N", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("smooth_nr", C.c_float), ("smooth_dr", C.c_float),
                ("x", C.c_void_p * 4), ("target", C.c_void_p), ("head_weight", C.c_float * 4),
                ("stats", C.c_void_p), ("loss", C.c_void_p), ("dx", C.c_void_p * 4),
                ("gscale", C.c_float), ("gscale_dev", C.c_void_p)]


class FocalArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("C", C.c_int32), ("alpha", C.c_float), ("gamma", C.c_float),
                ("x", C.c_void_p), ("target", C.c_void_p), ("weight", C.c_void_p),
                ("loss", C.c_void_p), ("dx", C.c_void_p), ("gscale", C.c_float), ("gscale_dev", C.c_void_p)]


class AdamArgs(C.Structure):
    _fields_ = [("n", C.c_int64), ("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("grad_scale", C.c_float), ("step", C.c_int32), ("zero_grad", C.c_int32), ("dynamic", C.c_void_p)]


class _PackArgs(C.Structure):
    _fields_ = [("w", C.c_void_p), ("packed", C.c_void_p), ("Cin", C.c_int32), ("Cout", C.c_int32),
                ("dgrad", C.c_int32), ("compute", C.c_int32)]


class _MixArgs(C.Structure):
    _fields_ = [("seg", C.c_void_p), ("cls", C.c_void_p), ("alpha", C.c_float), ("out4", C.c_void_p)]


class _MemsetArgs(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("bytes", C.c_size_t)]


class _CountsArgs(C.Structure):
    _fields_ = [("logits", C.c_void_p), ("target", C.c_void_p), ("n", C.c_int64), ("out3", C.c_void_p)]


class _C8PackArgs(C.Structure):
    _fields_ = [("src", C.c_void_p), ("src_batch_stride", C.c_int64), ("dst", C.c_void_p),
                ("N", C.c_int32), ("C", C.c_int32), ("HW", C.c_int32), ("compute", C.c_int32)]


class DparamDesc(C.Structure):
    """mtbc_dparam_desc (include/mtbc.h)."""
    _fields_ = [("part", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("dbias_pre", C.c_void_p),
                ("N", C.c_int32), ("C", C.c_int32), ("T", C.c_int32), ("accumulate", C.c_int32)]


class _SyncArgs(C.Structure):
    _fields_ = [("event", C.c_void_p), ("index", C.c_int32)]


class _WViewArgs(C.Structure):
    _fields_ = [("w", C.c_void_p), ("dst", C.c_void_p), ("Cout", C.c_int32), ("Cin", C.c_int32), ("ci_off", C.c_int32),
                ("ci_cnt", C.c_int32), ("mode", C.c_int32), ("k_off", C.c_int32), ("K", C.c_int32)]


class HeadFuseArgs(C.Structure):
    """mtbc_head_fuse_args (include/mtbc.h)."""
    _fields_ = [("Cin", C.c_int32), ("Cmid", C.c_int32), ("R", C.c_int32), ("k", C.c_int32),
                ("wT", C.c_void_p), ("bT", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p),
                ("Wc", C.c_void_p), ("bc", C.c_void_p), ("G", C.c_void_p), ("gb", C.c_void_p),
                ("dwT", C.c_void_p), ("dbT", C.c_void_p), ("dw1", C.c_void_p), ("db1", C.c_void_p),
                ("acc_wT", C.c_int32), ("acc_bT", C.c_int32), ("acc_w1", C.c_int32), ("acc_b1", C.c_int32)]


class _OpUnion(C.Union):
    _fields_ = [("conv3", Conv3x3Args), ("inorm", InstNormArgs), ("pool", MaxPoolArgs), ("convT", ConvTArgs),
                ("conv1", Conv1x1Args), ("gap", GapArgs), ("linear", LinearArgs), ("dice", DiceArgs),
                ("focal", FocalArgs), ("adam", AdamArgs), ("pack", _PackArgs), ("mix", _MixArgs),
                ("memset0", _MemsetArgs), ("counts", _CountsArgs), ("head", HeadFuseArgs), ("c8pack", _C8PackArgs), ("wview", _WViewArgs), ("dparam", DparamDesc), ("sync", _SyncArgs)]


class Op(C.Structure):
    _fields_ = [("kind", C.c_int32), ("tag", C.c_int32), ("u", _OpUnion)]


# op kinds -- keep in sync with the enum in include/mtbc.h
(OP_CONV3_FWD, OP_CONV3_DGRAD, OP_CONV3_WGRAD, OP_CONV3_PACK_FWD, OP_CONV3_PACK_DGRAD,
 OP_IN_FWD, OP_IN_BWD, OP_POOL_FWD, OP_POOL_BWD, OP_CONVT_FWD, OP_CONVT_DGRAD, OP_CONVT_WGRAD,
 OP_CONV1_FWD, OP_CONV1_DGRAD, OP_CONV1_WGRAD, OP_GAP_FWD, OP_GAP_BWD, OP_LINEAR_FWD, OP_LINEAR_BWD,
 OP_DICE_FWD, OP_DICE_BWD, OP_FOCAL, OP_LOSS_MIX, OP_ADAM, OP_MEMSET, OP_DICE_COUNTS, OP_CONV3_PACK_LP,
 OP_HEAD_COMBINE, OP_HEAD_EXPAND, OP_C8_PACK, OP_C8_PACK16, OP_CONV3_WVIEW,
 OP_SET_STREAM, OP_EVENT_RECORD, OP_EVENT_WAIT, OP_IN_DPARAM) = range(1, 37)

OP_UNION_FIELD = {
    OP_CONV3_FWD: "conv3", OP_CONV3_DGRAD: "conv3", OP_CONV3_WGRAD: "conv3",
    OP_CONV3_PACK_FWD: "pack", OP_CONV3_PACK_DGRAD: "pack",
    OP_IN_FWD: "inorm", OP_IN_BWD: "inorm", OP_POOL_FWD: "pool", OP_POOL_BWD: "pool",
    OP_CONVT_FWD: "convT", OP_CONVT_DGRAD: "convT", OP_CONVT_WGRAD: "convT",
    OP_CONV1_FWD: "conv1", OP_CONV1_DGRAD: "conv1", OP_CONV1_WGRAD: "conv1",
    OP_GAP_FWD: "gap", OP_GAP_BWD: "gap", OP_LINEAR_FWD: "linear", OP_LINEAR_BWD: "linear",
    OP_DICE_FWD: "dice", OP_DICE_BWD: "dice", OP_FOCAL: "focal", OP_LOSS_MIX: "mix", OP_ADAM: "adam",
    OP_MEMSET: "memset0", OP_DICE_COUNTS: "counts", OP_CONV3_PACK_LP: "pack",
    OP_HEAD_COMBINE: "head", OP_HEAD_EXPAND: "head", OP_C8_PACK: "c8pack", OP_C8_PACK16: "c8pack", OP_CONV3_WVIEW: "wview",
    OP_SET_STREAM: "sync", OP_EVENT_RECORD: "sync", OP_EVENT_WAIT: "sync", OP_IN_DPARAM: "dparam",
}

# every symbol include/mtbc.h declares (tests check the library exports all of them)
class PackDesc(C.Structure):
    """mtbc_pack_desc (include/mtbc.h)."""
    _fields_ = [("w", C.c_void_p), ("packed", C.c_void_p), ("Cin", C.c_int32), ("Cout", C.c_int32),
                ("kind", C.c_int32), ("compute", C.c_int32)]


class WViewDesc(C.Structure):
    """mtbc_wview_desc (include/mtbc.h)."""
    _fields_ = [("w", C.c_void_p), ("dst", C.c_void_p), ("Cout", C.c_int32), ("Cin", C.c_int32), ("ci_off", C.c_int32),
                ("ci_cnt", C.c_int32), ("mode", C.c_int32), ("k_off", C.c_int32), ("K", C.c_int32)]


EXPORTS = [
    "mtbc_version", "mtbc_strerror", "mtbc_arch",
    "mtbc_conv3x3_packed_elems", "mtbc_conv3x3_packed_dgrad_elems", "mtbc_conv3x3_pack_fwd",
    "mtbc_conv3x3_pack_dgrad", "mtbc_conv3x3_packed_lp_elems", "mtbc_conv3x3_pack_lp", "mtbc_conv3x3_pack_many", "mtbc_c8_pack", "mtbc_c8_unpack", "mtbc_c8_pack16", "mtbc_conv3x3_weight_view", "mtbc_conv3x3_weight_view_many", "mtbc_augment_flip_rotate", "mtbc_convT_head_combine", "mtbc_convT_head_expand", "mtbc_conv3x3_wgrad_workspace", "mtbc_conv3x3_wgrad_sync_bytes", "mtbc_conv3x3_stats_slots", "mtbc_conv3x3_fwd", "mtbc_conv3x3_dgrad",
    "mtbc_conv3x3_wgrad", "mtbc_instnorm_fwd_workspace", "mtbc_instnorm_coop_state_bytes", "mtbc_instnorm_coop_error_offset", "mtbc_instnorm_c8_supported", "mtbc_instnorm_bwd_team", "mtbc_instnorm_dparam_many", "mtbc_instnorm_lrelu_fwd", "mtbc_instnorm_lrelu_bwd", "mtbc_maxpool2_fwd",
    "mtbc_maxpool2_bwd", "mtbc_convT_wgrad_workspace", "mtbc_convT_fwd_c8_supported", "mtbc_convT_fwd", "mtbc_convT_dgrad", "mtbc_convT_wgrad",
    "mtbc_conv1x1_wgrad_workspace", "mtbc_conv1x1_fwd", "mtbc_conv1x1_dgrad", "mtbc_conv1x1_wgrad",
    "mtbc_gap_fwd", "mtbc_gap_bwd", "mtbc_linear_fwd", "mtbc_linear_bwd", "mtbc_dice_fwd", "mtbc_dice_bwd",
    "mtbc_focal_fwd_bwd", "mtbc_loss_mix", "mtbc_adam_step", "mtbc_adam_dynamic", "mtbc_dice_counts", "mtbc_program_run",
    "mtbc_program_run_ms", "mtbc_event_create", "mtbc_event_destroy",
]

ABI_VERSION = 202          # MTBC_VERSION of include/mtbc.h these mirrors follow
_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libmtbc_hip.so; raise MtbcError (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MtbcError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (or `make -C multi_task_breast_cancer_amd/csrc`). There is no CPU fallback.")
    # torch FIRST: it ships its own HIP runtime (torch/lib/libamdhip64.so) and the device memory / streams handed to this library come
    # from it.  Loaded before torch, libmtbc_hip.so would pull in the system runtime (/opt/rocm/lib) instead and the process would hold
    # two of them: kernels registered with one, streams created by the other -> every launch fails (seen as "kernel launch failed"
    # on the first op when __graft_entry__.build() and smoke() ran in one process).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    lib.mtbc_version.restype = C.c_int
    if lib.mtbc_version() != ABI_VERSION:        # the ctypes mirrors below are one struct layout: a stale library would read garbage
        raise MtbcError(f"{LIB_PATH} reports ABI version {lib.mtbc_version()}, this binding is written for {ABI_VERSION}: rebuild "
                        f"(`make -C multi_task_breast_cancer_amd/csrc`)")
    lib.mtbc_strerror.restype = C.c_char_p
    lib.mtbc_strerror.argtypes = [C.c_int]
    lib.mtbc_arch.restype = C.c_char_p
    for name in ("mtbc_conv3x3_packed_elems", "mtbc_conv3x3_packed_dgrad_elems"):
        getattr(lib, name).restype = C.c_size_t
        getattr(lib, name).argtypes = [C.c_int32, C.c_int32]
    lib.mtbc_conv3x3_packed_lp_elems.restype = C.c_size_t
    lib.mtbc_conv3x3_packed_lp_elems.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    for _n in ("mtbc_convT_head_combine", "mtbc_convT_head_expand"):
        getattr(lib, _n).restype = C.c_int
        getattr(lib, _n).argtypes = [C.POINTER(HeadFuseArgs), C.c_void_p]
    lib.mtbc_conv3x3_stats_slots.restype = C.c_int32
    lib.mtbc_conv3x3_stats_slots.argtypes = [C.POINTER(Conv3x3Args)]
    lib.mtbc_instnorm_fwd_workspace.restype = C.c_size_t
    lib.mtbc_instnorm_fwd_workspace.argtypes = [C.POINTER(InstNormArgs)]
    lib.mtbc_augment_flip_rotate.restype = C.c_int
    lib.mtbc_augment_flip_rotate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.mtbc_conv3x3_pack_many.restype = C.c_int
    lib.mtbc_conv3x3_pack_many.argtypes = [C.POINTER(PackDesc), C.c_int32, C.c_void_p]
    lib.mtbc_c8_pack.restype = C.c_int
    lib.mtbc_c8_pack.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.mtbc_conv3x3_weight_view.restype = C.c_int
    lib.mtbc_conv3x3_weight_view.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int32] * 7 + [C.c_void_p]
    lib.mtbc_conv3x3_weight_view_many.restype = C.c_int
    lib.mtbc_conv3x3_weight_view_many.argtypes = [C.POINTER(WViewDesc), C.c_int32, C.c_void_p]
    lib.mtbc_c8_pack16.restype = C.c_int
    lib.mtbc_c8_pack16.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.mtbc_c8_unpack.restype = C.c_int
    lib.mtbc_c8_unpack.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.mtbc_conv3x3_pack_lp.restype = C.c_int
    lib.mtbc_conv3x3_pack_lp.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    for name in ("mtbc_conv3x3_pack_fwd", "mtbc_conv3x3_pack_dgrad"):
        getattr(lib, name).restype = C.c_int
        getattr(lib, name).argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    for name, typ in (("mtbc_conv3x3_wgrad_workspace", Conv3x3Args), ("mtbc_conv3x3_wgrad_sync_bytes", Conv3x3Args), ("mtbc_convT_wgrad_workspace", ConvTArgs),
                      ("mtbc_conv1x1_wgrad_workspace", Conv1x1Args)):
        getattr(lib, name).restype = C.c_size_t
        getattr(lib, name).argtypes = [C.POINTER(typ)]
    for name, typ in (("mtbc_conv3x3_fwd", Conv3x3Args), ("mtbc_conv3x3_dgrad", Conv3x3Args),
                      ("mtbc_conv3x3_wgrad", Conv3x3Args), ("mtbc_instnorm_lrelu_fwd", InstNormArgs),
                      ("mtbc_instnorm_lrelu_bwd", InstNormArgs), ("mtbc_maxpool2_fwd", MaxPoolArgs),
                      ("mtbc_maxpool2_bwd", MaxPoolArgs), ("mtbc_convT_fwd", ConvTArgs),
                      ("mtbc_convT_dgrad", ConvTArgs), ("mtbc_convT_wgrad", ConvTArgs),
                      ("mtbc_conv1x1_fwd", Conv1x1Args), ("mtbc_conv1x1_dgrad", Conv1x1Args),
                      ("mtbc_conv1x1_wgrad", Conv1x1Args), ("mtbc_gap_fwd", GapArgs), ("mtbc_gap_bwd", GapArgs),
                      ("mtbc_linear_fwd", LinearArgs), ("mtbc_linear_bwd", LinearArgs),
                      ("mtbc_dice_fwd", DiceArgs), ("mtbc_dice_bwd", DiceArgs), ("mtbc_focal_fwd_bwd", FocalArgs),
                      ("mtbc_adam_step", AdamArgs)):
        getattr(lib, name).restype = C.c_int
        getattr(lib, name).argtypes = [C.POINTER(typ), C.c_void_p]
    lib.mtbc_instnorm_coop_state_bytes.restype = C.c_size_t
    lib.mtbc_instnorm_coop_state_bytes.argtypes = []
    lib.mtbc_instnorm_coop_error_offset.restype = C.c_size_t
    lib.mtbc_instnorm_coop_error_offset.argtypes = []
    lib.mtbc_instnorm_bwd_team.restype = C.c_int32
    lib.mtbc_instnorm_bwd_team.argtypes = [C.POINTER(InstNormArgs)]
    lib.mtbc_instnorm_dparam_many.restype = C.c_int
    lib.mtbc_instnorm_dparam_many.argtypes = [C.POINTER(DparamDesc), C.c_int32, C.c_void_p]
    lib.mtbc_instnorm_c8_supported.restype = C.c_int
    lib.mtbc_instnorm_c8_supported.argtypes = [C.POINTER(InstNormArgs), C.c_int32]
    lib.mtbc_convT_fwd_c8_supported.restype = C.c_int
    lib.mtbc_convT_fwd_c8_supported.argtypes = [C.POINTER(ConvTArgs)]
    lib.mtbc_loss_mix.restype = C.c_int
    lib.mtbc_loss_mix.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
    lib.mtbc_dice_counts.restype = C.c_int
    lib.mtbc_dice_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.mtbc_adam_dynamic.restype = C.c_int
    lib.mtbc_adam_dynamic.argtypes = [C.POINTER(AdamArgs), C.POINTER(C.c_float * 3)]
    lib.mtbc_program_run.restype = C.c_int
    lib.mtbc_program_run.argtypes = [C.POINTER(Op), C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_int32)]
    lib.mtbc_program_run_ms.restype = C.c_int
    lib.mtbc_program_run_ms.argtypes = [C.POINTER(Op), C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_int32)]
    lib.mtbc_event_create.restype = C.c_int
    lib.mtbc_event_create.argtypes = [C.POINTER(C.c_void_p)]
    lib.mtbc_event_destroy.restype = C.c_int
    lib.mtbc_event_destroy.argtypes = [C.c_void_p]
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().mtbc_strerror(rc).decode()
        raise MtbcError(f"libmtbc_hip: {what or 'call'} failed: {msg} ({rc})")


def require_gpu() -> None:
    import torch
    if not torch.cuda.is_available():
        raise MtbcError("no HIP device visible: the multi-task training path runs on MI355X only "
                        "(there is no CPU fallback in the product path)")
