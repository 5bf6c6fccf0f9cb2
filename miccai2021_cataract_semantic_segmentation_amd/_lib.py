"""ctypes binding of libcatseg_hip.so (C ABI declared in include/catseg.h).

There is no CPU fallback: importing this module without the built library raises.
PyTorch is used only to own device memory and streams; every kernel is launched
through the C ABI with raw pointers.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CATSEG_LIB") or os.path.join(_HERE, "libcatseg_hip.so")  # CATSEG_LIB: A/B builds of the library

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libcatseg_hip.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C miccai2021_cataract_semantic_segmentation_amd/csrc`. There is no CPU fallback." % LIB_PATH)

lib = C.CDLL(LIB_PATH)

P, I, L, F, SZ = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_size_t


class ConvDesc(C.Structure):
    _fields_ = [(n, I) for n in ("B", "H", "W", "Cin", "Ho", "Wo", "Cout", "kh", "kw", "stride", "pad", "dil",
                                 "ldx", "ldy", "stem4", "groups")]


_SIGS = {
    "catseg_last_error": (C.c_char_p, []),
    "catseg_version": (I, []),
    "catseg_conv2d_fwd": (I, [P, P, P, P, P, I, P]),
    "catseg_conv2d_fwd_bnstats": (I, [P, P, P, P, P, I, P, SZ, P, P, P]),
    "catseg_conv2d_fwd_bf16x3_bnstats": (I, [P, P, P, P, P, I, P, SZ, P, P, P]),
    "catseg_bn_finalize": (I, [P, I, L, L, I, P, F, F, P, P, P, P, P]),
    "catseg_bn_finalize_counts": (I, [P, I, P, L, I, P, F, F, P, P, P, P, P]),
    "catseg_dconv3_supported": (I, [I]),
    "catseg_dconv3_wimg_bytes": (SZ, [I]),
    "catseg_dconv3_tiles": (I, [I, I, I, I, P, P]),
    "catseg_dconv3_prep": (I, [P, I, I, P, P]),
    "catseg_dconv3_layout": (I, [I, P, P]),
    "catseg_dwgrad3_supported": (I, [I]),
    "catseg_dwgrad3_workspace": (SZ, [I, I, I, I]),
    "catseg_dwgrad3": (I, [I, I, I, I, P, I, P, I, P, P, SZ, P]),
    "catseg_debug_set_dwgrad3_blocks": (I, [I]),
    "catseg_dconv3_prep_batch": (I, [P, I, P, P, P]),
    "catseg_dconv3": (I, [I, I, I, I, P, I, P, P, P, I, I, P, SZ, P, P]),
    "catseg_dconv3_bnbwd": (I, [I, I, I, I, P, I, P, P, I, P, I, P, P, P, P, SZ, P]),
    "catseg_dwgrad3_f16x2": (I, [I, I, I, I, P, I, P, P, I, P, P, P, SZ, P]),
    "catseg_dconv3_f16x2_wimg_bytes": (SZ, [I]),
    "catseg_dconv3_f16x2_prep_batch": (I, [P, I, P, P, P, P]),
    "catseg_dconv3_f16x2": (I, [I, I, I, I, P, I, P, P, P, P, P, I, I, P, SZ, P, P]),
    "catseg_dconv3_bnbwd_f16x2": (I, [I, I, I, I, P, I, P, P, P, P, I, P, I, P, P, P, P, SZ, P]),
    "catseg_dconv3_pl_supported": (I, [I]),
    "catseg_dconv3_pl_rows": (I, [I, I, I, I]),
    "catseg_planes_bytes": (SZ, [L, I]),
    "catseg_planes_from_f32": (I, [P, I, L, I, P, P, I, P]),
    "catseg_dconv3_pl": (I, [I, I, I, I, P, P, P, P, P, P, I, I, P, SZ, P, P, P]),
    "catseg_dconv3_pl_bnbwd": (I, [I, I, I, I, P, P, P, P, P, I, P, I, P, P, P, P, SZ, P, P]),
    "catseg_dwgrad3_pl_supported": (I, [I]),
    "catseg_dwgrad3_pl_workspace": (SZ, [I, I, I, I]),
    "catseg_dwgrad3_pl": (I, [I, I, I, I, P, P, P, P, P, P, SZ, P]),
    "catseg_bn_finalize_counts_bound": (I, [P, I, P, L, I, P, P, F, F, P, P, P, P, P, P, P]),
    "catseg_bn_apply_planes": (I, [P, I, P, P, P, P, I, P, P, I, P, L, I, I, P, P]),
    "catseg_add_n_act_planes": (I, [P, P, P, I, P, I, P, L, I, I, P, P]),
    "catseg_bn_backward_planes": (I, [P, I, P, I, P, I, P, P, P, L, I, I, P, P, P, P, P, P, P, I, I, P, SZ, P]),
    "catseg_bn_backward_pre_planes": (I, [P, I, P, I, P, P, P, I, L, I, P, P, P, P, P, P, P, SZ, P]),
    "catseg_bn_backward_h2_workspace": (SZ, [L, I]),
    "catseg_bn_backward_h2": (I, [P, I, P, I, P, I, P, P, P, L, I, I, P, P, P, P, P, P, P, P, P, SZ, P]),
    "catseg_head_fwd": (I, [P, I, P, P, P, P, P, I, L, I, P, I, I, P]),
    "catseg_head_backward_workspace": (SZ, [L, I]),
    "catseg_head_backward": (I, [P, I, P, I, P, P, P, P, I, L, I, P, P, P, P, P, P, P, P, P, P, P, SZ, P]),
    "catseg_maxpool2x2_fwd": (I, [P, I, P, I, P, I, I, I, I, P]),
    "catseg_maxpool2x2_bwd": (I, [P, I, P, P, I, I, I, I, I, P]),
    "catseg_bias_rows": (I, [P, P, I, L, I, P]),
    "catseg_bn_apply_amax": (I, [P, I, P, P, P, P, I, P, I, L, I, I, P, P]),
    "catseg_bn_backward_amax": (I, [P, I, P, I, P, I, P, P, P, L, I, I, P, I, P, P, P, I, I, P, SZ, P, P]),
    "catseg_bn_mask_bytes": (SZ, [L, I]),
    "catseg_bn_apply_mask": (I, [P, I, P, P, P, P, I, P, I, L, I, P, P, P]),
    "catseg_bn_backward_mask": (I, [P, I, P, P, I, P, P, L, I, P, I, P, P, P, I, I, P, SZ, P, P]),
    "catseg_bn_apply_planes_mask": (I, [P, I, P, P, P, P, I, P, P, I, P, L, I, P, P, P]),
    "catseg_bn_backward_planes_mask": (I, [P, I, P, P, I, P, P, L, I, P, P, P, P, P, P, P, I, I, P, SZ, P]),
    "catseg_bn_backward_pre_amax": (I, [P, I, P, I, P, P, P, I, L, I, P, I, P, P, P, SZ, P, P]),
    "catseg_add_n_act_amax": (I, [P, P, I, P, I, L, I, I, P, P]),
    "catseg_conv2d_fwd_fused": (I, [P, P, P, P, P, I, I, P, P]),
    "catseg_fold_bn": (I, [P, P, P, P, P, P, F, I, I, P, P, P]),
    "catseg_conv2d_bwd_data": (I, [P, P, P, P, I, P]),
    "catseg_conv2d_bwd_weight_workspace": (SZ, [P]),
    "catseg_conv2d_bwd_weight": (I, [P, P, P, P, P, P, SZ, P]),
    "catseg_gemm_batched": (I, [I, I, I, I, I, P, I, L, P, I, L, P, I, L, I, I, P]),
    "catseg_sum_slabs": (I, [P, P, L, I, I, I, P]),
    "catseg_debug_set_tile": (I, [I, I]),
    "catseg_debug_set_splits": (I, [I]),
    "catseg_debug_set_strided_multi": (I, [I]),
    "catseg_debug_set_lovasz_prune": (I, [I]),
    "catseg_debug_set_wgrad_direct": (I, [I]),
    "catseg_debug_plan_conv": (I, [P, I, P]),
    "catseg_debug_set_b3_tile": (I, [I]),
    "catseg_debug_set_dconv3_blocks": (I, [I]),
    "catseg_debug_set_dconv3_pl_slots": (I, [I]),
    "catseg_debug_set_dconv3_pl_pair": (I, [I]),
    "catseg_debug_set_dwgrad3_pl_blocks": (I, [I]),
    "catseg_debug_dconv3_pl_occupancy": (I, [I, I]),
    "catseg_debug_dwgrad3_pl_occupancy": (I, [I]),
    "catseg_debug_set_dconv3_spec": (I, [I]),
    "catseg_debug_set_dconv3_alt96": (I, [I]),
    "catseg_aug_pad_flip_u8": (I, [P, P, I, I, I, I, P, I, I, P]),
    "catseg_aug_box_blur": (I, [P, P, I, I, I, I, P, P, P, P]),
    "catseg_aug_color_op": (I, [P, I, I, I, P, P, P, SZ, P]),
    "catseg_split3_elems": (SZ, [L, I]),
    "catseg_split3": (I, [P, I, L, I, P, P]),
    "catseg_split3_weight_t": (I, [P, I, I, I, P, P]),
    "catseg_split3_blocked_elems": (SZ, [L, I]),
    "catseg_split3_blocked": (I, [P, L, I, I, P, P, P]),
    "catseg_split3_weight_blocked": (I, [P, I, I, I, P, P]),
    "catseg_split3_weight_t_blocked": (I, [P, I, I, I, P, P]),
    "catseg_conv2d_fwd_bf16x3_blocked": (I, [P, P, P, P, P, I, P, SZ, P, P, P]),
    "catseg_conv2d_bwd_data_bf16x3_blocked": (I, [P, P, P, P, I, P]),
    "catseg_conv2d_fwd_fused_bf16x3_blocked": (I, [P, P, P, P, P, I, I, P, P]),
    "catseg_split2h_blocked_elems": (SZ, [L, I]),
    "catseg_split2h_planar_elems": (SZ, [L, I]),
    "catseg_split2h": (I, [P, L, I, I, P, P, P, P]),
    "catseg_split2h_bound": (I, [P, L, I, I, P, P, P, P, P, P, P, P]),
    "catseg_conv2d_bwd_weight_f16x2_workspace": (SZ, [P]),
    "catseg_conv2d_bwd_weight_f16x2": (I, [P, P, P, P, P, P, P, SZ, P]),
    "catseg_conv2d_bwd_weight_f16x2_blocked": (I, [P, P, P, P, P, P, P, SZ, P]),
    "catseg_concat_bilinear_split2h": (I, [I, P, P, P, P, P, P, I, I, I, P, P, P]),
    "catseg_stem3_supported": (I, [I, I, I]),
    "catseg_stem3_partial_rows": (I, [I, I, I]),
    "catseg_stem3_wgrad_workspace": (SZ, []),
    "catseg_stem3_fwd": (I, [P, L, L, L, L, I, I, I, P, P, P, I, P, P, P]),
    "catseg_stem3_bwd_weight": (I, [P, L, L, L, L, I, I, I, P, I, P, P, SZ, P]),
    "catseg_stem7_supported": (I, [I, I, I]),
    "catseg_stem7_partial_rows": (I, [I, I, I]),
    "catseg_stem7_fwd": (I, [P, L, L, L, L, I, I, I, P, P, P, I, P, P, P]),
    "catseg_split2h_weight_blocked": (I, [P, I, I, I, P, P, P]),
    "catseg_split2h_weight_t_blocked": (I, [P, I, I, I, P, P, P]),
    "catseg_conv2d_fwd_f16x2_blocked": (I, [P, P, P, P, P, P, P, I, P, SZ, P, P, P]),
    "catseg_conv2d_fwd_fused_f16x2_blocked": (I, [P, P, P, P, P, P, P, I, I, P, P]),
    "catseg_conv2d_bwd_data_f16x2_blocked": (I, [P, P, P, P, P, P, I, P]),
    "catseg_conv2d_fwd_bf16x3": (I, [P, P, P, P, P, I, P]),
    "catseg_conv2d_bwd_data_bf16x3": (I, [P, P, P, P, I, P]),
    "catseg_conv2d_bwd_weight_bf16x3_workspace": (SZ, [P]),
    "catseg_conv2d_bwd_weight_bf16x3": (I, [P, P, P, P, P, SZ, P]),
    "catseg_bias_grad": (I, [P, I, L, I, P, P, SZ, P]),
    "catseg_bn_workspace": (SZ, [L, I]),
    "catseg_bn_train_stats": (I, [P, L, I, I, P, F, F, P, P, P, P, P, SZ, P]),
    "catseg_bn_eval_scale": (I, [I, P, P, F, P, P]),
    "catseg_bn_apply": (I, [P, I, P, P, P, P, I, P, I, L, I, I, P]),
    "catseg_bn_backward": (I, [P, I, P, I, P, I, P, P, P, L, I, I, P, I, P, P, P, I, I, P, SZ, P]),
    "catseg_bn_backward_pre": (I, [P, I, P, I, P, P, P, I, L, I, P, I, P, P, P, SZ, P]),
    "catseg_nchw3_to_nhwc4": (I, [P, P, I, I, I, P]),
    "catseg_stem_pack_weight": (I, [P, P, I, P]),
    "catseg_stem_unpack_grad": (I, [P, P, I, P]),
    "catseg_axpy2d": (I, [P, I, P, I, L, I, F, I, P]),
    "catseg_add_n_act": (I, [P, P, I, P, I, L, I, I, P]),
    "catseg_relu_bwd": (I, [P, I, P, I, P, I, L, I, P]),
    "catseg_weight_pad_cin": (I, [P, P, I, I, I, I, I, P]),
    "catseg_scale_by_device_scalar": (I, [P, L, P, P]),
    "catseg_maxpool3x3s2_fwd": (I, [P, I, P, I, P, I, I, I, I, I, I, P]),
    "catseg_maxpool3x3s2_bwd": (I, [P, I, P, P, I, I, I, I, I, I, I, P]),
    "catseg_bilinear_fwd": (I, [P, I, P, I, I, I, I, I, I, I, I, I, P]),
    "catseg_bilinear_bwd": (I, [P, I, P, I, I, I, I, I, I, I, I, I, I, P, SZ, P]),
    "catseg_global_avgpool_fwd": (I, [P, I, P, I, I, I, P]),
    "catseg_global_avgpool_bwd": (I, [P, P, I, I, I, I, I, P]),
    "catseg_adaptive_avgpool_fwd": (I, [P, I, P, I, I, I, I, I, P]),
    "catseg_adaptive_avgpool_bwd": (I, [P, P, I, I, I, I, I, I, I, P]),
    "catseg_softmax_spatial_workspace": (SZ, [I, I]),
    "catseg_softmax_spatial_fwd": (I, [P, P, I, I, I, I, P, SZ, P]),
    "catseg_softmax_spatial_bwd": (I, [P, P, P, I, I, I, I, I, P, SZ, P]),
    "catseg_softmax_rows_fwd": (I, [P, P, L, I, I, F, P]),
    "catseg_softmax_rows_bwd": (I, [P, P, P, L, I, I, F, P]),
    "catseg_lovasz_workspace": (SZ, [L, I]),
    "catseg_lovasz_softmax": (I, [P, P, L, I, F, P, P, I, P, SZ, P]),
    "catseg_lovasz_softmax_fwd": (I, [P, P, L, I, F, P, I, P, SZ, P]),
    "catseg_lovasz_softmax_bwd": (I, [P, L, I, F, P, P, I, P, SZ, P]),
    "catseg_ce_workspace": (SZ, [L]),
    "catseg_cross_entropy": (I, [P, P, L, I, L, F, P, P, P, SZ, P]),
    "catseg_ohem_workspace": (SZ, [L]),
    "catseg_ohem_cross_entropy": (I, [P, P, L, I, L, F, L, F, P, P, P, SZ, P]),
    "catseg_ingest_u8": (I, [P, P, I, I, I, P, P, I, I, P, P, P, P, P, P]),
    "catseg_resize_nearest": (I, [P, I, P, I, I, I, I, I, I, I, I, I, F, P]),
    "catseg_confusion_matrix": (I, [P, P, L, I, P, P]),
    "catseg_adam_step": (I, [P, P, P, P, L, F, F, F, F, I, F, P]),
    "catseg_pconv1_supported": (I, [I, I]),
    "catseg_pconv1_wimg_bytes": (SZ, [I, I]),
    "catseg_pconv1_prep_batch": (I, [P, I, P, P, P, P]),
    "catseg_pconv1": (I, [L, I, I, P, I, P, P, P, P, P, I, I, P, SZ, P, P, P]),
    "catseg_pconv1_wgrad_supported": (I, [I, I]),
    "catseg_pconv1_wgrad_workspace": (SZ, [L, I, I]),
    "catseg_pconv1_wgrad": (I, [L, I, I, P, I, P, P, I, P, P, P, SZ, P]),
    "catseg_debug_set_pconv1_wgrad_blocks": (I, [I]),
    "catseg_debug_set_h2w_waves": (I, [I]),
    "catseg_debug_set_h2w_slow_epilogue": (I, [I]),
    "catseg_debug_set_bilinear_bwd_fused": (I, [I]),
    "catseg_gconv_supported": (I, [P]),
    "catseg_gconv_class_bytes": (SZ, [P]),
    "catseg_gconv_wimg_bytes": (SZ, [P, I]),
    "catseg_gconv_entries": (I, [P, I, L, L, P]),
    "catseg_gconv_fwd": (I, [P, P, P, P, P, P, P, P, SZ, P, P, P]),
    "catseg_gconv_bwd_data": (I, [P, P, P, P, P, P, I, P]),
    "catseg_gconv_wgrad_supported": (I, [P]),
    "catseg_gconv_wgrad_workspace": (SZ, [P]),
    "catseg_gconv_bwd_weight": (I, [P, P, P, P, P, P, P, SZ, P]),
    "catseg_adam_hyper": (None, [F, F, F, I, F, P]),
    "catseg_adam_step_dev": (I, [P, P, P, P, L, P, F, F, F, P]),
}
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)  # AttributeError here = header / library mismatch
    _fn.restype = _res
    _fn.argtypes = _args

EXPORTS = tuple(_SIGS)

if os.environ.get("CATSEG_SYNC"):  # debugging aid: synchronise after every entry point so a device fault names its launch
    class _SyncLib:
        def __init__(self, inner):
            self._inner = inner

        def __getattr__(self, name):
            fn = getattr(self._inner, name)
            if name in ("catseg_last_error", "catseg_version") or name.endswith("_workspace"):
                return fn

            def call(*a):
                desc = [[getattr(x._obj, f) for f, _ in x._obj._fields_] for x in a if isinstance(getattr(x, "_obj", None), ConvDesc)]
                print("[catseg]", name, desc, [x for x in a if isinstance(x, (int, float))], flush=True)
                rc = fn(*a)
                torch.cuda.synchronize()
                return rc
            return call

    lib = _SyncLib(lib)


class CatsegError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise CatsegError("libcatseg_hip: error %d: %s" % (rc, lib.catseg_last_error().decode()))


def stream():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return 0 if t is None else t.data_ptr()
