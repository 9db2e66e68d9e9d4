"""ctypes binding of ``libesam3_hip.so`` (C ABI declared in ``include/esam3.h``).

There is deliberately no fallback: if the shared library is missing or fails to load the
import raises -- the product path never routes through PyTorch CPU/eager code.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libesam3_hip.so")

ESAM3_F32 = 0
ESAM3_BF16 = 1


class Esam3Error(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("dtype", C.c_int), ("backbone", C.c_int), ("model_name", C.c_char * 16),
                ("device", C.c_int), ("interactive", C.c_int), ("fuse_linear_chains", C.c_int)]


class ImageFeatures(C.Structure):
    _fields_ = [("sam3_fpn_dev", C.c_void_p * 3), ("sam2_fpn_dev", C.c_void_p * 3),
                ("trunk_dev", C.c_void_p), ("stages_dev", C.c_void_p * 5)]


class GroundIn(C.Structure):
    _fields_ = [("sam3_fpn_dev", C.c_void_p * 3), ("n_images", C.c_int), ("language_features_dev", C.c_void_p),
                ("language_mask_dev", C.c_void_p), ("n_tokens", C.c_int),
                ("n_points", C.c_int), ("points_dev", C.c_void_p), ("point_labels_dev", C.c_void_p),
                ("point_mask_dev", C.c_void_p),
                ("n_boxes", C.c_int), ("boxes_dev", C.c_void_p), ("box_labels_dev", C.c_void_p),
                ("box_mask_dev", C.c_void_p)]


class GroundOut(C.Structure):
    _fields_ = [("pred_logits_dev", C.c_void_p), ("pred_boxes_dev", C.c_void_p), ("presence_logit_dev", C.c_void_p),
                ("pred_masks_dev", C.c_void_p), ("semantic_seg_dev", C.c_void_p)]


class Prompts(C.Structure):
    _fields_ = [("sam2_fpn_dev", C.c_void_p * 3), ("n_images", C.c_int), ("n_prompts", C.c_int),
                ("prompt_image_dev", C.c_void_p), ("coords_dev", C.c_void_p),
                ("labels_dev", C.c_void_p), ("n_points", C.c_int), ("mask_input_dev", C.c_void_p),
                ("multimask_output", C.c_int)]


class DecodeOut(C.Structure):
    _fields_ = [("low_res_dev", C.c_void_p), ("iou_dev", C.c_void_p), ("obj_score_dev", C.c_void_p)]


_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_F = C.c_float
_FP = C.POINTER(C.c_float)

# name -> (restype, argtypes); must list every symbol include/esam3.h declares
SIGNATURES = {
    "esam3_last_error": (C.c_char_p, []),
    "esam3_create": (_I, [C.POINTER(Config), C.POINTER(_P)]),
    "esam3_destroy": (None, [_P]),
    "esam3_load_weight": (_I, [_P, C.c_char_p, _P, C.POINTER(_L), _I]),
    "esam3_finalize": (_I, [_P]),
    "esam3_release_host_weights": (C.c_int64, [_P]),
    "esam3_encode_image": (_I, [_P, _P, _I, C.POINTER(ImageFeatures), _P]),
    "esam3_decode": (_I, [_P, C.POINTER(Prompts), C.POINTER(DecodeOut), _P]),
    "esam3_postprocess_masks": (_I, [_P, _P, _I, _I, _I, _F, _F, _P, _P, _P]),
    "esam3_clamp_f32": (_I, [_P, _P, _L, _F, _F, _P]),
    "esam3_set_scope_hooks": (None, [_P, _P]),
    "esam3_profile_enable": (_I, [_P, _I]),
    "esam3_profile_tag": (_I, [_P, C.c_char_p]),
    "esam3_profile_report": (_I, [_P, C.c_char_p, _L]),
    "esam3_workspace_bytes": (_L, [_P]),
    "esam3_elem_size": (_I, [_P]),
    "esam3_ground": (_I, [_P, _P, _P, _P]),
    "esam3_encode_text": (_I, [_P, _P, _I, _I, _P, _P, _P]),
    "esam3_preprocess_u8": (_I, [_P, _P, _I, _I, _I, _P]),
    "esam3_preprocess_resize_u8": (_I, [_P, _I, _I, _P, _I, _I, _P]),
    "esam3_preprocess_resize_u8_batch": (_I, [_P, _I, _I, _I, _P, _I, _I, _P]),
    "esam3_preprocess_resize_rgbx_batch": (_I, [_P, _I, _I, _I, _P, _I, _I, _P]),
    "esam3_op_linear": (_I, [_I, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P]),
    "esam3_op_fused_mlp": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P]),
    "esam3_op_resize_shuffle": (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "esam3_op_conv2d": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "esam3_op_conv3x3_s2": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "esam3_op_window_attention": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "esam3_op_attn_window": (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "esam3_rle_scratch_bytes": (_L, [_I, _I, _I, _L]),
    "esam3_rle_encode": (_I, [_P, _I, _I, _I, _P, _L, _P, _P, _L, _P]),
    "esam3_rle_to_string": (_L, [_P, _L, _P, _L]),
    "esam3_host_widen_u8_f32": (_I, [_P, _P, C.c_int64]),
    "esam3_rle_from_string": (_L, [_P, _L, _P, _L]),
    "esam3_stage1_preprocess_shape": (None, [_I, _I, _I, _P, _P]),
    "esam3_stage1_preprocess_u8": (_I, [_P, _I, _I, _P, _I, _P, _P, _P, _P, _P]),
    "esam3_act_forward": (_I, [_I, _P, _P, _L, _I, _P]),
    "esam3_act_backward": (_I, [_I, _P, _P, _P, _L, _I, _P]),
    "esam3_linear_wgrad_workspace": (_L, [_L, _I, _I]),
    "esam3_linear_wgrad": (_I, [_I, _P, _P, _L, _I, _I, _P, _P, _P, _P]),
    "esam3_conv3x3_wgrad_workspace": (_L, [_I, _I, _I, _I, _I, _I]),
    "esam3_conv3x3_wgrad": (_I, [_I, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "esam3_colsum_workspace": (_L, [_L, _I]),
    "esam3_colsum": (_I, [_I, _P, _L, _I, _P, _P, _P]),
    "esam3_channel_scale": (_I, [_I, _P, _P, _I, C.c_float, _P, _I, C.c_float, _P, _P, _I, _L, _I, _P]),
    "esam3_batched_coldot_workspace": (_L, [_I, _I]),
    "esam3_batched_coldot": (_I, [_I, _P, _P, _I, _L, _I, C.c_float, _P, _P, _P]),
    "esam3_train_conv3x3_s2": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "esam3_ln_train_forward": (_I, [_I, _P, _P, _L, _I, _P, _P, C.c_float, _P, _P, _P]),
    "esam3_ln_train_workspace": (_L, [_I]),
    "esam3_ln_train_backward": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P, _P]),
    "esam3_win_attn_train_forward": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, C.c_float, _P]),
    "esam3_win_attn_train_backward": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, C.c_float, _P]),
    "esam3_window_partition": (_I, [_I, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "esam3_win_attn_train_forward_tab": (_I, [_I, _P, _P, _P, _I, _P, _P, _I, _I, C.c_float, _P]),
    "esam3_win_attn_train_backward_tab": (_I, [_I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, C.c_float, _P]),
    "esam3_attn_bias_gather_sum": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "esam3_dwconv_wgrad_workspace": (_L, [_I]),
    "esam3_dwconv_wgrad": (_I, [_I, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "esam3_lite_mla_backward": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "esam3_lite_mla_backward_workspace": (_L, [_I, _I, _I, _I]),
    "esam3_lite_mla_backward_ws": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, C.c_float, _P, _P]),
    "esam3_lite_mla_backward_ws2": (_I, [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, C.c_float, _P, _P]),
    "esam3_dwconv_dgrad": (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "esam3_train_pack_bytes": (_L, [_I, _I, _I]),
    "esam3_train_linear": (_I, [_I, _P, _P, _P, _P, _L, _I, _I, _I, _P, _P]),
    "esam3_train_conv3x3": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "esam3_train_conv3x3_workspace": (_L, [_I, _I, _I, _I, _I, _I]),
    "esam3_train_conv3x3_ws": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _L, _P]),
    "esam3_train_dwconv": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "esam3_train_dwconv_dgrad": (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "esam3_stem_im2col": (_I, [_I, _P, _P, _I, _I, _I, _P]),
    "esam3_train_stem": (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "esam3_resize_bilinear_backward": (_I, [_I, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "esam3_bn_train_workspace": (_L, [_I]),
    "esam3_bn_train_forward": (_I, [_I, _P, _P, _L, _I, _P, _P, _P, _P, C.c_double, C.c_double, _P, _P, _P, _P]),
    "esam3_bn_train_backward": (_I, [_I, _P, _P, _P, _L, _I, _P, _P, _P, _P, _P, _P, _P]),
    "esam3_bn_act_train_forward": (_I, [_I, _P, _P, _P, _I, _L, _I, _P, _P, _P, _P, C.c_double, C.c_double, _P, _P, _P, _P]),
    "esam3_bn_act_train_backward": (_I, [_I, _P, _P, _P, _I, _P, _L, _I, _P, _P, _P, _P, _P, _P, _P]),
    "esam3_bn_act_train_backward_rc": (_I, [_I, _P, _P, _I, _P, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "esam3_bn_train_stats": (_I, [_I, _P, _L, _I, C.c_double, _P, _P, _P, _P, _P]),
    "esam3_bn_train_apply": (_I, [_I, _P, _P, _L, _I, _P, _P, _P, _P, _P]),
    "esam3_bn_train_backward_sums": (_I, [_I, _P, _P, _L, _I, _P, _P, _P, _P, _P, _P]),
    "esam3_bn_train_backward_apply": (_I, [_I, _P, _P, _P, _L, _I, _P, _P, _P, _P, _P, C.c_double, _P]),
    "esam3_stage1_update_workspace": (_L, [_L]),
    "esam3_stage1_update": (_I, [_P, _P, _P, _P, _L, _P, _P, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _F, _P, _F, _F,
                            _I, _I, _I, _P, _P, _P]),
    "esam3_distill_loss": (_I, [_I, _P, _I, _P, _P, _I, _I, _I, _P, _P, _P]),
    "esam3_distill_loss_backward": (_I, [_I, _P, _I, _P, _P, _I, _I, _I, C.c_float, C.c_float, _P, _P, _P]),
    "esam3_distill_loss_backward_ds": (_I, [_I, _P, _I, _P, _P, _I, _I, _I, C.c_float, C.c_float, _P, _P, _P, _P]),
    "esam3_set_text_causal": (_I, [_P, _I]),
    "esam3_op_mha": (_I, [_I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _I, _I, _P]),
    "esam3_op_vit_rope": (_I, [_I, _P, _P, _L, _I, _I, _I, _I, _P]),
    "esam3_op_squeeze_excite": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "esam3_op_conv3x3_padded": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "esam3_op_upconv": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "esam3_op_conv_transpose2x2": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "esam3_op_mbconv_fused": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "esam3_op_mbconv3": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "esam3_op_lite_mla_block": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "esam3_resize_axis_tables": (_I, [_I, _I, _P, _P, _P]),
    "esam3_op_rowlin256": (_I, [_P, _P, _P, _P, _I, _P, _L, _P]),
    "esam3_op_i2t_block": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "esam3_op_dwconv": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "esam3_op_stem": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "esam3_op_stem_dsconv": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "esam3_op_lite_mla": (_I, [_I, _P, _P, _I, _I, _I, _I, _P]),
    "esam3_op_grouped_pw": (_I, [_I, _P, _P, _P, _L, _I, _I, _P]),
    "esam3_op_resize_bilinear": (_I, [_I, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "esam3_op_layernorm": (_I, [_I, _P, _P, _P, _P, _P, _L, _I, _F, _I, _P]),
    "esam3_op_attention": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "esam3_op_fill_holes": (_I, [_P, _P, _I, _I, _I, _F, _F, _P]),
    "esam3_op_upsample_masks": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "esam3_op_cast": (_I, [_I, _I, _P, _P, _L, _P]),
}

_lib = None


def load() -> C.CDLL:
    """Load the HIP extension (once).  Raises ``Esam3Error`` if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("ESAM3_DEV_LIB") or LIB_PATH   # development aid: an A/B build of the SAME sources (make dev / tools/dev_variants.sh)
    if path != LIB_PATH:
        if not os.path.exists(path):
            raise Esam3Error(f"ESAM3_DEV_LIB={path} does not exist")
        lib = C.CDLL(os.path.abspath(path))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _install_scope_hooks(lib)
        _lib = lib
        return lib
    if not os.path.exists(LIB_PATH):
        raise Esam3Error(
            f"{LIB_PATH} is missing: build it with `python -c \"import __graft_entry__ as g; "
            "g.build()\"` (or `make -C efficientsam3_amd/csrc`).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _install_scope_hooks(lib)
    _lib = lib
    return lib


_SCOPE_PUSH = C.CFUNCTYPE(None, C.c_char_p)
_SCOPE_POP = C.CFUNCTYPE(None)
_scope_keep = []     # the ctypes callback objects must outlive the library's pointers to them
_scope_open = []     # record_function ranges opened by the engine's scopes, innermost last (per process: the engine is not re-entrant)


def _install_scope_hooks(lib):
    """The engine's phase scopes (include/esam3.h, esam3_set_scope_hooks) become torch.profiler.record_function ranges with the
    reference's names (sam3/model/sam3_image.py:449-479) -- only while a torch profiler is recording; otherwise a callback costs a
    flag test."""
    try:
        import torch
        from torch.autograd import profiler as _prof
    except Exception:  # noqa: BLE001  (no torch: nothing to forward to; roctx still sees the scopes)
        return

    def push(name):
        if _prof._is_profiler_enabled:
            rf = torch.profiler.record_function(name.decode())
            rf.__enter__()
            _scope_open.append(rf)
        else:
            _scope_open.append(None)

    def pop():
        rf = _scope_open.pop() if _scope_open else None
        if rf is not None:
            rf.__exit__(None, None, None)

    cbs = (_SCOPE_PUSH(push), _SCOPE_POP(pop))
    _scope_keep.extend(cbs)
    lib.esam3_set_scope_hooks(C.cast(cbs[0], C.c_void_p), C.cast(cbs[1], C.c_void_p))


def check(rc: int, what: str = "esam3 call"):
    if rc != 0:
        msg = load().esam3_last_error()
        raise Esam3Error(f"{what} failed: {msg.decode() if msg else 'unknown error'}")
