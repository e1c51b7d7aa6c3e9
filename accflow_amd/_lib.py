"""ctypes binding of libaccflow_hip.so (the C-ABI declared in include/accflow_hip.h).

There is deliberately NO fallback: if the shared library is missing or fails to load, importing the
ops raises.  The product path never routes through PyTorch operators or the CPU oracle.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ACCFLOW_HIP_LIB") or os.path.join(_HERE, "lib", "libaccflow_hip.so")

ABI_VERSION = 20
c_f = ctypes.c_void_p      # device pointers travel as void*
c_ll = ctypes.c_longlong
c_i = ctypes.c_int


class ConvSrc(ctypes.Structure):
    """Mirror of accflow_conv_src: one source of the multi-source S16 form."""
    _fields_ = [
        ("ptr", c_f), ("bs", c_ll),
        ("C", c_i), ("Hs", c_i), ("Ws", c_i),
        ("step", c_i), ("oy", c_i), ("ox", c_i),
        ("KH", c_i), ("KW", c_i), ("padH", c_i), ("padW", c_i),
        ("reserved", c_i),
    ]


MAX_SRC = 4


class ConvDesc(ctypes.Structure):
    """Mirror of accflow_conv_desc (include/accflow_hip.h)."""
    _fields_ = [
        ("in0", c_f), ("in1", c_f),
        ("in0_bs", c_ll), ("in1_bs", c_ll),
        ("C0", c_i), ("C1", c_i),
        ("B", c_i), ("H", c_i), ("W", c_i),
        ("OH", c_i), ("OW", c_i),
        ("KH", c_i), ("KW", c_i), ("stride", c_i), ("padH", c_i), ("padW", c_i),
        ("Cout", c_i),
        ("wpack", c_f), ("ktab", c_f),
        ("Kpad", c_i), ("CoutPad", c_i),
        ("bias", c_f),
        ("out", c_f), ("out_bs", c_ll),
        ("act", c_i), ("epi", c_i),
        ("e0", c_f), ("e0_bs", c_ll),
        ("e1", c_f), ("e1_bs", c_ll),
        ("out2", c_f), ("out2_bs", c_ll),
        ("offset", c_f), ("offset_bs", c_ll),
        ("dmask", c_f), ("dmask_bs", c_ll),
        ("wsplit", c_f), ("mode", c_i),
        ("wpatch", c_f),
        ("wsplit_bs", c_ll),
        ("kws", c_f), ("kws_elems", c_ll),
        ("wpatch16", c_f), ("guard", c_f),
        ("wscale16", c_f), ("wsplit16", c_f), 
        ("stats", c_f), ("stat_slots", c_i), ("pre", c_f), ("pre_bs", c_ll), ("in_norm", c_f), ("acc_scale", ctypes.c_float),
        ("in_fmt", c_i), ("out16", c_f), ("out16_bs", c_ll),
        ("cb", c_i), ("out_cbs", c_ll), ("e0_cbs", c_ll), ("out16_cbs", c_ll),
        ("nsrc", c_i), ("src", ConvSrc * MAX_SRC),
        ("e0_fmt", c_i),
        ("tg_w16", c_f), ("tg_scale", c_f), ("tg_out", c_f), ("tg_out_bs", c_ll), ("tg_out_ps", c_ll),
        ("tg_rows", c_i), ("tg_coutpad", c_i),
        ("split_c0", c_i), ("p32", c_i),
    ]


# name -> argtypes; every function returns int (0 = ok, else hipError_t)
SIGNATURES = {
    "accflow_abi_version": [],
    "accflow_conv_desc_bytes": [],
    "accflow_conv_src_bytes": [],
    "accflow_s16_item_words": [c_i, c_i, c_i],
    "accflow_to_s16_f32": [c_f, c_ll, c_f, c_ll, c_f, c_i, c_i, c_i, c_f],
    "accflow_conv_kpad": [c_i, c_i, c_i],
    "accflow_conv_coutpad": [c_i],
    "accflow_conv_pack_f32": [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f],
    "accflow_conv_pack_bf16s": [c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_f],
    "accflow_conv_patch_elems": [c_i, c_i, c_i, c_i],
    "accflow_conv_pack_patch": [c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_f],
    "accflow_conv_pack_patch16": [c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_f, c_f],
    "accflow_conv_pack_split16": [c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_f, c_f],
    "accflow_conv_multi_pack_elems": [c_i, c_i, ctypes.POINTER(c_i), ctypes.POINTER(c_i), ctypes.POINTER(c_i)],
    "accflow_conv_pack_multi16": [ctypes.POINTER(c_f), c_f, c_i, c_i, ctypes.POINTER(c_i), ctypes.POINTER(c_i), ctypes.POINTER(c_i),
                                  c_f, c_f, c_f],
    "accflow_conv2d_f32": [ctypes.POINTER(ConvDesc), c_f],
    "accflow_conv_stat_slots": [ctypes.POINTER(ConvDesc)],
    "accflow_conv_in_norm_supported": [ctypes.POINTER(ConvDesc)],
    "accflow_instance_stats_finalize_f32": [c_f, c_i, c_f, c_i, c_i, ctypes.c_float, c_f],
    "accflow_instance_norm_apply_f32": [c_f, c_f, c_i, c_f, c_f, c_f, c_i, c_i, c_i, ctypes.c_float, c_i, c_f],
    "accflow_instance_norm_apply_s16_f32": [c_f, c_f, c_i, c_f, c_f, c_f, c_f, c_ll, c_f, c_i, c_i, c_i, ctypes.c_float, c_i, c_f],
    "accflow_instance_norm_apply_s16res_f32": [c_f, c_f, c_i, c_f, c_f, c_ll, c_f, c_f, c_ll, c_f, c_i, c_i, c_i, ctypes.c_float, c_f],
    "accflow_instance_stats_finalize_sub_f32": [c_f, c_i, c_i, c_i, c_f, c_i, c_i, ctypes.c_float, c_f],
    "accflow_instance_norm_apply_s16proj_f32": [c_f, c_f, c_i, c_f, c_ll, c_f, c_i, c_i, c_i, c_f, c_f, c_ll, c_f, c_i, c_i, c_i,
                                                ctypes.c_float, c_f],
    "accflow_get_occ_s16": [c_f, c_ll, c_f, c_ll, c_f, c_ll, c_f, c_ll, c_f, c_i, c_i, c_i, c_i, c_i, c_f],
    "accflow_blend_s16": [c_f, c_f, c_f, c_f, c_ll, c_f, c_i, c_i, c_i, c_f],
    "accflow_deform_columns_s16": [c_f, c_ll, c_f, c_ll, c_f, c_ll, c_i, c_f, c_ll, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f],
    "accflow_corr_volume_f32": [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f],
    "accflow_corr_volume_ws_bytes": [c_i, c_i, c_i],
    "accflow_corr_volume_split_f32": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f],
    "accflow_corr_lookup_f32": [c_f, c_f, c_f, c_f, c_f, c_f, c_ll, c_i, c_i, c_i, c_f],
    "accflow_corr_disp_supported": [c_i, c_i],
    "accflow_corr_disp_level_elems": [c_i, c_i, c_i],
    "accflow_corr_volume_disp_f32": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_f, c_i, c_i, c_i, c_i, c_f],
    "accflow_corr_pack_bytes": [c_i, c_i, c_i],
    "accflow_corr_pack_f32": [c_f, c_f, c_i, c_f, c_i, c_i, c_i, c_i, c_f],
    "accflow_corr_volume_disp_packed_f32": [c_f, c_i, ctypes.POINTER(c_i), ctypes.POINTER(c_i), c_f, c_f, c_f, c_f, c_i, c_f, c_i, c_i, c_i,
                                            c_i, c_f],
    "accflow_corr_disp_pool_f32": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f],
    "accflow_corr_lookup_disp_f32": [c_f, c_f, c_f, c_f, c_f, c_f, c_ll, c_i, c_i, c_i, c_f],
    "accflow_corr_lookup_disp_s16": [c_f, c_f, c_f, c_f, c_f, c_f, c_ll, c_f, c_i, c_i, c_i, c_f],
    "accflow_corr_lookup_convc1_kpad": [],
    "accflow_corr_lookup_convc1_s16": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_ll, c_f, c_ll, c_i, c_f, c_i, c_i, c_i, c_i, c_f],
    "accflow_flow_from_coords_s16": [c_f, c_f, c_ll, c_f, c_ll, c_f, c_ll, c_f, c_ll, c_i, c_f, c_i, c_i, c_i, c_i, c_f],
    "accflow_convex_upsample_f32": [c_f, c_ll, c_f, c_ll, c_f, c_i, c_i, c_i, c_f],
    "accflow_backwarp_f32": [c_f, c_ll, c_f, c_ll, c_f, c_ll, c_i, c_i, c_i, c_i, c_f],
    "accflow_compose_flow_f32": [c_f, c_ll, c_f, c_ll, c_f, c_ll, c_i, c_i, c_i, c_f],
    "accflow_get_occ_f32": [c_f, c_ll, c_f, c_ll, c_f, c_ll, c_f, c_ll, c_i, c_i, c_i, c_i, c_i, c_f],
    "accflow_downflow8_f32": [c_f, c_f, c_i, c_i, c_i, c_i, c_f],
    "accflow_instance_norm_f32": [c_f, c_f, c_f, c_i, c_i, c_i, ctypes.c_float, c_i, c_f],
    "accflow_split_tanh_relu_f32": [c_f, c_f, c_ll, c_f, c_ll, c_i, c_i, c_i, c_i, c_f],
    "accflow_split_tanh_relu_idx_f32": [c_f, c_i, ctypes.POINTER(c_i), c_f, c_ll, c_f, c_ll, c_i, c_i, c_i, c_i, c_f],
    "accflow_coords_grid_f32": [c_f, c_f, c_i, c_i, c_i, c_f],
    "accflow_flow_from_coords_f32": [c_f, c_f, c_ll, c_f, c_ll, c_f, c_i, c_i, c_i, c_i, c_f],
    "accflow_deform_columns_f32": [c_f, c_ll, c_f, c_ll, c_f, c_ll, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f],
    "accflow_conv_pack_all_f32": [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f],
    "accflow_tap_sum_f32": [c_f, c_f, c_f, c_ll, c_f, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f],
    "accflow_tap_sum_parts_f32": [c_f, c_i, c_ll, c_ll, c_f, c_f, c_ll, c_f, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f],
    "accflow_blend_f32": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f],
    "accflow_activation_f32": [c_f, c_ll, c_i, c_i, c_i, c_i, c_f],
    "accflow_copy_f32": [c_f, c_ll, c_f, c_ll, c_i, c_i, c_i, c_f],
    "accflow_gma_attention_f32": [c_f, c_f, c_i, c_i, c_i, ctypes.c_float, c_f],
    "accflow_gma_attention_ws_bytes": [c_i, c_i, c_i],
    "accflow_gma_attention_t_f32": [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, ctypes.c_float, c_f],
    "accflow_gma_aggregate_ws_bytes": [c_i, c_i, c_i],
    "accflow_gma_attention_s16_ws_bytes": [c_i, c_i, c_i],
    "accflow_gma_attention_s16": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, ctypes.c_float, c_f],
    "accflow_gma_aggregate_s16_ws_bytes": [c_i, c_i, c_i],
    "accflow_gma_aggregate_s16": [c_f, c_f, c_f, c_ll, c_f, c_f, c_ll, c_f, c_ll, c_f, c_f, c_i, c_i, c_i, c_i, c_f],
    "accflow_gma_aggregate_t_f32": [c_f, c_f, c_f, c_f, c_f, c_ll, c_f, c_i, c_f, c_i, c_i, c_i, c_i, c_f],
    "accflow_gma_aggregate_f32": [c_f, c_f, c_f, c_f, c_f, c_ll, c_i, c_i, c_i, c_f],
    "accflow_act_backward_f32": [c_f, c_ll, c_f, c_ll, c_f, c_ll, c_i, c_ll, c_i, c_f],
    "accflow_add_f32": [c_f, c_ll, c_f, c_ll, c_i, c_ll, c_f],
    "accflow_l1_grad_f32": [c_f, c_f, c_f, c_ll, ctypes.c_float, c_f],
    "accflow_dilate_f32": [c_f, c_ll, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f],
    "accflow_blend_backward_f32": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f],
    "accflow_convex_upsample_backward_f32": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f],
    "accflow_conv_wgrad_f32": [c_f, c_ll, c_f, c_ll, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f],
    "accflow_deform_conv_backward_f32": [c_f, c_ll, c_f, c_ll, c_f, c_ll, c_f, c_f, c_ll, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                                         c_i, c_f],
}

_lock = threading.Lock()
_lib = None


def _promote_hip_runtime():
    """Make the process's HIP runtime symbols global so the (runtime-less) kernel library binds to
    the SAME libamdhip64 PyTorch uses; loading a second runtime would split streams/allocations."""
    import torch  # noqa: F401  (loads torch/lib/libamdhip64.so)
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    else:  # system ROCm build of torch
        ctypes.CDLL("libamdhip64.so", mode=ctypes.RTLD_GLOBAL)


def load():
    """Load (once) and return the ctypes handle; raises RuntimeError if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "accflow_amd: %s not found - run `python -m accflow_amd.build` (hipcc, gfx950). "
                "There is no non-HIP fallback." % LIB_PATH)
        _promote_hip_runtime()
        lib = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
            fn.argtypes = argtypes
            fn.restype = ctypes.c_longlong if name in ("accflow_conv_patch_elems", "accflow_conv_multi_pack_elems", "accflow_corr_pack_bytes", "accflow_corr_volume_ws_bytes",
                                                    "accflow_gma_aggregate_ws_bytes", "accflow_gma_attention_ws_bytes", "accflow_gma_aggregate_s16_ws_bytes", "accflow_gma_attention_s16_ws_bytes", "accflow_corr_disp_level_elems",
                                                    "accflow_s16_item_words") else ctypes.c_int
        if lib.accflow_abi_version() != ABI_VERSION:
            raise RuntimeError("accflow_amd: ABI version mismatch")
        if (lib.accflow_conv_desc_bytes() != ctypes.sizeof(ConvDesc) or lib.accflow_conv_src_bytes() != ctypes.sizeof(ConvSrc)):
            raise RuntimeError("accflow_amd: accflow_conv_desc is %d / %d bytes in the library, %d / %d in _lib.py's mirror" % (
                lib.accflow_conv_desc_bytes(), lib.accflow_conv_src_bytes(), ctypes.sizeof(ConvDesc), ctypes.sizeof(ConvSrc)))
        _lib = lib
    return _lib
