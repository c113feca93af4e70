"""ctypes binding of libagenda_hip.so (C ABI in include/agenda_hip.h).

There is NO fallback: if the HIP library is missing or fails to load, importing the product path
raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AGD_LIB: developer override of the library path (tools/ point it at the experiments build, `make -C agenda_amd/csrc exp`);
# whatever the path, a missing or unloadable library raises -- there is no fallback
LIB_PATH = os.environ.get("AGD_LIB") or os.path.join(_HERE, "libagenda_hip.so")
AGD_MAX_LEVELS = 8
AGD_N_CLASSES = 11


class AgdConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int),
        ("in_channels", C.c_int), ("out_channels", C.c_int), ("n_levels", C.c_int),
        ("block_out_channels", C.c_int * AGD_MAX_LEVELS),
        ("down_cross", C.c_int * AGD_MAX_LEVELS),
        ("num_heads", C.c_int * AGD_MAX_LEVELS),
        ("layers_per_block", C.c_int), ("cross_attention_dim", C.c_int),
        ("use_linear_projection", C.c_int), ("norm_num_groups", C.c_int),
        ("vae_latent_channels", C.c_int), ("vae_out_channels", C.c_int), ("vae_n_levels", C.c_int),
        ("vae_block_out_channels", C.c_int * AGD_MAX_LEVELS),
        ("vae_layers_per_block", C.c_int), ("vae_norm_num_groups", C.c_int),
        ("vae_scaling_factor", C.c_float),
        ("max_tokens", C.c_int),
        ("prediction_type", C.c_int),
        ("workspace_bytes", C.c_longlong),
        ("text_hidden", C.c_int), ("text_layers", C.c_int), ("text_heads", C.c_int), ("text_intermediate", C.c_int),
        ("text_vocab", C.c_int), ("text_max_pos", C.c_int), ("text_act", C.c_int), ("text_eps", C.c_float),
    ]


class AgendaHipError(RuntimeError):
    pass


_P = C.c_void_p
_SIGS = {
    "agd_create": (_P, [C.c_int, C.POINTER(AgdConfig)]),
    "agd_destroy": (None, [_P]),
    "agd_last_error": (C.c_char_p, [_P]),
    "agd_load_tensor": (C.c_int, [_P, C.c_char_p, _P, C.c_int, C.c_int, C.POINTER(C.c_longlong)]),
    "agd_finalize": (C.c_int, [_P]),
    "agd_set_context": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "agd_text_encode": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P]),
    "agd_text_set_embedding_row": (C.c_int, [_P, C.c_int, _P]),
    "agd_unet_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_float, _P, _P]),
    "agd_unet_forward_ts": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(C.c_float), _P, _P]),
    "agd_cfg_ddim_step": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, _P]),
    "agd_denoise": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                              C.POINTER(C.c_float), C.c_float, _P]),
    "agd_denoise_plms": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                   C.POINTER(C.c_float), C.c_float, _P]),
    "agd_vae_decode": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P]),
    "agd_vae_encode": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P]),
    "agd_set_option": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "agd_record_config": (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    "agd_record_reset": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "agd_daam_global": (C.c_int, [_P, C.c_int, C.c_int, _P, _P]),
    "agd_hook_global": (C.c_int, [_P, _P, _P]),
    "agd_hook_count": (C.c_int, [_P]),
    "agd_hook_last_map": (C.c_int, [_P, _P, C.c_int, _P]),
    "agd_cross_attn": (C.c_int, [_P, C.c_char_p, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P]),
    "agd_attn_processor": (C.c_int, [_P, C.c_char_p, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P]),
    "agd_hook_reset": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "agd_hook_num_maps": (C.c_int, [_P]),
    "agd_hook_map_dims": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int)]),
    "agd_hook_map": (C.c_int, [_P, C.c_int, _P, _P]),
    "agd_op_attn_reg_loss": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P, _P, C.c_float, _P, _P, _P]),
    "agd_attn_processor_backward": (C.c_int, [_P, C.c_char_p, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "agd_op_conv2d": (C.c_int, [_P, _P, _P, _P] + [C.c_int] * 9 + [_P]),
    "agd_op_conv2d_ex": (C.c_int, [_P, _P, _P, _P] + [C.c_int] * 10 + [_P]),
    "agd_op_linear": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "agd_op_groupnorm": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _P]),
    "agd_op_conv_groupnorm": (C.c_int, [_P] * 6 + [C.c_int] * 6 + [C.c_float, C.c_int, C.c_int, _P]),
    "agd_op_layernorm": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_float, _P]),
    "agd_op_ff_fused": (C.c_int, [_P] * 8 + [C.c_int, C.c_int, C.c_float, _P]),
    "agd_op_attn_chain": (C.c_int, [_P] * 9 + [C.c_int] * 5 + [C.c_float, _P]),
    "agd_op_xattn_premul": (C.c_int, [_P] * 9 + [C.c_int] * 5 + [C.c_float, _P]),
    "agd_op_attention": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P]),
    "agd_op_attention_headsum": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P]),
    "agd_op_bicubic_clamp_mean": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "agd_op_heatmap_u8": (C.c_int, [_P, C.c_int, C.c_int, _P, _P]),
    "agd_op_resize_u8_pil": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "agd_op_stack_heatmaps": (C.c_int, [_P, _P, _P, C.c_longlong, _P, _P, _P]),
    "agd_profile_begin": (C.c_int, [_P]),
    "agd_profile_end": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "agd_profile_end_ex": (C.c_int, [_P, C.c_double, C.c_double] + [C.POINTER(C.c_double)] * 5 + [C.POINTER(C.c_longlong)]),
    "agd_profile_class_name": (C.c_char_p, [C.c_int]),
    "agd_version": (C.c_char_p, []),
}

EXPORTS = tuple(_SIGS.keys())
_lib = None


def load():
    """Load the library once; raise (never fall back) if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AgendaHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C agenda_amd/csrc`. There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, ctx=None, what: str = ""):
    if rc != 0:
        msg = load().agd_last_error(ctx)
        raise AgendaHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    """Raw device/host pointer of a contiguous torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), "tensor must be contiguous"
    return C.c_void_p(t.data_ptr())


def current_stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
