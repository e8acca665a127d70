"""ctypes binding of liblvattn.so.  The library is built in-tree by
``make -C long_vita_amd/csrc`` (or ``__graft_entry__.build()``); a missing or
unloadable library is a hard error: there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblvattn.so")

LV_FWD_BLOCK_M = 256
LV_BWD_BLOCK_N = 128
LV_DESC_INTS = 8

_lib = None

_i32p = C.POINTER(C.c_int32)
_i64 = C.c_int64
_vp = C.c_void_p
_int = C.c_int
_f = C.c_float

_SIGNATURES = {
    "lvattn_abi_version": (C.c_int, []),
    "lvattn_last_error": (C.c_char_p, []),
    "lvattn_fwd_block_m": (C.c_int, []),
    "lv_build_fwd_blocks": (_int, [_i32p, _int, _int, _i32p, _int]),
    "lv_build_bwd_blocks": (_int, [_i32p, _int, _int, _i32p, _int]),
    "lv_build_fwd_blocks_rect": (_int, [_int, _int, _int, _int, _int, _int, _i32p, _int]),
    "lv_build_bwd_blocks_rect": (_int, [_int, _int, _int, _int, _int, _int, _i32p, _int]),
    "lvattn_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int,
                          _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f, _int, _vp]),
    "lvattn_fwd_hinted": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int,
                                 _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f, _int, _int, _vp]),
    "lv_mean_keys_per_item": (_int, [_i32p, _int]),
    "lvattn_fwd_dense": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int,
                                _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f, _int, _vp]),
    "lvattn_bwd_delta": (_int, [_vp, _vp, _vp, _int, _int, _int, _i64, _i64, _i64, _i64, _i64, _vp]),
    "lvattn_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _int, _int, _int, _int, _int, _int,
                          _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                          _f, _int, _int, _vp]),
    "lv_build_fwd_blocks_band": (_int, [_i32p, _int, _int, _i32p, _int]),
    "lv_build_bwd_blocks_band": (_int, [_i32p, _int, _int, _i32p, _int]),
    "lv_build_fwd_blocks_rect_band": (_int, [_int, _int, _int, _int, _int, _int, _i32p, _int]),
    "lv_plan_fwd_split": (_int, [_i32p, _int, _int, _int, _int]),
    "lv_plan_bwd_split": (_int, [_i32p, _int, _int, _int, _int, _int]),
    "lv_build_fwd_blocks_split": (_int, [_i32p, _int, _int, _int, _i32p, _int, _i32p, _int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "lv_build_bwd_blocks_split": (_int, [_i32p, _int, _int, _int, _i32p, _int, _i32p, _int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "lvattn_combine_lse": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _i64, _i64, _i64, _i64, _vp]),
    "lvattn_combine_sum": (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _i64, _i64, _vp]),
    "lvattn_cast_f32_bf16": (_int, [_vp, _vp, _int, _int, _int, _i64, _i64, _vp]),
    "lvattn_merge": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _i64, _i64, _i64, _vp]),
    "lvattn_merge_init": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _i64, _i64, _i64, _vp]),
    "lv_clock_probe": (_int, [_vp, _vp, _vp, _int, _int, _vp]),
    "lv_pixel_shuffle": (_int, [_vp, _vp, _int, _int, _int, _int, _i64, _vp]),
    "lv_pixel_shuffle_v2": (_int, [_vp, _vp, _int, _int, _int, _int, _int, _i64, _vp]),
    "lv_scatter_rows": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _i64, _i64, _vp]),
    "lv_gather_rows": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    "lvattn_select_fwd_kernel": (_int, [C.c_char_p]),
    "lvattn_select_bwd_kernels": (_int, [C.c_char_p, C.c_char_p]),
    "lvattn_decode_workspace_bytes": (_i64, [_int, _int, _int, _int]),
    "lvattn_decode": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _int,
                             _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f, _int, _vp]),
    "lvattn_decode_noalloc_max_bytes": (_i64, []),
    "lvattn_decode_noalloc": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int,
                                     _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f, _int, _vp]),
    "lv_layernorm": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _i64, _i64, _f, _vp]),
    "lv_bias_gelu": (_int, [_vp, _vp, _vp, _i64, _int, _vp]),
    "lv_bias_gelu_bwd": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _vp]),
    "lv_layernorm_bwd_workspace_bytes": (_i64, [_i64, _int]),
    "lv_layernorm_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _int, _i64, _i64, _i64, _f, _vp]),
    "lv_bias_scale_residual": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _vp]),
    "lv_bias_scale_residual_layernorm": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _f, _vp]),
    "lv_patchify": (_int, [_vp, _vp, _int, _int, _int, _int, _int, _int, _vp]),
    "lv_embed_tokens": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _vp]),
    "lv_rope": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _i64, _i64, _i64, _i64, _i64, _int, _int, _vp]),
    "lvattn_fwd_rope": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int,
                               _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f, _int,
                               _vp, _vp, _vp, _i64, _int, _vp, _i64, _i64, _vp]),
    "lv_rope_pack_kv": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _i64, _i64, _i64, _i64, _i64, _int, _vp]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class LvAttnError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle.  Raises if the library is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LvAttnError(
                f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                "(there is no CPU fallback for the attention path)")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(code: int, what: str):
    if code < 0:
        msg = lib().lvattn_last_error().decode(errors="replace")
        raise LvAttnError(f"{what} failed ({code}): {msg}")
    return code
