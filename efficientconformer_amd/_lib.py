"""ctypes binding of the C ABI in include/effconf.h (the only way Python reaches the HIP kernels).

There is deliberately no fallback: if libeffconf.so is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libeffconf.so")

ABI_VERSION = 3


class EcBlock(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dim_model", "dim_expand", "ff_ratio", "num_heads", "kernel_size",
                                            "group_size", "max_pos", "conv_stride")]


class EcConfig(C.Structure):
    _fields_ = [("n_mels", C.c_int32), ("sample_rate", C.c_int32), ("n_fft", C.c_int32), ("win_length", C.c_int32),
                ("hop_length", C.c_int32), ("normalize", C.c_int32), ("mean", C.c_float), ("std", C.c_float),
                ("sub_layers", C.c_int32), ("sub_filters", C.c_int32 * 4), ("num_blocks", C.c_int32),
                ("blocks", C.POINTER(EcBlock)), ("vocab_size", C.c_int32),
                ("causal", C.c_int32), ("left_context", C.c_int32), ("right_context", C.c_int32)]


class EcRnntConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dim_encoder", "dim_decoder", "dim_joint", "vocab_size", "num_layers",
                                            "max_consec_dec_step", "joint_mode", "joint_act")]


# name -> (restype, argtypes); the symbol list tests/test_abi.py checks against include/effconf.h
_P, _I32, _I64P, _F32P, _SZ = C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t
SIGNATURES = {
    "effconf_abi_version": (C.c_int, []),
    "effconf_last_error": (C.c_char_p, []),
    "effconf_encoder_create": (_P, [C.POINTER(EcConfig)]),
    "effconf_encoder_destroy": (None, [_P]),
    "effconf_encoder_load_tensor": (C.c_int, [_P, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), _I32]),
    "effconf_encoder_finalize": (C.c_int, [_P]),
    "effconf_encoder_workspace_bytes": (_SZ, [_P, _I32, _I32, _I32]),
    "effconf_encoder_out_frames": (_I32, [_P, _I32, _I32]),
    "effconf_encoder_forward": (C.c_int, [_P, _F32P, _I64P, _I32, _I32, _F32P, _I64P, _P, _SZ, _P]),
    "effconf_encoder_forward_mel": (C.c_int, [_P, _F32P, _I64P, _I32, _I32, _F32P, _I64P, _P, _SZ, _P]),
    "effconf_encoder_workspace_bytes_ragged": (_SZ, [_P, _P, _I32, _I32, _I32]),
    "effconf_encoder_forward_ragged": (C.c_int, [_P, _F32P, _I64P, _P, _I32, _I32, _I32, _F32P, _I32, _I64P, _P, _SZ, _P]),
    "effconf_mel_frontend": (C.c_int, [_P, _F32P, _I32, _I32, _F32P, _P]),
    "effconf_encoder_attention_dims": (C.c_int, [_P, _I32, _I32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "effconf_encoder_set_attention_outputs": (C.c_int, [_P, C.POINTER(C.c_void_p), _I32]),
    "effconf_host_pack_rows": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int64), _I32, C.c_void_p, C.c_int64, _I32, _I32]),
    "effconf_ctc_greedy": (C.c_int, [_P, _F32P, _I64P, _I32, _I32, _P, _P, _F32P, _P, _SZ, _P]),
    "effconf_ctc_greedy_bf16": (C.c_int, [_P, _P, _I64P, _I32, _I32, _P, _P, _F32P, _P, _SZ, _P]),
    "effconf_rnnt_create": (_P, [C.POINTER(EcRnntConfig)]),
    "effconf_rnnt_destroy": (None, [_P]),
    "effconf_rnnt_load_tensor": (C.c_int, [_P, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), _I32]),
    "effconf_rnnt_finalize": (C.c_int, [_P]),
    "effconf_rnnt_workspace_bytes": (_SZ, [_P, _I32, _I32]),
    "effconf_rnnt_max_tokens": (_I32, [_P, _I32]),
    "effconf_rnnt_set_option": (C.c_int, [_P, C.c_char_p, _I32]),
    "effconf_rnnt_greedy": (C.c_int, [_P, _F32P, _I64P, _I32, _I32, _P, _P, _I32, _P, _SZ, _P]),
    "effconf_encoder_set_option": (C.c_int, [_P, C.c_char_p, _I32]),
    "effconf_profile_enable": (C.c_int, [_P, _I32]),
    "effconf_profile_read": (C.c_int, [_P, _I32, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                       C.POINTER(C.c_double)]),
    "effconf_relpos_attention": (C.c_int, [_P, _P, _P, _P, _F32P, _I32, _P, _I32, _I32, _I32, _I32, _I32, _P, _I32, _I32, _P]),
    "effconf_module_workspace_bytes": (_SZ, [_P, _I32, _I32]),
    "effconf_ffn": (C.c_int, [_P, _I32, _I32, _F32P, _I32, _F32P, _P, _SZ, _P]),
    "effconf_conv_module": (C.c_int, [_P, _I32, _F32P, _I32, _I32, _F32P, _P, _SZ, _P]),
    "effconf_subsample": (C.c_int, [_P, _F32P, _I32, _I32, _F32P, _P, _SZ, _P]),
    "effconf_layernorm_residual": (C.c_int, [_P, _I32, _I32, _F32P, _F32P, C.c_float, _I32, _F32P, _P]),
    "effconf_encoder_set_trace": (C.c_int, [_P, _P, _SZ]),
    "effconf_encoder_trace_count": (_I32, [_P]),
    "effconf_encoder_trace_entry": (C.c_int, [_P, _I32, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                               C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
}

# include/effconf_debug.h: exported by libeffconf_debug.so only (tests / tools; load_debug())
DEBUG_SIGNATURES = {
    "effconf_debug_mel": (C.c_int, [_P, _I32, _I32, _F32P, _I32, _I32, _F32P, _P, _P]),
    "effconf_debug_neighbour": (C.c_int, [_I32, _I32, _I32, _I32, _F32P, _SZ, _P]),
    "effconf_debug_victim": (C.c_int, [_I32, _I32, _I32, _F32P, _P]),
    "effconf_debug_spin": (C.c_int, [C.c_double, _P]),
    "effconf_debug_pack_dwconv_mfma": (C.c_int, [_F32P, _I32, _I32, _P, _SZ]),
    "effconf_debug_lds_fill": (C.c_int, [_I32, _I32, _I32, _P, _SZ, _I32, _I32, _P, _P]),
    "effconf_debug_sx_gemm": (C.c_int, [_F32P, _I32, _P, _P, _I32, _F32P, _I32, _I32, _I32, _I32, _F32P, _I32, _F32P, _I32, C.c_float, _P]),
    "effconf_debug_dwconv": (C.c_int, [_P, _I32, _I32, _I32, _I32, _F32P, _F32P, _I32, _I32, _I32, _I32, _P, _P]),
    "effconf_debug_sxf_ffn": (C.c_int, [_P, _I32, _I32, _F32P, _I32, _F32P, _I32, _I32, _P]),
    "effconf_debug_gemm": (C.c_int, [_P, _I32, _P, _I32, _P, _I32, _I32, _I32, _I32, _I32, _P, _I32, _P, _I32, C.c_float, _P]),
}

_lib = None
_lib_debug = None


class EffconfError(RuntimeError):
    pass


def load():
    """dlopen libeffconf.so and bind every symbol; raises if the library is missing (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EffconfError("libeffconf.so not built: run `python -m efficientconformer_amd._build` "
                           "(or __graft_entry__.build()); there is no non-HIP fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.effconf_abi_version() != ABI_VERSION:
        raise EffconfError("libeffconf ABI version mismatch")
    _lib = lib
    return lib


def load_debug():
    """dlopen libeffconf_debug.so (the product objects + the diagnostics of include/effconf_debug.h) and bind the product AND the diagnostic symbols.
    Tests and tools only: nothing of the package calls this."""
    global _lib_debug
    if _lib_debug is not None:
        return _lib_debug
    path = os.path.join(HERE, "libeffconf_debug.so")
    if not os.path.exists(path):
        raise EffconfError("libeffconf_debug.so not built: run `python -m efficientconformer_amd._build`")
    lib = C.CDLL(path)
    for name, (res, args) in list(SIGNATURES.items()) + list(DEBUG_SIGNATURES.items()):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib_debug = lib
    return lib


def check(rc: int, what: str, lib=None):
    if rc != 0:
        msg = (lib or load()).effconf_last_error()
        if not msg and _lib_debug is not None:
            msg = _lib_debug.effconf_last_error()
        raise EffconfError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))
