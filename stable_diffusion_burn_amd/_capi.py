"""ctypes binding of libsdmi.so (include/sdmi.h).

There is NO fallback: if the shared library is missing or a call fails, an
exception is raised.  The product path never routes through oracle/ or any
CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "lib" / "libsdmi.so"


class SdmiError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"sdmi status {status}: {message}")
        self.status = status


class SdmiConfig(C.Structure):
    _fields_ = [
        ("device", C.c_int32), ("model_channels", C.c_int32), ("n_head", C.c_int32), ("ctx_dim", C.c_int32),
        ("latent_h", C.c_int32), ("latent_w", C.c_int32), ("vae_ch", C.c_int32), ("max_batch", C.c_int32),
        ("precision", C.c_int32), ("clip_layers", C.c_int32), ("clip_heads", C.c_int32), ("clip_vocab", C.c_int32),
        ("clip_ctx", C.c_int32), ("reserved", C.c_int32 * 3),
    ]


_F = C.POINTER(C.c_float)
_U8 = C.POINTER(C.c_uint8)
_CTX = C.c_void_p
_TOK = C.c_void_p
_I32 = C.POINTER(C.c_int32)

# name -> (restype, argtypes); every symbol include/sdmi.h declares
SIGNATURES = {
    "sdmi_default_config": (C.c_int, [C.POINTER(SdmiConfig)]),
    "sdmi_create": (C.c_int, [C.POINTER(_CTX), C.POINTER(SdmiConfig)]),
    "sdmi_destroy": (None, [_CTX]),
    "sdmi_last_error": (C.c_char_p, []),
    "sdmi_synchronize": (C.c_int, [_CTX]),
    "sdmi_version": (C.c_char_p, []),
    "sdmi_set_stream": (C.c_int, [_CTX, C.c_void_p, C.c_int32]),
    "sdmi_set_weight": (C.c_int, [_CTX, C.c_char_p, _F, C.c_int32, C.POINTER(C.c_int64)]),
    "sdmi_weight_count": (C.c_int, [_CTX]),
    "sdmi_weight_info": (C.c_int, [_CTX, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "sdmi_load_weights_dir": (C.c_int, [_CTX, C.c_char_p]),
    "sdmi_load_weights_mpk": (C.c_int, [_CTX, C.c_char_p]),
    "sdmi_mpk_list": (C.c_int, [C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "sdmi_load_weights_packed": (C.c_int, [_CTX, _F, C.c_size_t, C.c_int32]),
    "sdmi_packed_size": (C.c_int64, [_CTX, C.c_int32]),
    "sdmi_finalize_weights": (C.c_int, [_CTX]),
    "sdmi_unet_forward": (C.c_int, [_CTX, _F, C.c_int32, _F, C.c_int32, C.c_int32, _F]),
    "sdmi_sample_latent": (C.c_int, [_CTX, _F, C.c_int32, C.c_int32, _F, C.c_int32, C.c_double, C.c_size_t, _F, C.c_uint64, _F]),
    "sdmi_decode_latent": (C.c_int, [_CTX, _F, C.c_int32, _F]),
    "sdmi_latent_to_image": (C.c_int, [_CTX, _F, C.c_int32, _U8]),
    "sdmi_sample_image": (C.c_int, [_CTX, _F, C.c_int32, C.c_int32, _F, C.c_int32, C.c_double, C.c_size_t, _F, C.c_uint64, _U8]),
    "sdmi_qkv_attention": (C.c_int, [_CTX, _F, _F, _F, _F, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _F]),
    "sdmi_tokenizer_create": (C.c_int, [C.POINTER(_TOK), C.c_char_p]),
    "sdmi_tokenizer_destroy": (None, [_TOK]),
    "sdmi_tokenizer_vocab_size": (C.c_int, [_TOK]),
    "sdmi_tokenizer_encode": (C.c_int, [_TOK, C.c_char_p, _I32, C.c_int32, _I32]),
    "sdmi_tokenizer_decode": (C.c_int, [_TOK, _I32, C.c_int32, C.c_char_p, C.c_int32, _I32]),
    "sdmi_clip_forward": (C.c_int, [_CTX, _I32, C.c_int32, C.c_int32, _F]),
    "sdmi_context": (C.c_int, [_CTX, _TOK, C.c_char_p, _F, C.c_int32, _I32]),
    "sdmi_encode_image": (C.c_int, [_CTX, _F, C.c_int32, _F]),
    "sdmi_write_png": (C.c_int, [C.c_char_p, _U8, C.c_int32, C.c_int32]),
    "sdmi_sample_latent_dev": (C.c_int, [_CTX, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_double, C.c_size_t, C.c_void_p, C.c_void_p]),
    "sdmi_latent_to_image_dev": (C.c_int, [_CTX, C.c_void_p, C.c_int32, C.c_void_p]),
    "sdmi_sample_image_dev": (C.c_int, [_CTX, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_double, C.c_size_t, C.c_void_p, C.c_void_p]),
    "sdmi_create_multi": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(SdmiConfig), _I32, C.c_int32]),
    "sdmi_destroy_multi": (None, [C.c_void_p]),
    "sdmi_multi_size": (C.c_int32, [C.c_void_p]),
    "sdmi_multi_ctx": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "sdmi_multi_load_weights": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "sdmi_sample_image_sharded": (C.c_int, [C.c_void_p, _F, C.c_int32, _F, C.c_int32, C.c_double, C.c_size_t, C.c_int32, _F, C.c_uint64, _U8]),
    "sdmi_shard_range": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _I32, _I32]),
    "sdmi_multi_broadcast_count": (C.c_int64, [C.c_void_p]),
    "sdmi_selftest_rank_errors": (C.c_int, [C.c_int32, C.c_int32]),
    "sdmi_op_group_norm": (C.c_int, [_CTX, _F, _F, _F, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, _F]),
    "sdmi_op_group_norm_fp8": (C.c_int, [_CTX, _F, _F, _F, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, _F]),
    "sdmi_op_layer_norm": (C.c_int, [_CTX, _F, _F, _F, C.c_int32, C.c_int32, C.c_float, _F]),
    "sdmi_op_conv2d": (C.c_int, [_CTX, _F, _F, _F, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _F]),
    "sdmi_op_linear": (C.c_int, [_CTX, _F, _F, _F, C.c_int32, C.c_int32, C.c_int32, _F]),
    "sdmi_op_geglu_forward": (C.c_int, [_CTX, _F, _F, _F, C.c_int32, C.c_int32, C.c_int32, _F]),
    "sdmi_op_geglu": (C.c_int, [_CTX, _F, C.c_int32, C.c_int32, _F]),
    "sdmi_op_timestep_embedding": (C.c_int, [_CTX, C.c_int32, C.c_int32, _F]),
    "sdmi_set_option": (C.c_int, [_CTX, C.c_char_p, C.c_char_p]),
    "sdmi_last_call_stats": (C.c_int, [_CTX, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "sdmi_profile_stats": (C.c_int, [_CTX, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "sdmi_profile_overhead": (C.c_int, [_CTX, C.POINTER(C.c_double)]),
    "sdmi_bench_conv": (C.c_int, [_CTX, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double)]),
    "sdmi_bench_attention": (C.c_int, [_CTX, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double)]),
}

_lib = None


def load_library() -> C.CDLL:
    """dlopen libsdmi.so and bind every declared symbol; raises if anything is missing.

    If torch is already imported its bundled libamdhip64 (same SONAME) is
    reused by the loader, so device pointers from torch tensors are valid in
    libsdmi; when torch will be used in the process, import it BEFORE this.
    """
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m stable_diffusion_burn_amd.build` "
            "(libsdmi has no CPU or PyTorch fallback)")
    lib = C.CDLL(str(LIB_PATH), mode=C.RTLD_GLOBAL if "torch" in sys.modules else C.RTLD_LOCAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != 0:
        msg = load_library().sdmi_last_error()
        raise SdmiError(status, msg.decode("utf-8", "replace") if msg else "")
