"""ctypes binding of libegot2x.so (C ABI declared in include/egot2x.h).

The product path has no CPU fallback: if the HIP library is missing or fails to load, every call raises.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# EGX_LIB (development aid): load a variant build (egot2_amd/_variants/lib_<name>.so, tools/build_variant.py) instead
LIB_PATH = os.environ.get("EGX_LIB") or os.path.join(_PKG, "libegot2x.so")

EGX_ABI_VERSION = 16
EGX_MAX_SEGMENTS = 8
EGX_F32, EGX_BF16, EGX_F32_SPLIT = 0, 1, 2
EGX_IMPL_AUTO, EGX_IMPL_GENERIC, EGX_IMPL_FUSED, EGX_IMPL_WIDE, EGX_IMPL_TILED = 0, 1, 2, 3, 4

_fp = C.c_void_p  # all device pointers travel as raw addresses


class Segment(C.Structure):
    _fields_ = [("feat", _fp), ("T", C.c_int), ("d_in", C.c_int), ("proj_w", _fp), ("proj_b", _fp),
                ("add_vec", _fp), ("pos", _fp), ("pos_stride", C.c_int), ("feat_bf16", C.c_int), ("pool", C.c_int)]


class SegmentGrads(C.Structure):
    _fields_ = [("proj_w", _fp), ("proj_b", _fp), ("add_vec", _fp), ("pos", _fp), ("feat", _fp)]


_LAYER_FIELDS = ["in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "lin1_w", "lin1_b", "lin2_w", "lin2_b",
                 "norm1_w", "norm1_b", "norm2_w", "norm2_b"]


class Layer(C.Structure):
    _fields_ = [(n, _fp) for n in _LAYER_FIELDS]


class LayerGrads(C.Structure):
    _fields_ = [(n, _fp) for n in _LAYER_FIELDS]


class Head(C.Structure):
    _fields_ = [("ln_w", _fp), ("ln_b", _fp), ("W", _fp), ("b", _fp), ("n_out", C.c_int)]


class HeadGrads(C.Structure):
    _fields_ = [("ln_w", _fp), ("ln_b", _fp), ("W", _fp), ("b", _fp)]


class Config(C.Structure):
    _fields_ = [("d_model", C.c_int), ("n_heads", C.c_int), ("d_ff", C.c_int), ("n_layers", C.c_int),
                ("n_segments", C.c_int), ("ln_eps", C.c_float), ("compute", C.c_int), ("impl", C.c_int),
                ("p_drop", C.c_float), ("p_pos", C.c_float), ("p_feat", C.c_float), ("seed_ptr", _fp),
                ("advance_seed", C.c_int), ("zero_buf", _fp), ("zero_bytes", C.c_size_t), ("bwd_stage", C.c_int),
                ("deterministic", C.c_int), ("out_tokens", C.c_int), ("bucket_cb", C.c_void_p), ("bucket_user", C.c_void_p),
                # ABI v15: persistent packed-weight cache, upstream gradient of a fused loss, fused weighted cross entropy
                ("weight_cache", _fp), ("weight_cache_valid", C.c_int), ("d_logits_scale", _fp), ("ce", C.c_void_p),
                ("token_ce", C.c_void_p)]


class Ce(C.Structure):
    """egx_ce: weighted cross entropy on the pooled head's logits, evaluated by the translator forward itself."""
    _fields_ = [("target", _fp), ("class_weight", _fp), ("loss", _fp), ("d_logits", _fp)]


class TokenCe(C.Structure):
    """egx_token_ce: per-token classifier + weighted cross entropy on the tokens the encoder returns (the ASD task's lossAV)."""
    _fields_ = [("W", _fp), ("b", _fp), ("target", _fp), ("class_weight", _fp), ("C", C.c_int), ("logits", _fp), ("probs", _fp), ("pred", _fp),
                ("loss", _fp), ("correct", _fp), ("d_logits", _fp), ("d_W", _fp), ("d_b", _fp)]


BUCKET_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int)


class DecConfig(C.Structure):
    _fields_ = [("d_model", C.c_int), ("n_heads", C.c_int), ("d_ff", C.c_int), ("n_layers", C.c_int), ("vocab", C.c_int),
                ("sy", C.c_int), ("S", C.c_int), ("ln_eps", C.c_float), ("compute", C.c_int), ("p_drop", C.c_float),
                ("p_pos", C.c_float), ("seed_ptr", C.c_void_p)]


_DEC_LAYER_FIELDS = ["sa_in_w", "sa_in_b", "sa_out_w", "sa_out_b", "norm1_w", "norm1_b", "ca_in_w", "ca_in_b", "ca_out_w",
                     "ca_out_b", "norm2_w", "norm2_b", "lin1_w", "lin1_b", "lin2_w", "lin2_b", "norm3_w", "norm3_b"]


class DecLayer(C.Structure):
    _fields_ = [(n, _fp) for n in _DEC_LAYER_FIELDS]


class DecLayerGrads(C.Structure):
    _fields_ = [(n, _fp) for n in _DEC_LAYER_FIELDS]


# symbol -> (restype, argtypes); every symbol include/egot2x.h declares
SIGNATURES = {
    "egx_abi_version": (C.c_int, []),
    "egx_last_error": (C.c_char_p, []),
    "egx_launch_count": (C.c_longlong, [C.c_int]),
    "egx_tuning_reload": (None, []),
    "egx_weight_cache_bytes": (C.c_size_t, [C.POINTER(Config), C.POINTER(Segment)]),
    "egx_encoder_workspace": (C.c_int, [C.POINTER(Config), C.POINTER(Segment), C.c_int,
                                        C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "egx_encoder_uses_fused": (C.c_int, [C.POINTER(Config), C.POINTER(Segment), C.c_int]),
    "egx_encoder_token_ce_ok": (C.c_int, [C.POINTER(Config), C.POINTER(Segment), C.c_int]),
    "egx_encoder_impl": (C.c_int, [C.POINTER(Config), C.POINTER(Segment), C.c_int]),
    "egx_encoder_slices": (C.c_int, [C.POINTER(Config), C.POINTER(Segment), C.c_int]),
    "egx_wide_gemm_scratch": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "egx_wide_gemm": (C.c_int, [C.c_int, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, _fp, _fp]),
    "egx_wide_attention_fwd": (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64, _fp]),
    "egx_wide_attention_bwd": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64, _fp]),
    "egx_encoder_fwd": (C.c_int, [C.POINTER(Config), C.POINTER(Segment), _fp, _fp, C.POINTER(Layer), C.c_int,
                                  _fp, _fp, _fp, C.c_int, C.c_uint64, _fp]),
    "egx_encoder_bwd": (C.c_int, [C.POINTER(Config), C.POINTER(Segment), _fp, _fp, C.POINTER(Layer), C.c_int,
                                  _fp, _fp, _fp, C.POINTER(SegmentGrads), _fp, _fp, C.POINTER(LayerGrads),
                                  C.c_int, C.c_uint64, _fp]),
    "egx_translator_workspace": (C.c_int, [C.POINTER(Config), C.POINTER(Segment), C.c_int,
                                           C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "egx_translator_fwd": (C.c_int, [C.POINTER(Config), C.POINTER(Segment), _fp, _fp, C.POINTER(Layer), C.POINTER(Head),
                                     C.c_int, _fp, _fp, _fp, _fp, C.c_int, C.c_uint64, _fp]),
    "egx_translator_bwd": (C.c_int, [C.POINTER(Config), C.POINTER(Segment), _fp, _fp, C.POINTER(Layer), C.POINTER(Head),
                                     C.c_int, _fp, _fp, _fp, C.POINTER(SegmentGrads), _fp, _fp, C.POINTER(LayerGrads),
                                     C.POINTER(HeadGrads), C.c_int, C.c_uint64, _fp]),
    "egx_small_attention_fwd": (C.c_int, [_fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint32, _fp]),
    "egx_small_attention_bwd": (C.c_int, [_fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, _fp, _fp, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint32, _fp]),
    "egx_embed_pos_fwd": (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_float, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                    C.c_uint64, _fp]),
    "egx_embed_pos_bwd": (C.c_int, [_fp, _fp, _fp, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64, _fp]),
    "egx_decoder_workspace": (C.c_int, [C.POINTER(DecConfig), C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "egx_decoder_fwd": (C.c_int, [C.POINTER(DecConfig), _fp, _fp, _fp, _fp, C.c_int, C.POINTER(DecLayer), _fp, _fp, C.c_int, _fp, _fp,
                                  _fp, C.c_int, C.c_uint64, _fp]),
    "egx_decoder_bwd": (C.c_int, [C.POINTER(DecConfig), _fp, C.POINTER(DecLayer), _fp, C.c_int, _fp, _fp, _fp, _fp, _fp,
                                  C.POINTER(DecLayerGrads), _fp, _fp, _fp, C.c_size_t, C.c_int, C.c_uint64, _fp]),
    "egx_comm_unique_id": (C.c_int, [_fp]),
    "egx_comm_create": (C.c_int, [_fp, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "egx_comm_size": (C.c_int, [_fp]),
    "egx_allreduce": (C.c_int, [_fp, _fp, C.c_size_t, C.c_int, C.c_int, _fp]),
    "egx_comm_destroy": (C.c_int, [_fp]),
    "egx_comm_library": (C.c_char_p, []),
    "egx_relu_mask": (C.c_int, [_fp, _fp, C.c_size_t, _fp]),
    "egx_dropout": (C.c_int, [_fp, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint32, _fp]),
    "egx_pool_pack": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                _fp, C.c_int, C.c_longlong, _fp]),
    "egx_weighted_ce": (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, _fp, _fp, _fp]),
    "egx_linear_ce_scratch": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "egx_linear_ce_fwd": (C.c_int, [_fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    "egx_linear_ce_bwd": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp]),
    "egx_counter_add": (C.c_int, [_fp, C.c_int64, _fp]),
    "egx_adam_step": (C.c_int, [_fp, _fp, _fp, _fp, C.c_size_t, _fp, C.c_float, C.c_float, C.c_float, C.c_float,
                                C.c_float, C.c_int, C.c_float, _fp]),
    "egx_pool_head_fwd": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_float, _fp, _fp, C.c_int,
                                    _fp, _fp, _fp]),
    "egx_pool_head_bwd": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_float, _fp, C.c_int,
                                    _fp, _fp, _fp, _fp, _fp, _fp]),
    "egx_linear_fwd": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "egx_linear_residual_fwd": (C.c_int, [_fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "egx_gelu_fwd": (C.c_int, [_fp, _fp, C.c_size_t, _fp]),
    "egx_gelu_bwd": (C.c_int, [_fp, _fp, _fp, C.c_size_t, _fp]),
    "egx_linear_bwd_scratch": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "egx_linear_bwd": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "egx_gemm": (C.c_int, [C.c_int, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, C.c_int, _fp,
                           C.c_size_t, _fp]),
    "egx_layernorm_fwd": (C.c_int, [_fp, _fp, _fp, _fp, C.c_float, _fp, _fp, _fp, C.c_int, C.c_int, _fp]),
    "egx_layernorm_bwd": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, _fp]),
    "egx_attention_fwd": (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64, _fp]),
    "egx_slices_stolen": (C.c_longlong, [C.c_int]),
    "egx_debug_stamps": (C.c_int, [_fp, C.c_int]),
    "egx_seed_advance": (C.c_int, [_fp, _fp]),
    "egx_timing_enable": (None, [C.c_int]),
    "egx_timing_read": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "egx_ffn_dw_scratch": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "egx_ffn_dw": (C.c_int, [_fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64, _fp, _fp, _fp,
                             C.c_int, _fp, _fp]),
    "egx_attention_bwd": (C.c_int, [_fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                    C.c_uint64, _fp]),
}

_lib = None


class EgxError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libegot2x.so (once). Raises loudly when the HIP extension is missing: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EgxError(
            f"{LIB_PATH} is missing: build it with `python -m egot2_amd.build` (hipcc --offload-arch=gfx950). "
            "egot2_amd has no CPU or PyTorch fallback for the translator.")
    # PyTorch owns the in-process HIP runtime (its wheel bundles libamdhip64): import it first so that
    # libegot2x.so binds to the SAME runtime instance; loading ours first splits the process across two runtimes.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    # EGX_LIB=<path> loads a variant build (A/B aid). It goes through the SAME ABI-version and symbol checks: a stale variant
    # with other struct layouts would corrupt memory instead of failing. EGX_LIB_UNSAFE=1 (development only) skips them.
    variant = bool(os.environ.get("EGX_LIB"))
    unsafe = variant and os.environ.get("EGX_LIB_UNSAFE") == "1"
    if variant:
        import warnings
        warnings.warn(f"egot2_amd: EGX_LIB is set, loading {LIB_PATH} instead of the product library"
                      + (" WITHOUT ABI checks (EGX_LIB_UNSAFE=1)" if unsafe else ""))
    for name, (res, args) in SIGNATURES.items():
        if unsafe and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    ver = lib.egx_abi_version()
    if ver != EGX_ABI_VERSION and not unsafe:
        raise EgxError(f"libegot2x ABI version {ver} != binding version {EGX_ABI_VERSION}; rebuild the library")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().egx_last_error()
        raise EgxError("libegot2x: " + (msg.decode() if msg else f"error code {rc}"))


def ptr(t) -> int | None:
    """Device address of a contiguous fp32 tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()
