"""ctypes binding of libf5hip.so (C ABI: include/f5hip.h).  Fails loudly when the library or the GPU is missing."""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must be imported first: libf5hip resolves libamdhip64.so.7 to the runtime torch already loaded)

_HERE = os.path.dirname(os.path.abspath(__file__))
# (F5HIP_LIB: developer override, another in-tree build of the library for same-box A/B runs of two builds)
LIB_PATH = os.environ.get("F5HIP_LIB") or os.path.join(_HERE, "lib", "libf5hip.so")

F5_PREC_BF16, F5_PREC_FP32 = 0, 1
F5_ODE_EULER, F5_ODE_MIDPOINT = 0, 1
F5_ROPE_ADJACENT, F5_ROPE_HALF_SPLIT = 0, 1
F5_BACKBONE_DIT, F5_BACKBONE_UNETT, F5_BACKBONE_MMDIT = 0, 1, 2
F5_SKIP = {"concat": 0, "add": 1, "none": 2}
SITES = ("qkv", "attention", "attn_out", "ff1", "ff2", "ln1", "ln2", "conv31", "input_proj")  # F5_SITE_* order
F5_MEL_VOCOS, F5_MEL_BIGVGAN = 0, 1
ACT = {"none": 0, "gelu_tanh": 1, "gelu_erf": 2, "mish": 3}


class F5HipError(RuntimeError):
    pass


class DitConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dim", "depth", "heads", "dim_head", "ff_inner", "mel_dim", "text_num_embeds", "text_dim",
                                         "conv_layers", "text_mask_padding", "pe_attn_head", "qk_norm", "long_skip", "precision", "rope_layout",
                                         "backbone", "skip_connect")]


class VocosConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_mels", "dim", "inter_dim", "layers", "n_fft", "hop")]


class BigVGANConfig(C.Structure):  # struct f5_bigvgan_config
    _fields_ = [("num_mels", C.c_int32), ("upsample_initial_channel", C.c_int32), ("num_upsamples", C.c_int32), ("upsample_rates", C.c_int32 * 8),
                ("upsample_kernel_sizes", C.c_int32 * 8), ("num_kernels", C.c_int32), ("resblock_kernel_sizes", C.c_int32 * 4),
                ("resblock_dilations", (C.c_int32 * 3) * 4), ("snake_logscale", C.c_int32), ("use_tanh_at_final", C.c_int32),
                ("use_bias_at_final", C.c_int32)]


class MelConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_fft", "hop", "win", "n_mels", "sample_rate", "mel_type")]


class DurationWeights(C.Structure):  # struct f5_duration_weights
    _fields_ = [(n, C.c_void_p) for n in ("text_embed", "conv1_w", "conv1_b", "norm1_w", "norm1_b", "conv2_w", "conv2_b", "norm2_w", "norm2_b",
                                          "proj_w", "proj_b")] + [(n, C.c_int32) for n in ("vocab_rows", "in_channels", "filter_channels", "kernel_size")] + \
               [("cond_w", C.c_void_p), ("cond_b", C.c_void_p), ("gin_channels", C.c_int32)]


_P, _I, _F = C.c_void_p, C.c_int, C.c_float
_PROTOS = {
    "f5_last_error": (C.c_char_p, []),
    "f5_version": (_I, []),
    "f5_device_count": (_I, [C.c_char_p]),
    "f5_model_create": (_I, [C.POINTER(DitConfig), C.POINTER(_P)]),
    "f5_model_set_tensor": (_I, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), _I]),
    "f5_model_has_tensor": (_I, [_P, C.c_char_p, C.POINTER(C.c_int64)]),
    "f5_model_finalize": (_I, [_P]),
    "f5_model_destroy": (_I, [_P]),
    "f5_plan_create": (_I, [_P, _I, _I, _I, C.POINTER(_P)]),
    "f5_plan_destroy": (_I, [_P]),
    "f5_plan_workspace_bytes": (C.c_int64, [_P]),
    "f5_sample": (_I, [_P, _I, _I, _P, _P, _I, _P, _P, _P, _P, _I, _F, _I, _P, _P, _I, _P]),
    "f5_sample_finish": (_I, [_P, _P]),
    "f5_text_embed": (_I, [_P, _I, _I, _P, _I, _I, _P, _P]),
    "f5_dit_forward": (_I, [_P, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P]),
    "f5_sample_ragged": (_I, [_P, _I, _P, _P, _P, _I, _P, _P, _P, _I, C.c_float, _I, _P, _P]),
    "f5_mmdit_forward": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P]),
    "f5_plan_timing_begin": (_I, [_P, _I]),
    "f5_plan_timing_end": (_I, [_P, C.POINTER(C.c_float), C.POINTER(C.c_int), _P]),
    "f5_plan_timing_site": (_I, [_P, _I, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "f5_plan_set_tap": (_I, [_P, C.c_char_p, _P]),
    "f5_plan_set_option": (_I, [_P, C.c_char_p, _I]),
    "f5_plan_get_option": (_I, [_P, C.c_char_p, C.POINTER(C.c_int)]),
    "f5_duration_predict": (_I, [C.POINTER(DurationWeights), _I, _I, _P, _I, _P, _P, _P, _P]),
    "f5_duration_predict_g": (_I, [C.POINTER(DurationWeights), _I, _I, _P, _I, _P, _P, _I, _P, _P, _P]),
    "f5_op_linear": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P]),
    "f5_op_linear_fused": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P, _I, _I, _P, _P]),
    "f5_op_ln_fold": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _P, _P, _P]),
    "f5_op_layernorm_modulate": (_I, [_I, _I, _P, _P, _P, _P, _P]),
    "f5_op_attention": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "f5_op_conv_pos_embed": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "f5_bench_gemm_site": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, C.POINTER(C.c_float), _P]),
    "f5_bench_attention": (_I, [_I, _I, _I, _I, _I, C.POINTER(C.c_float), _P]),
    "f5_bench_mfma_rate": (_I, [_I, C.POINTER(C.c_float), _P]),
    "f5_tuning_set": (_I, [C.c_char_p, _I]),
    "f5_debug_attn_stamps": (_I, [_P]),
    "f5_debug_gemm_clock": (_I, [_P]),
    "f5_vocoder_create": (_I, [C.POINTER(VocosConfig), C.POINTER(_P)]),
    "f5_vocoder_set_tensor": (_I, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), _I]),
    "f5_vocoder_has_tensor": (_I, [_P, C.c_char_p, C.POINTER(C.c_int64)]),
    "f5_vocoder_finalize": (_I, [_P]),
    "f5_vocoder_destroy": (_I, [_P]),
    "f5_vocoder_decode": (_I, [_P, _I, _I, _P, _P, _P]),
    "f5_vocoder_istft_head": (_I, [_P, _I, _I, _P, _P, _P]),
    "f5_bigvgan_create": (_I, [C.POINTER(BigVGANConfig), C.POINTER(_P)]),
    "f5_bigvgan_set_tensor": (_I, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), _I]),
    "f5_bigvgan_has_tensor": (_I, [_P, C.c_char_p, C.POINTER(C.c_int64)]),
    "f5_bigvgan_finalize": (_I, [_P]),
    "f5_bigvgan_destroy": (_I, [_P]),
    "f5_bigvgan_forward": (_I, [_P, _I, _I, _P, _P, _P]),
    "f5_frontend_create": (_I, [C.POINTER(MelConfig), C.POINTER(_P)]),
    "f5_frontend_destroy": (_I, [_P]),
    "f5_frontend_mel": (_I, [_P, _I, _I, _P, _P, _P]),
    "f5_frontend_resample": (_I, [_P, _I, _I, _I, _I, _P, _P, _P]),
}
EXPORTS = tuple(_PROTOS)

_lib = None


def load(build_if_missing: bool = False):
    """Load libf5hip.so (no compute).  Raises F5HipError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        if build_if_missing:
            from . import build as _b
            _b.build(verbose=False)
        else:
            raise F5HipError(f"{LIB_PATH} not found: build it with `python -m eraxvif5tts_amd.build` (hipcc, gfx950). "
                             "There is no CPU / eager-PyTorch fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    # developer knobs for A/B runs: F5HIP_TUNING="gemm_nt=33,attn_variant=1" -> f5_tuning_set(key, value)
    for item in filter(None, os.environ.get("F5HIP_TUNING", "").split(",")):
        key, _, val = item.partition("=")
        if lib.f5_tuning_set(key.strip().encode(), int(val)) != 0:
            raise F5HipError(f"F5HIP_TUNING: {lib.f5_last_error().decode('utf-8', 'replace')}")
    return lib


def last_error() -> str:
    return load().f5_last_error().decode("utf-8", "replace")


def check(rc: int, what: str = ""):
    if rc != 0:
        raise F5HipError(f"libf5hip {what} failed (code {rc}): {last_error()}")


def require_gpu():
    """The product path needs a real MI355X; anything else is an error, never a silent fallback."""
    lib = load()
    if not torch.cuda.is_available():
        raise F5HipError("no ROCm device visible to PyTorch: the HIP hot path cannot run (no CPU fallback)")
    name = C.create_string_buffer(64)
    if lib.f5_device_count(name) <= 0:
        raise F5HipError("no gfx950 (MI355X) device usable by libf5hip")
    return name.value.decode()


def ptr(t):
    """Device/host pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "libf5hip takes contiguous tensors"
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def set_tensors(handle, setter, has, state: dict):
    """Upload every tensor of `state` the native handle knows (strict=False semantics); returns the names used."""
    lib = load()
    used = []
    for name, t in state.items():
        if not getattr(lib, has)(handle, name.encode(), None):
            continue
        h = t.detach().to("cpu", torch.float32).contiguous()
        shape = (C.c_int64 * h.ndim)(*h.shape)
        check(getattr(lib, setter)(handle, name.encode(), C.c_void_p(h.data_ptr()), shape, h.ndim), f"set_tensor({name})")
        used.append(name)
    return used
