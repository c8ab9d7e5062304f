"""Device front-end of the reference audio: log-mel spectrogram and sample-rate conversion through libf5hip
(``f5_frontend_mel`` / ``f5_frontend_resample`` in include/f5hip.h), replacing the two torchaudio transforms on the path
(reference model/modules.py:75-143, infer/f5tts_wrapper.py:338-341).  One native handle per mel configuration, created on first use."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

_handles = {}


def _handle(n_fft, hop, win, n_mels, sample_rate):
    key = (n_fft, hop, win, n_mels, sample_rate)
    h = _handles.get(key)
    if h is None:
        _lib.require_gpu()
        lib = _lib.load()
        cfg = _lib.MelConfig(n_fft=n_fft, hop=hop, win=win, n_mels=n_mels, sample_rate=sample_rate)
        h = C.c_void_p()
        _lib.check(lib.f5_frontend_create(C.byref(cfg), C.byref(h)), "frontend_create")
        _handles[key] = h
    return h


@torch.no_grad()
def mel_spectrogram(wave: torch.Tensor, n_fft=1024, hop_length=256, win_length=1024, n_mel_channels=100, target_sample_rate=24000) -> torch.Tensor:
    """wave [b, nw] on the GPU -> log-mel [b, n_mels, nw // hop + 1] (float32), all arithmetic in libf5hip."""
    lib = _lib.load()
    w = wave.to(dtype=torch.float32).contiguous()
    b, nw = w.shape
    out = torch.empty(b, n_mel_channels, nw // hop_length + 1, device=w.device, dtype=torch.float32)
    _lib.check(lib.f5_frontend_mel(_handle(n_fft, hop_length, win_length, n_mel_channels, target_sample_rate), b, nw, _lib.ptr(w), _lib.ptr(out),
                                   _lib.stream_ptr()), "frontend_mel")
    return out


@torch.no_grad()
def resample(wave: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    """wave [..., n] on the GPU -> [..., ceil(new * n / orig)] (torchaudio.transforms.Resample semantics: sinc_interp_hann, width 6, rolloff 0.99)."""
    lib = _lib.load()
    shape = wave.shape
    w = wave.to(dtype=torch.float32).reshape(-1, shape[-1]).contiguous()
    import math
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    target = -(-new * shape[-1] // orig)
    out = torch.empty(w.shape[0], target, device=w.device, dtype=torch.float32)
    _lib.check(lib.f5_frontend_resample(_handle(1024, 256, 1024, 100, 24000), w.shape[0], shape[-1], int(orig_freq), int(new_freq), _lib.ptr(w),
                                        _lib.ptr(out), _lib.stream_ptr()), "frontend_resample")
    return out.reshape(shape[:-1] + (target,)).to(wave.dtype)
