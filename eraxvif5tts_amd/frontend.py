"""Device front-end of the reference audio: log-mel spectrogram and sample-rate conversion through libf5hip
(``f5_frontend_mel`` / ``f5_frontend_resample`` in include/f5hip.h), replacing the two torchaudio transforms on the path
(reference model/modules.py:75-143, infer/f5tts_wrapper.py:338-341).  One native handle per (device, mel configuration), created on first
use with that device current (its DFT / filterbank / resampling tables and workspace live there) and destroyed at interpreter exit."""
from __future__ import annotations

import atexit
import ctypes as C

import torch

from . import _lib

_handles = {}


def _device_index(t: torch.Tensor) -> int:
    if not t.is_cuda:
        raise _lib.F5HipError("the device front-end takes tensors that live on the GPU (there is no CPU path in libf5hip)")
    return t.device.index if t.device.index is not None else torch.cuda.current_device()


def _handle(device_index, n_fft, hop, win, n_mels, sample_rate, mel_type=_lib.F5_MEL_VOCOS):
    key = (device_index, n_fft, hop, win, n_mels, sample_rate, mel_type)
    h = _handles.get(key)
    if h is None:
        _lib.require_gpu()
        lib = _lib.load()
        cfg = _lib.MelConfig(n_fft=n_fft, hop=hop, win=win, n_mels=n_mels, sample_rate=sample_rate, mel_type=mel_type)
        h = C.c_void_p()
        with torch.cuda.device(device_index):  # the handle's tables and workspace are allocated on the current device
            _lib.check(lib.f5_frontend_create(C.byref(cfg), C.byref(h)), "frontend_create")
        _handles[key] = h
    return h


@atexit.register
def _destroy_handles():
    if not _handles:
        return
    try:
        lib = _lib.load()
        for h in _handles.values():
            lib.f5_frontend_destroy(h)
    except Exception:  # noqa: BLE001  (interpreter shutdown: the runtime may already be gone)
        pass
    _handles.clear()


@torch.no_grad()
def mel_spectrogram(wave: torch.Tensor, n_fft=1024, hop_length=256, win_length=1024, n_mel_channels=100, target_sample_rate=24000,
                    mel_type="vocos") -> torch.Tensor:
    """wave [b, nw] on the GPU -> log-mel (float32), all arithmetic in libf5hip.  mel_type "vocos": [b, n_mels, nw // hop + 1] (torchaudio
    MelSpectrogram, reference modules.py:75-101); "bigvgan": [b, n_mels, (nw + 2 pad - n_fft) // hop + 1] with pad = (n_fft - hop) // 2
    (get_bigvgan_mel_spectrogram, modules.py:29-72)."""
    lib = _lib.load()
    w = wave.to(dtype=torch.float32).contiguous()
    b, nw = w.shape
    dev = _device_index(w)
    mt = {"vocos": _lib.F5_MEL_VOCOS, "bigvgan": _lib.F5_MEL_BIGVGAN}[mel_type]
    frames = nw // hop_length + 1 if mt == _lib.F5_MEL_VOCOS else (nw + 2 * ((n_fft - hop_length) // 2) - n_fft) // hop_length + 1
    out = torch.empty(b, n_mel_channels, frames, device=w.device, dtype=torch.float32)
    with torch.cuda.device(dev):
        _lib.check(lib.f5_frontend_mel(_handle(dev, n_fft, hop_length, win_length, n_mel_channels, target_sample_rate, mt), b, nw, _lib.ptr(w),
                                       _lib.ptr(out), _lib.stream_ptr()), "frontend_mel")
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
    dev = _device_index(w)
    out = torch.empty(w.shape[0], target, device=w.device, dtype=torch.float32)
    with torch.cuda.device(dev):  # (the resampler shares the default mel handle of its device: one handle type in the C ABI)
        _lib.check(lib.f5_frontend_resample(_handle(dev, 1024, 256, 1024, 100, 24000), w.shape[0], shape[-1], int(orig_freq), int(new_freq),
                                            _lib.ptr(w), _lib.ptr(out), _lib.stream_ptr()), "frontend_resample")
    return out.reshape(shape[:-1] + (target,)).to(wave.dtype)
