"""Mel-spectrogram front-end of the sampler (reference modules.py:75-143).

Only the ``vocos`` mel type of the hot path is provided (the bigvgan variant belongs to another vocoder, out of scope).
torchaudio is not a dependency: ``MelSpectrogram(sr 24000, n_fft 1024, win 1024, hop 256, n_mels 100, power=1, center=True,
norm=None, mel_scale='htk')`` is written out with ``torch.stft`` (periodic Hann, reflect padding) and the HTK triangular
filterbank, then ``clamp(min=1e-5).log()``.  A waveform on the GPU -- the inference path: the wrapper keeps the prompt on the device --
goes through libf5hip (``f5_frontend_mel``); the ``torch.stft`` form below only serves host-resident tensors.
"""
from __future__ import annotations

import math

import torch
from torch import nn


def _hz_to_mel(f):
    return 2595.0 * math.log10(1.0 + f / 700.0)


def htk_filterbank(n_freqs, n_mels, sample_rate, f_min=0.0, f_max=None):
    f_max = sample_rate / 2 if f_max is None else f_max
    freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m = torch.linspace(_hz_to_mel(f_min), _hz_to_mel(f_max), n_mels + 2)
    f = 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    width = f[1:] - f[:-1]
    slope = f[None, :] - freqs[:, None]
    return torch.clamp(torch.minimum(-slope[:, :-2] / width[:-1], slope[:, 2:] / width[1:]), min=0.0)  # [n_freqs, n_mels]


def get_vocos_mel_spectrogram(waveform, n_fft=1024, n_mel_channels=100, target_sample_rate=24000, hop_length=256, win_length=1024):
    if waveform.ndim == 3:
        waveform = waveform.squeeze(1)
    assert waveform.ndim == 2
    if waveform.is_cuda:  # the hot path: HIP kernels (frames -> DFT on the fp32-input MFMA -> magnitude -> HTK filterbank -> log), csrc/frontend.hip
        from ..frontend import mel_spectrogram
        return mel_spectrogram(waveform, n_fft=n_fft, hop_length=hop_length, win_length=win_length, n_mel_channels=n_mel_channels,
                               target_sample_rate=target_sample_rate).to(waveform.dtype)
    # host-resident waveforms (the reference is CPU-runnable too; only the CPU tests come here)
    wav = waveform.float()
    window = torch.hann_window(win_length, periodic=True, device=wav.device)
    spec = torch.stft(wav, n_fft, hop_length=hop_length, win_length=win_length, window=window, center=True, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True).abs()
    fb = htk_filterbank(n_fft // 2 + 1, n_mel_channels, target_sample_rate).to(wav.device)
    mel = torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)
    return mel.clamp(min=1e-5).log().to(waveform.dtype)


class MelSpec(nn.Module):
    def __init__(self, n_fft=1024, hop_length=256, win_length=1024, n_mel_channels=100, target_sample_rate=24_000, mel_spec_type="vocos"):
        super().__init__()
        assert mel_spec_type in ["vocos", "bigvgan"], "We only support two extract mel backend: vocos or bigvgan"
        if mel_spec_type != "vocos":
            raise NotImplementedError("only the vocos mel front-end is on the MI355X path")
        self.n_fft, self.hop_length, self.win_length = n_fft, hop_length, win_length
        self.n_mel_channels, self.target_sample_rate = n_mel_channels, target_sample_rate
        self.register_buffer("dummy", torch.tensor(0), persistent=False)

    def forward(self, wav):
        return get_vocos_mel_spectrogram(wav, n_fft=self.n_fft, n_mel_channels=self.n_mel_channels,
                                         target_sample_rate=self.target_sample_rate, hop_length=self.hop_length, win_length=self.win_length)
