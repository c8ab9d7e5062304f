"""Mel-spectrogram front-end of the sampler (reference modules.py:75-143).

Both mel types are provided: ``vocos`` (the hot path) and, since round 4, ``bigvgan`` (``get_bigvgan_mel_spectrogram``, modules.py:29-72: reflect
padding of (n_fft - hop) / 2, ``center=False``, ``sqrt(power + 1e-9)``, librosa's Slaney-scale area-normalised filterbank) -- the front-end of the
other vocoder plug; the BigVGAN network itself is a third-party checkout absent from the reference tree.
torchaudio is not a dependency: ``MelSpectrogram(sr 24000, n_fft 1024, win 1024, hop 256, n_mels 100, power=1, center=True,
norm=None, mel_scale='htk')`` is written out with ``torch.stft`` (periodic Hann, reflect padding) and the HTK triangular
filterbank, then ``clamp(min=1e-5).log()``.  A waveform on the GPU -- the inference path: the wrapper keeps the prompt on the device --
goes through libf5hip (``f5_frontend_mel``).  A host-resident waveform raises: the package has no CPU path (round 4: the ``torch.stft``
form that used to serve the CPU tests is gone; its restatement lives in the oracle, ``oracle/cpu_ref.mel_spectrogram``).
"""
from __future__ import annotations

import torch
from torch import nn


def get_vocos_mel_spectrogram(waveform, n_fft=1024, n_mel_channels=100, target_sample_rate=24000, hop_length=256, win_length=1024):
    if waveform.ndim == 3:
        waveform = waveform.squeeze(1)
    assert waveform.ndim == 2
    if waveform.is_cuda:  # the hot path: HIP kernels (frames -> DFT on the fp32-input MFMA -> magnitude -> HTK filterbank -> log), csrc/frontend.hip
        from ..frontend import mel_spectrogram
        return mel_spectrogram(waveform, n_fft=n_fft, hop_length=hop_length, win_length=win_length, n_mel_channels=n_mel_channels,
                               target_sample_rate=target_sample_rate).to(waveform.dtype)
    raise RuntimeError("MelSpec: the waveform must be on the MI355X (libf5hip f5_frontend_mel); this package has no CPU path")


def get_bigvgan_mel_spectrogram(waveform, n_fft=1024, n_mel_channels=100, target_sample_rate=24000, hop_length=256, win_length=1024, fmin=0, fmax=None,
                                center=False):
    """modules.py:29-72 on the device (csrc/frontend.hip, mel_type F5_MEL_BIGVGAN).  fmin / fmax / center keep the reference's defaults."""
    if fmin != 0 or fmax is not None or center:
        raise NotImplementedError("get_bigvgan_mel_spectrogram: only the reference's defaults fmin=0, fmax=None, center=False are on the MI355X path")
    if waveform.ndim == 3:
        waveform = waveform.squeeze(1)
    assert waveform.ndim == 2
    if waveform.is_cuda:
        from ..frontend import mel_spectrogram
        return mel_spectrogram(waveform, n_fft=n_fft, hop_length=hop_length, win_length=win_length, n_mel_channels=n_mel_channels,
                               target_sample_rate=target_sample_rate, mel_type="bigvgan").to(waveform.dtype)
    raise RuntimeError("MelSpec: the waveform must be on the MI355X (libf5hip f5_frontend_mel); this package has no CPU path")


class MelSpec(nn.Module):
    def __init__(self, n_fft=1024, hop_length=256, win_length=1024, n_mel_channels=100, target_sample_rate=24_000, mel_spec_type="vocos"):
        super().__init__()
        assert mel_spec_type in ["vocos", "bigvgan"], "We only support two extract mel backend: vocos or bigvgan"
        self.extractor = get_vocos_mel_spectrogram if mel_spec_type == "vocos" else get_bigvgan_mel_spectrogram  # modules.py:125-128
        self.n_fft, self.hop_length, self.win_length = n_fft, hop_length, win_length
        self.n_mel_channels, self.target_sample_rate = n_mel_channels, target_sample_rate
        self.register_buffer("dummy", torch.tensor(0), persistent=False)

    def forward(self, wav):
        return self.extractor(wav, n_fft=self.n_fft, n_mel_channels=self.n_mel_channels, target_sample_rate=self.target_sample_rate,
                              hop_length=self.hop_length, win_length=self.win_length)
