"""Reference-audio front-end on the device (csrc/frontend.hip through the C ABI): log-mel spectrogram and sample-rate conversion against the
CPU restatements (oracle/cpu_ref.mel_spectrogram = torch.stft + HTK filterbank; oracle/cpu_ref.resample = torchaudio's sinc FIR as a float64 polyphase sum).

Tolerances: the DFT runs on the fp32-input MFMA (exact fp32 products, fp32 sums over 1024 terms): |log-mel difference| <= 2e-3 wherever the mel
energy is above the 1e-5 floor by a factor of 10 (below it, the log of a difference of rounding noise is meaningless); resampling rel-L2 <= 1e-5."""
import math

import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import cpu_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    from eraxvif5tts_amd import _lib
    _lib.require_gpu()


@pytest.mark.parametrize("B,nw", [(1, 24000), (2, 48001), (3, 5000), (1, 600)])
def test_mel_spectrogram_matches_oracle(B, nw):
    from eraxvif5tts_amd.model.modules import MelSpec
    g = torch.Generator().manual_seed(nw)
    t = torch.arange(nw) / 24000.0
    wav = 0.3 * torch.sin(2 * math.pi * (200 + 150 * torch.arange(B)[:, None]) * t[None]) + 0.05 * torch.randn(B, nw, generator=g)
    wav[:, : nw // 5] *= 0.01  # a quiet stretch
    ref = cpu_ref.mel_spectrogram(wav)
    out = MelSpec()(wav.cuda()).cpu()
    assert out.shape == ref.shape == (B, 100, nw // 256 + 1)
    strong = ref > math.log(1e-4)
    assert (out - ref)[strong].abs().max() < 2e-3
    assert (out - ref).abs().max() < 0.2 and (out >= math.log(1e-5) - 1e-6).all()
    assert rel_l2(out.exp(), ref.exp()) < 1e-4


def test_mel_inside_sample_uses_the_device_path():
    """CFM.sample with a raw waveform prompt (cfm.py:103-105): the prompt mel is computed by f5_frontend_mel and is what comes back in the
    prompt frames of the output."""
    import gpu_helpers as G
    arch = dict(dim=128, depth=2, heads=2, ff_mult=2, text_dim=64, conv_layers=1, pe_attn_head=1, text_mask_padding=False)
    V = 40
    W = cpu_ref.random_dit_weights(arch, V, seed=51)
    cfm = G.make_cfm(arch, V, W, "fp32")
    g = torch.Generator().manual_seed(52)
    wav = 0.2 * torch.randn(1, 256 * 30, generator=g)
    text = torch.randint(0, V, (1, 15), generator=g)
    out, _ = cfm.sample(cond=wav.cuda(), text=text.cuda(), duration=80, steps=2, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=3, return_trajectory=False)
    mel = cpu_ref.mel_spectrogram(wav).permute(0, 2, 1)  # [1, 31, 100]
    assert out.shape == (1, 80, 100)
    got = out[0, :31].cpu()
    strong = mel[0] > math.log(1e-4)
    assert (got - mel[0])[strong].abs().max() < 2e-3


@pytest.mark.parametrize("orig,new", [(16000, 24000), (44100, 24000), (48000, 24000), (22050, 24000), (24000, 24000)])
def test_resample_matches_oracle(orig, new):
    """f5_frontend_resample against oracle/cpu_ref.resample, the float64 polyphase restatement of torchaudio's sinc_interp_hann resampler
    (width 6, rolloff 0.99; f5tts_wrapper.py:338-341)."""
    from eraxvif5tts_amd.infer import audio
    g = torch.Generator().manual_seed(orig)
    n = 12345
    t = torch.arange(n) / orig
    wav = torch.stack([0.5 * torch.sin(2 * math.pi * 440 * t) + 0.1 * torch.randn(n, generator=g), 0.2 * torch.randn(n, generator=g)])
    ref = cpu_ref.resample(wav, orig, new)
    out = audio.resample(wav.cuda(), orig, new).cpu()  # device tensor: f5_frontend_resample
    assert out.shape == ref.shape == (2, math.ceil(new // math.gcd(orig, new) * n / (orig // math.gcd(orig, new))))
    assert rel_l2(out, ref) < 1e-5


@pytest.mark.parametrize("nw", [24000, 7777, 2048])
def test_bigvgan_mel_matches_oracle(nw):
    """MelSpec(mel_spec_type="bigvgan") on the device (f5_frontend_mel with mel_type F5_MEL_BIGVGAN: reflect padding of (n_fft - hop) / 2, no centring,
    sqrt(power + 1e-9), Slaney filterbank built in C++) against oracle/cpu_ref.bigvgan_mel_spectrogram, which is pinned to the reference's own
    get_bigvgan_mel_spectrogram (tests/golden/bigvgan_mel.npz).  Same tolerance as the vocos mel: |d log-mel| <= 2e-3 above the floor."""
    from eraxvif5tts_amd.model.modules import MelSpec
    g = torch.Generator().manual_seed(nw)
    t = torch.arange(nw) / 24000.0
    wav = torch.stack([0.3 * torch.sin(2 * math.pi * 200 * t) * (1 + 0.4 * torch.sin(2 * math.pi * 4 * t)) + 0.02 * torch.randn(nw, generator=g),
                       0.1 * torch.randn(nw, generator=g)])
    ref = cpu_ref.bigvgan_mel_spectrogram(wav)
    out = MelSpec(mel_spec_type="bigvgan")(wav.cuda()).cpu()
    assert out.shape == ref.shape == (2, 100, (nw + 768 - 1024) // 256 + 1)
    strong = ref > math.log(1e-4)
    assert (out - ref)[strong].abs().max() < 2e-3 and (out - ref).abs().max() < 5e-2
