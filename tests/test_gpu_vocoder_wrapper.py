"""Vocos vocoder (plug point B) and the F5TTSWrapper facade on the GPU, through the C ABI.

Tolerances: the vocoder runs on the fp32-input MFMA (exact fp32 products, fp32 accumulation): rel-L2 <= 1e-4 against the
CPU oracle (sum order + exp/sin/cos ulps); the ISTFT head alone <= 1e-5 against torch.istft."""
import math
import os

import numpy as np
import pytest
import torch
import yaml

from conftest import rel_l2
from oracle import cpu_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    from eraxvif5tts_amd import _lib
    _lib.require_gpu()


def _vocos(V, **hp):
    from eraxvif5tts_amd.vocos import Vocos
    v = Vocos(**hp)
    v.load_state_dict({k: t for k, t in V.items() if k in v.state_dict()}, strict=False)
    return v.cuda()


# (2, 683) / (1, 2731): the C5 lengths -- the generated part of a C2 utterance (1024 - 341 frames) and of a C4 one (4096 - 1365)
@pytest.mark.parametrize("B,T", [(1, 40), (2, 129), (3, 7), (2, 683), (1, 2731)])
def test_vocos_decode_matches_oracle(B, T):
    V = cpu_ref.random_vocos_weights(seed=3)
    voc = _vocos(V)
    mel = torch.randn(B, 100, T, generator=torch.Generator().manual_seed(T)) * 2 - 3
    ref = cpu_ref.vocos_decode(V, mel)
    out = voc.decode(mel.cuda()).cpu()
    assert out.shape == ref.shape == (B, (T - 1) * 256)
    assert rel_l2(out, ref) < 1e-4


def test_istft_head_matches_torch_istft_and_round_trips():
    V = cpu_ref.random_vocos_weights(seed=4)
    voc = _vocos(V)
    g = torch.Generator().manual_seed(0)
    # (1) random head activations vs torch.istft on the same spectrum
    B, T = 2, 33
    head = torch.cat([torch.randn(B, T, 513, generator=g) * 0.5, torch.randn(B, T, 513, generator=g) * 3.0], dim=-1)
    mag = torch.exp(head[..., :513]).clamp(max=1e2).transpose(1, 2)
    ph = head[..., 513:].transpose(1, 2)
    spec = torch.complex(mag * torch.cos(ph), mag * torch.sin(ph))
    spec[:, 0].imag.zero_()
    spec[:, -1].imag.zero_()  # irfft ignores the imaginary part of DC / Nyquist
    ref = torch.istft(spec, 1024, hop_length=256, win_length=1024, window=torch.hann_window(1024), center=True)
    out = voc.istft_head(head.cuda()).cpu()
    assert rel_l2(out, ref) < 1e-5
    # (2) size-independent property at the benchmark length: ISTFT(STFT(x)) == x  (T = 683 frames = generated part of C2)
    x = torch.randn(1, 682 * 256, generator=g) * 0.1
    S = torch.stft(x, 1024, hop_length=256, win_length=1024, window=torch.hann_window(1024), center=True, return_complex=True)
    head = torch.cat([torch.log(S.abs().clamp_min(1e-7)), torch.angle(S)], dim=1).transpose(1, 2).contiguous()
    y = voc.istft_head(head.cuda()).cpu()
    assert y.shape == x.shape and (y - x).abs().max() < 2e-4


def test_istft_head_fft_equals_dense_dft():
    """The two forms of the ISTFT head in the library: the LDS FFT (production, n_fft = 1024) against the dense inverse-DFT GEMM it replaced
    (kept behind the `vocos_fft` knob as the cross-check), on random head activations including clipped magnitudes (exp > 1e2)."""
    from eraxvif5tts_amd import _lib
    V = cpu_ref.random_vocos_weights(seed=8)
    voc = _vocos(V)
    g = torch.Generator().manual_seed(3)
    B, T = 3, 57
    head = torch.cat([torch.randn(B, T, 513, generator=g) * 2.0 + 1.0, torch.randn(B, T, 513, generator=g) * 3.0], dim=-1)
    head[0, 5, :40] = 7.0  # exp(7) > 100: the clip is live
    fft = voc.istft_head(head.cuda()).cpu()
    _lib.check(_lib.load().f5_tuning_set(b"vocos_fft", 0))
    try:
        dft = voc.istft_head(head.cuda()).cpu()
    finally:
        _lib.check(_lib.load().f5_tuning_set(b"vocos_fft", 1))
    assert torch.isfinite(fft).all() and rel_l2(fft, dft) < 2e-6


def _write_tiny_assets(tmp, arch, V, W, vocos_hp, VW, backbone="DiT"):
    cfg = {"model": {"name": "tiny_custom", "backbone": backbone, "arch": arch,
                     "mel_spec": {"target_sample_rate": 24000, "n_mel_channels": 100, "hop_length": 256, "win_length": 1024, "n_fft": 1024,
                                  "mel_spec_type": "vocos"}}}
    cfg_path = os.path.join(tmp, "tiny_custom.yaml")
    with open(cfg_path, "w") as f:
        yaml.safe_dump(cfg, f)
    ema = {"ema_model.transformer." + k: v for k, v in W.items()}
    ema.update({"initted": torch.tensor(True), "step": torch.tensor(7), "ema_model.mel_spec.mel_stft.spectrogram.window": torch.hann_window(1024)})
    ckpt = os.path.join(tmp, "model_7.pt")
    torch.save({"ema_model_state_dict": ema}, ckpt)
    vdir = os.path.join(tmp, "vocos")
    os.makedirs(vdir)
    with open(os.path.join(vdir, "config.yaml"), "w") as f:
        yaml.safe_dump({"feature_extractor": {"init_args": {"n_fft": 1024, "hop_length": 256, "n_mels": 100}},
                        "backbone": {"init_args": {"input_channels": 100, "dim": vocos_hp["dim"], "intermediate_dim": vocos_hp["intermediate_dim"],
                                                   "num_layers": vocos_hp["num_layers"]}},
                        "head": {"init_args": {"dim": vocos_hp["dim"], "n_fft": 1024, "hop_length": 256}}}, f)
    torch.save({**VW, "feature_extractor.mel_spec.spectrogram.window": torch.hann_window(1024)}, os.path.join(vdir, "pytorch_model.bin"))
    vocab = os.path.join(tmp, "vocab.txt")
    with open(vocab, "w", encoding="utf-8") as f:
        f.write(" \n" + "\n".join(list("abcdefghijklmnopqrstuvwxyz.,!?'")) + "\n")
    return cfg_path, ckpt, vdir, vocab


@pytest.mark.parametrize("backbone", ["UNetT", "MMDiT"])
def test_wrapper_with_the_other_backbones(tmp_path, backbone):
    """plug point A through the facade: a config whose `model.backbone` is UNetT / MMDiT (the reference resolves f5_tts.model.<backbone>,
    infer/f5tts_wrapper.py:134; configs/E2TTS_*.yaml) + an EMA checkpoint with that class's tensor names -> preprocess_reference -> generate()."""
    from eraxvif5tts_amd.infer import audio
    from eraxvif5tts_amd.infer.f5tts_wrapper import F5TTSWrapper
    from eraxvif5tts_amd import model as _model
    V = 32
    if backbone == "UNetT":
        arch = dict(dim=128, depth=4, heads=2, ff_mult=2, pe_attn_head=1, text_mask_padding=False)
        W = cpu_ref.random_unett_weights(arch, V, seed=45)
    else:
        arch = dict(dim=128, depth=3, heads=2, ff_mult=2, text_mask_padding=True)
        W = cpu_ref.random_mmdit_weights(arch, V, seed=47)
    hp = dict(dim=64, intermediate_dim=128, num_layers=2)
    VW = cpu_ref.random_vocos_weights(seed=46, dim=64, inter=128, layers=2)
    cfg_path, ckpt, vdir, vocab = _write_tiny_assets(str(tmp_path), arch, V, W, hp, VW, backbone=backbone)
    sr = 24000
    t = np.arange(int(2.0 * sr)) / sr
    ref_wav = os.path.join(str(tmp_path), "ref.wav")
    audio.write_wav(ref_wav, 0.2 * np.sin(2 * np.pi * 170 * t), sr)
    tts = F5TTSWrapper(model_name=cfg_path, ckpt_path=ckpt, vocab_file=vocab, use_local_vocoder=True, vocoder_path=vdir, precision="fp32")
    assert type(tts.model.transformer) is getattr(_model, backbone)
    tts.preprocess_reference(ref_wav, "a steady tone")
    torch.manual_seed(3)
    wave, rate, spec = tts.generate("hello there.", nfe_step=3, return_numpy=True, return_spectrogram=True)
    assert rate == 24000 and np.isfinite(wave).all() and spec.shape[0] == 100 and len(wave) == (spec.shape[1] - 1) * 256
    # the bundled E2-TTS configs resolve to the same class
    import yaml as _yaml
    from eraxvif5tts_amd.infer.f5tts_wrapper import _CONFIG_DIR
    for name in ("E2TTS_Base", "E2TTS_Small"):
        with open(os.path.join(_CONFIG_DIR, name + ".yaml")) as f:
            assert _yaml.safe_load(f)["model"]["backbone"] == "UNetT"


def test_wrapper_end_to_end(tmp_path):
    """checkpoint (.pt with ema_model.* keys) + local Vocos dir + vocab file -> preprocess_reference -> generate, and the same
    mel through the oracle vocoder."""
    from eraxvif5tts_amd.infer import audio
    from eraxvif5tts_amd.infer.f5tts_wrapper import F5TTSWrapper
    arch = dict(dim=128, depth=2, heads=2, ff_mult=2, text_dim=64, conv_layers=2, pe_attn_head=1, text_mask_padding=False)
    V = 32
    W = cpu_ref.random_dit_weights(arch, V, seed=5)
    hp = dict(dim=64, intermediate_dim=128, num_layers=2)
    VW = cpu_ref.random_vocos_weights(seed=6, dim=64, inter=128, layers=2)
    cfg_path, ckpt, vdir, vocab = _write_tiny_assets(str(tmp_path), arch, V, W, hp, VW)
    # 2.2 s synthetic prompt: silence + quiet tone + silence (exercises edge trimming and the rms boost)
    sr = 24000
    t = np.arange(int(1.8 * sr)) / sr
    tone = 0.02 * np.sin(2 * np.pi * 180 * t) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t))
    wav = np.concatenate([np.zeros(int(0.2 * sr)), tone, np.zeros(int(0.2 * sr))])
    ref_wav = os.path.join(str(tmp_path), "ref.wav")
    audio.write_wav(ref_wav, wav, sr)

    tts = F5TTSWrapper(model_name=cfg_path, ckpt_path=ckpt, vocab_file=vocab, use_local_vocoder=True, vocoder_path=vdir, precision="fp32")
    with pytest.raises(ValueError):
        tts.generate("hello")
    aud, ref_text = tts.preprocess_reference(ref_wav, "a quiet tone")
    assert ref_text == "a quiet tone. "
    assert aud.shape[0] == 1 and abs(float(torch.sqrt(torch.mean(aud ** 2))) - 0.1) < 2e-3  # boosted to the target rms
    assert tts.ref_audio_len == aud.shape[-1] // 256 and abs(tts.get_current_audio_length() - 1.85) < 0.05
    text = "hello there, this is a test. " * 3 + "and one more sentence to force a second chunk, because the budget is small."
    wave, rate, spec = tts.generate(text, nfe_step=4, return_numpy=True, return_spectrogram=True)
    assert rate == 24000 and wave.ndim == 1 and np.isfinite(wave).all() and spec.shape[0] == 100
    # chunking + cross-fade bookkeeping: every chunk contributes (frames - 1) * 256 samples, overlaps of 3600 removed
    from eraxvif5tts_amd.infer.utils_infer import chunk_text
    n_chunks = len(chunk_text(text, max_chars=int(len(ref_text.encode()) / tts.get_current_audio_length() * (22 - tts.get_current_audio_length()))))
    assert n_chunks >= 2
    assert len(wave) == (spec.shape[1] - n_chunks) * 256 - (n_chunks - 1) * 3600
    # the vocoder half against the oracle on the very same mel
    ref_wave = cpu_ref.vocos_decode({k: v for k, v in VW.items()}, torch.from_numpy(spec[None, :, : 40]).float())
    got = tts.vocoder.decode(torch.from_numpy(spec[None, :, : 40]).float().cuda()).cpu()
    assert rel_l2(got, ref_wave) < 1e-4
    out_path = os.path.join(str(tmp_path), "out", "gen.wav")
    assert tts.generate("short one.", output_path=out_path, nfe_step=2) == out_path and os.path.getsize(out_path) > 44
    # duration predictor plug point (f5tts_wrapper.py:165-170, 381-406, 486-498): absent -> the flag is dropped with a warning ...
    assert not tts.has_duration_predictor
    from eraxvif5tts_amd.model import DurationPredictor
    torch.manual_seed(3)
    tts.model.duration_predictor = DurationPredictor(text_num_embeds=V, in_channels=16, filter_channels=24, kernel_size=3, p_dropout=0.1).eval().cuda()
    tts.has_duration_predictor = True
    # ... present: ref_audio_len + int(exp(log_duration) / speed), exactly the reference's arithmetic (one token: the only case its .item() accepts)
    tok, ln = torch.tensor([[5]], device="cuda"), torch.tensor([1], device="cuda")
    logd = cpu_ref.duration_predictor({k: v.detach().cpu() for k, v in tts.model.duration_predictor.state_dict().items()}, tok.cpu(), torch.ones(1, 1))
    assert tts.calculate_duration_with_predictor(tok, ln, local_speed=0.5) == tts.ref_audio_len + int(float(torch.exp(logd)) / 0.5)
    with pytest.raises((RuntimeError, ValueError)):  # the reference's reduction leaves [nt] values: .item() refuses more than one token
        tts.calculate_duration_with_predictor(torch.tensor([[5, 6, 7]], device="cuda"), torch.tensor([3], device="cuda"))


@pytest.mark.parametrize("prec,ragged", [("fp32", True), ("fp32", False), ("bf16", True)])
def test_generate_end_to_end_matches_the_oracle_chain(tmp_path, prec, ragged):
    """a1 as a whole (f5tts_wrapper.py:408-607): wav file -> preprocess_reference -> generate() with two text chunks (so the cross-fade fires) on the HIP
    path, against the CPU oracle chain cpu_ref.generate_chain = mel_spectrogram -> CFM.sample -> Vocos.decode -> rms rule -> cross-fade on the same
    wav / text / seed.  The noise is drawn as the reference's CPU path draws it (torch's CPU generator, one randn per chunk in chunk order:
    `noise_device = "cpu"`), so both sides start from the same y0 after the same torch.manual_seed.
    Stated tolerances: generated mel rel-L2 <= 2e-4 (fp32 mode) / 2e-2 (bf16 mode), as everywhere; waveform rel-L2 <= 2e-3 (fp32) / 1e-1 (bf16):
    Vocos exponentiates its head (exp, clip 1e2), which amplifies a mel deviation by about 5x at these sizes."""
    from eraxvif5tts_amd.infer import audio
    from eraxvif5tts_amd.infer.f5tts_wrapper import F5TTSWrapper
    from eraxvif5tts_amd.infer.utils_infer import chunk_text
    from eraxvif5tts_amd.model.utils import convert_char_to_pinyin, list_str_to_idx
    arch = dict(dim=128, depth=2, heads=2, ff_mult=2, text_dim=64, conv_layers=2, pe_attn_head=1, text_mask_padding=False)
    V = 32
    W = cpu_ref.random_dit_weights(arch, V, seed=25)
    hp = dict(dim=64, intermediate_dim=128, num_layers=2)
    VW = cpu_ref.random_vocos_weights(seed=26, dim=64, inter=128, layers=2)
    cfg_path, ckpt, vdir, vocab = _write_tiny_assets(str(tmp_path), arch, V, W, hp, VW)
    sr = 24000
    t = np.arange(int(2.0 * sr)) / sr
    wav = 0.03 * np.sin(2 * np.pi * 190 * t + 0.7) * (1 + 0.3 * np.sin(2 * np.pi * 5 * t)) + 0.01 * np.sin(2 * np.pi * 1370 * t)
    ref_wav = os.path.join(str(tmp_path), "ref.wav")
    audio.write_wav(ref_wav, wav, sr)
    # the stored prompt, computed independently: 16-bit PCM as read back, + 50 ms of silence (:313), boosted to rms 0.1 (:334-336)
    pcm = np.clip(np.round(wav * 32767.0), -32768, 32767).astype(np.float32) / np.float32(32768.0)
    prompt = torch.from_numpy(np.concatenate([pcm, np.zeros(int(0.05 * sr), np.float32)]))[None]
    prompt = prompt * 0.1 / torch.sqrt(torch.mean(torch.square(prompt)))

    tts = F5TTSWrapper(model_name=cfg_path, ckpt_path=ckpt, vocab_file=vocab, use_local_vocoder=True, vocoder_path=vdir, precision=prec)
    tts.model.noise_device = "cpu"
    if not ragged:
        tts.ragged_chunks = 0
    aud, ref_text = tts.preprocess_reference(ref_wav, "a quiet tone")
    assert aud.shape == prompt.shape and (aud.cpu() - prompt).abs().max() < 1e-6
    text = "hello there, this is a test. " * 3 + "and one more sentence to force a second chunk, because the budget is small."
    max_chars = int(len(ref_text.encode()) / (prompt.shape[-1] / sr) * (22 - prompt.shape[-1] / sr))
    chunks = chunk_text(text, max_chars=max_chars)
    assert len(chunks) == 2
    torch.manual_seed(1234)
    wave, rate, spec = tts.generate(text, nfe_step=4, return_numpy=True, return_spectrogram=True)

    vmap = tts.vocab_char_map
    oracle_chunks = [(list_str_to_idx(convert_char_to_pinyin([ref_text + c]), vmap), len(c.encode("utf-8"))) for c in chunks]
    torch.manual_seed(1234)
    ref_wave, ref_mels = cpu_ref.generate_chain(W, arch, VW, prompt, len(ref_text.encode("utf-8")), oracle_chunks, nfe_step=4)
    ref_spec = np.concatenate(ref_mels, axis=1)
    assert spec.shape == ref_spec.shape and wave.shape == ref_wave.shape and min(m.shape[1] for m in ref_mels) >= 256
    mel_err, wave_err = rel_l2(spec, ref_spec), rel_l2(wave, ref_wave)
    print(f"generate() vs oracle chain [{prec}, ragged={ragged}]: mel rel-L2 {mel_err:.2e}, wave rel-L2 {wave_err:.2e}")
    assert mel_err < {"fp32": 2e-4, "bf16": 2e-2}[prec]
    assert wave_err < {"fp32": 2e-3, "bf16": 1e-1}[prec]


def test_infer_process_batch_process_and_safetensors_checkpoint(tmp_path, monkeypatch):
    """utils_infer.load_model on a .safetensors EMA checkpoint (reference utils_infer.py:184-226, f5tts_wrapper.py:224-229),
    infer_process / infer_batch_process (reference utils_infer.py:366-563): chunk bookkeeping, the streaming generator, and the
    ORIGINAL-rms rule of infer_batch_process (:440-442,491-492), all over the HIP sampler + HIP vocoder."""
    from safetensors.torch import save_file
    from eraxvif5tts_amd.infer import audio, utils_infer as U
    from eraxvif5tts_amd.model import DiT
    arch = dict(dim=128, depth=2, heads=2, ff_mult=2, text_dim=64, conv_layers=2, pe_attn_head=1, text_mask_padding=False)
    V = 32
    W = cpu_ref.random_dit_weights(arch, V, seed=15)
    hp = dict(dim=64, intermediate_dim=128, num_layers=2)
    VW = cpu_ref.random_vocos_weights(seed=16, dim=64, inter=128, layers=2)
    cfg_path, ckpt_pt, vdir, vocab = _write_tiny_assets(str(tmp_path), arch, V, W, hp, VW)
    ema = {"ema_model.transformer." + k: v.contiguous() for k, v in W.items()}
    ema.update({"initted": torch.tensor(True), "step": torch.tensor(7)})
    ckpt_st = os.path.join(str(tmp_path), "model_7.safetensors")
    save_file(ema, ckpt_st)
    model = U.load_model(DiT, {**arch, "precision": "fp32"}, ckpt_st, vocab_file=vocab, device="cuda")
    model_pt = U.load_model(DiT, {**arch, "precision": "fp32"}, ckpt_pt, vocab_file=vocab, device="cuda")
    sd, sd_pt = model.state_dict(), model_pt.state_dict()
    assert set(sd) == set(sd_pt)
    for k in sd:
        assert torch.equal(sd[k], sd_pt[k]), k
        if k.startswith("transformer.") and k[len("transformer."):] in W:
            assert torch.equal(sd[k].cpu(), W[k[len("transformer."):]]), k
    vocoder = U.load_vocoder(is_local=True, local_path=vdir, device="cuda")

    sr = 24000
    t = np.arange(int(1.5 * sr)) / sr
    wav = 0.02 * np.sin(2 * np.pi * 200 * t) * (1 + 0.4 * np.sin(2 * np.pi * 2 * t))  # quieter than target_rms = 0.1
    ref_wav = os.path.join(str(tmp_path), "ref2.wav")
    audio.write_wav(ref_wav, wav, sr)
    ref_file, ref_text = U.preprocess_ref_audio_text(ref_wav, "a quiet tone")
    assert ref_text == "a quiet tone. "
    gen_text = "hello there, this is a test. " * 6 + "and one more sentence to force a second chunk, because the budget is small."
    a, rate = U._load_audio(ref_file)
    n_chunks = len(U.chunk_text(gen_text, max_chars=int(len(ref_text.encode()) / (a.shape[-1] / rate) * (22 - a.shape[-1] / rate))))
    assert n_chunks >= 2
    torch.manual_seed(5)
    wave, rate_out, spec = U.infer_process(ref_file, ref_text, gen_text, model, vocoder, nfe_step=4, device="cuda")
    assert rate_out == 24000 and np.isfinite(wave).all() and spec.shape[0] == 100
    assert len(wave) == (spec.shape[1] - n_chunks) * 256 - (n_chunks - 1) * 3600

    # streaming generator: the chunks of every text batch concatenate to the batch's whole wave (no cross-fade in streaming mode)
    batches = U.chunk_text(gen_text, max_chars=60)[:2]
    torch.manual_seed(6)
    chunks = list(U.infer_batch_process((a, rate), ref_text, batches, model, vocoder, nfe_step=2, device="cuda", streaming=True, chunk_size=1000))
    assert all(c[1] == 24000 and 0 < len(c[0]) <= 1000 for c in chunks)
    torch.manual_seed(6)
    whole, _, _ = next(U.infer_batch_process((a, rate), ref_text, batches, model, vocoder, nfe_step=2, device="cuda", cross_fade_duration=0))
    assert np.array_equal(np.concatenate([c[0] for c in chunks]), whole)
    # (the non-streaming call sampled its two batches as ONE ragged batch -- both have >= 256 frames; the serial order gives the same samples)
    monkeypatch.setenv("F5HIP_RAGGED_CHUNKS", "0")
    torch.manual_seed(6)
    serial, _, _ = next(U.infer_batch_process((a, rate), ref_text, batches, model, vocoder, nfe_step=2, device="cuda", cross_fade_duration=0))
    monkeypatch.delenv("F5HIP_RAGGED_CHUNKS")
    assert np.array_equal(serial, whole)

    # original-rms rule: a prompt quieter than the target is boosted for conditioning and the output scaled back by rms / target
    rms = float(torch.sqrt(torch.mean(torch.square(a))))
    assert rms < 0.1
    torch.manual_seed(7)
    quiet, _, _ = next(U.infer_batch_process((a, rate), ref_text, batches[:1], model, vocoder, nfe_step=2, device="cuda"))
    torch.manual_seed(7)
    loud, _, _ = next(U.infer_batch_process((a * 0.1 / rms, rate), ref_text, batches[:1], model, vocoder, nfe_step=2, device="cuda"))
    assert np.abs(quiet - loud * (rms / 0.1)).max() <= 2e-3 * np.abs(loud).max() + 1e-6


def test_streaming_wire_over_the_hip_wrapper(tmp_path):
    """SURVEY 8(f).1 on the real path: streaming.wire.stream_audio over the HIP F5TTSWrapper (HIP sampler + HIP vocoder) emits the 44-byte
    unknown-size WAV header and then, per text chunk, exactly the int16 PCM of generate()'s output for that chunk (reference
    src/streaming/f5tts-fastapi-server.py:173-204,246-250,270-421); the reference cache installs / clears the wrapper's state."""
    from eraxvif5tts_amd.infer import audio
    from eraxvif5tts_amd.infer.f5tts_wrapper import F5TTSWrapper
    from eraxvif5tts_amd.streaming.wire import ReferenceCache, create_wave_header, pcm16_bytes, stream_audio
    arch = dict(dim=128, depth=2, heads=2, ff_mult=2, text_dim=64, conv_layers=2, pe_attn_head=1, text_mask_padding=False)
    V = 32
    W = cpu_ref.random_dit_weights(arch, V, seed=25)
    hp = dict(dim=64, intermediate_dim=128, num_layers=2)
    VW = cpu_ref.random_vocos_weights(seed=26, dim=64, inter=128, layers=2)
    cfg_path, ckpt, vdir, vocab = _write_tiny_assets(str(tmp_path), arch, V, W, hp, VW)
    sr = 24000
    t = np.arange(int(1.6 * sr)) / sr
    ref_wav = os.path.join(str(tmp_path), "spk.wav")
    audio.write_wav(ref_wav, 0.05 * np.sin(2 * np.pi * 220 * t), sr)
    tts = F5TTSWrapper(model_name=cfg_path, ckpt_path=ckpt, vocab_file=vocab, use_local_vocoder=True, vocoder_path=vdir)
    cache = ReferenceCache()
    entry = cache.add(tts, "spk", ref_wav, text="a steady tone")
    assert entry["loaded"] is True and entry["processed_text"] == "a steady tone. " and tts.ref_audio_processed is None
    bad = cache.add(tts, "broken", os.path.join(str(tmp_path), "missing.wav"), text="x")
    assert bad["loaded"] is False and bad["error"]
    with pytest.raises(LookupError, match="not ready"):
        next(stream_audio(tts, cache, "broken", ["x"]))
    chunks = ["hello there.", "   ", "this is the second chunk..", "and a third one, a little longer than the others."]
    torch.manual_seed(11)
    parts = list(stream_audio(tts, cache, "spk", chunks, nfe_step=4))
    assert tts.ref_audio_processed is None and tts.ref_text is None  # cleared after the request (:419-421)
    assert parts[0] == create_wave_header(24000) and len(parts[0]) == 44 and len(parts) == 4  # the blank chunk yields nothing
    # the same requests one by one through generate(), same noise stream
    cache.install(tts, "spk")
    torch.manual_seed(11)
    for text, got in zip(("hello there.", "this is the second chunk.", "and a third one, a little longer than the others."), parts[1:]):
        wave, rate = tts.generate(text=text, return_numpy=True, nfe_step=4)
        assert rate == 24000 and got == pcm16_bytes(wave) and len(got) == 2 * len(wave) and np.isfinite(wave).all()
    cache.clear(tts)


def test_generate_long_form_chunks_and_cross_fade(tmp_path):
    """C4's chunk-and-cross-fade variant through generate() (reference infer/f5tts_wrapper.py:459-461,541-575): an 8 s prompt and a text long
    enough for several chunks of <= 22 s each, i.e. sequences of ~2000 frames per sample() call; the chunks are synthesised one by one
    and joined with 0.15 s (3600-sample) linear cross-fades.  Checked: chunk count = chunk_text's, every chunk's frame budget follows
    the byte-ratio duration rule, total length = sum of chunks - overlaps, the fade region is the linear blend of the two neighbours."""
    from eraxvif5tts_amd.infer import audio
    from eraxvif5tts_amd.infer.f5tts_wrapper import F5TTSWrapper
    from eraxvif5tts_amd.infer.utils_infer import chunk_text
    arch = dict(dim=128, depth=2, heads=2, ff_mult=2, text_dim=64, conv_layers=2, pe_attn_head=1, text_mask_padding=False)
    V = 32
    W = cpu_ref.random_dit_weights(arch, V, seed=35)
    hp = dict(dim=64, intermediate_dim=128, num_layers=2)
    VW = cpu_ref.random_vocos_weights(seed=36, dim=64, inter=128, layers=2)
    cfg_path, ckpt, vdir, vocab = _write_tiny_assets(str(tmp_path), arch, V, W, hp, VW)
    sr = 24000
    t = np.arange(8 * sr) / sr
    ref_wav = os.path.join(str(tmp_path), "long_ref.wav")
    audio.write_wav(ref_wav, 0.2 * np.sin(2 * np.pi * 150 * t) * (1 + 0.3 * np.sin(2 * np.pi * 2 * t)), sr)
    tts = F5TTSWrapper(model_name=cfg_path, ckpt_path=ckpt, vocab_file=vocab, use_local_vocoder=True, vocoder_path=vdir)
    ref_text = "this is the reference sentence that goes with the eight second prompt, spoken at an even pace."
    tts.preprocess_reference(ref_wav, ref_text, clip_short=False)
    sentence = "the quick brown fox jumps over the lazy dog, and then it does so once more. "
    text = sentence * 6
    secs = tts.get_current_audio_length()
    max_chars = int(len(tts.ref_text.encode()) / secs * (22 - secs))
    chunks = chunk_text(text, max_chars=max_chars)
    assert len(chunks) >= 3
    # per-chunk waves with cross-fade off, then the joined wave with the default 0.15 s fade (same noise: same seeds per call)
    torch.manual_seed(21)
    plain, rate, spec = tts.generate(text, nfe_step=2, cross_fade_duration=0.0, return_numpy=True, return_spectrogram=True)
    torch.manual_seed(21)
    faded, _ = tts.generate(text, nfe_step=2, return_numpy=True)
    frames = [tts.ref_audio_len + int(tts.ref_audio_len / len(tts.ref_text.encode()) * len(c.encode()) / 1.0) for c in chunks]
    assert max(frames) > 1500 and max(frames) <= 2100  # <= 22 s per sample() call
    lens = [(f - tts.ref_audio_len - 1) * 256 for f in frames]  # generated frames minus the overlap frame, vocoded
    assert rate == 24000 and len(plain) == sum(lens) and spec.shape[1] == sum(f - tts.ref_audio_len for f in frames)
    n = 3600
    assert len(faded) == sum(lens) - n * (len(chunks) - 1)
    first, second = plain[: lens[0]], plain[lens[0]: lens[0] + lens[1]]
    blend = first[-n:] * np.linspace(1, 0, n) + second[:n] * np.linspace(0, 1, n)
    assert np.allclose(faded[lens[0] - n: lens[0]], blend, atol=1e-6) and np.array_equal(faded[: lens[0] - n], first[:-n])
    assert np.isfinite(faded).all()
    # the chunks of a call are sampled as one ragged batch by default (every chunk here has >= 256 frames); one chunk per HIP stream, and one
    # after the other (the reference's order), give the same samples
    assert tts.ragged_chunks >= 2 and tts.chunk_streams > 1
    tts.ragged_chunks = 0
    torch.manual_seed(21)
    streamed, _ = tts.generate(text, nfe_step=2, return_numpy=True)
    assert np.array_equal(streamed, faded)
    tts.chunk_streams = 1
    torch.manual_seed(21)
    serial, _ = tts.generate(text, nfe_step=2, return_numpy=True)
    assert np.array_equal(serial, faded)


BIGVGAN_TINY = dict(num_mels=100, upsample_initial_channel=192, upsample_rates=[4, 4, 2, 2, 2, 2], upsample_kernel_sizes=[8, 8, 4, 4, 4, 4],
                    resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], snake_logscale=True,
                    use_tanh_at_final=False, use_bias_at_final=False)


@pytest.mark.parametrize("T,variant", [(13, "v2"), (40, "v2"), (77, "tanh_bias"), (5, "v2")])
def test_bigvgan_forward_matches_oracle(T, variant):
    """SURVEY 8(f).4, PARITY UNPINNED (the BigVGAN checkout is absent from the reference tree): csrc/bigvgan.hip against the oracle's restatement of the
    published BigVGAN-v2 generator (oracle/cpu_ref.bigvgan_forward: torch conv1d / conv_transpose1d, anti-aliased SnakeBeta with the Kaiser-sinc
    filters) on a narrow network of the v2 topology (6 stages x 3 AMP blocks, x256), random weights scaled so that the final clamp is not saturated;
    fp32-input MFMA: rel-L2 <= 1e-4.  Variant tanh_bias: use_tanh_at_final / use_bias_at_final (the v1 form)."""
    from eraxvif5tts_amd.bigvgan import BigVGAN
    hp = dict(BIGVGAN_TINY)
    if variant == "tanh_bias":
        hp.update(use_tanh_at_final=True, use_bias_at_final=True, snake_logscale=False)
    W = cpu_ref.random_bigvgan_weights(hp, seed=T)
    if not hp["snake_logscale"]:
        for k in W:
            if k.endswith(".alpha") or k.endswith(".beta"):
                W[k] = W[k].abs() + 0.5
    W["conv_post.weight"] = W["conv_post.weight"] * 0.0015
    mel = torch.randn(2, 100, T, generator=torch.Generator().manual_seed(T + 1)) * 2 - 3
    ref = cpu_ref.bigvgan_forward(W, hp, mel)
    assert ref.shape == (2, 1, T * 256) and float((ref.abs() >= 0.999).float().mean()) < 0.01  # (almost) never saturated: the comparison sees the whole network
    voc = BigVGAN(hp)
    voc.load_state_dict(W)
    voc = voc.eval().cuda()
    out = voc(mel.cuda()).cpu()
    assert out.shape == ref.shape and rel_l2(out, ref) < 1e-4
    assert voc.remove_weight_norm() is voc


def test_bigvgan_checkpoint_directory_and_weight_norm(tmp_path):
    """from_pretrained on a local directory in the layout of nvidia/bigvgan_v2_24khz_100band_256x (config.json + bigvgan_generator.pt with
    {"generator": ...}), weight-normed convolutions (weight_g / weight_v) and the Activation1d filter buffers folded at load time, through
    utils_infer.load_vocoder("bigvgan", is_local=True) as the reference's branch does (utils_infer.py:125-138)."""
    import json

    from eraxvif5tts_amd.infer import utils_infer as U
    hp = dict(BIGVGAN_TINY, resblock="1", activation="snakebeta")
    W = cpu_ref.random_bigvgan_weights(hp, seed=9)
    W["conv_post.weight"] = W["conv_post.weight"] * 0.0015
    ck = {}
    g = torch.Generator().manual_seed(10)
    for k, v in W.items():
        if k.endswith(".weight"):  # weight = g * v / ||v||: store a rescaled direction and the matching magnitude
            vv = v * (0.5 + torch.rand(v.shape[0], generator=g)).reshape(-1, *([1] * (v.ndim - 1)))
            ck[k[:-len("weight")] + "weight_v"] = vv
            ck[k[:-len("weight")] + "weight_g"] = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.ndim - 1)))
        else:
            ck[k] = v
    filt = cpu_ref.kaiser_sinc_filter1d(0.25, 0.3, 12).reshape(1, 1, 12)
    ck["activation_post.upsample.filter"] = filt
    ck["activation_post.downsample.lowpass.filter"] = filt
    ck["resblocks.0.activations.0.upsample.filter"] = filt
    ck["resblocks.0.activations.0.downsample.lowpass.filter"] = filt
    d = str(tmp_path)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(hp, f)
    torch.save({"generator": ck}, os.path.join(d, "bigvgan_generator.pt"))
    voc = U.load_vocoder("bigvgan", is_local=True, local_path=d, device="cuda")
    mel = torch.randn(1, 100, 21, generator=g) * 2 - 3
    assert rel_l2(voc(mel.cuda()).cpu(), cpu_ref.bigvgan_forward(W, hp, mel)) < 1e-4
    with pytest.raises(RuntimeError):
        U.load_vocoder("bigvgan", is_local=False)
