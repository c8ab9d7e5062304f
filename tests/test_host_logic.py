"""CPU tests of the host side: tokenizer / text front-end against vectors captured from the reference, chunk_text and the
duration / cross-fade rules against hand-derived cases, the C-ABI surface of libf5hip.so, and CFM's host logic (driven over
a torch backbone built from the CPU oracle) against the reference golden vectors."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_arch, golden_weights, load_golden, rel_l2


def test_tokenizer_and_ids_match_reference():
    from eraxvif5tts_amd.infer.utils_infer import DEFAULT_VOCAB
    from eraxvif5tts_amd.model.utils import get_tokenizer, lens_to_mask, list_str_to_idx
    z = load_golden("host_logic")
    vmap, vsize = get_tokenizer(DEFAULT_VOCAB, "custom")
    assert vsize == int(z["vocab_size"]) == 2545 and vmap[" "] == 0
    for tok, idx in zip(z["probe_tokens"], z["probe_ids"]):
        assert vmap.get(str(tok), -1) == int(idx)
    ids = list_str_to_idx([list(str(t)) for t in z["texts"]], vmap)
    assert np.array_equal(ids.numpy(), z["ids"])  # OOV -> 0, batch padding -1
    lens = torch.from_numpy(z["lens"])
    assert np.array_equal(lens_to_mask(lens).numpy(), z["mask"])
    assert np.array_equal(lens_to_mask(lens, length=9).numpy(), z["mask_len9"])
    with pytest.raises(FileNotFoundError):
        get_tokenizer("/nonexistent/vocab.txt", "custom")


def test_convert_char_to_pinyin_non_han_rules():
    from eraxvif5tts_amd.model.utils import convert_char_to_pinyin
    out = convert_char_to_pinyin(["hello world; “quoted” it’s"])[0]
    assert "".join(out) == 'hello world, "quoted" it\'s'  # ; -> , and curly -> straight quotes (utils.py:249-251)
    # a multi-letter ASCII run directly after a non-ASCII letter gets a space in front (utils.py:263-266)
    assert "".join(convert_char_to_pinyin(["trường"])[0]) == "trườ ng"
    assert convert_char_to_pinyin(["a b"])[0] == ["a", " ", "b"]


def test_chunk_text_byte_budget():
    from eraxvif5tts_amd.infer.utils_infer import chunk_text
    text = "Xin chào các bạn. Hôm nay trời đẹp, chúng ta đi chơi nhé! Được không? Tất nhiên rồi."
    chunks = chunk_text(text, max_chars=40)
    assert chunks == ["Xin chào các bạn.", "Hôm nay trời đẹp,", "chúng ta đi chơi nhé!", "Được không? Tất nhiên rồi."]
    assert all(len(c.encode("utf-8")) <= 40 + 1 for c in chunks)
    assert chunk_text("one. two. three.", max_chars=1000) == ["one. two. three."]
    assert chunk_text("", max_chars=10) == []
    assert chunk_text("你好。世界。", max_chars=6) == ["你好。", "世界。"]  # CJK punctuation splits without whitespace


def test_cross_fade_rule():
    from eraxvif5tts_amd.infer.utils_infer import cross_fade_concat
    a, b = np.ones(10000, np.float32), np.zeros(8000, np.float32)
    out = cross_fade_concat([a, b], 0.15)
    n = 3600  # 0.15 s * 24000
    assert len(out) == 10000 + 8000 - n
    assert np.allclose(out[10000 - n: 10000], np.linspace(1, 0, n))
    assert len(cross_fade_concat([a, b], 0.0)) == 18000
    short = np.ones(100, np.float32)
    assert len(cross_fade_concat([short, b], 0.15)) == 8000  # fade limited to the shorter chunk


def test_c_abi_exports_every_declared_symbol():
    from eraxvif5tts_amd import _lib
    header = open(os.path.join(ROOT, "include", "f5hip.h")).read()
    declared = sorted(set(re.findall(r"\b(f5_[a-z0-9_]+)\s*\(", header)))
    lib = _lib.load(build_if_missing=True)
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(_lib.EXPORTS) == declared
    assert lib.f5_version() == 400


def test_product_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from eraxvif5tts_amd import _lib
    from eraxvif5tts_amd.model import CFM, DiT
    m = DiT(dim=128, depth=1, heads=2, ff_mult=2, text_dim=64, conv_layers=1, text_num_embeds=10, mel_dim=100)
    c = CFM(transformer=m, mel_spec_kwargs={"mel_spec_type": "vocos"})
    with pytest.raises(_lib.F5HipError):
        c.sample(cond=torch.randn(1, 10, 100), text=torch.randint(0, 10, (1, 5)), duration=20, steps=2)
    with pytest.raises(_lib.F5HipError):
        m(x=torch.randn(1, 8, 100), cond=torch.randn(1, 8, 100), text=torch.zeros(1, 4, dtype=torch.long), time=torch.tensor(0.5),
          drop_audio_cond=False, drop_text=False)


class _OracleBackbone(torch.nn.Module):
    """plug point A accepts any module: a CPU torch backbone evaluated by the oracle lets CFM's host logic (duration rule,
    masks, noise, sway grid, CFG, solver loop, final where) run on CPU against the reference golden vectors."""

    def __init__(self, W, cfg):
        super().__init__()
        from oracle import cpu_ref
        self.W, self.cfg, self.ref, self.dim = W, cfg, cpu_ref, cfg["dim"]
        self.p = torch.nn.Parameter(torch.zeros(1))
        self.cleared = 0

    def forward(self, x, cond, text, time, drop_audio_cond, drop_text, mask=None, cache=False):
        return self.ref.dit_forward(self.W, self.cfg, x, cond, text, time, drop_audio_cond, drop_text, mask=mask)

    def clear_cache(self):
        self.cleared += 1


@pytest.mark.parametrize("name", ["tiny_base", "tiny_v1"])
def test_cfm_host_logic_against_reference_golden(name):
    from eraxvif5tts_amd.model import CFM
    z = load_golden(name)
    bb = _OracleBackbone(golden_weights(z), golden_arch(z))
    c = CFM(transformer=bb, mel_spec_kwargs={"mel_spec_type": "vocos"})
    out, traj = c.sample(cond=torch.from_numpy(z["cond"]), text=torch.from_numpy(z["text"]), duration=torch.from_numpy(z["duration"]),
                         lens=torch.from_numpy(z["lens"]), steps=int(z["steps"]), cfg_strength=float(z["cfg_strength"]),
                         sway_sampling_coef=float(z["sway"]), seed=int(z["seed"]))
    assert torch.equal(traj[0], torch.from_numpy(z["y0"]))  # per-sample manual_seed + randn, zero padded (cfm.py:178-183)
    assert rel_l2(out, z["out"]) < 2e-5 and rel_l2(traj, z["traj"]) < 2e-5
    assert bb.cleared == 1


def test_cfm_midpoint_and_duration_rule():
    from eraxvif5tts_amd.model import CFM
    z = load_golden("tiny_b1_midpoint")
    bb = _OracleBackbone(golden_weights(z), golden_arch(z))
    c = CFM(transformer=bb, mel_spec_kwargs={"mel_spec_type": "vocos"}, odeint_kwargs={"method": "midpoint"})
    out, traj = c.sample(cond=torch.from_numpy(z["cond"]), text=torch.from_numpy(z["text"]), duration=int(z["duration"]), steps=int(z["steps"]),
                         cfg_strength=0.0, sway_sampling_coef=float(z["sway"]), seed=int(z["seed"]))
    assert out.shape[1] == 41 and rel_l2(out, z["out_cfg0"]) < 2e-5


def test_front_end_has_no_host_path():
    """the package computes no path arithmetic in PyTorch: host-resident waveforms raise (the device kernels are checked against the oracle in
    tests/test_gpu_frontend.py)."""
    from eraxvif5tts_amd.infer import audio
    from eraxvif5tts_amd.model.modules import MelSpec
    wav = torch.randn(2, 6000, generator=torch.Generator().manual_seed(0)) * 0.1
    with pytest.raises(RuntimeError, match="no CPU path"):
        MelSpec()(wav)
    with pytest.raises(RuntimeError, match="no CPU path"):
        audio.resample(wav, 16000, 24000)
    assert audio.resample(wav, 24000, 24000) is wav


def test_audio_front_end():
    from eraxvif5tts_amd.infer import audio
    sr = 24000
    t = np.arange(int(2.0 * sr)) / sr
    tone = (0.3 * np.sin(2 * np.pi * 220 * t) * 32767).astype(np.int32)
    x = np.concatenate([np.zeros(int(0.3 * sr), np.int32), tone, np.zeros(int(0.4 * sr), np.int32)])
    seg = audio.Segment(x, sr, 2)
    assert len(seg) == 2700 and abs(seg.slice_ms(300, 2300).dBFS - 20 * np.log10(0.3 / np.sqrt(2))) < 0.05
    trimmed = audio.remove_silence_edges(seg)
    assert abs(len(trimmed) - 2000) <= 10
    assert audio.detect_leading_silence(seg, -42) == 300
    assert audio.detect_nonsilent(seg, min_silence_len=100, silence_thresh=-40, seek_step=10)[0][0] in range(290, 311)
    # wav round trip
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".wav") as f:
        audio.write_wav(f.name, trimmed.samples[:, 0] / 32768.0, sr)
        back, sr2, width = audio.read_wav(f.name)
        assert sr2 == sr and width == 2 and np.abs(back[:, 0] - trimmed.samples[:, 0]).max() <= 1


@pytest.mark.parametrize("orig,new", [(16000, 24000), (44100, 24000), (48000, 24000), (22050, 24000)])
def test_resample_oracle(orig, new):
    """oracle/cpu_ref.resample (float64 polyphase restatement of torchaudio's sinc_interp_hann, width 6, rolloff 0.99): a band-limited tone
    must come out as the same tone at the new rate (analytic truth).  (The HIP resampler is checked against it in tests/test_gpu_frontend.py.)"""
    from oracle import cpu_ref
    n = 9000
    t = torch.arange(n, dtype=torch.float64) / orig
    tone = (0.4 * torch.sin(2 * np.pi * 440 * t) + 0.2 * torch.sin(2 * np.pi * 3100 * t + 0.3)).float()
    noise = 0.1 * torch.randn(n, generator=torch.Generator().manual_seed(orig))
    wav = torch.stack([tone, noise])
    ref = cpu_ref.resample(wav, orig, new)
    g = np.gcd(orig, new)
    assert ref.shape == (2, int(np.ceil((new // g) * n / (orig // g)))) and ref.dtype == torch.float32
    tn = torch.arange(ref.shape[1], dtype=torch.float64) / new
    want = 0.4 * torch.sin(2 * np.pi * 440 * tn) + 0.2 * torch.sin(2 * np.pi * 3100 * tn + 0.3)
    assert (ref[0, 300:-300].double() - want[300:-300]).abs().max() < 4e-3  # pass-band ripple of the 6-zero-crossing Hann-windowed sinc
    assert cpu_ref.resample(wav, new, new) is wav


def test_streaming_wire_format():
    """44-byte header with unknown size, int16 PCM by truncation, reference-cache install/clear semantics."""
    import struct
    from eraxvif5tts_amd.streaming.wire import ReferenceCache, create_wave_header, pcm16_bytes, stream_audio
    h = create_wave_header(24000)
    assert len(h) == 44 and h[:4] == b"RIFF" and h[8:16] == b"WAVEfmt " and h[36:40] == b"data"
    assert struct.unpack("<I", h[4:8])[0] == 36 and struct.unpack("<I", h[40:44])[0] == 0
    assert struct.unpack("<HHIIHH", h[20:36]) == (1, 1, 24000, 48000, 2, 16)
    # the reference's data_size > 0 branch goes through the same wave.open/close: close() re-patches the sizes to the (zero)
    # bytes actually written, so the header is identical -- reproduced as is
    assert create_wave_header(24000, data_size=4800) == h
    assert pcm16_bytes(np.array([0.0, 0.5, -0.5, 0.99999, -1.0], np.float32)) == np.array([0, 16383, -16383, 32766, -32767], np.int16).tobytes()

    class FakeWrapper:
        device, target_sample_rate = "cpu", 24000
        ref_audio_processed = ref_text = ref_audio_len = None

        def preprocess_reference(self, ref_audio_path, ref_text, clip_short):
            if "bad" in ref_audio_path:
                raise FileNotFoundError(ref_audio_path)
            self.ref_audio_processed, self.ref_text, self.ref_audio_len = torch.ones(1, 2560), ref_text + ". ", 10
            return self.ref_audio_processed, self.ref_text

        def generate(self, text, return_numpy, **kw):
            assert self.ref_text == "hi. " and self.ref_audio_len == 10
            return np.full(len(text), 0.25, np.float32), 24000

    m, cache = FakeWrapper(), ReferenceCache()
    assert cache.add(m, "male", "ok.wav", "hi")["loaded"] is True and m.ref_text is None
    assert cache.add(m, "ghost", "bad.wav")["loaded"] is False
    with pytest.raises(LookupError, match="not ready"):
        next(stream_audio(m, cache, "ghost", ["x"]))
    parts = list(stream_audio(m, cache, "male", ["abc", "  ", "hello.."]))
    assert parts[0] == h and len(parts) == 3
    assert parts[1] == np.full(3, 8191, np.int16).tobytes() and len(parts[2]) == 2 * len("hello.")
    assert m.ref_text is None and m.ref_audio_processed is None  # state cleared after the request


def test_bench_refuses_a_rank_count_that_contradicts_gpus():
    """bench.py --gpus N: a launcher-provided WORLD_SIZE that differs from N is an error before any GPU work (the flag used to
    be parsed and ignored, so `--gpus 8` could silently measure one GPU)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "contradicts WORLD_SIZE=4" in r.stderr


def test_inference_prompt_buckets_follow_the_reference_rule():
    """eval/prompts.get_inference_prompt against a direct restatement of utils_eval.py:72-204 on a synthetic set: same duration rule, same bucket
    index, a batch emitted when a bucket has accumulated `infer_batch_size` FRAMES, residues at the end, seeded shuffle; padded_mel_batch."""
    import math
    import random

    from eraxvif5tts_amd.eval import prompts as P
    from oracle import cpu_ref
    meta = P.synthetic_metainfo(40, seed=3, min_secs=3.0, max_secs=20.0)
    got = P.get_inference_prompt(meta, tokenizer="char", infer_batch_size=3000, num_buckets=20, min_secs=3, max_secs=40, device="cpu",
                                 mel_spec_module=cpu_ref.mel_spectrogram)
    # restatement
    min_tok, max_tok, nb = 3 * 24000 // 256, 40 * 24000 // 256, 20
    acc, cur, want = [0] * nb, [[] for _ in range(nb)], []
    for utt, ptxt, (wav, sr), gtxt, _ in meta:
        if len(ptxt[-1].encode()) == 1:
            ptxt = ptxt + " "
        ref_len = wav.shape[-1] // 256
        total = ref_len + int(ref_len / len(ptxt.encode()) * len(gtxt.encode()) / 1.0)
        b = math.floor((total - min_tok) / (max_tok - min_tok + 1) * nb)
        cur[b].append((utt, ref_len, total, ptxt + gtxt))
        acc[b] += total
        if acc[b] >= 3000:
            want.append(cur[b])
            acc[b], cur[b] = 0, []
    want += [cur[b] for b in range(nb) if acc[b] > 0]
    random.seed(666)
    random.shuffle(want)
    assert len(got) == len(want) and sum(len(b[0]) for b in got) == 40
    for g, w in zip(got, want):
        utts, rms, mels, ref_lens, totals, texts = g
        assert utts == [x[0] for x in w] and ref_lens == [x[1] for x in w] and totals == [x[2] for x in w] and texts == [x[3] for x in w]
        assert mels.shape == (len(w), max(ref_lens) + 1, 100) and all(float(r) < 0.1 for r in rms)  # quiet prompts: boosted for conditioning
        # the rows past a prompt's own mel are zero padding
        for i, n in enumerate(ref_lens):
            assert float(mels[i, n + 1:].abs().max() if n + 1 < mels.shape[1] else 0.0) == 0.0
    kw = P.sample_kwargs(got[0], device="cpu", nfe_step=4, seed=7)
    assert kw["cond"].shape[0] == len(got[0][0]) and kw["duration"].tolist() == got[0][4] and kw["lens"].tolist() == got[0][3] and kw["seed"] == 7
