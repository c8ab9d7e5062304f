"""Batched-inference front of the reference's evaluation path: length-bucketed, frame-budgeted batches over the HIP sampler.

Reference: ``f5_tts/eval/utils_eval.py:58-204`` (``padded_mel_batch``, ``get_inference_prompt``) builds the batch list, and
``f5_tts/eval/eval_infer_batch.py:160-196`` consumes it -- split over the processes, ``CFM.sample`` per padded batch, then per utterance
``generated[ref_mel_len:total_mel_len]`` -> vocoder -> rms rule.  Same functions, argument names and tuple layout here, so the reference's
script logic reads unchanged; what differs is underneath:

  * the prompt waveforms are decoded by ``infer/audio.py`` (torchaudio is absent) and the mel / resampling run on the MI355X;
  * ``infer_prompts(..., mode="ragged")`` hands a bucket to ``CFM.sample_ragged`` -- one set of launches over the utterances concatenated
    along the token axis, a prompt per utterance, NO padding to the bucket's longest utterance and no key mask: every utterance gets the
    arithmetic of its own batch-1 ``sample()`` (bit-identical to it from 256 frames on).  ``mode="padded"`` is the reference's form
    (one padded batch, key-padding mask);
  * ``sample_kwargs`` turns a bucket into the keyword arguments ``eval/sharded.sample_sharded`` splits over the GPUs of a node.
"""
from __future__ import annotations

import math
import random
from typing import Iterable, List, Optional, Sequence

import torch
import torch.nn.functional as F

from ..infer import audio as _audio
from ..model.modules import MelSpec
from ..model.utils import convert_char_to_pinyin


def padded_mel_batch(ref_mels):
    """utils_eval.py:58-66: [mel, T_i] prompts -> [b, max T, mel], zero-padded."""
    max_mel_length = max(int(mel.shape[-1]) for mel in ref_mels)
    padded = [F.pad(mel, (0, max_mel_length - mel.shape[-1]), value=0) for mel in ref_mels]
    return torch.stack(padded).permute(0, 2, 1)


def _load_prompt_audio(prompt_wav, device):
    """(waveform [1, n] float32 on `device`, sample rate).  A path is decoded from RIFF/WAVE; a (tensor, rate) pair is taken as it is
    (synthetic prompts: bench.py, tests)."""
    if isinstance(prompt_wav, (tuple, list)):
        wav, sr = prompt_wav
        wav = torch.as_tensor(wav, dtype=torch.float32)
        if wav.ndim == 1:
            wav = wav[None]
        return wav.to(device), int(sr)
    seg = _audio.Segment.from_file(prompt_wav)
    return _audio.segment_to_float(seg).to(device), seg.frame_rate


def get_inference_prompt(metainfo, speed=1.0, tokenizer="pinyin", polyphone=True, target_sample_rate=24000, n_fft=1024, win_length=1024,
                         n_mel_channels=100, hop_length=256, mel_spec_type="vocos", target_rms=0.1, use_truth_duration=False, infer_batch_size=1,
                         num_buckets=200, min_secs=3, max_secs=40, device="cuda", mel_spec_module=None):
    """utils_eval.py:72-204.  metainfo: (utt, prompt_text, prompt_wav, gt_text, gt_wav) per utterance.  Utterances are dropped into
    `num_buckets` length buckets by their total mel length; a bucket is emitted as a batch as soon as it has accumulated `infer_batch_size`
    FRAMES (not utterances), the residue of every bucket at the end; the batch list is shuffled with a fixed seed.
    Returns [(utts, ref_rms_list, padded ref mels [b, max nc, mel], ref_mel_lens, total_mel_lens, final_text_list)].
    `device`: where the prompts are resampled and turned into mels (the HIP front-end); `mel_spec_module`: another callable wave -> mel
    (plug point as CFM's; the CPU tests pass the oracle's, the package itself has no host mel)."""
    prompts_all = []
    min_tokens = min_secs * target_sample_rate // hop_length
    max_tokens = max_secs * target_sample_rate // hop_length
    batch_accum = [0] * num_buckets
    utts, ref_rms_list, ref_mels, ref_mel_lens, total_mel_lens, final_text_list = ([[] for _ in range(num_buckets)] for _ in range(6))
    mel_spectrogram = mel_spec_module or MelSpec(n_fft=n_fft, hop_length=hop_length, win_length=win_length, n_mel_channels=n_mel_channels,
                                                 target_sample_rate=target_sample_rate, mel_spec_type=mel_spec_type)

    def emit(b):
        prompts_all.append((utts[b], ref_rms_list[b], padded_mel_batch(ref_mels[b]), ref_mel_lens[b], total_mel_lens[b], final_text_list[b]))

    for utt, prompt_text, prompt_wav, gt_text, gt_wav in metainfo:
        # audio
        ref_audio, ref_sr = _load_prompt_audio(prompt_wav, device)
        if ref_audio.shape[0] > 1:
            ref_audio = ref_audio.mean(dim=0, keepdim=True)
        ref_rms = torch.sqrt(torch.mean(torch.square(ref_audio)))
        if ref_rms < target_rms:
            ref_audio = ref_audio * target_rms / ref_rms
        assert ref_audio.shape[-1] > 5000, f"Empty prompt wav: {prompt_wav}"
        if ref_sr != target_sample_rate:
            ref_audio = _audio.resample(ref_audio, ref_sr, target_sample_rate)
        # text
        if len(prompt_text[-1].encode("utf-8")) == 1:
            prompt_text = prompt_text + " "
        text = [prompt_text + gt_text]
        text_list = convert_char_to_pinyin(text, polyphone=polyphone) if tokenizer == "pinyin" else text
        # duration, mel frame length
        ref_mel_len = ref_audio.shape[-1] // hop_length
        if use_truth_duration:
            gt_audio, gt_sr = _load_prompt_audio(gt_wav, device)
            if gt_sr != target_sample_rate:
                gt_audio = _audio.resample(gt_audio, gt_sr, target_sample_rate)
            total_mel_len = ref_mel_len + int(gt_audio.shape[-1] / hop_length / speed)
        else:
            ref_text_len = len(prompt_text.encode("utf-8"))
            gen_text_len = len(gt_text.encode("utf-8"))
            total_mel_len = ref_mel_len + int(ref_mel_len / ref_text_len * gen_text_len / speed)
        ref_mel = mel_spectrogram(ref_audio).squeeze(0)
        # bucket
        assert infer_batch_size > 0, "infer_batch_size should be greater than 0."
        assert min_tokens <= total_mel_len <= max_tokens, (
            f"Audio {utt} has duration {total_mel_len * hop_length // target_sample_rate}s out of range [{min_secs}, {max_secs}].")
        bucket_i = math.floor((total_mel_len - min_tokens) / (max_tokens - min_tokens + 1) * num_buckets)
        utts[bucket_i].append(utt)
        ref_rms_list[bucket_i].append(ref_rms)
        ref_mels[bucket_i].append(ref_mel)
        ref_mel_lens[bucket_i].append(ref_mel_len)
        total_mel_lens[bucket_i].append(total_mel_len)
        final_text_list[bucket_i].extend(text_list)
        batch_accum[bucket_i] += total_mel_len
        if batch_accum[bucket_i] >= infer_batch_size:
            emit(bucket_i)
            batch_accum[bucket_i] = 0
            utts[bucket_i], ref_rms_list[bucket_i], ref_mels[bucket_i] = [], [], []
            ref_mel_lens[bucket_i], total_mel_lens[bucket_i], final_text_list[bucket_i] = [], [], []
    for bucket_i, bucket_frames in enumerate(batch_accum):  # residue
        if bucket_frames > 0:
            emit(bucket_i)
    random.seed(666)  # "not only leave easy work for last workers" (utils_eval.py:201-203)
    random.shuffle(prompts_all)
    return prompts_all


def sample_kwargs(prompt, device="cuda", nfe_step=32, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=None, no_ref_audio=False, **extra):
    """One batch of `get_inference_prompt` -> the keyword arguments of ``CFM.sample`` as eval_infer_batch.py:163-183 passes them
    (``eval/sharded.sample_sharded`` takes a list of these)."""
    _, _, ref_mels, ref_mel_lens, total_mel_lens, final_text_list = prompt
    return dict(cond=ref_mels.to(device), text=final_text_list, duration=torch.tensor(total_mel_lens, dtype=torch.long, device=device),
                lens=torch.tensor(ref_mel_lens, dtype=torch.long, device=device), steps=nfe_step, cfg_strength=cfg_strength,
                sway_sampling_coef=sway_sampling_coef, no_ref_audio=no_ref_audio, seed=seed, **extra)


def ragged_sample_fn(cfm):
    """A ``sample_fn`` for ``sample_sharded`` that runs every batch as ONE ragged launch set (no padding, no key mask) and returns it in the
    padded [b, max N, mel] layout the gather slices -- rows past an utterance's own length are zero and never read."""
    def fn(cond, text, duration, lens, steps=32, cfg_strength=2.0, sway_sampling_coef=None, seed=None, no_ref_audio=False, max_duration=4096, **_):
        if no_ref_audio:
            cond = torch.zeros_like(cond)
        outs = cfm.sample_ragged(cond, text, [int(d) for d in duration], lens=lens, steps=steps, cfg_strength=cfg_strength,
                                 sway_sampling_coef=sway_sampling_coef, seed=seed, max_duration=max_duration)
        n = max(o.shape[1] for o in outs)
        return torch.cat([F.pad(o, (0, 0, 0, n - o.shape[1])) for o in outs]), None
    return fn


def ragged_ok(cfm, prompt, min_frames=256, max_rows=16384):
    """Can this bucket take the ragged sampler?  DiT backbone, every utterance long enough for the tuned kernels (from 256 frames on a
    batch-1 call runs them too, so ragged == batch-1 bit for bit), and the concatenation within one launch set's row budget."""
    tr = getattr(cfm, "transformer", None)
    total = prompt[4]
    return (hasattr(tr, "native_sample_ragged") and getattr(tr, "BACKBONE", None) == 0 and min(total) >= min_frames and max(total) <= 4096 and
            sum(-(-(int(t) + 16) // 16) * 16 for t in total) <= max_rows)


@torch.no_grad()
def infer_prompts(cfm, prompts_all: Sequence, vocoder=None, mode="ragged", nfe_step=32, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=None,
                  no_ref_audio=False, target_rms=0.1, device="cuda") -> Iterable:
    """eval_infer_batch.py:163-195 over the HIP path: for every batch of `prompts_all`, sample, then per utterance cut
    ``generated[ref_mel_len:total_mel_len]``, decode (when a vocoder is given) and apply the rms rule.  Yields
    ``(utt, mel [1, mel, frames] f32, wave [1, samples] | None)`` in batch order.
    mode "ragged": a bucket the ragged sampler can take runs unpadded (`ragged_ok`), the others padded; "padded": the reference's form."""
    for prompt in prompts_all:
        utts, ref_rms_list, _, ref_mel_lens, total_mel_lens, _ = prompt
        kw = sample_kwargs(prompt, device, nfe_step, cfg_strength, sway_sampling_coef, seed, no_ref_audio)
        if mode == "ragged" and ragged_ok(cfm, prompt):
            generated, _ = ragged_sample_fn(cfm)(**kw)
        else:
            generated, _ = cfm.sample(return_trajectory=False, **kw)
        for i, gen in enumerate(generated):
            gen = gen[ref_mel_lens[i]: total_mel_lens[i], :].unsqueeze(0)
            gen_mel_spec = gen.permute(0, 2, 1).to(torch.float32)
            wave = None
            if vocoder is not None:
                wave = vocoder.decode(gen_mel_spec) if hasattr(vocoder, "decode") else vocoder(gen_mel_spec).squeeze(0)  # eval_infer_batch.py:186-189
                if ref_rms_list[i] < target_rms:
                    wave = wave * ref_rms_list[i] / target_rms
            yield utts[i], gen_mel_spec, wave


def synthetic_metainfo(n_utts, seed=0, min_secs=3.0, max_secs=20.0, sample_rate=24000, prompt_secs=(2.0, 6.0), chars_per_sec=14.0) -> List[tuple]:
    """A synthetic evaluation set with a stated, realistic length distribution (there is no dataset offline): total utterance length uniform in
    [min_secs, max_secs], prompt length uniform in `prompt_secs`, text lengths that make the reference's byte-ratio duration rule
    (utils_eval.py:139-141) land on that total.  Prompts are band-limited noise bursts given as (tensor, rate) pairs."""
    g = torch.Generator().manual_seed(seed)
    letters = "abcdefghijklmnopqrstuvwxyz"
    meta = []
    for i in range(n_utts):
        total = float(torch.empty(1).uniform_(min_secs, max_secs, generator=g))
        ps = float(torch.empty(1).uniform_(*prompt_secs, generator=g))
        ps = min(ps, total * 0.5)
        n = int(ps * sample_rate)
        t = torch.arange(n) / sample_rate
        f0 = float(torch.empty(1).uniform_(90.0, 240.0, generator=g))
        wav = 0.08 * torch.sin(2 * math.pi * f0 * t) * (1 + 0.4 * torch.sin(2 * math.pi * 3.1 * t)) + 0.01 * torch.randn(n, generator=g)
        n_ref = max(8, int(ps * chars_per_sec))
        n_gen = max(8, int((total - ps) * chars_per_sec))

        def words(k):
            idx = torch.randint(0, 26, (k,), generator=g).tolist()
            s = "".join(letters[j] for j in idx)
            return " ".join(s[a:a + 5] for a in range(0, k, 6))[:k]
        meta.append((f"utt{i:04d}", words(n_ref) + ".", (wav, sample_rate), " " + words(n_gen) + ".", None))
    return meta
