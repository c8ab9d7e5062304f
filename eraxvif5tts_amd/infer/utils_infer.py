"""Shared inference helpers: drop-in for the hot-path functions of ``f5_tts/infer/utils_infer.py``.

``chunk_text`` (:70-97), ``load_vocoder`` (:101-139), ``load_checkpoint`` (:184-226), ``load_model`` (:232-266),
``preprocess_ref_audio_text`` (:292-360, without the ASR branch), ``infer_process`` (:366-414) and ``infer_batch_process``
(:417-563) keep the reference's names, arguments, defaults and return values.  Out of scope here: the Whisper
ASR pipeline (``transcribe``: a by-name hub download), BigVGAN, the Gradio/CLI shells.
"""
from __future__ import annotations

import hashlib
import os
import re
import tempfile

import numpy as np
import torch

from ..model import CFM
from ..model.utils import convert_char_to_pinyin, get_tokenizer
from . import audio as _audio

device = "cuda" if torch.cuda.is_available() else "cpu"

# -----------------------------------------
target_sample_rate = 24000
n_mel_channels = 100
hop_length = 256
win_length = 1024
n_fft = 1024
mel_spec_type = "vocos"
target_rms = 0.1
cross_fade_duration = 0.15
ode_method = "euler"
nfe_step = 32  # 16, 32
cfg_strength = 2.0
sway_sampling_coef = -1.0
speed = 1.0
fix_duration = None
# -----------------------------------------

_ref_audio_cache = {}
DEFAULT_VOCAB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "examples", "vocab.txt")


def chunk_text(text, max_chars=135):
    """Split `text` into chunks of at most `max_chars` utf-8 BYTES at sentence punctuation (reference :70-97)."""
    chunks, current = [], ""
    sentences = re.split(r"(?<=[;:,.!?])\s+|(?<=[；：，。！？])", text)
    for sentence in sentences:
        piece = sentence + " " if sentence and len(sentence[-1].encode("utf-8")) == 1 else sentence
        if len(current.encode("utf-8")) + len(sentence.encode("utf-8")) <= max_chars:
            current += piece
        else:
            if current:
                chunks.append(current.strip())
            current = piece
    if current:
        chunks.append(current.strip())
    return chunks


def load_vocoder(vocoder_name="vocos", is_local=False, local_path="", device=device, hf_cache_dir=None):
    """Plug point B.  Only local weights can be loaded (there is no network): BigVGAN from ``{local_path}/config.json`` + ``bigvgan_generator.pt``; Vocos from ``{local_path}/config.yaml`` +
    ``{local_path}/pytorch_model.bin`` as the reference's is_local branch reads them (:104-107,113-124)."""
    if vocoder_name == "bigvgan":  # reference :125-138 (parity unpinned: the BigVGAN checkout is absent from the reference tree, see eraxvif5tts_amd/bigvgan.py)
        from ..bigvgan import BigVGAN
        if not is_local:
            raise RuntimeError("snapshot_download of nvidia/bigvgan_v2_24khz_100band_256x is not possible offline: pass is_local=True with a local directory "
                               "(config.json + bigvgan_generator.pt)")
        vocoder = BigVGAN.from_pretrained(local_path, use_cuda_kernel=False)
        vocoder.remove_weight_norm()
        return vocoder.eval().to(device)
    if vocoder_name != "vocos":
        raise NotImplementedError(f"vocoder {vocoder_name}: vocos and bigvgan are the reference's two")
    from ..vocos import Vocos
    if not is_local:
        raise RuntimeError("Download Vocos from huggingface charactr/vocos-mel-24khz is not possible offline: pass "
                           "use_local_vocoder=True / is_local=True with a local vocos-mel-24khz directory")
    print(f"Load vocos from local path {local_path}")
    config_path, model_path = f"{local_path}/config.yaml", f"{local_path}/pytorch_model.bin"
    vocoder = Vocos.from_hparams(config_path)
    state_dict = torch.load(model_path, map_location="cpu", weights_only=True)
    own = vocoder.state_dict()
    vocoder.load_state_dict({k: v for k, v in state_dict.items() if k in own}, strict=False)  # feature_extractor.* buffers are not used by decode()
    return vocoder.eval().to(device)


def load_checkpoint(model, ckpt_path, device: str, dtype=None, use_ema=True):
    """Reference :184-226.  The checkpoint dtype policy of the reference (fp16 on CUDA) does not apply: the HIP backbone keeps
    fp32 master weights and converts them once to its own bf16 kernel layouts."""
    ckpt_type = ckpt_path.split(".")[-1]
    if ckpt_type == "safetensors":
        from safetensors.torch import load_file
        checkpoint = load_file(ckpt_path, device="cpu")
    else:
        checkpoint = torch.load(ckpt_path, map_location="cpu", weights_only=True)
    if use_ema:
        if ckpt_type == "safetensors":
            checkpoint = {"ema_model_state_dict": checkpoint}
        checkpoint["model_state_dict"] = {k.replace("ema_model.", ""): v for k, v in checkpoint["ema_model_state_dict"].items()
                                          if k not in ["initted", "step"]}
        for key in ["mel_spec.mel_stft.mel_scale.fb", "mel_spec.mel_stft.spectrogram.window"]:  # backward compatibility, as the reference
            checkpoint["model_state_dict"].pop(key, None)
    elif ckpt_type == "safetensors":
        checkpoint = {"model_state_dict": checkpoint}
    model.load_state_dict(checkpoint["model_state_dict"], strict=False)
    del checkpoint
    return model.to(device)


def load_model(model_cls, model_cfg, ckpt_path, mel_spec_type=mel_spec_type, vocab_file="", ode_method=ode_method, use_ema=True,
               device=device):
    if vocab_file == "":
        vocab_file = DEFAULT_VOCAB
    tokenizer = "custom"
    print("\nvocab : ", vocab_file)
    print("token : ", tokenizer)
    print("model : ", ckpt_path, "\n")
    vocab_char_map, vocab_size = get_tokenizer(vocab_file, tokenizer)
    model = CFM(
        transformer=model_cls(**model_cfg, text_num_embeds=vocab_size, mel_dim=n_mel_channels),
        mel_spec_kwargs=dict(n_fft=n_fft, hop_length=hop_length, win_length=win_length, n_mel_channels=n_mel_channels,
                             target_sample_rate=target_sample_rate, mel_spec_type=mel_spec_type),
        odeint_kwargs=dict(method=ode_method),
        vocab_char_map=vocab_char_map,
    ).to(device)
    return load_checkpoint(model, ckpt_path, device, use_ema=use_ema)


def remove_silence_edges(audio, silence_threshold=-42):
    return _audio.remove_silence_edges(audio, silence_threshold)


def preprocess_ref_audio_text(ref_audio_orig, ref_text, clip_short=True, show_info=print):
    show_info("Converting audio...")
    aseg = _audio.Segment.from_file(ref_audio_orig)
    if clip_short:
        aseg = _audio.clip_reference(aseg, show_info)
    aseg = remove_silence_edges(aseg)
    aseg = aseg + aseg.silent_like(50)
    with tempfile.NamedTemporaryFile(delete=False, suffix=".wav") as f:
        ref_audio = f.name
    _audio.write_wav(ref_audio, _audio.segment_to_float(aseg).mean(dim=0).numpy(), aseg.frame_rate)
    with open(ref_audio, "rb") as fh:
        audio_hash = hashlib.md5(fh.read()).hexdigest()
    if not ref_text.strip():
        if audio_hash in _ref_audio_cache:
            show_info("Using cached reference text...")
            ref_text = _ref_audio_cache[audio_hash]
        else:
            raise RuntimeError("No reference text provided and the ASR model (openai/whisper-large-v3-turbo, a network download) "
                               "is not part of this build: pass ref_text")
    else:
        show_info("Using custom reference text...")
    if not ref_text.endswith(". ") and not ref_text.endswith("。"):
        ref_text += " " if ref_text.endswith(".") else ". "
    print("\nref_text  ", ref_text)
    return ref_audio, ref_text


def _load_audio(path):
    seg = _audio.Segment.from_file(path)
    return _audio.segment_to_float(seg), seg.frame_rate


def cross_fade_concat(waves, cross_fade_duration, sample_rate=target_sample_rate):
    """Linear cross-fade of consecutive chunks (reference :519-555 / f5tts_wrapper.py:541-575)."""
    if cross_fade_duration <= 0:
        return np.concatenate(waves)
    final = waves[0]
    for nxt in waves[1:]:
        n = min(int(cross_fade_duration * sample_rate), len(final), len(nxt))
        if n <= 0:
            final = np.concatenate([final, nxt])
            continue
        mixed = final[-n:] * np.linspace(1, 0, n) + nxt[:n] * np.linspace(0, 1, n)
        final = np.concatenate([final[:-n], mixed, nxt[n:]])
    return final


def infer_process(ref_audio, ref_text, gen_text, model_obj, vocoder, mel_spec_type=mel_spec_type, show_info=print, progress=None,
                  target_rms=target_rms, cross_fade_duration=cross_fade_duration, nfe_step=nfe_step, cfg_strength=cfg_strength,
                  sway_sampling_coef=sway_sampling_coef, speed=speed, fix_duration=fix_duration, device=device):
    audio, sr = _load_audio(ref_audio)
    max_chars = int(len(ref_text.encode("utf-8")) / (audio.shape[-1] / sr) * (22 - audio.shape[-1] / sr))
    gen_text_batches = chunk_text(gen_text, max_chars=max_chars)
    for i, t in enumerate(gen_text_batches):
        print(f"gen_text {i}", t)
    print("\n")
    show_info(f"Generating audio in {len(gen_text_batches)} batches...")
    return next(infer_batch_process((audio, sr), ref_text, gen_text_batches, model_obj, vocoder, mel_spec_type=mel_spec_type,
                                    progress=progress, target_rms=target_rms, cross_fade_duration=cross_fade_duration,
                                    nfe_step=nfe_step, cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef, speed=speed,
                                    fix_duration=fix_duration, device=device))


def infer_batch_process(ref_audio, ref_text, gen_text_batches, model_obj, vocoder, mel_spec_type="vocos", progress=None, target_rms=0.1,
                        cross_fade_duration=0.15, nfe_step=32, cfg_strength=2.0, sway_sampling_coef=-1, speed=1, fix_duration=None,
                        device=None, streaming=False, chunk_size=2048):
    audio, sr = ref_audio
    if audio.shape[0] > 1:
        audio = torch.mean(audio, dim=0, keepdim=True)
    rms = torch.sqrt(torch.mean(torch.square(audio)))  # the ORIGINAL (pre-boost) rms scales the output back (reference :440-442,491-492)
    if rms < target_rms:
        audio = audio * target_rms / rms
    audio = audio.to(device)
    if sr != target_sample_rate:
        audio = _audio.resample(audio, sr, target_sample_rate)  # on the device (f5_frontend_resample)
    if len(ref_text[-1].encode("utf-8")) == 1:
        ref_text = ref_text + " "

    ref_audio_len = audio.shape[-1] // hop_length

    def plan_batch(gen_text):  # host side of one text batch: tokens and the frame budget (reference :455-470)
        local_speed = 0.3 if len(gen_text.encode("utf-8")) < 10 else speed
        final_text_list = convert_char_to_pinyin([ref_text + gen_text])
        if fix_duration is not None:
            duration = int(fix_duration * target_sample_rate / hop_length)
        else:
            ref_text_len, gen_text_len = len(ref_text.encode("utf-8")), len(gen_text.encode("utf-8"))
            duration = ref_audio_len + int(ref_audio_len / ref_text_len * gen_text_len / local_speed)
        return final_text_list, duration

    def finish_batch(generated):  # mel -> wave (reference :481-497)
        generated = generated.to(torch.float32)[:, ref_audio_len:, :].permute(0, 2, 1)
        generated_wave = vocoder.decode(generated) if mel_spec_type == "vocos" else vocoder(generated)  # reference :485-488
        if rms < target_rms:
            generated_wave = generated_wave * rms / target_rms
        return generated_wave.squeeze().cpu().numpy(), generated[0].cpu().numpy()

    def process_batch(gen_text):
        final_text_list, duration = plan_batch(gen_text)
        with torch.inference_mode():
            generated, _ = model_obj.sample(cond=audio, text=final_text_list, duration=duration, steps=nfe_step, cfg_strength=cfg_strength,
                                            sway_sampling_coef=sway_sampling_coef, return_trajectory=False)
            generated_wave, mel = finish_batch(generated)
            if streaming:
                for j in range(0, len(generated_wave), chunk_size):
                    yield generated_wave[j: j + chunk_size], target_sample_rate
            else:
                yield generated_wave, mel

    batches = progress.tqdm(gen_text_batches) if progress is not None else gen_text_batches
    if streaming:
        for gen_text in batches:
            for chunk in process_batch(gen_text):
                yield chunk
        return
    generated_waves, spectrograms = [], []
    # Several text batches, each long enough for the tuned kernels: ONE ragged batch per group (F5TTSWrapper.generate does the same; every
    # utterance keeps the arithmetic -- and the noise draw order -- of its own batch-1 sample() call, so the audio is bit-identical).
    transformer = getattr(model_obj, "transformer", None)
    jobs = [plan_batch(t) for t in gen_text_batches]
    group_max = int(os.environ.get("F5HIP_RAGGED_CHUNKS", "8"))
    if (group_max >= 2 and len(jobs) >= 2 and hasattr(model_obj, "sample_ragged") and hasattr(transformer, "native_sample_ragged")
            and getattr(transformer, "BACKBONE", None) == 0 and min(d for _, d in jobs) >= 256 and max(d for _, d in jobs) <= 4096):
        with torch.inference_mode():
            i = 0
            while i < len(jobs):
                group, rows = [], 0
                while i < len(jobs) and len(group) < group_max and (not group or rows + jobs[i][1] <= 16384):
                    group.append(jobs[i])
                    rows += jobs[i][1]
                    i += 1
                for generated in model_obj.sample_ragged(audio, [j[0][0] for j in group], [j[1] for j in group], steps=nfe_step,
                                                         cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef):
                    wave, mel = finish_batch(generated)
                    generated_waves.append(wave)
                    spectrograms.append(mel)
        batches = []
    for gen_text in batches:  # the reference's thread pool resolves to serial generators on the caller thread
        wave, mel = next(process_batch(gen_text))
        generated_waves.append(wave)
        spectrograms.append(mel)
    if generated_waves:
        yield cross_fade_concat(generated_waves, cross_fade_duration), target_sample_rate, np.concatenate(spectrograms, axis=1)
    else:
        yield None, target_sample_rate, None
