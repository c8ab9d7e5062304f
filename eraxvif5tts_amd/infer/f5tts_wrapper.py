"""F5TTSWrapper: preprocess one reference voice once, synthesize many texts -- the reference's inference facade
(``f5_tts/infer/f5tts_wrapper.py:28-621``) over the MI355X HIP backbone, sampler and vocoder.

Same constructor kwargs/defaults (:34-52), ``preprocess_reference(ref_audio_path, ref_text, clip_short)`` (:256-354),
``generate(text, output_path, nfe_step, cfg_strength, sway_sampling_coef, speed, fix_duration, cross_fade_duration,
use_duration_predictor, return_numpy, return_spectrogram)`` (:408-607), ``get_current_audio_length`` (:618-621), the same
stored fields, error types and the three return shapes.

Provenance of this file: the host-side control flow of ``generate()`` / ``preprocess_reference()`` is a condensed restatement of the
reference's, statement for statement where its behaviour is observable (local names, the duration rule, the progress prints, the rms and
cross-fade arithmetic) -- the north star asks for an identical API and identical host behaviour, and these lines have one spelling.  Nothing
the reference computes on the model path is reused: the backbone, sampler, vocoder, mel and resampling all run in libf5hip.

Differences forced by the offline, ROCm-only image (each raises instead of silently approximating):
  * no Whisper ASR is initialised (reference :100 downloads openai/whisper-large-v3-turbo): ``ref_text`` must be given;
  * checkpoints and the Vocos weights must be local files (reference :125 / utils_infer.py:110-112 fetch from the hub);
  * ``device`` must be a ROCm GPU: the backbone has no CPU path.
The optional conv duration predictor (:381-406) runs on the HIP path (``model/duration_predictor.py`` over ``f5_duration_predict``)
when the loaded model carries one; otherwise the reference's own fallback branch is taken (:166-168).
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch
import yaml

from ..model import CFM
from ..model import backbones as _backbones
from ..model.utils import convert_char_to_pinyin, get_tokenizer, list_str_to_idx
from . import audio as _audio
from .utils_infer import DEFAULT_VOCAB, chunk_text, cross_fade_concat, load_checkpoint, load_vocoder

_CONFIG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs")


class F5TTSWrapper:
    """A wrapper class for F5-TTS that preprocesses reference audio once and allows for repeated TTS generation."""

    def __init__(self, model_name: str = "F5TTS_v1_Base", ckpt_path: Optional[str] = None, vocab_file: Optional[str] = None,
                 vocoder_name: str = "vocos", use_local_vocoder: bool = False, vocoder_path: Optional[str] = None,
                 device: Optional[str] = None, hf_cache_dir: Optional[str] = None, target_sample_rate: int = 24000,
                 n_mel_channels: int = 100, hop_length: int = 256, win_length: int = 1024, n_fft: int = 1024,
                 ode_method: str = "euler", use_ema: bool = True, use_duration_predictor: bool = False,
                 vocoder=None, precision: Optional[str] = None):
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self.device = device
        self.target_sample_rate, self.n_mel_channels = target_sample_rate, n_mel_channels
        self.hop_length, self.win_length, self.n_fft = hop_length, win_length, n_fft
        self.mel_spec_type = vocoder_name
        self.ode_method = ode_method
        self.use_duration_predictor = use_duration_predictor

        # model configuration: a bundled config name, or (reference :128-131) a YAML path when "custom" is in the name
        config_path = os.path.join(_CONFIG_DIR, f"{model_name}.yaml") if "custom" not in model_name.lower() else model_name
        with open(config_path, "r") as f:
            model_cfg = yaml.safe_load(f)
        model_cls = getattr(_backbones, model_cfg["model"]["backbone"], None)  # plug point A (reference :134)
        if model_cls is None:
            raise NotImplementedError(f"backbone {model_cfg['model']['backbone']} is not on the MI355X path (DiT, UNetT and MMDiT are)")
        model_arc = dict(model_cfg["model"]["arch"])
        if precision is not None:
            model_arc["precision"] = precision

        if vocab_file is None:
            vocab_file = DEFAULT_VOCAB
        self.vocab_char_map, vocab_size = get_tokenizer(vocab_file, "custom")

        self.model = CFM(
            transformer=model_cls(**model_arc, text_num_embeds=vocab_size, mel_dim=n_mel_channels),
            mel_spec_kwargs=dict(n_fft=n_fft, hop_length=hop_length, win_length=win_length, n_mel_channels=n_mel_channels,
                                 target_sample_rate=target_sample_rate, mel_spec_type=vocoder_name),
            odeint_kwargs=dict(method=ode_method),
            vocab_char_map=self.vocab_char_map,
        ).to(self.device)

        if ckpt_path is None:
            raise FileNotFoundError("ckpt_path is required: the reference's default (hf://SWivid/F5-TTS/...) is a network download")
        self._load_checkpoint(self.model, ckpt_path, use_ema=use_ema)

        self.has_duration_predictor = hasattr(self.model, "duration_predictor") and self.model.duration_predictor is not None
        if self.use_duration_predictor and not self.has_duration_predictor:
            print("Warning: Duration predictor requested but not found in model. Using fallback duration calculation.")
            self.use_duration_predictor = False
        elif self.has_duration_predictor:
            print("Duration predictor found in model.")

        if vocoder is not None:  # plug point B: any object with .decode(mel[b, 100, T])
            self.vocoder = vocoder
        else:
            if vocoder_path is None:
                vocoder_path = "../checkpoints/vocos-mel-24khz"
            self.vocoder = load_vocoder(vocoder_name=vocoder_name, is_local=use_local_vocoder, local_path=vocoder_path,
                                        device=self.device, hf_cache_dir=hf_cache_dir)

        self.ref_audio_processed = None
        self.ref_text = None
        self.ref_audio_len = None
        self.target_rms = 0.1
        self.cross_fade_duration = 0.15
        # text chunks of one generate() call sampled concurrently (HIP streams); 1 = one after the other as the reference does.  Not a
        # constructor argument (the reference's signature is kept); set the attribute or F5HIP_CHUNK_STREAMS
        self.chunk_streams = int(os.environ.get("F5HIP_CHUNK_STREAMS", "4"))
        # text chunks of one generate() call sampled as ONE ragged batch (libf5hip f5_sample_ragged): up to this many utterances per launch set
        # (0 = off: chunks run one per stream as above).  Applies when every chunk has at least 256 frames (the row count from which a batch-1
        # call takes the tuned kernels too, so both paths compute bit-identical mels).
        self.ragged_chunks = int(os.environ.get("F5HIP_RAGGED_CHUNKS", "8"))
        # "cpu": draw every chunk's initial noise from torch's CPU generator, in chunk order -- the numbers the reference's CPU path draws after
        # the same torch.manual_seed (reference model/cfm.py:178-183 with device = cpu); None = on the GPU, as the reference's GPU path does
        self.model.noise_device = os.environ.get("F5HIP_NOISE_DEVICE") or None
        self.nfe_step = 32
        self.cfg_strength = 2.0
        self.sway_sampling_coef = -1.0
        self.speed = 1.0
        self.fix_duration = None

    def _load_checkpoint(self, model, ckpt_path, dtype=None, use_ema=True):
        return load_checkpoint(model, ckpt_path, self.device, dtype=dtype, use_ema=use_ema)

    # ------------------------------------------------------------------ reference voice
    def preprocess_reference(self, ref_audio_path: str, ref_text: str = "", clip_short: bool = True):
        print("Converting audio...")
        aseg = _audio.Segment.from_file(ref_audio_path)
        if clip_short:
            aseg = _audio.clip_reference(aseg)
        aseg = self._remove_silence_edges(aseg)
        aseg = aseg + aseg.silent_like(50)
        if not ref_text.strip():
            raise RuntimeError("No reference text provided: automatic transcription (Whisper) needs a network model download; pass ref_text")
        print("Using custom reference text...")
        if not ref_text.endswith(". ") and not ref_text.endswith("。"):
            ref_text += " " if ref_text.endswith(".") else ". "
        print("\nReference text:", ref_text)

        audio, sr = _audio.segment_to_float(aseg), aseg.frame_rate
        if audio.shape[0] > 1:
            audio = torch.mean(audio, dim=0, keepdim=True)
        rms = torch.sqrt(torch.mean(torch.square(audio)))
        if rms < self.target_rms:  # boosted only when quieter than the target (reference :334-336)
            audio = audio * self.target_rms / rms
        audio = audio.to(self.device)
        if sr != self.target_sample_rate:
            audio = _audio.resample(audio, sr, self.target_sample_rate)  # on the device (f5_frontend_resample)
        self.ref_audio_processed = audio
        self.ref_text = ref_text
        self.ref_audio_len = audio.shape[-1] // self.hop_length
        return audio, ref_text

    def _remove_silence_edges(self, audio, silence_threshold=-42):
        return _audio.remove_silence_edges(audio, silence_threshold)

    def calculate_duration_with_predictor(self, text_tokens, text_lengths, local_speed=1.0):
        """Reference :381-406, op for op: token mask from the lengths, ``model.duration_predictor(tokens, mask)`` -> [b, 1, nt]
        log-durations, ``exp(...).squeeze(-1).sum(dim=1)`` and ``durations[0].item()``.  (As in the reference the reduction runs
        over the singleton channel axis, so ``.item()`` only succeeds for a one-token text; the quirk is kept, not repaired.)"""
        b, nt = text_tokens.shape
        range_tensor = torch.arange(nt, device=self.device).unsqueeze(0)
        text_tokens_mask = (range_tensor < text_lengths.unsqueeze(1)).int()
        with torch.inference_mode():
            log_durations = self.model.duration_predictor(text_tokens, text_tokens_mask)
            durations = torch.exp(log_durations).squeeze(-1).sum(dim=1)
        return self.ref_audio_len + int(durations[0].item() / local_speed)

    # ------------------------------------------------------------------ synthesis
    def generate(self, text: str, output_path: Optional[str] = None, nfe_step: Optional[int] = None, cfg_strength: Optional[float] = None,
                 sway_sampling_coef: Optional[float] = None, speed: Optional[float] = None, fix_duration: Optional[float] = None,
                 cross_fade_duration: Optional[float] = None, use_duration_predictor: Optional[bool] = None,
                 return_numpy: bool = False, return_spectrogram: bool = False):
        if self.ref_audio_processed is None or self.ref_text is None:
            raise ValueError("Reference audio not preprocessed. Call preprocess_reference() first.")
        nfe_step = nfe_step if nfe_step is not None else self.nfe_step
        cfg_strength = cfg_strength if cfg_strength is not None else self.cfg_strength
        sway_sampling_coef = sway_sampling_coef if sway_sampling_coef is not None else self.sway_sampling_coef
        speed = speed if speed is not None else self.speed
        fix_duration = fix_duration if fix_duration is not None else self.fix_duration
        cross_fade_duration = cross_fade_duration if cross_fade_duration is not None else self.cross_fade_duration
        use_predictor = use_duration_predictor if use_duration_predictor is not None else self.use_duration_predictor
        can_use_predictor = use_predictor and self.has_duration_predictor

        audio_len = self.ref_audio_processed.shape[-1] / self.target_sample_rate
        max_chars = int(len(self.ref_text.encode("utf-8")) / audio_len * (22 - audio_len))
        text_batches = chunk_text(text, max_chars=max_chars)
        for i, text_batch in enumerate(text_batches):
            print(f"Text batch {i}: {text_batch}")
        print("\n")

        generated_waves, spectrograms = [], []
        # The text chunks of one call are independent (the reference samples them one after the other, :476-533).  Here up to `chunk_streams`
        # of them are in flight at once, each on a HIP stream (and a libf5hip plan) of its own: a single-utterance sample() fills a fraction
        # of the 256 CUs, so the streams overlap.  Every chunk runs exactly the launches of the serial path -- same shapes, same kernels --
        # so its mel is bit-identical to the serial result; noise is drawn in chunk order either way.
        transformer = getattr(self.model, "transformer", None)
        jobs = []  # (token list, frames asked for) per chunk -- host work of the reference's loop (:476-510), before any sampling
        for i, text_batch in enumerate(text_batches):
            local_speed = 0.3 if len(text_batch.encode("utf-8")) < 10 else speed
            final_text_list = convert_char_to_pinyin([self.ref_text + text_batch])
            if fix_duration is not None:
                duration = int(fix_duration * self.target_sample_rate / self.hop_length)
                print(f"Using fixed duration: {fix_duration}s ({duration} frames)")
            elif can_use_predictor:
                if isinstance(final_text_list[0], str):
                    text_tokens = list_str_to_idx(final_text_list, self.vocab_char_map).to(self.device)
                else:
                    text_tokens = torch.tensor(final_text_list, device=self.device)
                text_lengths = torch.tensor([len(t) for t in final_text_list], device=self.device)
                duration = self.calculate_duration_with_predictor(text_tokens, text_lengths, local_speed)
                print(f"Duration predictor output: {duration} frames")
            else:
                ref_text_len, gen_text_len = len(self.ref_text.encode("utf-8")), len(text_batch.encode("utf-8"))
                duration = self.ref_audio_len + int(self.ref_audio_len / ref_text_len * gen_text_len / local_speed)
                print(f"Calculated duration based on text ratio: {duration} frames")
            jobs.append((final_text_list, int(duration)))

        # Several chunks, each long enough for the tuned kernels: ONE ragged batch per group of chunks -- the utterances concatenated along the
        # token axis, no padding to a common length, every chunk with the arithmetic of its own batch-1 call (bit-identical mels, same noise
        # order).  A single-utterance sample() fills a fraction of the 256 CUs; the concatenation fills them.
        ragged = (int(self.ragged_chunks) >= 2 and len(jobs) >= 2 and torch.cuda.is_available() and hasattr(transformer, "native_sample_ragged")
                  and getattr(transformer, "BACKBONE", None) == 0 and min(d for _, d in jobs) >= 256 and max(d for _, d in jobs) <= 4096)
        n_streams = 1
        if not ragged and torch.cuda.is_available() and hasattr(transformer, "finish_pending"):
            n_streams = max(1, min(len(text_batches), int(self.chunk_streams)))
        main = torch.cuda.current_stream() if n_streams > 1 else None
        streams = self._chunk_stream_pool(n_streams) if n_streams > 1 else []
        for st in streams:
            st.wait_stream(main)  # the preprocessed prompt was produced on the caller's stream
        mels = []
        if ragged:
            group, rows = [], 0
            groups = [group]
            for job in jobs:
                if group and (len(group) >= int(self.ragged_chunks) or rows + job[1] > 16384):
                    group, rows = [], 0
                    groups.append(group)
                group.append(job)
                rows += job[1]
            with torch.inference_mode():
                for group in groups:
                    mels += self.model.sample_ragged(self.ref_audio_processed, [j[0][0] for j in group], [j[1] for j in group], steps=nfe_step,
                                                     cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef)
        for i, (final_text_list, duration) in enumerate(jobs if not ragged else []):
            with torch.inference_mode():
                if n_streams > 1:
                    with torch.cuda.stream(streams[i % n_streams]):
                        generated, _ = self.model.sample(cond=self.ref_audio_processed, text=final_text_list, duration=duration, steps=nfe_step,
                                                         cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef, return_trajectory=False,
                                                         defer_guard=True)
                else:
                    generated, _ = self.model.sample(cond=self.ref_audio_processed, text=final_text_list, duration=duration, steps=nfe_step,
                                                     cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef, return_trajectory=False)
                mels.append(generated)
        if n_streams > 1:
            transformer.finish_pending()  # reads every chunk's fp16 range-guard flag (and lets the library redo a chunk whose guard fired)
            # finish_pending() synchronises a stream only when its plan has a guard to read (bf16 mode with fp16 storage); the vocoder below runs
            # on the caller's stream, so that stream must be ordered behind EVERY chunk stream whatever the precision (round 4: the fp32 mode
            # decoded mels that were still being written -- found by test_generate_end_to_end_matches_the_oracle_chain[fp32-False])
            for st in streams:
                main.wait_stream(st)
        with torch.inference_mode():
            for generated in mels:
                if n_streams > 1:
                    generated.record_stream(main)
                generated = generated.to(torch.float32)[:, self.ref_audio_len:, :].permute(0, 2, 1)
                if self.mel_spec_type == "vocos":
                    generated_wave = self.vocoder.decode(generated)
                elif self.mel_spec_type == "bigvgan":
                    generated_wave = self.vocoder(generated)
                rms = torch.sqrt(torch.mean(torch.square(self.ref_audio_processed)))  # of the stored, already boosted prompt (:529-531)
                if rms < self.target_rms:
                    generated_wave = generated_wave * rms / self.target_rms
                generated_waves.append(generated_wave.squeeze().cpu().numpy())
                if return_spectrogram or output_path is not None:
                    spectrograms.append(generated.squeeze().cpu().numpy())

        if not generated_waves:
            raise RuntimeError("No audio generated")
        final_wave = cross_fade_concat(generated_waves, cross_fade_duration, self.target_sample_rate)
        combined_spectrogram = np.concatenate(spectrograms, axis=1) if spectrograms else None
        if output_path is not None:
            output_dir = os.path.dirname(output_path)
            if output_dir and not os.path.exists(output_dir):
                os.makedirs(output_dir)
            _audio.write_wav(output_path, final_wave, self.target_sample_rate)
            if return_spectrogram:
                np.save(os.path.splitext(output_path)[0] + "_spec.npy", combined_spectrogram)  # matplotlib is not in the image
            if not return_numpy:
                return output_path
        if return_spectrogram:
            return final_wave, self.target_sample_rate, combined_spectrogram
        return final_wave, self.target_sample_rate

    def _chunk_stream_pool(self, n):
        pool = getattr(self, "_chunk_streams_pool", None) or []
        while len(pool) < n:
            pool.append(torch.cuda.Stream())
        self._chunk_streams_pool = pool
        return pool[:n]

    def get_current_audio_length(self):
        if self.ref_audio_processed is None:
            return 0
        return self.ref_audio_processed.shape[-1] / self.target_sample_rate
