"""Wire behaviour of the reference's streaming front-end (``src/streaming/f5tts-fastapi-server.py``), so the HIP wrapper can sit
under the shipped server / clients unchanged (a "next" row after the hot path).

* ``create_wave_header`` (:173-204): the 44-byte RIFF/WAVE header Python's ``wave`` module writes for mono 16-bit PCM; with
  ``data_size=0`` the size fields are 36 / 0 ("unknown size", players read until the stream ends).
* ``pcm16_bytes`` (``process_chunk`` :246-250): ``(audio * 32767).astype(np.int16)`` -- truncation towards zero, no clipping.
* ``ReferenceCache`` (:106-170, 300-340): processed prompt (waveform tensor, normalised text, length in frames) stored once per
  speaker id and installed into the wrapper's ``ref_*`` fields per request, cleared afterwards (:419-421).
* ``stream_audio`` (:270-421): header first, then the int16 PCM of each text chunk as soon as it is synthesised.
Text normalisation (vinorm) and the langchain splitter are third-party and stay with the caller: the generator takes chunks.
"""
from __future__ import annotations

import io
import wave
from typing import Iterable, Iterator, Optional

import numpy as np


def create_wave_header(sample_rate, num_channels=1, bits_per_sample=16, data_size=0) -> bytes:
    buf = io.BytesIO()
    with wave.open(buf, "wb") as wf:
        wf.setnchannels(num_channels)
        wf.setsampwidth(bits_per_sample // 8)
        wf.setframerate(sample_rate)
        if data_size > 0:
            wf.setnframes(data_size // (num_channels * (bits_per_sample // 8)))
        wf.writeframes(b"")
    return buf.getvalue()


def pcm16_bytes(audio: np.ndarray) -> bytes:
    return (np.asarray(audio) * 32767).astype(np.int16).tobytes()


class ReferenceCache:
    """speaker id -> processed reference, with the reference server's entry layout and status strings."""

    def __init__(self):
        self.entries = {}

    def add(self, model, ref_id, audio_path, text="", name=None, clip_short=False):
        try:
            _, processed_text = model.preprocess_reference(ref_audio_path=audio_path, ref_text=text.strip(), clip_short=clip_short)
            self.entries[ref_id] = {"ref_audio_path": audio_path, "ref_text_original": text, "loaded": True, "name": name or ref_id,
                                    "processed_mel": model.ref_audio_processed.clone().detach(), "processed_text": model.ref_text,
                                    "processed_mel_len": model.ref_audio_len, "error": None}
        except Exception as e:  # noqa: BLE001  (the server records the failure and keeps serving the other speakers)
            self.entries[ref_id] = {"loaded": False, "name": name or ref_id, "error": str(e)}
        finally:
            model.ref_audio_processed = model.ref_text = model.ref_audio_len = None
        return self.entries[ref_id]

    def install(self, model, ref_id):
        """Set the wrapper's reference state from the cache (no preprocessing); raises LookupError with the server's detail text."""
        e = self.entries.get(ref_id)
        if not e or e.get("loaded") is not True or any(k not in e for k in ("processed_mel", "processed_text", "processed_mel_len")):
            status = e.get("loaded", "Not Found") if e else "Not Found"
            detail = f"Reference speaker '{ref_id}' is not ready. Status: {status}."
            if e and e.get("error"):
                detail += f" Error during processing: {e['error']}"
            raise LookupError(detail)
        model.ref_audio_processed = e["processed_mel"].to(model.device) if hasattr(e["processed_mel"], "to") else e["processed_mel"]
        model.ref_text = e["processed_text"]
        model.ref_audio_len = e["processed_mel_len"]

    @staticmethod
    def clear(model):
        model.ref_audio_processed = model.ref_text = model.ref_audio_len = None


def process_chunk(chunk_text: str, model, **gen_kwargs) -> Optional[bytes]:
    chunk_text = chunk_text.strip()
    if not chunk_text:
        return None
    if chunk_text.endswith(".."):
        chunk_text = chunk_text[:-1].strip()
    if not chunk_text:
        return None
    audio, _sr = model.generate(text=chunk_text, return_numpy=True, **gen_kwargs)
    if audio is None or np.size(audio) == 0:
        return None
    return pcm16_bytes(audio)


def stream_audio(model, cache: ReferenceCache, speaker: str, text_chunks: Iterable[str], **gen_kwargs) -> Iterator[bytes]:
    """WAV header (unknown size) then one PCM block per synthesised chunk; the wrapper's reference state is always cleared."""
    cache.install(model, speaker)
    try:
        yield create_wave_header(sample_rate=model.target_sample_rate, data_size=0)
        for chunk in text_chunks:
            data = process_chunk(chunk, model, **gen_kwargs)
            if data:
                yield data
    finally:
        cache.clear(model)
