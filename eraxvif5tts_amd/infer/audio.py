"""Reference-audio front-end without pydub / torchaudio / ffmpeg (none of them exists in the target image).

Restates, on integer PCM sample arrays, the pieces of those third-party packages the reference calls from
``F5TTSWrapper.preprocess_reference`` (f5tts_wrapper.py:256-379) and ``preprocess_ref_audio_text`` (utils_infer.py:292-360):
pydub's millisecond slicing / dBFS / ``split_on_silence`` / ``detect_leading_silence`` and torchaudio's
``transforms.Resample`` (windowed-sinc, Hann, lowpass_filter_width 6, rolloff 0.99).  These are "parity unpinned"
restatements (the packages are absent from the reference tree); only RIFF/WAVE input is decoded here.
"""
from __future__ import annotations

import math
import struct

import numpy as np
import torch


# ----------------------------------------------------------------------------- WAV I/O
def read_wav(path):
    """-> (int samples [n, channels] as int32, sample_rate, sample_width_bytes, float_flag)"""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: only RIFF/WAVE files can be decoded without ffmpeg")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
            if fmt[0] == 0xFFFE and len(body) >= 26:  # WAVE_FORMAT_EXTENSIBLE: real tag in the sub-format GUID
                fmt = (struct.unpack("<H", body[24:26])[0],) + fmt[1:]
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError(f"{path}: missing fmt/data chunk")
    tag, ch, sr, _, _, bits = fmt
    if tag == 3:  # IEEE float
        arr = np.frombuffer(pcm, dtype="<f4" if bits == 32 else "<f8").astype(np.float64)
        ints = np.clip(np.round(arr * 32767.0), -32768, 32767).astype(np.int32)  # pydub converts float wavs to 16-bit PCM via ffmpeg
        width = 2
    elif tag == 1:
        width = bits // 8
        if width == 1:
            ints = np.frombuffer(pcm, dtype=np.uint8).astype(np.int32) - 128
        elif width == 2:
            ints = np.frombuffer(pcm, dtype="<i2").astype(np.int32)
        elif width == 3:
            b = np.frombuffer(pcm[: len(pcm) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            ints = (b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16))
            ints = np.where(ints >= 1 << 23, ints - (1 << 24), ints)
        elif width == 4:
            ints = np.frombuffer(pcm, dtype="<i4").astype(np.int64).astype(np.int32)
        else:
            raise ValueError(f"unsupported PCM width {bits}")
    else:
        raise ValueError(f"unsupported WAV format tag {tag}")
    n = len(ints) // ch
    return ints[: n * ch].reshape(n, ch), sr, width


def write_wav(path, wave, sample_rate):
    """float waveform [n] or [1, n] in [-1, 1] -> 16-bit PCM WAV (what torchaudio.save writes for float input by default is
    32-bit float; 16-bit keeps the files small and is what the streaming front-ends of the reference send)."""
    x = np.asarray(wave, dtype=np.float64).reshape(-1)
    pcm = np.clip(np.round(x * 32767.0), -32768, 32767).astype("<i2").tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, sample_rate, sample_rate * 2, 2, 16))
        f.write(b"data" + struct.pack("<I", len(pcm)) + pcm)


# ----------------------------------------------------------------------------- pydub-like segment on int samples
class Segment:
    """Minimal stand-in for pydub.AudioSegment: int samples [n, ch], millisecond indexing, rms / dBFS."""

    def __init__(self, samples, frame_rate, sample_width):
        a = np.asarray(samples, dtype=np.int32)
        self.samples = a if a.ndim == 2 else a.reshape(len(a), 1)
        self.frame_rate, self.sample_width = frame_rate, sample_width

    @classmethod
    def from_file(cls, path):
        s, sr, w = read_wav(path)
        return cls(s, sr, w)

    def __len__(self):  # milliseconds, as pydub rounds
        return int(round(1000.0 * self.samples.shape[0] / self.frame_rate))

    @property
    def duration_seconds(self):
        return self.samples.shape[0] / self.frame_rate

    def _frame(self, ms):
        return int(ms * (self.frame_rate / 1000.0))

    def slice_ms(self, start, end):
        n = len(self)
        start = 0 if start is None else (start if start >= 0 else n + start)
        end = n if end is None else (end if end >= 0 else n + end)
        start, end = min(max(start, 0), n), min(max(end, 0), n)
        return Segment(self.samples[self._frame(start): self._frame(end)], self.frame_rate, self.sample_width)

    def __add__(self, other):
        return Segment(np.concatenate([self.samples, other.samples]) if other.samples.size else self.samples, self.frame_rate, self.sample_width)

    def silent_like(self, duration_ms):
        return Segment(np.zeros((self._frame(duration_ms), self.samples.shape[1]), np.int32), self.frame_rate, self.sample_width)

    @property
    def max_possible_amplitude(self):
        return float(1 << (8 * self.sample_width - 1))

    @property
    def rms(self):  # audioop.rms: integer sqrt of the mean square over all interleaved samples
        if self.samples.size == 0:
            return 0
        return int(math.sqrt(float(np.mean(self.samples.astype(np.float64) ** 2))))

    @property
    def dBFS(self):
        r = self.rms
        return -float("inf") if r == 0 else 20.0 * math.log10(r / self.max_possible_amplitude)


def detect_silence(seg, min_silence_len=1000, silence_thresh=-16, seek_step=1):
    n = len(seg)
    if n < min_silence_len:
        return []
    thresh = (10 ** (silence_thresh / 20.0)) * seg.max_possible_amplitude
    last = n - min_silence_len
    starts = list(range(0, last + 1, seek_step))
    if last % seek_step:
        starts.append(last)
    silent = [i for i in starts if seg.slice_ms(i, i + min_silence_len).rms <= thresh]
    if not silent:
        return []
    ranges, prev, cur = [], silent[0], silent[0]
    for i in silent[1:]:
        if not (i == prev + seek_step) and i > prev + min_silence_len:
            ranges.append([cur, prev + min_silence_len])
            cur = i
        prev = i
    ranges.append([cur, prev + min_silence_len])
    return ranges


def detect_nonsilent(seg, min_silence_len=1000, silence_thresh=-16, seek_step=1):
    silent = detect_silence(seg, min_silence_len, silence_thresh, seek_step)
    n = len(seg)
    if not silent:
        return [[0, n]]
    if silent[0][0] == 0 and silent[0][1] == n:
        return []
    out, prev_end = [], 0
    for s, e in silent:
        out.append([prev_end, s])
        prev_end = e
    if prev_end != n:
        out.append([prev_end, n])
    if out and out[0] == [0, 0]:
        out.pop(0)
    return out


def split_on_silence(seg, min_silence_len=1000, silence_thresh=-16, keep_silence=100, seek_step=1):
    ranges = [[s - keep_silence, e + keep_silence] for s, e in detect_nonsilent(seg, min_silence_len, silence_thresh, seek_step)]
    for a, b in zip(ranges, ranges[1:]):
        if b[0] < a[1]:
            a[1] = (a[1] + b[0]) // 2
            b[0] = a[1]
    n = len(seg)
    return [seg.slice_ms(max(s, 0), min(e, n)) for s, e in ranges]


def detect_leading_silence(seg, silence_threshold=-50.0, chunk_size=10):
    trim, n = 0, len(seg)
    while trim < n and seg.slice_ms(trim, trim + chunk_size).dBFS < silence_threshold:
        trim += chunk_size
    return min(trim, n)


def remove_silence_edges(seg, silence_threshold=-42):
    """f5tts_wrapper.py:356-379 / utils_infer.py:269-287."""
    seg = seg.slice_ms(detect_leading_silence(seg, silence_threshold), None)
    end = seg.duration_seconds
    for ms in range(len(seg) - 1, -1, -1):
        if seg.slice_ms(ms, ms + 1).dBFS > silence_threshold:
            break
        end -= 0.001
    return seg.slice_ms(0, int(end * 1000))


def clip_reference(seg, show_info=print):
    """The three clipping attempts of f5tts_wrapper.py:272-301 (<= 12 s, cut on long then short silences)."""
    def gather(min_len, thresh):
        out = seg.silent_like(0)
        for part in split_on_silence(seg, min_silence_len=min_len, silence_thresh=thresh, keep_silence=1000, seek_step=10):
            if len(out) > 6000 and len(out + part) > 12000:
                return out, True
            out = out + part
        return out, False

    wave, clipped = gather(1000, -50)
    if clipped:
        show_info("Audio is over 12s, clipping short. (1)")
    if len(wave) > 12000:
        wave, clipped = gather(100, -40)
        if clipped:
            show_info("Audio is over 12s, clipping short. (2)")
    if len(wave) > 12000:
        wave = wave.slice_ms(0, 12000)
        show_info("Audio is over 12s, clipping short. (3)")
    return wave


def segment_to_float(seg):
    """What torchaudio.load gives for the exported 16-bit wav: float32 [channels, n] = int / 2^(bits-1)."""
    x = seg.samples.astype(np.float32) / np.float32(seg.max_possible_amplitude)
    return torch.from_numpy(np.ascontiguousarray(x.T))


# ----------------------------------------------------------------------------- torchaudio.transforms.Resample (sinc_interp_hann)
def resample(waveform, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    if orig_freq == new_freq:
        return waveform
    if waveform.is_cuda and lowpass_filter_width == 6 and rolloff == 0.99:  # device path: csrc/frontend.hip (same kernel table, fp32 FIR)
        from ..frontend import resample as _hip_resample
        return _hip_resample(waveform, orig_freq, new_freq)
    raise RuntimeError("resample: the waveform must be on the MI355X (libf5hip f5_frontend_resample, torchaudio's sinc_interp_hann with width 6 / "
                       "rolloff 0.99); this package has no CPU path (the float64 restatement lives in oracle/cpu_ref.resample)")
