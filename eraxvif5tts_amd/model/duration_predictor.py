"""Duration predictor: drop-in for ``f5_tts/model/duration_predictor.py``.

Same constructor, parameter names (``text_embed``, ``conv_1``, ``norm_1``, ``conv_2``, ``norm_2``, ``proj``, optional ``cond``) and
``forward`` / ``phoneme_forward`` signatures as the reference (:5-26, :28-46, :48-68), so a checkpoint's ``duration_predictor.*`` tensors
load unchanged; the arithmetic runs in libf5hip (``f5_duration_predict``, csrc/duration.hip).  There is no eager fallback.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from .. import _lib


class DurationPredictor(nn.Module):
    def __init__(self, text_num_embeds, in_channels, filter_channels, kernel_size, p_dropout, gin_channels=0):
        super().__init__()
        text_dim = in_channels
        self.text_embed = nn.Embedding(text_num_embeds + 1, text_dim)  # use 0 as filler token
        self.in_channels, self.filter_channels, self.kernel_size = in_channels, filter_channels, kernel_size
        self.p_dropout, self.gin_channels = p_dropout, gin_channels
        self.drop = nn.Dropout(p_dropout)  # identity at inference; kept for state_dict / attribute compatibility
        self.conv_1 = nn.Conv1d(in_channels, filter_channels, kernel_size, padding=kernel_size // 2)
        self.norm_1 = nn.GroupNorm(1, filter_channels)
        self.conv_2 = nn.Conv1d(filter_channels, filter_channels, kernel_size, padding=kernel_size // 2)
        self.norm_2 = nn.GroupNorm(1, filter_channels)
        self.proj = nn.Conv1d(filter_channels, 1, 1)
        if gin_channels != 0:
            self.cond = nn.Conv1d(gin_channels, in_channels, 1)

    def _run(self, x, x_mask, g, add_one):
        if g is not None and self.gin_channels == 0:
            raise AttributeError("DurationPredictor: `g` given but the net was built with gin_channels = 0 (no `cond` layer, reference :25-26)")
        if self.training and self.p_dropout > 0:
            raise RuntimeError("DurationPredictor: inference only (call .eval(); Dropout is the identity there)")
        _lib.require_gpu()
        lib = _lib.load()
        dev = self.text_embed.weight.device
        if dev.type != "cuda":
            raise _lib.F5HipError("DurationPredictor must live on the GPU (no CPU path)")
        b, nt = x.shape
        tok = x.to(device=dev, dtype=torch.int32).contiguous()
        msk = x_mask.to(device=dev, dtype=torch.int32).contiguous()
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        tensors = [f32(self.text_embed.weight), f32(self.conv_1.weight), f32(self.conv_1.bias), f32(self.norm_1.weight), f32(self.norm_1.bias),
                   f32(self.conv_2.weight), f32(self.conv_2.bias), f32(self.norm_2.weight), f32(self.norm_2.bias),
                   f32(self.proj.weight).reshape(-1), f32(self.proj.bias)]
        gdev, g_nt = None, 0
        if g is not None:  # x = x + self.cond(g)  (reference :33-35); g [b, gin, 1] or [b, gin, nt]
            gdev = g.detach().to(device=dev, dtype=torch.float32).contiguous()
            if gdev.ndim != 3 or gdev.shape[0] != b or gdev.shape[1] != self.gin_channels or gdev.shape[2] not in (1, nt):
                raise ValueError(f"g must be [batch, gin_channels, 1 | nt], got {tuple(gdev.shape)}")
            g_nt = int(gdev.shape[2])
            tensors += [f32(self.cond.weight).reshape(self.in_channels, self.gin_channels), f32(self.cond.bias)]
        ptrs = [C.c_void_p(t.data_ptr()) for t in tensors]
        w = _lib.DurationWeights(*ptrs[:11], self.text_embed.num_embeddings, self.in_channels, self.filter_channels, self.kernel_size,
                                 ptrs[11] if g is not None else None, ptrs[12] if g is not None else None, self.gin_channels if g is not None else 0)
        scratch = torch.empty(2 * b * self.filter_channels * nt + b * self.in_channels * max(g_nt, 1), device=dev, dtype=torch.float32)
        out = torch.empty(b, nt, device=dev, dtype=torch.float32)
        _lib.check(lib.f5_duration_predict_g(C.byref(w), b, nt, _lib.ptr(tok), add_one, _lib.ptr(msk), _lib.ptr(gdev), g_nt, _lib.ptr(scratch),
                                             _lib.ptr(out), _lib.stream_ptr()), "f5_duration_predict")
        return out.unsqueeze(1)  # [b, 1, nt] like the reference

    def forward(self, x, x_mask, g=None):
        """x: token ids [b, nt] (batch pad -1), x_mask [b, nt] -> log-durations [b, 1, nt] (reference :28-46)."""
        return self._run(x, x_mask, g, 1)

    def phoneme_forward(self, phoneme_indices, phoneme_mask, g=None):
        """Same network on ids that are already embedding rows (no +1 shift, reference :48-68)."""
        return self._run(phoneme_indices, phoneme_mask, g, 0)
