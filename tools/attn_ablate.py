#!/usr/bin/env python3
"""Timing-only ablations of the flash-attention kernel at C2 size (64 x 16 heads of 1024 x 64)."""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from eraxvif5tts_amd import _lib
_lib.require_gpu()
lib = _lib.load()
names = {0: "full", 1: "no softmax math", 2: "no K/V tile refresh / barrier", 3: "no PV MFMAs", 4: "no QK^T MFMAs"}
for rnd in range(2):
    for v in names:
        _lib.check(lib.f5_tuning_set(b"attn_ablate", v))
        ms = C.c_float()
        _lib.check(lib.f5_bench_attention(1, 64, 1024, 16, 10, C.byref(ms), _lib.stream_ptr()))
        print(f"ablation {v} ({names[v]:30s}): {ms.value*1e3:7.1f} us  ({4.0*1024*1024*64*16*64/ms.value/1e9:6.1f} TF-equivalent)")
_lib.check(lib.f5_tuning_set(b"attn_ablate", 0))
