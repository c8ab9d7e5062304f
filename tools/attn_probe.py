#!/usr/bin/env python3
"""A/B the two flash-attention kernels (32 vs 64 queries per wave) at C2 / C4 sizes."""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from eraxvif5tts_amd import _lib
_lib.require_gpu()
lib = _lib.load()
for rnd in range(2):
    for v in (1, 2):
        _lib.check(lib.f5_tuning_set(b"attn_variant", v))
        for (B, N, H) in ((64, 1024, 16), (16, 4096, 16)):
            ms = C.c_float()
            _lib.check(lib.f5_bench_attention(1, B, N, H, 10, C.byref(ms), _lib.stream_ptr()))
            print(f"variant {v} B={B} N={N}: {ms.value*1e3:7.1f} us  {4.0*N*N*64*H*B/ms.value/1e9:6.1f} TFLOP/s")
