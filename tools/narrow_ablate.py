#!/usr/bin/env python3
"""Timing ablations of the narrow-tile (plain ring) GEMM loop at single-utterance size (2 048 token rows): what is a K-step made of?"""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from eraxvif5tts_amd import _lib
_lib.require_gpu()
lib = _lib.load()
rows, seq = 2048, 1024
names = {0: "full", 1: "no MFMA", 2: "no DMA refill", 4: "fragments read once", 8: "no barrier", 3: "no MFMA, no DMA", 5: "no MFMA, no reads", 6: "no DMA, no reads",
         7: "barrier + waits only", 15: "loop skeleton only"}
for site, (label, nk) in {3: ("outp N1024 K1024 (128 tiles 256x64)", 32), 2: ("ff2 N1024 K2048 (128 tiles)", 64), 1: ("ff1 N2048 K1024 (256 tiles)", 32), 0: ("qkv N3072 K1024 (192 tiles 256x128)", 32)}.items():
    for a, nm in names.items():
        _lib.check(lib.f5_tuning_set(b"gemm_fast_ablate", a))
        ms = C.c_float()
        for _ in range(2):
            _lib.check(lib.f5_bench_gemm_site(1, site, rows, seq, 1024, 16, 2048, 20, C.byref(ms), _lib.stream_ptr()))
        print(f"{label:38s} {nm:24s}: {ms.value * 1e3:6.1f} us = {ms.value * 1e3 / nk:5.3f} us/K-step", flush=True)
_lib.check(lib.f5_tuning_set(b"gemm_fast_ablate", 0))
