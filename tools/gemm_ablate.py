#!/usr/bin/env python3
"""Timing-only ablations of the tuned GEMM main loop on the FF2 / out-projection sites (variants 10-12 compute wrong results)."""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from eraxvif5tts_amd import _lib
_lib.require_gpu()
lib = _lib.load()
names = {1: "full", 10: "no DMA in loop", 11: "fragments read once", 12: "no MFMA", 13: "no MFMA, 128-B-row pieces", 0: "plain ring", 14: "plain ring, no MFMA", 20: "half-slab ring", 21: "half-slab ring, no MFMA"}
for rnd in range(2):
    for v in (1, 12, 20, 21):
        _lib.check(lib.f5_tuning_set(b"gemm_variant", v))
        out = []
        for site in (2, 3):
            ms = C.c_float()
            _lib.check(lib.f5_bench_gemm_site(1, site, 65536, 1024, 1024, 16, 2048, 10, C.byref(ms), _lib.stream_ptr()))
            out.append(ms.value * 1e3)
        per_stage = (out[0] - out[1]) / 4 / 32   # 4 rounds of blocks, 32 extra K-steps
        print(f"variant {v:2d} ({names[v]:22s}): FF2 {out[0]:7.1f} us  out-proj {out[1]:7.1f} us  -> {per_stage:.3f} us per 32-deep K-step, fixed {out[1]/4 - 32*per_stage:.1f} us per tile")
_lib.check(lib.f5_tuning_set(b"gemm_variant", 1))
