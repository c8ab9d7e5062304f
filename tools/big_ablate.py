#!/usr/bin/env python3
"""Timing ablations of gemm_big.hip on the attention-out site (K=1024) and FF2 (K=2048): which resource bounds the K-step?"""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from eraxvif5tts_amd import _lib
_lib.require_gpu()
lib = _lib.load()
rows, seq = 65536, 1024
names = {0: "full", 8: "no epilogue", 9: "no epilogue, no MFMA", 10: "no epilogue, no DMA", 12: "no epilogue, no ds_read", 14: "no epi/DMA/ds_read (MFMA only)",
         11: "no epi/MFMA/DMA (ds_read only)", 13: "no epi/MFMA/ds_read (DMA only)", 15: "sync skeleton only"}
for site, (label, fl, nk) in {3: ("outp K=1024", 2.0 * rows * 1024 * 1024, 32), 2: ("ff2 K=2048", 2.0 * rows * 1024 * 2048, 64)}.items():
    for a, nm in names.items():
        _lib.check(lib.f5_tuning_set(b"gemm_big_ablate", a))
        ms = C.c_float()
        for _ in range(2):
            _lib.check(lib.f5_bench_gemm_site(1, site, rows, seq, 1024, 16, 2048, 10, C.byref(ms), _lib.stream_ptr()))
        tile_us = ms.value * 1e3 / 4.0  # 1024 tiles on 256 CUs = 4 rounds
        print(f"{label} {nm:36s}: {ms.value * 1e3:7.1f} us  {fl / ms.value / 1e9:7.1f} TFLOP/s-equiv  tile {tile_us:5.1f} us  = {tile_us / nk:5.3f} us/K-step")
_lib.check(lib.f5_tuning_set(b"gemm_big_ablate", 0))
