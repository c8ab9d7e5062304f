#!/usr/bin/env python3
"""Persistent GEMM grid: does starting every other workgroup late (de-synchronised store bursts) pay?  ticks are 10 ns."""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from eraxvif5tts_amd import _lib
_lib.require_gpu()
lib = _lib.load()
rows, seq = 65536, 1024
sites = {0: ("qkv  N3072 K1024", 2.0 * rows * 3072 * 1024), 1: ("ff1  N2048 K1024", 2.0 * rows * 2048 * 1024),
         2: ("ff2  N1024 K2048", 2.0 * rows * 1024 * 2048), 3: ("outp N1024 K1024", 2.0 * rows * 1024 * 1024)}
for skew in (0, 500, 1000, 1500, 2000, 3000, 0):
    _lib.check(lib.f5_tuning_set(b"gemm_skew", skew))
    line = []
    for s, (name, fl) in sites.items():
        ms = C.c_float()
        for _ in range(2):
            _lib.check(lib.f5_bench_gemm_site(1, s, rows, seq, 1024, 16, 2048, 10, C.byref(ms), _lib.stream_ptr()))
        line.append(f"{name.split()[0]} {ms.value * 1e3:6.1f} us ({fl / ms.value / 1e9:6.1f})")
    print(f"skew {skew / 100:5.1f} us: " + "  ".join(line), flush=True)
_lib.check(lib.f5_tuning_set(b"gemm_skew", 0))
