#!/usr/bin/env python3
"""A/B the L2-patch tile order (gemm_group) of the tuned GEMM on the four DiT call sites at C2 size."""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from eraxvif5tts_amd import _lib
_lib.require_gpu()
lib = _lib.load()
rows, seq = 65536, 1024
sites = {0: ("qkv  N3072 K1024", 2.0 * rows * 3072 * 1024), 1: ("ff1  N2048 K1024", 2.0 * rows * 2048 * 1024),
         2: ("ff2  N1024 K2048", 2.0 * rows * 1024 * 2048), 3: ("outp N1024 K1024", 2.0 * rows * 1024 * 1024)}
res = {}
for rnd in range(2):
    for gm in (1, 4, 8, 16, 32):
        _lib.check(lib.f5_tuning_set(b"gemm_group", gm))
        for s, (name, fl) in sites.items():
            ms = C.c_float()
            _lib.check(lib.f5_bench_gemm_site(1, s, rows, seq, 1024, 16, 2048, 10, C.byref(ms), _lib.stream_ptr()))
            res.setdefault((gm, s), []).append(fl / ms.value / 1e9)
for (gm, s), tf in sorted(res.items()):
    print(f"group {gm:2d} {sites[s][0]}: TFLOP/s {[round(x, 1) for x in tf]}")
