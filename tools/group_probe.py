#!/usr/bin/env python3
"""L2 patch height (token tiles per XCD patch) of the persistent GEMM on the four C2 call sites."""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from eraxvif5tts_amd import _lib
_lib.require_gpu()
lib = _lib.load()
rows, seq = 65536, 1024
sites = {0: ("qkv", 2.0 * rows * 3072 * 1024), 1: ("ff1", 2.0 * rows * 2048 * 1024), 2: ("ff2", 2.0 * rows * 1024 * 2048), 3: ("outp", 2.0 * rows * 1024 * 1024)}
for rnd in range(2):
    for gm in (1, 2, 4, 8, 16, 32):
        _lib.check(lib.f5_tuning_set(b"gemm_group", gm))
        line = []
        for s, (name, fl) in sites.items():
            ms = C.c_float()
            for _ in range(2):
                _lib.check(lib.f5_bench_gemm_site(1, s, rows, seq, 1024, 16, 2048, 10, C.byref(ms), _lib.stream_ptr()))
            line.append(f"{name} {fl / ms.value / 1e9:6.1f}")
        print(f"group {gm:2d}: " + "  ".join(line), flush=True)
_lib.check(lib.f5_tuning_set(b"gemm_group", 8))
