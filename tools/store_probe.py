#!/usr/bin/env python3
"""Store flavours of the lean GEMM epilogue (0 default, 1 nt, 2 sc0 sc1, 3 sc0 sc1 nt) on the four C2 call sites."""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from eraxvif5tts_amd import _lib
_lib.require_gpu()
lib = _lib.load()
rows, seq = 65536, 1024
sites = {0: ("qkv", 2.0 * rows * 3072 * 1024), 1: ("ff1", 2.0 * rows * 2048 * 1024), 2: ("ff2", 2.0 * rows * 1024 * 2048), 3: ("outp", 2.0 * rows * 1024 * 1024)}
for rnd in range(2):
    for flav, knob in ((0, 0), (1, 63), (2, 2 << 8), (3, 3 << 8)):
        _lib.check(lib.f5_tuning_set(b"gemm_nt", knob))
        line = []
        for s, (name, fl) in sites.items():
            ms = C.c_float()
            for _ in range(2):
                _lib.check(lib.f5_bench_gemm_site(1, s, rows, seq, 1024, 16, 2048, 10, C.byref(ms), _lib.stream_ptr()))
            line.append(f"{name} {fl / ms.value / 1e9:6.1f}")
        print(f"store flavour {flav}: " + "  ".join(line), flush=True)
_lib.check(lib.f5_tuning_set(b"gemm_nt", 0))
