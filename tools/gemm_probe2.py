#!/usr/bin/env python3
"""A/B ring depth (4 / 5 slots) x leading-dimension padding of the tuned GEMM on the four DiT call sites at C2 size."""
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
    for var in (1, 20):
        for st in (4, 5):
            if var == 20 and st == 5:
                continue
            for pad in (0,):
                for k, v in ((b"gemm_variant", var), (b"gemm_stages", st), (b"bench_pad_a", pad), (b"bench_pad_w", pad)):
                    _lib.check(lib.f5_tuning_set(k, v))
                for s, (name, fl) in sites.items():
                    ms = C.c_float()
                    _lib.check(lib.f5_bench_gemm_site(1, s, rows, seq, 1024, 16, 2048, 10, C.byref(ms), _lib.stream_ptr()))
                    res.setdefault((var, st, pad, s), []).append(fl / ms.value / 1e9)
for (var, st, pad, s), tf in sorted(res.items()):
    print(f"variant {var} stages {st} pad {pad:3d} {sites[s][0]}: TFLOP/s {[round(x, 1) for x in tf]}")
