#!/usr/bin/env python3
"""Does the leading dimension (power-of-two row stride -> L2 channel aliasing) limit the LDS-DMA rate?"""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from eraxvif5tts_amd import _lib
_lib.require_gpu()
lib = _lib.load()
for v in (1, 12):
    _lib.check(lib.f5_tuning_set(b"gemm_variant", v))
    for pa, pw in ((0, 0), (64, 0), (0, 64), (64, 64), (32, 32), (128, 128), (8, 8)):
        _lib.check(lib.f5_tuning_set(b"bench_pad_a", pa)); _lib.check(lib.f5_tuning_set(b"bench_pad_w", pw))
        out = []
        for site in (2, 3):
            ms = C.c_float()
            _lib.check(lib.f5_bench_gemm_site(1, site, 65536, 1024, 1024, 16, 2048, 10, C.byref(ms), _lib.stream_ptr()))
            out.append(ms.value * 1e3)
        print(f"variant {v:2d} pad_a {pa:3d} pad_w {pw:3d}: FF2 {out[0]:7.1f} us ({2*65536*1024*2048/out[0]/1e6:7.1f} TF)  out-proj {out[1]:7.1f} us ({2*65536*1024*1024/out[1]/1e6:7.1f} TF)")
