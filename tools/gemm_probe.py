#!/usr/bin/env python3
"""A/B the tuned GEMM main-loop variants on the four DiT call sites at C2 size (interleaved rounds, one process)."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from eraxvif5tts_amd import _lib
_lib.require_gpu()
lib = _lib.load()
rows, seq = 65536, 1024
sites = {0: ("qkv  N3072 K1024", 2.0 * rows * 3072 * 1024), 1: ("ff1  N2048 K1024", 2.0 * rows * 2048 * 1024),
         2: ("ff2  N1024 K2048", 2.0 * rows * 1024 * 2048), 3: ("outp N1024 K1024", 2.0 * rows * 1024 * 1024)}
import os
if os.environ.get("LEAN") is not None:
    _lib.check(lib.f5_tuning_set(b"gemm_lean", int(os.environ["LEAN"])))
if os.environ.get("BIG") is not None:
    _lib.check(lib.f5_tuning_set(b"gemm_big", int(os.environ["BIG"])))
variants = [int(v) for v in (sys.argv[1:] or ["0", "1"])]
res = {}
for rnd in range(3):
    for v in variants:
        _lib.check(lib.f5_tuning_set(b"gemm_variant", v)); globals().__setitem__("_v", v)
        for s, (name, fl) in sites.items():
            ms = C.c_float()
            _lib.check(lib.f5_bench_gemm_site(1, s, rows, seq, 1024, 16, 2048, 10, C.byref(ms), _lib.stream_ptr()))
            res.setdefault((v, s), []).append(fl / ms.value / 1e9)
for (v, s), tf in sorted(res.items()):
    print(f"variant {v} {sites[s][0]}: TFLOP/s rounds = {[round(x, 1) for x in tf]}  median {sorted(tf)[len(tf)//2]:.1f}")
for k in (0, 1):
    ms = C.c_float()
    rc = lib.f5_bench_attention(k, 64, 1024, 16, 5 if k else 1, C.byref(ms), _lib.stream_ptr())
    if rc == 0:
        print(f"attention kernel {k}: {ms.value:.3f} ms = {4.0 * 1024 * 1024 * 64 * 16 * 64 / ms.value / 1e9:.1f} TFLOP/s")
