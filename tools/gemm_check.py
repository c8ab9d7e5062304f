#!/usr/bin/env python3
"""Correctness of every tuned-GEMM main-loop variant against fp64 torch on a few shapes: python tools/gemm_check.py 1 2"""
import math, sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from eraxvif5tts_amd import _lib
import gpu_helpers as G
_lib.require_gpu()
lib = _lib.load()
bad = 0
import os
if os.environ.get("BIG") is not None:
    _lib.check(lib.f5_tuning_set(b"gemm_big", int(os.environ["BIG"])))
for v in [int(x) for x in sys.argv[1:]]:
    _lib.check(lib.f5_tuning_set(b"gemm_variant", v))
    for (M, N, K) in [(256, 256, 32), (512, 1024, 1024), (300, 3072, 128), (2048, 2048, 2048), (1000, 256, 96), (16384, 3072, 256), (16640, 2048, 64), (40960, 1024, 32), (40960, 1024, 1024), (41000, 2048, 96), (65536, 256, 64)]:
        g = torch.Generator().manual_seed(M + N + K)
        A, W, b = G.bf16_round(torch.randn(M, K, generator=g)), G.bf16_round(torch.randn(N, K, generator=g) / math.sqrt(K)), torch.randn(N, generator=g)
        ref = (A.double() @ W.double().t() + b.double()).float()
        out = G.op_linear(0, 1, A, W, b, "none")
        err = float((out - ref).norm() / ref.norm())
        print(f"variant {v} M={M} N={N} K={K}: rel-L2 {err:.2e}")
        bad += err > 2e-5
print("FAIL" if bad else "OK")
sys.exit(1 if bad else 0)
