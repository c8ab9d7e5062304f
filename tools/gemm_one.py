#!/usr/bin/env python3
"""Launch ONE tuned-GEMM call site a few times (for rocprofv3 --pmc passes): python tools/gemm_one.py <site> <variant> [iters]"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from eraxvif5tts_amd import _lib
_lib.require_gpu()
lib = _lib.load()
site, variant = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
_lib.check(lib.f5_tuning_set(b"gemm_variant", variant))
ms = C.c_float()
_lib.check(lib.f5_bench_gemm_site(1, site, 65536, 1024, 1024, 16, 2048, iters, C.byref(ms), _lib.stream_ptr()))
print(f"site {site} variant {variant}: {ms.value:.4f} ms/launch")
