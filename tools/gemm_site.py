"""One DiT block GEMM call site, a few back-to-back launches on random operands (target of rocprofv3 passes):
   python tools/gemm_site.py SITE [rows seq iters]      SITE: 0 fused QKV + RoPE, 1 FF1 + GELU, 2 FF2 x gate, 3 attention out x gate"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eraxvif5tts_amd import _lib  # noqa: E402

lib = _lib.load()
_lib.require_gpu()
site = int(sys.argv[1])
rows, seq, iters = [int(x) for x in sys.argv[2:5]] if len(sys.argv) >= 5 else (65536, 1024, 5)
ms = C.c_float(0.0)
_lib.check(lib.f5_bench_gemm_site(1, site, rows, seq, 1024, 16, 2048, iters, C.byref(ms), _lib.stream_ptr()))
N, K = {0: (3072, 1024), 1: (2048, 1024), 2: (1024, 2048), 3: (1024, 1024)}[site]
print(f"site {site} M={rows} N={N} K={K}: {ms.value * 1e3:.1f} us = {2.0 * rows * N * K / ms.value / 1e9:.0f} TFLOP/s")
