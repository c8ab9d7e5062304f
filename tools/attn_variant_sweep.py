"""wide (64 queries per wave, variant 2) vs pipelined (32 queries per wave, variant 5) attention kernel over small grids: where should the launcher switch?
   python tools/attn_variant_sweep.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eraxvif5tts_amd import _lib  # noqa: E402

lib = _lib.load()
_lib.require_gpu()
H = 16
for nb, N in ((2, 768), (2, 1024), (2, 1280), (2, 1536), (2, 1792), (2, 2048), (4, 768), (4, 1024), (4, 1536), (6, 1024), (6, 1536), (8, 1024), (3, 1024)):
    res = {}
    for v in (0, 2, 5):
        best = 1e9
        for rnd in range(3):
            _lib.check(lib.f5_tuning_set(b"attn_variant", v))
            ms = C.c_float(0.0)
            _lib.check(lib.f5_bench_attention(1, nb, N, H, 20, C.byref(ms), _lib.stream_ptr()))
            best = min(best, ms.value)
        res[v] = best
    _lib.check(lib.f5_tuning_set(b"attn_variant", 0))
    wgs = nb * H * ((N + 255) // 256)
    print(f"nb={nb} N={N} (256-query blocks {wgs}): auto {res[0]*1e3:.1f} us  wide {res[2]*1e3:.1f} us  pipelined {res[5]*1e3:.1f} us", flush=True)
