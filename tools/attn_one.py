"""A few launches of one attention variant at one shape, for rocprofv3 counter passes:  python3 tools/attn_one.py <variant> <B*2> <N> <H> <launches>"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eraxvif5tts_amd import _lib  # noqa: E402

v, B, N, H, it = (int(a) for a in sys.argv[1:6])
lib = _lib.load()
_lib.require_gpu()
_lib.check(lib.f5_tuning_set(b"attn_variant", v))
ms = C.c_float(0.0)
_lib.check(lib.f5_bench_attention(1, B, N, H, it, C.byref(ms), _lib.stream_ptr()))
print(f"variant {v}: {ms.value * 1e3:.1f} us per launch")
