"""A/B of the attention start stagger (tuning knob attn_stagger, 10-ns ticks of delay per 64-key tile for the second resident workgroup of
every CU) on the one-item-per-workgroup kernel (attn_variant 2) and the persistent grid (6): interleaved rounds in ONE process, random data.
   python tools/attn_stagger_probe.py [stagger values...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eraxvif5tts_amd import _lib  # noqa: E402

lib = _lib.load()
_lib.require_gpu()
staggers = [int(v) for v in sys.argv[1:]] or [0, 25, 50, 100, 150]
for (B, N, H, tag) in ((64, 1024, 16, "C2"), (16, 4096, 16, "C4")):
    flops = 4.0 * B * H * N * N * 64
    res = {(v, s): [] for v in (2, 6) for s in staggers}
    for rnd in range(4):
        for v in (2, 6):
            for s in staggers:
                _lib.check(lib.f5_tuning_set(b"attn_variant", v))
                _lib.check(lib.f5_tuning_set(b"attn_stagger", s))
                ms = C.c_float(0.0)
                _lib.check(lib.f5_bench_attention(1, B, N, H, 20, C.byref(ms), _lib.stream_ptr()))
                res[(v, s)].append(ms.value)
    _lib.check(lib.f5_tuning_set(b"attn_variant", 0))
    _lib.check(lib.f5_tuning_set(b"attn_stagger", 0))
    for v in (2, 6):
        print(f"{tag} variant {v}: " + "  ".join(f"stagger {s}: {min(res[(v, s)]) * 1e3:.1f} us (med {sorted(res[(v, s)])[len(res[(v, s)]) // 2] * 1e3:.1f}, {flops / min(res[(v, s)]) / 1e9:.0f} TF)"
                                                 for s in staggers), flush=True)
