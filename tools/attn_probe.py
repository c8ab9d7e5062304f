"""A/B of the attention schedules in ONE process (interleaved rounds, random data): f5_bench_attention per variant at the C2 and C4 shapes,
plus a quick parity check of every variant against the fp64 softmax.   python tools/attn_probe.py [variants...]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from eraxvif5tts_amd import _lib  # noqa: E402

lib = _lib.load()
_lib.require_gpu()
variants = [int(v) for v in sys.argv[1:]] or [2, 5]


def parity(v):
    import gpu_helpers as G
    from test_gpu_ops import _attn_ref
    worst = 0.0
    for (B, N, H, masked) in ((2, 200, 3, True), (1, 1024, 2, False), (2, 1024, 2, True), (1, 2050, 2, False), (3, 333, 1, True), (1, 64, 2, False), (1, 65, 1, False)):
        g = torch.Generator().manual_seed(N + H)
        qkv = G.bf16_round(torch.randn(B, N, 3, H, 64, generator=g) * 1.5)
        mask = None
        if masked:
            lens = torch.tensor([N, max(1, N - 13), max(1, N // 2)][:B])
            mask = torch.arange(N)[None, :] < lens[:, None]
        ref = _attn_ref(qkv, mask)
        _lib.check(lib.f5_tuning_set(b"attn_variant", v))
        out = G.op_attention(0, 1, qkv, mask)
        _lib.check(lib.f5_tuning_set(b"attn_variant", 0))
        valid = slice(None) if mask is None else mask
        err = float((out[valid] - ref[valid]).norm() / ref[valid].norm())
        worst = max(worst, err)
        if not (err < 6e-3 and torch.isfinite(out).all()):
            print(f"  variant {v}: PARITY FAIL at B={B} N={N} H={H} masked={masked}: rel-L2 {err:.3e}")
    return worst


for v in variants:
    print(f"variant {v}: worst rel-L2 vs fp64 softmax {parity(v):.3e}", flush=True)

for (B, N, H, tag) in ((64, 1024, 16, "C2"), (16, 4096, 16, "C4"))[:int(os.environ.get("PROBE_SHAPES", "4"))] + (((8, 1024, 16, "B=4"), (2, 1024, 16, "B=1")) if int(os.environ.get("PROBE_SHAPES", "4")) > 2 else ()):
    flops = 4.0 * B * H * N * N * 64
    best = {v: [] for v in variants}
    for rnd in range(3):
        for v in variants:
            _lib.check(lib.f5_tuning_set(b"attn_variant", v))
            ms = C.c_float(0.0)
            _lib.check(lib.f5_bench_attention(1, B, N, H, 20, C.byref(ms), _lib.stream_ptr()))
            best[v].append(ms.value)
    _lib.check(lib.f5_tuning_set(b"attn_variant", 0))
    print(f"{tag} (B*2={B}, N={N}): " + "  ".join(f"v{v}: {min(t) * 1e3:.1f} us = {flops / min(t) / 1e9:.0f} TF (med {sorted(t)[1] * 1e3:.1f})" for v, t in best.items()), flush=True)
