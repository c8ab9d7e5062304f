"""A/B of GEMM tuning knobs at the four block call sites in ONE process (interleaved rounds, random operands):
   python tools/gemm_ab.py ROWS key=value[,key=value] ...   (each argument after ROWS is one configuration; "base" = defaults)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eraxvif5tts_amd import _lib  # noqa: E402

lib = _lib.load()
_lib.require_gpu()
rows = int(sys.argv[1])
configs = sys.argv[2:] or ["base"]
DEFAULTS = {"gemm_variant": 1, "gemm_persist": 1, "gemm_lean": 1, "gemm_group": 0, "gemm_tile": 0, "gemm_bm128": 1, "gemm_w4": 1}


def apply(cfg):
    kv = dict(DEFAULTS)
    if cfg != "base":
        for part in cfg.split(","):
            k, v = part.split("=")
            kv[k] = int(v)
    for k, v in kv.items():
        _lib.check(lib.f5_tuning_set(k.encode(), v))


SITES = {0: ("qkv", 3072, 1024), 1: ("ff1", 2048, 1024), 2: ("ff2", 1024, 2048), 3: ("out", 1024, 1024)}
for site, (name, N, K) in SITES.items():
    best = {c: [] for c in configs}
    for rnd in range(3):
        for c in configs:
            apply(c)
            ms = C.c_float(0.0)
            _lib.check(lib.f5_bench_gemm_site(1, site, rows, 1024, 1024, 16, 2048, 10, C.byref(ms), _lib.stream_ptr()))
            best[c].append(ms.value)
    apply("base")
    print(f"{name} M={rows} N={N} K={K}: " + "  ".join(f"[{c}] {min(t) * 1e3:.1f} us = {2.0 * rows * N * K / min(t) / 1e9:.0f} TF" for c, t in best.items()), flush=True)
