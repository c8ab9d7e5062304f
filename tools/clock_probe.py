"""The clock the chip holds under the hot kernels (MI355X lowers it under load): d(s_memtime) / d(s_memrealtime) x 100 MHz per workgroup, measured
inside the tuned GEMM at its four call sites (after >= 1 s of back-to-back launches on random operands) and inside the attention kernel.
   python tools/clock_probe.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eraxvif5tts_amd import _lib  # noqa: E402

lib = _lib.load()
_lib.require_gpu()
rows, seq, D, H, ff = 65536, 1024, 1024, 16, 2048
buf = torch.zeros(4096 * 4, dtype=torch.int64, device="cuda")
names = ["QKV + RoPE (N 3072, K 1024)", "FF1 + GELU (N 2048, K 1024)", "FF2 x gate (N 1024, K 2048)", "out-projection x gate (N 1024, K 1024)"]
flops = [2.0 * rows * 3072 * 1024, 2.0 * rows * 2048 * 1024, 2.0 * rows * 1024 * 2048, 2.0 * rows * 1024 * 1024]
for site in range(4):
    ms = C.c_float(0.0)
    _lib.check(lib.f5_bench_gemm_site(1, site, rows, seq, D, H, ff, 1200 if site == 3 else 600, C.byref(ms), _lib.stream_ptr()))  # heat: ~0.2-0.3 s
    buf.zero_()
    _lib.check(lib.f5_debug_gemm_clock(C.c_void_p(buf.data_ptr())))
    _lib.check(lib.f5_bench_gemm_site(1, site, rows, seq, D, H, ff, 50, C.byref(ms), _lib.stream_ptr()))
    torch.cuda.synchronize()
    _lib.check(lib.f5_debug_gemm_clock(None))
    c = buf.cpu().view(-1, 4).double()
    c = c[c[:, 3] > c[:, 1]]
    ghz = (c[:, 2] - c[:, 0]) / (c[:, 3] - c[:, 1]) * 0.1
    tf = flops[site] / (ms.value * 1e-3) / 1e12
    peak_at_clock = 256 * 4 * (16 * 16 * 32 * 2 / 16.0) * float(ghz.median()) * 1e9 / 1e12  # 1024 SIMDs x 1024 FLOP per cycle
    print(f"{names[site]:42s} {ms.value * 1e3:7.1f} us = {tf:6.0f} TFLOP/s; clock held {float(ghz.median()):.3f} GHz (min {float(ghz.min()):.3f}, max {float(ghz.max()):.3f}) "
          f"-> MFMA peak at that clock {peak_at_clock:6.0f} TFLOP/s, matrix pipe busy {tf / peak_at_clock:.3f}", flush=True)
