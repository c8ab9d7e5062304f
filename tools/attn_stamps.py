"""Where an item of the persistent attention kernel spends its time: in-kernel shader-clock stamps of wave 0 (diagnostic build, never timed
for throughput).   python tools/attn_stamps.py [B N H]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eraxvif5tts_amd import _lib  # noqa: E402

lib = _lib.load()
_lib.require_gpu()
B, N, H = [int(a) for a in sys.argv[1:4]] if len(sys.argv) >= 4 else (64, 1024, 16)
G = 512
buf = torch.zeros(G * 64 + G * 4, dtype=torch.int64, device="cuda")
_lib.check(lib.f5_tuning_set(b"attn_variant", 6))
ms = C.c_float(0.0)
_lib.check(lib.f5_bench_attention(1, B, N, H, 5, C.byref(ms), _lib.stream_ptr()))  # warm
_lib.check(lib.f5_debug_attn_stamps(C.c_void_p(buf.data_ptr())))
_lib.check(lib.f5_bench_attention(1, B, N, H, 1, C.byref(ms), _lib.stream_ptr()))
torch.cuda.synchronize()
_lib.check(lib.f5_debug_attn_stamps(None))
_lib.check(lib.f5_tuning_set(b"attn_variant", 0))
raw = buf.cpu()
t = raw[: G * 64].view(G, 8, 8).double()
clk = raw[G * 64:].view(G, 4).double()
ghz = (clk[:, 2] - clk[:, 0]) / (clk[:, 3] - clk[:, 1]) * 0.1
print(f"in-kernel clock (d s_memtime / d s_memrealtime x 100 MHz), median over workgroups: {float(ghz.median()):.3f} GHz (min {float(ghz.min()):.3f}, max {float(ghz.max()):.3f})")
names = ["item start (Q frags in regs)", "Q prefetch issued", "tile 0 done", "tile 1 done", "tile 2 done", "last tile done", "epilogue done"]
nt = N // 64
print(f"B={B} N={N} H={H}: kernel {ms.value * 1e3:.1f} us (stamped build); shader-clock cycles, median over {G} workgroups (wave 0)")
for it in range(min(8, 8)):
    row = t[:, it, :]
    if (row[:, 0] == 0).all():
        break
    d = lambda a, b: float((row[:, a] - row[:, b]).median())
    prev_end = t[:, it - 1, 6] if it > 0 else None
    gap = float((row[:, 0] - prev_end).median()) if prev_end is not None else float("nan")
    print(f"item {it}: prev epilogue end -> start {gap:8.0f} | start -> Q issued {d(1, 0):6.0f} | -> tile0 done {d(2, 1):6.0f} | tile1 {d(3, 2):6.0f} | tile2 {d(4, 3):6.0f} | "
          f"tiles 3..{nt - 1} {d(5, 4):8.0f} ({d(5, 4) / max(nt - 3, 1):6.0f} each) | epilogue {d(6, 5):6.0f} | item total {d(6, 0):8.0f}")
span = t[:, :, 6].max(dim=1).values - t[:, 0, 0]
print(f"workgroup life (first start -> last epilogue): median {float(span.median()):.0f}, min {float(span.min()):.0f}, max {float(span.max()):.0f} cycles")
