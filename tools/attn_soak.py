"""Race screen of the attention kernels: the same launch repeated on ragged / masked / long shapes, every output compared bit for bit with the
first (the DMA ring's counted waits and barriers are placed by hand: a misplaced one shows up as rare run-to-run differences).
   python tools/attn_soak.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from eraxvif5tts_amd import _lib  # noqa: E402
import gpu_helpers as G  # noqa: E402

lib = _lib.load()
_lib.require_gpu()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for variant in (2, 5, 6):
    for (B, N, H, masked) in ((8, 1024, 16, False), (6, 1000, 16, True), (2, 4096, 16, False), (3, 4033, 8, True), (5, 333, 4, True), (64, 256, 16, False),
                              (1, 1000, 16, False), (2, 4033, 16, False), (20, 768, 16, False), (64, 1024, 16, False)):  # ragged N without a mask array: the MASKED build with mask == nullptr (single-utterance serving)
        g = torch.Generator().manual_seed(N + H)
        qkv = G.bf16_round(torch.randn(B, N, 3, H, 64, generator=g) * 1.5)
        mask = None
        if masked:
            lens = torch.tensor([N - 17 * i for i in range(B)])
            mask = torch.arange(N)[None, :] < lens[:, None]
        _lib.check(lib.f5_tuning_set(b"attn_variant", variant))
        q = qkv.cuda().float().contiguous()  # uploaded once; f5_op_attention converts and launches on the device
        mk = None if mask is None else mask.cuda().to(torch.uint8).contiguous()
        out = torch.empty(B, N, H * 64, device="cuda")
        ref, bad = None, 0
        for _ in range(reps):
            out.zero_()
            _lib.check(lib.f5_op_attention(0, 1, B, N, H, _lib.ptr(q), _lib.ptr(mk), _lib.ptr(out), _lib.stream_ptr()))
            if ref is None:
                ref = out.clone()
            elif not torch.equal(out, ref):
                bad += 1
        _lib.check(lib.f5_tuning_set(b"attn_variant", 0))
        print(f"variant {variant} B={B} N={N} H={H} masked={masked}: {reps} launches, {bad} differ, finite={bool(torch.isfinite(ref).all())}", flush=True)
        assert bad == 0
print("attention soak ok")
