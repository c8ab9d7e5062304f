#!/usr/bin/env python3
"""One-off measurements of the other BASELINE.json configs on one MI355X (not bench lines): C4 long-form, small batches."""
import math, sys, time
import torch
sys.path.insert(0, ".")
import bench
from eraxvif5tts_amd.model import CFM, DiT
model = bench.synth_weights(DiT(**bench.BASE_ARCH, text_num_embeds=bench.VOCAB, mel_dim=100, precision="bf16"))
cfm = CFM(transformer=model, mel_spec_kwargs={"mel_spec_type": "vocos"}).cuda()
for (B, N, nfe) in [(8, 4096, 32), (32, 1024, 32), (4, 1024, 32), (1, 1024, 32), (1, 512, 32), (1, 256, 8)]:
    cond, text, lens, dur = bench.synth_batch(B, N, "cuda", seed=1)
    ts = []
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out, _ = cfm.sample(cond=cond, text=text, duration=dur, lens=lens, steps=nfe, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0,
                            return_trajectory=False)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    assert torch.isfinite(out).all()
    t = min(ts[1:])
    audio_s = B * (N - N // 3) * 256 / 24000
    flops = (378.9e6 + 90112.0 * N) * B * N * 2 * nfe
    print(f"B={B:2d} N={N:4d} NFE={nfe:2d}: first call {ts[0]*1e3:8.1f} ms, steady {t*1e3:8.1f} ms, {B*N/t:9.0f} mel-frames/s, RTF {t/audio_s:.5f}, {flops/t/1e12:6.1f} TFLOP/s")
