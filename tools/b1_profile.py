#!/usr/bin/env python3
"""Single-utterance sample() (B=1, N=1024, NFE=32) for rocprofv3: where does the latency go?"""
import sys, time
import torch
sys.path.insert(0, ".")
import bench
from eraxvif5tts_amd.model import CFM, DiT
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model = bench.synth_weights(DiT(**bench.BASE_ARCH, text_num_embeds=bench.VOCAB, mel_dim=100, precision="bf16"))
cfm = CFM(transformer=model, mel_spec_kwargs={"mel_spec_type": "vocos"}).cuda()
cond, text, lens, dur = bench.synth_batch(1, N, "cuda", seed=1)
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out, _ = cfm.sample(cond=cond, text=text, duration=dur, lens=lens, steps=32, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0, return_trajectory=False)
    torch.cuda.synchronize(); print(f"call {it}: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
