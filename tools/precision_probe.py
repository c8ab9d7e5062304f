#!/usr/bin/env python3
"""bf16 production path vs the exact-fp32 mode of the same library at F5TTS_Base depth: error of one network evaluation and of
sample() as a function of NFE, sequence length and CFG strength."""
import sys
import torch
sys.path.insert(0, ".")
import bench
from eraxvif5tts_amd.model import CFM, DiT

def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())

models = {}
for prec in ("fp32", "bf16"):
    torch.manual_seed(1234)  # DiT's default init draws from the global RNG: same weights for both precisions
    m = bench.synth_weights(DiT(**bench.BASE_ARCH, text_num_embeds=bench.VOCAB, mel_dim=100, precision=prec), seed=0)
    models[prec] = CFM(transformer=m, mel_spec_kwargs={"mel_spec_type": "vocos"}).cuda()

for N in (256, 1024):
    B = 2
    cond, text, lens, dur = bench.synth_batch(B, N, "cuda", seed=9)
    g = torch.Generator().manual_seed(10)
    y0 = torch.randn(B, N, 100, generator=g)
    n_ref = cond.shape[1]
    # single evaluation
    x = y0.cuda()
    step_cond = torch.nn.functional.pad(cond, (0, 0, 0, N - n_ref))
    mask = torch.ones(B, N, dtype=torch.bool, device="cuda")
    outs = {p: models[p].transformer(x=x, cond=step_cond, text=text, time=torch.tensor(0.3).cuda(), mask=mask, drop_audio_cond=False, drop_text=False).cpu()
            for p in models}
    print(f"N={N}: one evaluation bf16 vs fp32: rel-L2 {rel(outs['bf16'], outs['fp32']):.3e}  |out| rms {float(outs['fp32'].pow(2).mean().sqrt()):.3f}")
    for cfg in (0.0, 2.0):
        for nfe in (1, 2, 4, 8, 16, 32):
            res = {}
            for p in models:
                out, _ = models[p].sample(cond=cond, text=text, duration=dur, lens=lens, steps=nfe, cfg_strength=cfg, sway_sampling_coef=-1.0, y0=y0,
                                          return_trajectory=False, use_graph=False)
                res[p] = out[:, n_ref:].cpu()
            print(f"N={N} cfg={cfg} NFE={nfe:2d}: rel-L2 {rel(res['bf16'], res['fp32']):.3e}  rms(out) {float(res['fp32'].pow(2).mean().sqrt()):.3f}")
