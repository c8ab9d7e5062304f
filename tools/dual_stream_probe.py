#!/usr/bin/env python3
"""Does running two half-batches on two streams beat one full batch?  (desynchronises the GEMM store bursts, hides launch gaps)
python tools/dual_stream_probe.py [B] [N] [nsplit]"""
import sys, time
import torch
sys.path.insert(0, ".")
import bench
from eraxvif5tts_amd import _lib
from eraxvif5tts_amd.model import CFM, DiT
_lib.require_gpu()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
S = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda", 0)

def make(b, seed):
    model = bench.synth_weights(DiT(**bench.BASE_ARCH, text_num_embeds=bench.VOCAB, mel_dim=100, precision="bf16"))
    cfm = CFM(transformer=model, mel_spec_kwargs={"mel_spec_type": "vocos"}, odeint_kwargs={"method": "euler"}).to(dev)
    return cfm, bench.synth_batch(b, N, dev, seed=seed)

def run(cfm, batch):
    cond, text, lens, duration = batch
    return cfm.sample(cond=cond, text=text, duration=duration, lens=lens, steps=32, cfg_strength=2.0, sway_sampling_coef=-1.0,
                      seed=0, return_trajectory=False, use_graph=True)[0]

full = make(B, 0)
for _ in range(2): run(*full)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): run(*full)
torch.cuda.synchronize()
t_full = (time.perf_counter() - t0) / 3
print(f"one stream  B={B}: {t_full * 1e3:.1f} ms/step = {B * N / t_full:.0f} mel-frames/s", flush=True)

parts = [make(B // S, i) for i in range(S)]
streams = [torch.cuda.Stream() for _ in range(S)]
def step_split():
    for st, pt in zip(streams, parts):
        with torch.cuda.stream(st):
            run(*pt)
for _ in range(2): step_split()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): step_split()
torch.cuda.synchronize()
t_split = (time.perf_counter() - t0) / 3
print(f"{S} streams x B={B // S}: {t_split * 1e3:.1f} ms/step = {B * N / t_split:.0f} mel-frames/s  ({t_full / t_split:.3f}x)")
