#!/usr/bin/env python3
"""Two half-batches on two CU-masked streams (each stream owns half of the CUs) vs one full batch on the whole chip.
python tools/cumask_probe.py [B] [N] [mode]   mode 0: lower/upper 128 mask bits, 1: even/odd bits"""
import ctypes as C, sys, time
import torch
sys.path.insert(0, ".")
import bench
from eraxvif5tts_amd import _lib
from eraxvif5tts_amd.model import CFM, DiT
_lib.require_gpu()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda", 0)
torch.cuda.init(); torch.zeros(1, device=dev)
hip = C.CDLL("libamdhip64.so")

def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)

if mode == 0:
    masks = [(1 << 128) - 1, ((1 << 128) - 1) << 128]
else:
    even = sum(1 << i for i in range(0, 256, 2))
    masks = [even, even << 1]

def make(b, seed):
    model = bench.synth_weights(DiT(**bench.BASE_ARCH, text_num_embeds=bench.VOCAB, mel_dim=100, precision="bf16"))
    cfm = CFM(transformer=model, mel_spec_kwargs={"mel_spec_type": "vocos"}, odeint_kwargs={"method": "euler"}).to(dev)
    return cfm, bench.synth_batch(b, N, dev, seed=seed)

def run(cfm, batch):
    cond, text, lens, duration = batch
    return cfm.sample(cond=cond, text=text, duration=duration, lens=lens, steps=32, cfg_strength=2.0, sway_sampling_coef=-1.0,
                      seed=0, return_trajectory=False, use_graph=True)[0]

def timeit(fn, n=3):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

full = make(B, 0)
t_full = timeit(lambda: run(*full))
print(f"whole chip, B={B}: {t_full * 1e3:.1f} ms/step = {B * N / t_full:.0f} mel-frames/s", flush=True)
parts = [make(B // 2, i) for i in range(2)]
streams = [masked_stream(m) for m in masks]
def one_half():
    with torch.cuda.stream(streams[0]):
        run(*parts[0])
t_half = timeit(one_half)
print(f"one masked stream alone, B={B // 2}: {t_half * 1e3:.1f} ms/step (half the CUs: expect ~{t_full * 1e3:.0f})", flush=True)
def both():
    for st, pt in zip(streams, parts):
        with torch.cuda.stream(st):
            run(*pt)
t_split = timeit(both)
print(f"2 masked streams x B={B // 2}: {t_split * 1e3:.1f} ms/step = {B * N / t_split:.0f} mel-frames/s  ({t_full / t_split:.3f}x)")
