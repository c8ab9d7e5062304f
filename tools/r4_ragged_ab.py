"""Same-process A/B of the 4-chunk generate() workload (serial batch-1 calls and one ragged batch) over the knobs ln_fold / gemm_pad_rows,
plus the 8 x 1001 (M % 256 != 0) bench row against 8 x 1024:  python3 tools/r4_ragged_ab.py"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from eraxvif5tts_amd import _lib  # noqa: E402
from eraxvif5tts_amd.model import CFM, DiT  # noqa: E402

lib = _lib.load()
model = bench.synth_weights(DiT(**bench.BASE_ARCH, text_num_embeds=bench.VOCAB, mel_dim=100, precision="bf16"))
cfm = CFM(transformer=model, mel_spec_kwargs={"mel_spec_type": "vocos"}).cuda()
g = torch.Generator().manual_seed(31)
cond1 = (torch.randn(1, 300, 100, generator=g) * 2 - 3).clamp(math.log(1e-5), 3.0).cuda()
durs = [760, 1010, 900, 1180]
texts = [torch.randint(0, bench.VOCAB, (1, d // 7), generator=g).cuda() for d in durs]
y0s = [torch.randn(1, d, 100, generator=g).cuda() for d in durs]
skw = dict(steps=32, cfg_strength=2.0, sway_sampling_coef=-1.0)


def serial():
    return [cfm.sample(cond=cond1, text=t, duration=d, y0=y, return_trajectory=False, use_graph=False, **skw)[0] for t, d, y in zip(texts, durs, y0s)]


def ragged():
    return cfm.sample_ragged(cond1, texts, durs, y0s=y0s, use_graph=False, **skw)


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


for rnd in range(2):
    for fold, split in ((1, 1), (1, 0), (0, 1), (0, 0)):
        _lib.check(lib.f5_tuning_set(b"ln_fold", fold))
        _lib.check(lib.f5_tuning_set(b"gemm_pad_rows", split))
        print(f"round {rnd} ln_fold={fold} pad_rows={split}: serial {timeit(serial):7.1f} ms   ragged {timeit(ragged):7.1f} ms", flush=True)
_lib.check(lib.f5_tuning_set(b"ln_fold", 1))
for N in (1024, 1001):
    cond, text, lens, dur = bench.synth_batch(8, N, "cuda", seed=0)
    kw = dict(cond=cond, text=text, duration=dur, lens=lens, steps=32, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0, return_trajectory=False, use_graph=True)
    for split in (1, 0, 1, 0):
        _lib.check(lib.f5_tuning_set(b"gemm_pad_rows", split))
        ms = timeit(lambda: cfm.sample(**kw))
        print(f"8 x {N} pad_rows={split}: {ms:7.1f} ms per sample() = {8 * N / ms * 1e3:8.0f} mel-frames/s ({ms / (8 * N) * 1e3:.3f} us per frame)", flush=True)
