"""Which call trips the fp16 range guard: repeated graph / eager sample() at the C2 shape with diagnostics.   python3 tools/guard_probe2.py"""
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from eraxvif5tts_amd import _lib  # noqa: E402
from eraxvif5tts_amd.model import CFM, DiT  # noqa: E402

lib = _lib.load()
B, N = 32, 1024
model = bench.synth_weights(DiT(**bench.BASE_ARCH, text_num_embeds=bench.VOCAB, mel_dim=100, precision="bf16"))
cfm = CFM(transformer=model, mel_spec_kwargs={"mel_spec_type": "vocos"}).cuda()
cond, text, lens, dur = bench.synth_batch(B, N, "cuda", seed=0)
for mode in (True, True, False):
    for i in range(5):
        plan = model.plan(B, N, 32)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            out, _ = cfm.sample(cond=cond, text=text, duration=dur, lens=lens, steps=32, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0,
                                return_trajectory=False, use_graph=mode)
            torch.cuda.synchronize()
        if w:  # back to fp16 storage for the next call (drops the captured graphs: the next graph call captures again)
            _lib.check(lib.f5_plan_set_option(plan, b"residual_f16", -1))
        print(f"graph={mode} call {i}: finite {bool(torch.isfinite(out).all())} max {float(out.abs().max()):.4g}  " + (str(w[0].message)[:300] if w else "no event"), flush=True)
