"""Diagnostic of the fp16 residual-stream range guard at a bench shape: which NFE trips it, and what it saw.
   python3 tools/guard_probe.py [B N]"""
import ctypes as C
import os
import struct
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from eraxvif5tts_amd import _lib  # noqa: E402
from eraxvif5tts_amd.model import CFM, DiT  # noqa: E402

B, N = [int(a) for a in sys.argv[1:3]] if len(sys.argv) >= 3 else (32, 1024)
lib = _lib.load()
model = bench.synth_weights(DiT(**bench.BASE_ARCH, text_num_embeds=bench.VOCAB, mel_dim=100, precision="bf16"))
cfm = CFM(transformer=model, mel_spec_kwargs={"mel_spec_type": "vocos"}).cuda()
cond, text, lens, dur = bench.synth_batch(B, N, "cuda", seed=0)


def opt(plan, key):
    v = C.c_int(0)
    _lib.check(lib.f5_plan_get_option(plan, key, C.byref(v)))
    return v.value


import warnings
warnings.simplefilter("ignore")
for nfe in (1, 2, 4, 8, 16, 32):
    plan = model.plan(B, N, 32)
    _lib.check(lib.f5_plan_set_option(plan, b"residual_f16", -1))
    before = opt(plan, b"residual_fallbacks")
    out, traj = cfm.sample(cond=cond, text=text, duration=dur, lens=lens, steps=nfe, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0,
                           return_trajectory=True, use_graph=False)
    torch.cuda.synchronize()
    fired = opt(plan, b"residual_fallbacks") - before
    amax = struct.unpack("f", struct.pack("I", opt(plan, b"residual_guard_amax_bits") & 0xffffffff))[0]
    print(f"NFE {nfe:2d}: guard fired {fired}  amax seen {amax:.4g}  nan {opt(plan, b'residual_guard_nan')}  |x(1)| max {float(traj[-1].abs().max()):.4g} "
          f"|out| max {float(out.abs().max()):.4g} finite {bool(torch.isfinite(out).all())}", flush=True)
    for s in range(traj.shape[0]):
        print(f"    state {s}: max |x| {float(traj[s].abs().max()):.4g}")
