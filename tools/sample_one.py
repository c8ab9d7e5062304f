"""One eager CFM.sample() at a bench shape with few Euler steps (target of rocprofv3 counter passes):
   python3 tools/sample_one.py [B N NFE] [--events] [--repeat=K]
   --events: the HIP event pairs of bench.py's in-situ timing pass around every block kernel; --repeat=K: K samples, a device sync after each"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from eraxvif5tts_amd.model import CFM, DiT  # noqa: E402

events = "--events" in sys.argv
repeat = max([int(a.split("=")[1]) for a in sys.argv if a.startswith("--repeat=")] or [1])
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
B, N, nfe = [int(a) for a in argv[:3]] if len(argv) >= 3 else (32, 1024, 2)
model = bench.synth_weights(DiT(**bench.BASE_ARCH, text_num_embeds=bench.VOCAB, mel_dim=100, precision="bf16"))
cfm = CFM(transformer=model, mel_spec_kwargs={"mel_spec_type": "vocos"}).cuda()
cond, text, lens, dur = bench.synth_batch(B, N, "cuda", seed=0)
if events:
    import ctypes as C
    from eraxvif5tts_amd import _lib
    plan = model.plan(B, N, nfe)
    _lib.check(_lib.load().f5_plan_timing_begin(plan, (7 * 22 + 4) * nfe))
for _rep in range(repeat - 1):
    cfm.sample(cond=cond, text=text, duration=dur, lens=lens, steps=nfe, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0, return_trajectory=False, use_graph=False)
    torch.cuda.synchronize()
out, _ = cfm.sample(cond=cond, text=text, duration=dur, lens=lens, steps=nfe, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0, return_trajectory=False,
                    use_graph=False)
torch.cuda.synchronize()
if events:
    ms, cnt = C.c_float(0.0), C.c_int(0)
    _lib.check(_lib.load().f5_plan_timing_end(plan, C.byref(ms), C.byref(cnt), _lib.stream_ptr()))
    print("qkv ms", ms.value, "launches", cnt.value)
print("finite", bool(torch.isfinite(out).all()))
