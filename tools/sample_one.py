"""One eager CFM.sample() at a bench shape with few Euler steps (target of rocprofv3 counter passes):
   python3 tools/sample_one.py [B N NFE]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from eraxvif5tts_amd.model import CFM, DiT  # noqa: E402

B, N, nfe = [int(a) for a in sys.argv[1:4]] if len(sys.argv) >= 4 else (32, 1024, 2)
model = bench.synth_weights(DiT(**bench.BASE_ARCH, text_num_embeds=bench.VOCAB, mel_dim=100, precision="bf16"))
cfm = CFM(transformer=model, mel_spec_kwargs={"mel_spec_type": "vocos"}).cuda()
cond, text, lens, dur = bench.synth_batch(B, N, "cuda", seed=0)
out, _ = cfm.sample(cond=cond, text=text, duration=dur, lens=lens, steps=nfe, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0, return_trajectory=False,
                    use_graph=False)
torch.cuda.synchronize()
print("finite", bool(torch.isfinite(out).all()))
