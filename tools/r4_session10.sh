#!/bin/bash
OUT=gpurun_out
python -m pytest tests/test_gpu_vocoder_wrapper.py -m gpu -q -x -k "bigvgan" > $OUT/r4n_bigvgan.log 2>&1; tail -25 $OUT/r4n_bigvgan.log
python - <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
from oracle import cpu_ref
from eraxvif5tts_amd.bigvgan import BigVGAN
hp = dict(cpu_ref.BIGVGAN_V2_24K_100BAND_256X)
W = cpu_ref.random_bigvgan_weights(hp, seed=1)
W["conv_post.weight"] = W["conv_post.weight"] * 0.05
voc = BigVGAN(hp); voc.load_state_dict(W); voc = voc.eval().cuda()
for T in (683, 2731):
    mel = (torch.randn(1, 100, T) * 2 - 3).cuda()
    out = voc(mel); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): out = voc(mel)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print(f"full-size BigVGAN-v2 (112 M parameters) T={T}: {ms:.1f} ms per utterance = {T / ms * 1e3:.0f} mel-frames/s, RTF {ms / 1e3 / (T * 256 / 24000):.5f}, finite {bool(torch.isfinite(out).all())}")
mel = (torch.randn(1, 100, 40) * 2 - 3)
ref = cpu_ref.bigvgan_forward(W, hp, mel)
out = voc(mel.cuda()).cpu()
print("full-size vs oracle T=40 rel-L2", float((out - ref).norm() / ref.norm()))
PY
