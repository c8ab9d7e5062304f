#!/usr/bin/env python3
"""Headline benchmark: mel-frames/s of the F5-TTS flow-matching sampler (CFM.sample: 32 Euler steps x (cond + uncond) DiT
evaluations, CFG 2, sway -1) on MI355X, at BASELINE.json's C2: F5TTS_Base, bf16, batch 32, seq_len 1024, NFE 32.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

A "step" is one full sample() over one synthetic batch of 32 fixed-length utterances that is already resident in HBM.
Multi-GPU: utterance batches are sharded over ranks (one process per GPU, full weight replica, no collective inside the
ODE loop); the finished mels are all-gathered with RCCL inside the timed region.  Weak scaling: 32 utterances per GPU.

One JSON line on rank 0 with the driver's contract fields plus
  "roofline"      the dominant kernel (fused QKV projection GEMM, one shape per launch): algorithmic FLOPs per launch /
                  mean launch time measured here with HIP events (f5_bench_gemm_site), vs the 2.5 PFLOP/s dense bf16 MFMA peak
  "cpu_baseline"  the CPU oracle (oracle/cpu_ref.py, plain fp32 torch; the reference itself cannot travel) timed on this
                  host's cores on a bounded sample, scaled linearly to NFE 32.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BASE_ARCH = dict(dim=1024, depth=22, heads=16, ff_mult=2, text_dim=512, text_mask_padding=False, conv_layers=4, pe_attn_head=1)
VOCAB = 2545
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
QKV_TRAFFIC_BYTES = int((320754.9 * 2 + 393299.5) * 1024)  # profiles/r1_08_c2_persistent_grid_kernel_stats.md (PMC passes)


def synth_weights(model, seed=0):
    """Random-init F5TTS_Base (no checkpoint exists offline); zero-initialised tensors re-randomised (sigma 0.02) so the
    network is not degenerate."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for _, p in model.named_parameters():
            if torch.count_nonzero(p) == 0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    return model


def synth_batch(B, N, device, seed=0):
    """cond mel ~ N(-3, 2^2) clipped to [ln 1e-5, 3], N_ref = N//3, text ids uniform, length N//6."""
    g = torch.Generator().manual_seed(seed)
    n_ref = N // 3
    cond = (torch.randn(B, n_ref, 100, generator=g) * 2 - 3).clamp(math.log(1e-5), 3.0)
    text = torch.randint(0, VOCAB, (B, N // 6), generator=g)
    lens = torch.full((B,), n_ref, dtype=torch.long)
    duration = torch.full((B,), N, dtype=torch.long)
    return cond.to(device), text.to(device), lens.to(device), duration.to(device)


def cpu_baseline(model, N, seconds_hint=20):
    """CPU oracle on the host cores: B=1, N, NFE=1 (2 network evaluations), scaled linearly in NFE to 32."""
    from oracle import cpu_ref  # checker, used here only as the timed CPU baseline
    # the GPU box exposes every host core but a 1-GPU job owns a 16-core share; oversubscribing torch's pool is far slower
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(16, avail))
    torch.set_num_threads(cores)
    W = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    n_ref = N // 3
    bs, nfe = 2, 2
    cond = (torch.randn(bs, n_ref, 100, generator=g) * 2 - 3).clamp(math.log(1e-5), 3.0)
    text = torch.randint(0, VOCAB, (bs, N // 6), generator=g)
    t0 = time.perf_counter()
    cpu_ref.sample(W, BASE_ARCH, cond, text, N, steps=nfe, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0, return_trajectory=False)
    dt = time.perf_counter() - t0
    value = bs * N / (dt * 32 / nfe)
    return {"value": round(value, 3), "unit": "mel-frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle/cpu_ref.sample fp32 torch, B={bs} N={N} NFE={nfe} CFG=2 ({2 * nfe} network evaluations of {bs} utterances, "
                      f"{dt:.1f} s), scaled linearly to NFE=32"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--seq-len", type=int, default=1024)
    ap.add_argument("--nfe", type=int, default=32)
    ap.add_argument("--cfg", type=float, default=2.0)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        ndev = torch.cuda.device_count()
        torch.cuda.set_device(local_rank % max(ndev, 1))
        backend = os.environ.get("F5_BENCH_BACKEND", "nccl")  # nccl == RCCL on ROCm; gloo only to rehearse the N>1 path on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group(backend)
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    from eraxvif5tts_amd import _lib
    from eraxvif5tts_amd.model import CFM, DiT
    _lib.require_gpu()
    lib = _lib.load()

    B, N, nfe = args.batch, args.seq_len, args.nfe
    model = synth_weights(DiT(**BASE_ARCH, text_num_embeds=VOCAB, mel_dim=100, precision=args.precision))
    cfm = CFM(transformer=model, mel_spec_kwargs={"mel_spec_type": "vocos"}, odeint_kwargs={"method": "euler"}).to(dev)
    cond, text, lens, duration = synth_batch(B, N, dev, seed=rank)

    gathered = [torch.empty(B, N, 100, device=dev) for _ in range(world)] if world > 1 else None

    def step():
        out, _ = cfm.sample(cond=cond, text=text, duration=duration, lens=lens, steps=nfe, cfg_strength=args.cfg,
                            sway_sampling_coef=-1.0, seed=0, return_trajectory=False, use_graph=not args.no_graph)
        if world > 1:
            dist.all_gather(gathered, out.contiguous())  # the only collective of the path: finished mels over RCCL/xGMI
        return out

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out).all(), "non-finite mel output"

    frames = B * N * args.steps * world
    value = frames / elapsed
    gen_audio_s = B * (N - N // 3) * 256 / 24000.0 * args.steps * world
    result = {
        "metric": "mel-frames/s", "value": round(value, 2), "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "rtf": round(elapsed / gen_audio_s, 6),
        "config": {"workload": f"C2 F5TTS_Base random-init, CFM.sample euler NFE={nfe} CFG={args.cfg:g} sway=-1, batch {B}/GPU x seq_len {N} "
                               f"(N_ref={N // 3}), hipGraph={'off' if args.no_graph else 'on'}",
                   "global_batch": B * world, "seq_len": N, "nfe": nfe, "parallelism": f"utterance-sharded dp{world}"},
    }

    if rank == 0:
        # ---- roofline of the dominant kernel: fused QKV projection (29 % of the FLOPs, one shape per launch)
        rows = 2 * B * N if args.cfg >= 1e-5 else B * N
        flops = 2.0 * rows * (3 * 16 * 64) * 1024
        # in situ: one extra eager sample() with a HIP event pair around every fused-QKV launch (22 blocks x nfe evaluations),
        # on the stream the kernels run on; this is the same launch rocprofv3 averages in profiles/
        plan = model.plan(B, N, nfe)
        ms, cnt = C.c_float(0.0), C.c_int(0)
        _lib.check(lib.f5_plan_timing_begin(plan, BASE_ARCH["depth"] * nfe), "timing_begin")
        cfm.sample(cond=cond, text=text, duration=duration, lens=lens, steps=nfe, cfg_strength=args.cfg, sway_sampling_coef=-1.0, seed=0,
                   return_trajectory=False, use_graph=False)
        _lib.check(lib.f5_plan_timing_end(plan, C.byref(ms), C.byref(cnt), _lib.stream_ptr()), "timing_end")
        achieved = flops / (ms.value * 1e-3) / 1e12
        # isolated: back-to-back launches of the same kernel on cold operands (f5_bench_gemm_site), for reference
        iso = C.c_float(0.0)
        iso_ok = lib.f5_bench_gemm_site(1, 0, rows, N, 1024, 16, 2048, 10, C.byref(iso), _lib.stream_ptr()) == 0
        result["roofline"] = {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4),
                              # HBM-side bytes per launch from the rocprofv3 PMC passes recorded in profiles/r1_08_* (FETCH_SIZE x 2 on gfx950
                              # + WRITE_SIZE); only meaningful for the default C2 shape, null otherwise
                              "traffic": QKV_TRAFFIC_BYTES if (rows == 65536 and N == 1024) else None,
                              "kernel": "gemm_fast_kernel<256,128,DENSE,QKV+RoPE epilogue,persistent grid>",
                              "launch": f"M={rows} N=3072 K=1024, {flops / 1e9:.1f} GFLOP, {ms.value:.4f} ms mean over {cnt.value} launches inside an "
                                        f"eager sample() (HIP event pairs on the launch stream)",
                              "isolated_tflops": round(flops / (iso.value * 1e-3) / 1e12, 2) if iso_ok else None}
        # what the matrix pipe itself sustains on THIS device (register-resident MFMA stream, no memory traffic): clock-limited with
        # zero operands, power-limited with realistic ones -- context for `frac`, whose denominator stays the data-sheet peak
        sus = {}
        for label, rnd in (("zero_operands", 0), ("random_operands", 1)):
            tf = C.c_float(0.0)
            if lib.f5_bench_mfma_rate(rnd, C.byref(tf), _lib.stream_ptr()) == 0:
                sus[label] = round(tf.value, 1)
        result["roofline"]["mfma_sustained_tflops"] = sus or None
        # whole-loop MFMA fraction from the algorithmic FLOPs of SURVEY.md 8(d)
        per_token = 378.9e6 + 90112.0 * N
        total_flops = per_token * B * N * (2 if args.cfg >= 1e-5 else 1) * nfe * args.steps * world
        result["loop_tflops"] = round(total_flops / elapsed / 1e12 / world, 2)
        result["loop_mfma_frac"] = round(total_flops / elapsed / 1e12 / world / MFMA_BF16_PEAK_TFLOPS, 4)
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only (the other ranks would sit in the barrier for its 15 s)
            result["cpu_baseline"] = cpu_baseline(model, N)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
