#!/usr/bin/env python3
"""Headline benchmark: mel-frames/s of the F5-TTS flow-matching sampler (CFM.sample: 32 Euler steps x (cond + uncond) DiT
evaluations, CFG 2, sway -1) on MI355X, at BASELINE.json's C2: F5TTS_Base, bf16, batch 32, seq_len 1024, NFE 32.

  python bench.py --gpus N --steps K --warmup W [--scaling strong|weak] [--workload C2|C4|vocos]

A "step" is one full sample() over one synthetic batch of fixed-length utterances that is already resident in HBM.

Multi-GPU (--gpus N > 1): one process per GPU over RCCL (torch.distributed backend "nccl").  Started under torchrun (the driver's
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`) the ranks are used as given; started as a plain
`python bench.py --gpus N`, the parent process -- before it makes ANY GPU call -- launches that same torchrun command as a child
process, relays rank 0's JSON line and exits with the child's code.  A WORLD_SIZE that contradicts --gpus is an error.
  --scaling strong (default, C3 of SURVEY.md 8d): the SAME 32 utterances are split contiguously over the ranks (32/N per GPU, the
      reference's eval_infer_batch.py:163 split) and the finished mels are all-gathered inside the timed region;
  --scaling weak: 32 utterances per GPU.
There is no collective inside the ODE loop; the all_gather of the finished mels is the only one.

One JSON line on rank 0 with the driver's contract fields plus
  "roofline"      the dominant kernel (fused QKV projection GEMM): algorithmic FLOPs per launch / mean launch time measured here
                  with HIP event pairs on the launch stream, vs the 2.5 PFLOP/s dense bf16 MFMA peak; "kernels" lists the same
                  in-situ measurement for every kernel of a DiT block (GEMMs and attention against MFMA peak, the LayerNorm passes
                  against the 8 TB/s HBM peak); "traffic" is read from the newest profiles/*traffic*.json (a recorded rocprofv3 PMC
                  pass, named in "traffic_source"), null when there is none for this shape;
  "cpu_baseline"  the CPU oracle (oracle/cpu_ref.py, plain fp32 torch; the reference itself cannot travel) timed on this
                  host's cores on a bounded sample, scaled linearly to the workload's NFE.
  --workload vocos (C5): Vocos.decode on [B, 100, T] (T = 683 at B = 32, or --seq-len T), frames/s, HBM roofline of the ISTFT head.
The default C2 run on one GPU also measures C4 (2 timed sample() calls + in-situ kernel table) and C5 afterwards and reports them under
"workloads": {"C4": {value, ms_per_step, attention_frac, ...}, "C5": {value, rtf, head_gbs, cpu_baseline, ...}} (--no-extra skips them);
the C2 fields are unchanged.
"""
from __future__ import annotations

import argparse
import ctypes as C
import glob
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BASE_ARCH = dict(dim=1024, depth=22, heads=16, ff_mult=2, text_dim=512, text_mask_padding=False, conv_layers=4, pe_attn_head=1)
VOCAB = 2545
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3    # fp32-input MFMA (the vocoder backbone)
HBM_PEAK_GBS = 8000.0           # HBM3E spec (6.3 TB/s achievable, same guide)
WORKLOADS = {"C2": (32, 1024), "C4": (8, 4096)}


def synth_weights(model, seed=0):
    """Random-init F5TTS_Base (no checkpoint exists offline); zero-initialised tensors re-randomised (sigma 0.02) so the
    network is not degenerate."""
    import torch
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for _, p in model.named_parameters():
            if torch.count_nonzero(p) == 0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    return model


def synth_batch(B, N, device, seed=0):
    """cond mel ~ N(-3, 2^2) clipped to [ln 1e-5, 3], N_ref = N//3, text ids uniform, length N//6."""
    import torch
    g = torch.Generator().manual_seed(seed)
    n_ref = N // 3
    cond = (torch.randn(B, n_ref, 100, generator=g) * 2 - 3).clamp(math.log(1e-5), 3.0)
    text = torch.randint(0, VOCAB, (B, N // 6), generator=g)
    lens = torch.full((B,), n_ref, dtype=torch.long)
    duration = torch.full((B,), N, dtype=torch.long)
    return cond.to(device), text.to(device), lens.to(device), duration.to(device)


def _host_cores():
    import torch
    # the GPU box exposes every host core but a 1-GPU job owns a 16-core share; oversubscribing torch's pool is far slower
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(16, avail))
    torch.set_num_threads(cores)
    return cores


def cpu_baseline(model, N, nfe_full):
    """CPU oracle on the host cores: B=2 (B=1 for the long form), N, NFE=2 (NFE=1), scaled linearly in NFE."""
    import torch
    from oracle import cpu_ref  # checker, used here only as the timed CPU baseline
    cores = _host_cores()
    W = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    n_ref = N // 3
    bs, nfe = (2, 2) if N <= 1024 else (1, 1)
    cond = (torch.randn(bs, n_ref, 100, generator=g) * 2 - 3).clamp(math.log(1e-5), 3.0)
    text = torch.randint(0, VOCAB, (bs, N // 6), generator=g)
    t0 = time.perf_counter()
    cpu_ref.sample(W, BASE_ARCH, cond, text, N, steps=nfe, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0, return_trajectory=False)
    dt = time.perf_counter() - t0
    value = bs * N / (dt * nfe_full / nfe)
    return {"value": round(value, 3), "unit": "mel-frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle/cpu_ref.sample fp32 torch, B={bs} N={N} NFE={nfe} CFG=2 ({2 * nfe} network evaluations of {bs} utterances, "
                      f"{dt:.1f} s), scaled linearly to NFE={nfe_full}"}


def recorded_traffic(kernel_key, rows, N):
    """HBM-side bytes per launch of `kernel_key` from the newest profiles/*traffic*.json (written by tools/pmc_traffic.py from separate
    rocprofv3 --pmc passes: FETCH_SIZE x 2 on gfx950 + WRITE_SIZE).  Returns (bytes | None, source | None)."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json")), reverse=True):
        try:
            with open(path) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            continue
        e = rec.get(kernel_key)
        if e and e.get("rows") == rows and e.get("seq_len") == N:
            return int(e["bytes_per_launch"]), os.path.relpath(path, ROOT)
    return None, None


def launch_children(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a torchrun child BEFORE this process touches the GPU."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        if out.lstrip().startswith('{"metric"'):
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the ranks exited without a result line\n")
        rc = 1
    if line is not None:
        print(line, flush=True)
    sys.exit(rc)


def bench_vocos(args, dev, T=None, B=None, steps=None, warmup=None):
    """C5: Vocos.decode on the generated part of a C2 (T = 683) or C4 (T = 2731) batch.  Returns the result record."""
    import torch
    from eraxvif5tts_amd import _lib
    from eraxvif5tts_amd.vocos import Vocos
    from oracle import cpu_ref
    lib = _lib.load()
    T = T or (args.seq_len if args.seq_len else 683)
    B = B or (args.batch if args.batch else (32 if T <= 1024 else 8))
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    V = cpu_ref.random_vocos_weights(seed=3)  # random init of the vocos-mel-24khz shape (no checkpoint offline)
    voc = Vocos()
    voc.load_state_dict({k: t for k, t in V.items() if k in voc.state_dict()}, strict=False)
    voc = voc.to(dev)
    g = torch.Generator().manual_seed(0)
    mel = (torch.randn(B, 100, T, generator=g) * 2 - 3).clamp(math.log(1e-5), 3.0).to(dev)
    for _ in range(max(warmup, 1)):
        wave = voc.decode(mel)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        wave = voc.decode(mel)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert torch.isfinite(wave).all()
    frames = B * T * steps
    audio_s = B * (T - 1) * 256 / 24000.0 * steps
    # ISTFT head alone (HBM-bound): HIP events around f5_vocoder_istft_head on this stream
    head = torch.randn(B, T, 1026, generator=g).to(dev) * 0.5
    hw = voc.istft_head(head)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 20
    e0.record()
    for _ in range(iters):
        hw = voc.istft_head(head)
    e1.record()
    torch.cuda.synchronize()
    head_ms = e0.elapsed_time(e1) / iters
    head_bytes = B * T * (1026 * 4 + 256 * 4)  # read 1026 coefficients + write 256 samples per frame = 5.1 KB
    achieved = head_bytes / (head_ms * 1e-3) / 1e9
    flops = 2.0 * B * T * (7 * 100 * 512 + 8 * (7 * 512 + 2 * 512 * 1536) + 512 * 1026)
    result = {
        "metric": "vocoder mel-frames/s", "value": round(frames / elapsed, 2), "unit": "mel-frames/s", "n_gpus": 1, "steps": steps,
        "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic", "rtf": round(elapsed / audio_s, 7),
        "config": {"workload": f"C5 Vocos.decode (vocos-mel-24khz shape, random init): mel [B={B}, 100, T={T}] -> wave [B, {(T - 1) * 256}]",
                   "global_batch": B, "frames": T},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": None, "kernel": "ISTFT head (exp/clip/cos/sin -> 1024-point inverse FFT in LDS -> window; overlap-add)",
                     "launch": f"{head_bytes / 1e6:.1f} MB algorithmic (5.1 KB per frame) in {head_ms:.4f} ms, mean of {iters} calls (HIP events)",
                     "backbone_tflops_f32": round(flops * steps / elapsed / 1e12, 2), "backbone_peak_f32": MFMA_F32_PEAK_TFLOPS},
    }
    if not args.no_cpu_baseline:
        cores = _host_cores()
        bs = 2 if T <= 1024 else 1
        cm = mel[:bs].cpu()
        t0 = time.perf_counter()
        cpu_ref.vocos_decode(V, cm)
        dt = time.perf_counter() - t0
        result["cpu_baseline"] = {"value": round(bs * T / dt, 2), "unit": "mel-frames/s", "cores": cores, "kind": "port",
                                  "sample": f"oracle/cpu_ref.vocos_decode fp32 torch, B={bs} T={T} ({dt:.2f} s)"}
    del voc
    torch.cuda.empty_cache()
    return result


def insitu_kernels(model, cfm, batch, B, N, nfe, args):
    """One extra EAGER sample() with a HIP event pair around every block kernel (7 per block + 3 per evaluation) on the stream the kernels
    run on -- the same launches rocprofv3 averages in profiles/.  Returns (mean ms of the fused QKV GEMM, its launch count, its FLOP per
    launch, the per-kernel table)."""
    from eraxvif5tts_amd import _lib
    lib = _lib.load()
    cond, text, lens, duration = batch
    cfg_on = args.cfg >= 1e-5
    rows = (2 if cfg_on else 1) * B * N
    inner, D, ff, depth = 16 * 64, 1024, 2048, BASE_ARCH["depth"]
    plan = model.plan(B, N, nfe)
    ms, cnt = C.c_float(0.0), C.c_int(0)
    _lib.check(lib.f5_plan_timing_begin(plan, (7 * depth + 4) * nfe), "timing_begin")
    cfm.sample(cond=cond, text=text, duration=duration, lens=lens, steps=nfe, cfg_strength=args.cfg, sway_sampling_coef=-1.0, seed=0,
               return_trajectory=False, use_graph=False)
    _lib.check(lib.f5_plan_timing_end(plan, C.byref(ms), C.byref(cnt), _lib.stream_ptr()), "timing_end")
    flops_qkv = 2.0 * rows * (3 * inner) * D

    def site(i):
        a, n = C.c_float(0.0), C.c_int(0)
        _lib.check(lib.f5_plan_timing_site(plan, i, C.byref(a), C.byref(n)), "timing_site")
        return a.value, n.value
    fold, w4 = C.c_int(0), C.c_int(0)
    _lib.check(lib.f5_plan_get_option(plan, b"ln_fold_active", C.byref(fold)), "plan_get_option")
    _lib.check(lib.f5_plan_get_option(plan, b"gemm_w4", C.byref(w4)), "plan_get_option")
    # (csrc/gemm_w4.hip, w4_tile_rows: whole 256- or 128-row tiles, at least one per CU -- the fused projection has 12 feature tiles per token tile)
    insitu_kernels.w4 = bool(w4.value) and args.precision == "bf16" and ((rows % 256 == 0 and (rows // 256) * 12 * 4 >= 3 * 256) or
                                                                         (rows % 128 == 0 and (rows // 128) * 12 * 2 >= 256))
    es = 2 if args.precision == "bf16" else 4
    xs = es  # storage bytes of a residual-stream element: fp16 in the bf16 production mode (blocks 1..21 of 22; the first reads fp32), fp32 otherwise
    work = {  # algorithmic work per launch: FLOP for the MFMA-bound kernels, bytes for the HBM-bound ones
        "qkv": ("mfma", flops_qkv), "attention": ("mfma", 4.0 * rows * N * inner), "attn_out": ("mfma", 2.0 * rows * D * inner),
        "ff1": ("mfma", 2.0 * rows * ff * D), "ff2": ("mfma", 2.0 * rows * D * ff), "conv31": ("mfma", 2.0 * rows * D * 64 * 31),
        # input projection W_x . x (K = 100 padded to 128): 0.2 FLOP per byte -- HBM-bound: reads the noisy mel rows (bf16, 128 wide) and the
        # hoisted part of the embedding, writes the GEMM operand (bf16) and the residual stream
        "input_proj": ("hbm", rows * (128 * es // 2 + D * (xs + es + xs))),
        # bf16 mode: the fp16 stream is updated in place by the out-projection / FF2 epilogues, the LayerNorm passes read it and write the
        # normalised rows; fp32 mode: the passes also read the stored branches and write the stream
        "ln1": ("hbm", rows * D * ((xs + es) if args.precision == "bf16" else (xs + es + es))),
        "ln2": ("hbm", rows * D * ((xs + es) if args.precision == "bf16" else (xs + es + es + xs + es))),
    }
    if fold.value:
        # LayerNorm fold (round 4): the two passes of a block are gone; what runs at these sites is stats_finalize_kernel -- per token row it reads the
        # D / 64 partial sums (float2) the in-place residual epilogue in front wrote and the row's pivot, and writes (mean, rstd).  Block 0's first
        # site is still a LayerNorm pass (1 of its 22 launches per evaluation); it is averaged in with its own byte count.
        fin = rows * ((D // 64) * 8 + 8 + 8)
        _, n_ln1 = site(_lib.SITES.index("ln1"))
        if n_ln1 > nfe:  # statistics launches ran at the site (not only block 0's pass)
            work["ln1"] = ("hbm", (fin * (depth - 1) + work["ln1"][1]) / depth)
        work["ln2"] = ("hbm", fin)
    kernels = []
    for i, name in enumerate(_lib.SITES):
        t_ms, n = site(i)
        if n == 0 or t_ms <= 0:
            continue
        bound, w = work[name]
        if bound == "mfma":
            a = w / (t_ms * 1e-3) / 1e12
            kernels.append({"kernel": name, "bound": "mfma", "work": round(w / 1e9, 2), "work_unit": "GFLOP", "ms": round(t_ms, 4),
                            "launches": n, "achieved": round(a, 1), "unit": "TFLOP/s", "frac": round(a / MFMA_BF16_PEAK_TFLOPS, 4)})
        else:
            a = w / (t_ms * 1e-3) / 1e9
            if fold.value and name in ("ln1", "ln2"):
                name += " (LayerNorm folded: row-statistics finalize)"
            kernels.append({"kernel": name, "bound": "hbm", "work": round(w / 1e6, 1), "work_unit": "MB", "ms": round(t_ms, 4),
                            "launches": n, "achieved": round(a, 1), "unit": "GB/s", "frac": round(a / HBM_PEAK_GBS, 4)})
    return ms.value, cnt.value, flops_qkv, kernels


def bucketed_workloads(args, dev, cfm, nfe):
    """Batched inference over length buckets (the reference's production-shaped multi-utterance caller: eval/utils_eval.py:72-204 builds
    frame-budgeted buckets, eval_infer_batch.py:160-196 samples each as ONE padded + masked batch) against the ragged sampler, on a synthetic set
    with a stated length distribution."""
    import torch
    out = {}
    try:
        from eraxvif5tts_amd.eval import prompts as P
        meta = P.synthetic_metainfo(48, seed=1, min_secs=3.0, max_secs=20.0)
        # two bucketings of the same set: fine (40 length buckets, >= 6 000 frames per batch: batches of 2 - 6 utterances of nearly equal length, 2 % of the
        # padded rows are padding) and coarse (4 buckets, >= 16 000 frames: bigger batches, mixed lengths -- where padding costs and the ragged form pays)
        for key, nb, budget in (("bucketed_eval", 40, 6000), ("bucketed_eval_coarse", 4, 16000)):
            buckets = P.get_inference_prompt(meta, tokenizer="char", infer_batch_size=budget, num_buckets=nb, min_secs=3, max_secs=40, device=dev)
            frames = sum(sum(b[4]) for b in buckets)
            padded_rows = sum(len(b[4]) * max(b[4]) for b in buckets)
            bkw = dict(nfe_step=nfe, cfg_strength=args.cfg, sway_sampling_coef=-1.0, seed=0)
            times = {}
            for mode in ("padded", "ragged"):
                for _pass in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    n_out = sum(1 for _ in P.infer_prompts(cfm, buckets, mode=mode, **bkw))
                    torch.cuda.synchronize()
                    times[mode] = time.perf_counter() - t0
                assert n_out == len(meta)
            out[key] = {"value": round(frames / times["ragged"], 2), "unit": "mel-frames/s", "ms_per_step": round(times["ragged"] * 1e3, 3),
                        "padded_value": round(frames / times["padded"], 2), "padded_ms": round(times["padded"] * 1e3, 3),
                        "speedup_vs_padded": round(times["padded"] / times["ragged"], 3), "utterances": len(meta), "buckets": len(buckets),
                        "frames": int(frames), "padded_rows": int(padded_rows),
                        "config": f"48 synthetic utterances, total length U(3, 20) s, prompt U(2, 6) s, {len(buckets)} batches from {nb} length buckets of >= {budget} "
                                  f"frames (utils_eval.get_inference_prompt), NFE={nfe} CFG={args.cfg:g} sway=-1, bf16; value = ragged sampler, padded_value = the "
                                  "reference's padded + masked batches; third pass of each (hipGraph replay of the recurring shapes)"}
            del buckets
        del meta
        torch.cuda.empty_cache()

    except Exception as e:  # noqa: BLE001
        import traceback
        traceback.print_exc()
        out["bucketed_eval_error"] = f"{type(e).__name__}: {e}"
    return out


def extra_workloads(args, dev, model, cfm):
    """The other single-GPU configurations of BASELINE.json, measured after the C2 line so that the driver's default run records them too:
    C4 (long form, 8 x 4096: 2 timed sample() calls + the in-situ kernel table), the 4 x 1024 shard shape of the 8-GPU run and the
    single-utterance shape (5 timed sample() calls each), and C5 (Vocos.decode on the generated part of C2)."""
    import torch
    out = {}
    B, N, nfe = WORKLOADS["C4"][0], WORKLOADS["C4"][1], args.nfe
    batch = synth_batch(B, N, dev, seed=0)
    cond, text, lens, duration = batch
    kw = dict(cond=cond, text=text, duration=duration, lens=lens, steps=nfe, cfg_strength=args.cfg, sway_sampling_coef=-1.0, seed=0,
              return_trajectory=False, use_graph=not args.no_graph)
    cfm.sample(**kw)  # warm-up (captures the graph)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 2
    for _ in range(steps):
        o, _ = cfm.sample(**kw)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    assert torch.isfinite(o).all()
    _, _, _, kernels = insitu_kernels(model, cfm, batch, B, N, nfe, args)
    attn = next((k for k in kernels if k["kernel"] == "attention"), None)
    per_token = 378.9e6 + 90112.0 * N
    flops = per_token * B * N * (2 if args.cfg >= 1e-5 else 1) * nfe * steps
    out["C4"] = {"value": round(B * N * steps / el, 2), "unit": "mel-frames/s", "ms_per_step": round(el / steps * 1e3, 3), "steps": steps,
                 "rtf": round(el / (B * (N - N // 3) * 256 / 24000.0 * steps), 6),
                 "config": f"C4 long form: {B} utterances x seq_len {N} (N_ref={N // 3}), NFE={nfe} CFG={args.cfg:g} sway=-1, bf16, hipGraph",
                 "attention_frac": attn["frac"] if attn else None, "attention_ms": attn["ms"] if attn else None,
                 "loop_mfma_frac": round(flops / el / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "kernels": kernels}
    del batch, cond, text, lens, duration, o
    torch.cuda.empty_cache()
    # the two small shapes the other claims rest on: the per-GPU shard of the 8-GPU strong-scaling run (4 x 1024) and one utterance (1 x 1024:
    # what F5TTSWrapper.generate() pays per text chunk when it cannot batch them)
    for key, b in (("shard_4x1024", 4), ("single_utterance", 1)):
        cond, text, lens, duration = synth_batch(b, 1024, dev, seed=0)
        kw = dict(cond=cond, text=text, duration=duration, lens=lens, steps=nfe, cfg_strength=args.cfg, sway_sampling_coef=-1.0, seed=0,
                  return_trajectory=False, use_graph=not args.no_graph)
        cfm.sample(**kw)
        cfm.sample(**kw)  # (the second call of a shape is the one that captures the graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            o, _ = cfm.sample(**kw)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / n
        assert torch.isfinite(o).all()
        out[key] = {"value": round(b * 1024 / el, 2), "unit": "mel-frames/s", "ms_per_step": round(el * 1e3, 3),
                    "rtf": round(el / (b * (1024 - 1024 // 3) * 256 / 24000.0), 6),
                    "config": f"{b} utterance(s) x seq_len 1024 (N_ref=341), NFE={nfe} CFG={args.cfg:g} sway=-1, bf16, hipGraph"}
        del cond, text, lens, duration, o
    torch.cuda.empty_cache()
    # the text chunks of one F5TTSWrapper.generate() call (reference infer/f5tts_wrapper.py:476-533 samples them one after the other): four
    # utterances of different lengths over one prompt, serial batch-1 calls against ONE ragged batch (f5_sample_ragged; bit-identical mels)
    g = torch.Generator().manual_seed(31)
    cond1 = (torch.randn(1, 300, 100, generator=g) * 2 - 3).clamp(math.log(1e-5), 3.0).to(dev)
    durs = [760, 1010, 900, 1180]
    texts = [torch.randint(0, VOCAB, (1, d // 7), generator=g).to(dev) for d in durs]
    y0s = [torch.randn(1, d, 100, generator=g).to(dev) for d in durs]
    skw = dict(steps=nfe, cfg_strength=args.cfg, sway_sampling_coef=-1.0)

    def serial():
        return [cfm.sample(cond=cond1, text=t, duration=d, y0=y, return_trajectory=False, use_graph=False, **skw)[0] for t, d, y in zip(texts, durs, y0s)]

    def ragged():
        return cfm.sample_ragged(cond1, texts, durs, y0s=y0s, **skw)

    ref, got = serial(), ragged()
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(got, ref))
    times = {}
    for name, fn in (("serial", serial), ("ragged", ragged), ("serial", serial), ("ragged", ragged)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        times.setdefault(name, []).append(time.perf_counter() - t0)
    ts, tr = min(times["serial"]), min(times["ragged"])
    out["generate_4_chunks"] = {"value": round(sum(durs) / tr, 2), "unit": "mel-frames/s", "ms_per_step": round(tr * 1e3, 3),
                                "serial_ms": round(ts * 1e3, 3), "speedup_vs_serial": round(ts / tr, 3), "bit_identical_to_serial": bool(same),
                                "config": f"4 utterances of {durs} frames over one 300-frame prompt, NFE={nfe} CFG={args.cfg:g} sway=-1, bf16, eager launches"}
    del ref, got, cond1, texts, y0s
    torch.cuda.empty_cache()
    # Batched inference over length buckets (the reference's production-shaped multi-utterance caller: eval/utils_eval.py:72-204 builds
    # frame-budgeted buckets, eval_infer_batch.py:160-196 samples each as ONE padded + masked batch).  Synthetic set with a stated length
    # distribution (no dataset offline): 48 utterances, total length uniform in 3..20 s, prompt 2..6 s; buckets of >= 6 000 frames.  Padded
    # form (the reference's) against the ragged sampler (no padding, no key mask, a prompt per utterance); third pass of each (the second
    # pass of a shape captures its hipGraph, the third replays it).
    out.update(bucketed_workloads(args, dev, cfm, nfe))
    try:  # the other vocoder of plug point B (parity unpinned, not tuned: DESIGN section 9): one 683-frame utterance through the BigVGAN-v2 generator
        from eraxvif5tts_amd.bigvgan import BigVGAN
        from oracle import cpu_ref as _cr  # (only its seeded weight generator: random init of the published shapes, no checkpoint offline)
        bw = _cr.random_bigvgan_weights(_cr.BIGVGAN_V2_24K_100BAND_256X, seed=1)
        bw["conv_post.weight"] = bw["conv_post.weight"] * 0.0015
        voc = BigVGAN()
        voc.load_state_dict(bw)
        voc = voc.eval().to(dev)
        bmel = (torch.randn(1, 100, 683, generator=torch.Generator().manual_seed(2)) * 2 - 3).clamp(math.log(1e-5), 3.0).to(dev)
        w = voc(bmel)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            w = voc(bmel)
        torch.cuda.synchronize()
        bms = (time.perf_counter() - t0) / 5 * 1e3
        assert torch.isfinite(w).all()
        out["bigvgan_decode"] = {"value": round(683 / bms * 1e3, 2), "unit": "mel-frames/s", "ms_per_step": round(bms, 3),
                                 "rtf": round(bms / 1e3 / (683 * 256 / 24000.0), 6),
                                 "config": "BigVGAN-v2 generator (bigvgan_v2_24khz_100band_256x shape, 112.4 M parameters, random init), mel [1, 100, 683] -> "
                                           "wave [1, 1, 174848], fp32-input MFMA; parity unpinned (source absent from the reference tree)"}
        del voc, bw, w, bmel
        torch.cuda.empty_cache()
    except Exception as e:  # noqa: BLE001
        out["bigvgan_decode_error"] = f"{type(e).__name__}: {e}"
    v = bench_vocos(args, dev, T=683, B=32, steps=20, warmup=3)
    out["C5"] = {"value": v["value"], "unit": v["unit"], "ms_per_step": v["ms_per_step"], "rtf": v["rtf"], "config": v["config"]["workload"],
                 "head_gbs": v["roofline"]["achieved"], "head_frac": v["roofline"]["frac"],
                 "backbone_tflops_f32": v["roofline"]["backbone_tflops_f32"], "cpu_baseline": v.get("cpu_baseline")}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C2", choices=["C2", "C4", "vocos"])
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"])
    ap.add_argument("--batch", type=int, default=0, help="utterances (per GPU with --scaling weak, in total with strong); 0 = the workload's")
    ap.add_argument("--seq-len", type=int, default=0)
    ap.add_argument("--nfe", type=int, default=32)
    ap.add_argument("--cfg", type=float, default=2.0)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="C2 run: skip the C4 / C5 measurements recorded under `workloads`")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus is None:
        args.gpus = int(env_world) if env_world else 1
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if env_world is None and args.gpus > 1:
        launch_children(args)  # never returns; nothing in this process has touched the GPU
    world = int(env_world or "1")
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} contradicts WORLD_SIZE={world} (launch with --nproc-per-node {args.gpus}, or drop --gpus)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    backend = "none"
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
    from eraxvif5tts_amd.eval.sharded import split_between_processes
    from eraxvif5tts_amd.model import CFM, DiT
    _lib.require_gpu()
    lib = _lib.load()

    if args.workload == "vocos":
        if world > 1:
            sys.exit("bench.py: --workload vocos is a single-GPU measurement")
        print(json.dumps(bench_vocos(args, dev)), flush=True)
        return

    wl_B, wl_N = WORKLOADS[args.workload]
    N, nfe = args.seq_len or wl_N, args.nfe
    B_req = args.batch or wl_B
    if args.scaling == "strong" and world > 1:
        # C3: the same B_req utterances, split contiguously over the ranks as eval_infer_batch.py:163 splits its list
        mine = split_between_processes(list(range(B_req)), rank, world)
        B, B_total = len(mine), B_req
        if B == 0:
            sys.exit(f"bench.py: {B_req} utterances cannot feed {world} ranks")
    else:
        mine = list(range(rank * B_req, (rank + 1) * B_req))
        B, B_total = B_req, B_req * world
    model = synth_weights(DiT(**BASE_ARCH, text_num_embeds=VOCAB, mel_dim=100, precision=args.precision))
    cfm = CFM(transformer=model, mel_spec_kwargs={"mel_spec_type": "vocos"}, odeint_kwargs={"method": "euler"}).to(dev)
    # one global synthetic batch; every rank keeps its contiguous share (identical utterances whatever the rank count)
    cond, text, lens, duration = [t[mine[0]:mine[-1] + 1].contiguous() for t in synth_batch(B_total, N, dev, seed=0)]

    bmax = -(-B_total // world)
    gathered = [torch.zeros(bmax, N, 100, device=dev) for _ in range(world)] if world > 1 else None
    pad = torch.zeros(bmax, N, 100, device=dev) if world > 1 else None

    # N > 1: HIP events (on the stream the sampler's launches and the collective are enqueued on) split every step into this rank's
    # sample() and the all_gather behind it, so that one record separates per-GPU batch efficiency, rank skew and gather time
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)] if world > 1 else None
    tick = {"i": -1}

    def step():
        e = ev[tick["i"]] if (world > 1 and 0 <= tick["i"] < args.steps) else None
        if e:
            e[0].record()
        out, _ = cfm.sample(cond=cond, text=text, duration=duration, lens=lens, steps=nfe, cfg_strength=args.cfg,
                            sway_sampling_coef=-1.0, seed=0, return_trajectory=False, use_graph=not args.no_graph)
        if world > 1:
            if e:
                e[1].record()
            pad[:B].copy_(out)
            dist.all_gather(gathered, pad)  # the only collective of the path: finished mels over RCCL/xGMI
            if e:
                e[2].record()
        return out

    debug = bool(os.environ.get("F5_BENCH_DEBUG"))
    if debug:
        import warnings
        warnings.simplefilter("always")
    for i in range(args.warmup):
        step()
        if debug:
            sys.stderr.write(f"[debug] warm-up {i}: residual fallbacks so far {model.residual_fallbacks()}\n")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        tick["i"] = i
        out = step()
        if debug:
            sys.stderr.write(f"[debug] step {i}: residual fallbacks so far {model.residual_fallbacks()}\n")
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0  # this rank's own work, before it waits for the others
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    devices = [torch.cuda.get_device_name(dev)]
    per_rank = None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        names = [None] * world
        dist.all_gather_object(names, f"rank{rank}:cuda{torch.cuda.current_device()}:{devices[0]}")
        devices = names
        # after the timed region: the collective alone, every rank entering it together (no skew inside the measurement)
        tick["i"] = -1
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.barrier()
        torch.cuda.synchronize()
        g0.record()
        for _ in range(5):
            dist.all_gather(gathered, pad)
        g1.record()
        torch.cuda.synchronize()
        mine_rec = {"rank": rank, "utterances": B, "elapsed_s": round(own_elapsed, 6),
                    "sample_ms": round(sum(e[0].elapsed_time(e[1]) for e in ev) / args.steps, 3),
                    "gather_ms_in_step": round(sum(e[1].elapsed_time(e[2]) for e in ev) / args.steps, 3),  # includes waiting for slower ranks
                    "gather_only_ms": round(g0.elapsed_time(g1) / 5, 3),
                    "mel_frames_per_s": round(B * N * args.steps / max(own_elapsed, 1e-9), 2)}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine_rec)
    assert torch.isfinite(out).all(), "non-finite mel output"

    frames = B_total * N * args.steps
    value = frames / elapsed
    gen_audio_s = B_total * (N - N // 3) * 256 / 24000.0 * args.steps
    result = {
        "metric": "mel-frames/s", "value": round(value, 2), "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": args.scaling,  # "strong" (default): the same 32 utterances whatever the rank count; "weak": 32 per GPU
        "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "rtf": round(elapsed / gen_audio_s, 6),
        "config": {"workload": f"{args.workload if (N, B_req) == (wl_N, wl_B) else 'custom'} F5TTS_Base random-init, CFM.sample euler NFE={nfe} "
                               f"CFG={args.cfg:g} sway=-1, {B_total} utterances x seq_len {N} (N_ref={N // 3}), "
                               f"{'split ' + str(B) + ' per GPU' if world > 1 else 'one GPU'}, hipGraph={'off' if args.no_graph else 'on'}",
                   "global_batch": B_total, "per_gpu_batch": B, "seq_len": N, "nfe": nfe, "parallelism": f"utterance-sharded dp{world}",
                   "key_mask": "all-true (fixed-length utterances): CFM.sample selects the unmasked kernels, same values as the reference's masked path",
                   "residual_stream": "fp16 storage, fp32 arithmetic" if args.precision == "bf16" else "fp32"},
        "distributed": {"world_size": dist.get_world_size() if world > 1 else 1, "backend": backend, "devices": devices,
                        # per rank: its own wall time, HIP-event time of sample() and of the all_gather per step, the collective alone
                        "ranks": per_rank, "gather_payload_bytes": (bmax * N * 100 * 4) if world > 1 else 0},
    }

    if rank == 0:
        cfg_on = args.cfg >= 1e-5
        rows = (2 if cfg_on else 1) * B * N
        ms_qkv, n_qkv, flops_qkv, kernels = insitu_kernels(model, cfm, (cond, text, lens, duration), B, N, nfe, args)
        achieved = flops_qkv / (ms_qkv * 1e-3) / 1e12
        traffic, traffic_src = recorded_traffic("qkv", rows, N)
        result["roofline"] = {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4),
                              "traffic": traffic, "traffic_source": traffic_src,
                              "kernel": (("gemm_w4_kernel<QKV+RoPE epilogue" if getattr(insitu_kernels, "w4", False) else "gemm_fast_kernel<256,128,DENSE,QKV+RoPE epilogue") +
                                         (",persistent grid,LayerNorm fold> (fp16 stream x per-time fp16 weights on v_mfma_f32_16x16x32_f16; 21 of the 22 launches per "
                                          "evaluation, block 0's runs the bf16 build)" if any("LayerNorm folded" in k["kernel"] for k in kernels) else ",persistent grid>") +
                                         (" [one wave per SIMD: 4 waves x 128x128, 256x256x64 stages, csrc/gemm_w4.hip]" if getattr(insitu_kernels, "w4", False) else "")),
                              "launch": f"M={rows} N=3072 K=1024, {flops_qkv / 1e9:.1f} GFLOP, {ms_qkv:.4f} ms mean over {n_qkv} launches inside an "
                                        f"eager sample() (HIP event pairs on the launch stream)",
                              "kernels": kernels}
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
        total_flops = per_token * B_total * N * (2 if cfg_on else 1) * nfe * args.steps
        result["loop_tflops"] = round(total_flops / elapsed / 1e12 / world, 2)
        result["loop_mfma_frac"] = round(total_flops / elapsed / 1e12 / world / MFMA_BF16_PEAK_TFLOPS, 4)
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only (the other ranks would sit in the barrier for its 15 s)
            result["cpu_baseline"] = cpu_baseline(model, N, nfe)
        if world == 1 and args.workload == "C2" and not args.no_extra and (N, B_req) == (wl_N, wl_B) and args.precision == "bf16":
            del cond, text, lens, duration
            try:
                result["workloads"] = extra_workloads(args, dev, model, cfm)
            except Exception as e:  # noqa: BLE001  (the C2 line above is complete; a failing side workload must not take it down)
                import traceback
                traceback.print_exc()
                result["workloads"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
