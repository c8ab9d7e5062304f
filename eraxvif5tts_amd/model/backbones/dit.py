"""DiT backbone (plug point A of the reference) executed by libf5hip on MI355X.

Drop-in for ``f5_tts.model.backbones.dit.DiT`` (reference dit.py:103-233): same constructor kwargs, same
``state_dict()`` names/shapes (so existing ``.pt`` / ``.safetensors`` checkpoints load with ``load_state_dict``),
same ``forward(x, cond, text, time, drop_audio_cond, drop_text, mask=None, cache=False)``, ``clear_cache()`` and ``.dim``.

The torch parameters are only the checkpoint-facing copy of the weights; all arithmetic happens in the HIP library
(``include/f5hip.h``).  Nothing here computes the network in PyTorch: without the built library or without an MI355X
``forward`` raises.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import torch
from torch import nn

from .. import utils as _utils  # noqa: F401  (kept importable like the reference package layout)
from ... import _lib


def _param_spec(dim, depth, heads, dim_head, ff_inner, mel_dim, vocab, text_dim, conv_layers):
    """(name, shape, init) for every tensor of the reference DiT.state_dict()."""
    inner = heads * dim_head
    spec = [("time_embed.time_mlp.0.weight", (dim, 256), "linear"), ("time_embed.time_mlp.0.bias", (dim,), ("bias", 256)),
            ("time_embed.time_mlp.2.weight", (dim, dim), "linear"), ("time_embed.time_mlp.2.bias", (dim,), ("bias", dim)),
            ("text_embed.text_embed.weight", (vocab + 1, text_dim), "normal")]
    for i in range(conv_layers):
        p = f"text_embed.text_blocks.{i}."
        spec += [(p + "dwconv.weight", (text_dim, 1, 7), "linear"), (p + "dwconv.bias", (text_dim,), ("bias", 7)),
                 (p + "norm.weight", (text_dim,), "ones"), (p + "norm.bias", (text_dim,), "zeros"),
                 (p + "pwconv1.weight", (2 * text_dim, text_dim), "linear"), (p + "pwconv1.bias", (2 * text_dim,), ("bias", text_dim)),
                 (p + "grn.gamma", (1, 1, 2 * text_dim), "zeros"), (p + "grn.beta", (1, 1, 2 * text_dim), "zeros"),
                 (p + "pwconv2.weight", (text_dim, 2 * text_dim), "linear"), (p + "pwconv2.bias", (text_dim,), ("bias", 2 * text_dim))]
    kin = 2 * mel_dim + text_dim
    spec += [("input_embed.proj.weight", (dim, kin), "linear"), ("input_embed.proj.bias", (dim,), ("bias", kin))]
    for i in (0, 2):
        spec += [(f"input_embed.conv_pos_embed.conv1d.{i}.weight", (dim, dim // 16, 31), "linear"),
                 (f"input_embed.conv_pos_embed.conv1d.{i}.bias", (dim,), ("bias", dim // 16 * 31))]
    for i in range(depth):
        p = f"transformer_blocks.{i}."
        spec += [(p + "attn_norm.linear.weight", (6 * dim, dim), "zeros"), (p + "attn_norm.linear.bias", (6 * dim,), "zeros")]
        for nm in ("to_q", "to_k", "to_v"):
            spec += [(p + f"attn.{nm}.weight", (inner, dim), "linear"), (p + f"attn.{nm}.bias", (inner,), ("bias", dim))]
        spec += [(p + "attn.to_out.0.weight", (dim, inner), "linear"), (p + "attn.to_out.0.bias", (dim,), ("bias", inner)),
                 (p + "ff.ff.0.0.weight", (ff_inner, dim), "linear"), (p + "ff.ff.0.0.bias", (ff_inner,), ("bias", dim)),
                 (p + "ff.ff.2.weight", (dim, ff_inner), "linear"), (p + "ff.ff.2.bias", (dim,), ("bias", ff_inner))]
    # zero-initialised output path, as the reference's initialize_weights (dit.py:162-172)
    spec += [("norm_out.linear.weight", (2 * dim, dim), "zeros"), ("norm_out.linear.bias", (2 * dim,), "zeros"),
             ("proj_out.weight", (mel_dim, dim), "zeros"), ("proj_out.bias", (mel_dim,), "zeros")]
    return spec


def _register(root: nn.Module, dotted: str, tensor: torch.Tensor, buffer=False):
    mod = root
    parts = dotted.split(".")
    for part in parts[:-1]:
        if part not in mod._modules:
            mod.add_module(part, nn.Module())
        mod = mod._modules[part]
    if buffer:
        mod.register_buffer(parts[-1], tensor)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


class DiT(nn.Module):
    BACKBONE = _lib.F5_BACKBONE_DIT

    def __init__(self, *, dim, depth=8, heads=8, dim_head=64, dropout=0.1, ff_mult=4, mel_dim=100, text_num_embeds=256,
                 text_dim=None, text_mask_padding=True, qk_norm=None, conv_layers=0, pe_attn_head=None,
                 long_skip_connection=False, checkpoint_activations=False, precision=None, rope_layout=None):
        super().__init__()
        self.long_skip = bool(long_skip_connection)  # dit.py:153: Linear(2 dim -> dim, no bias) on cat(x, input embedding) after the blocks
        self.checkpoint_activations = checkpoint_activations  # training-only knob; accepted and ignored
        self._setup(dim=dim, depth=depth, heads=heads, dim_head=dim_head, ff_mult=ff_mult, mel_dim=mel_dim, text_num_embeds=text_num_embeds,
                    text_dim=text_dim, text_mask_padding=text_mask_padding, qk_norm=qk_norm, conv_layers=conv_layers, pe_attn_head=pe_attn_head,
                    precision=precision, rope_layout=rope_layout)

    def _spec(self):
        spec = _param_spec(self.dim, self.depth, self.heads, self.dim_head, self.ff_inner, self.mel_dim, self.text_num_embeds, self.text_dim,
                           self.conv_layers)
        if self.qk_norm == "rms_norm":  # modules.py:394-396: RMSNorm(dim_head, eps=1e-6) on q and k
            for i in range(self.depth):
                spec += [(f"transformer_blocks.{i}.attn.q_norm.weight", (self.dim_head,), "ones"), (f"transformer_blocks.{i}.attn.k_norm.weight", (self.dim_head,), "ones")]
        if self.long_skip:
            spec += [("long_skip_connection.weight", (self.dim, 2 * self.dim), "linear")]
        return spec

    def _setup(self, *, dim, depth, heads, dim_head, ff_mult, mel_dim, text_num_embeds, text_dim, text_mask_padding, qk_norm, conv_layers,
               pe_attn_head, precision, rope_layout, skip_connect_type="concat"):
        if text_dim is None:
            text_dim = mel_dim
        if qk_norm not in (None, "rms_norm"):
            raise ValueError(f"Unimplemented qk_norm: {qk_norm}")  # modules.py:398
        if (qk_norm is not None or getattr(self, "long_skip", False)) and self.BACKBONE != _lib.F5_BACKBONE_DIT:
            raise NotImplementedError("qk_norm / long_skip_connection are built for the DiT backbone (null / False in every shipped config)")
        self.qk_norm = qk_norm
        self.dim, self.depth, self.heads, self.dim_head = dim, depth, heads, dim_head
        self.ff_inner = int(dim * ff_mult)
        self.mel_dim, self.text_num_embeds, self.text_dim = mel_dim, text_num_embeds, text_dim
        self.text_mask_padding, self.conv_layers, self.pe_attn_head = bool(text_mask_padding), conv_layers, pe_attn_head
        self.skip_connect_type = skip_connect_type
        prec = precision or os.environ.get("F5HIP_PRECISION", "bf16")
        self.precision = {"bf16": _lib.F5_PREC_BF16, "fp32": _lib.F5_PREC_FP32}[prec]
        # x_transformers (dit.py:16,134) is not vendored in the reference tree: "adjacent" rotates feature pairs (2j, 2j+1), the form the
        # pinned >= 1.31 releases publish (default); "half_split" rotates (j, j+32).  ONE switch, also settable as F5HIP_ROPE_LAYOUT.
        layout = rope_layout or os.environ.get("F5HIP_ROPE_LAYOUT", "adjacent")
        self.rope_layout = {"adjacent": _lib.F5_ROPE_ADJACENT, "half_split": _lib.F5_ROPE_HALF_SPLIT}[layout]

        for name, shape, init in self._spec():
            t = torch.empty(shape)
            if init == "zeros":
                t.zero_()
            elif init == "ones":
                t.fill_(1.0)
            elif init == "normal":
                t.normal_()
            elif init == "linear":  # torch's default Linear/Conv init: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
                bound = 1.0 / math.sqrt(math.prod(shape[1:]))
                t.uniform_(-bound, bound)
            else:
                bound = 1.0 / math.sqrt(init[1])
                t.uniform_(-bound, bound)
            _register(self, name, t)
        inv_freq = 1.0 / (10000.0 ** (torch.arange(0, dim_head, 2).float() / dim_head))
        _register(self, "rotary_embed.inv_freq", inv_freq, buffer=True)

        self.text_cond, self.text_uncond = None, None  # text cache (reference dit.py:131)
        self._native = None
        self._native_versions = None
        self._plans = []
        self._seen_shapes = {}
        self._fallbacks_seen = {}
        self._pending = {}  # plan handle -> stream of a sample() whose range-guard check was deferred (finish_pending)
        self.register_load_state_dict_post_hook(lambda module, _keys: module._drop_native())

    # ------------------------------------------------------------------ native handle management
    def _drop_native(self):
        lib = _lib.load()
        for _, h in self._plans:
            lib.f5_plan_destroy(h)
        self._plans = []
        self._pending = {}
        if self._native is not None:
            lib.f5_model_destroy(self._native)
            self._native = None
        self.clear_cache()

    def __del__(self):
        try:
            self._drop_native()
        except Exception:  # noqa: BLE001
            pass

    def _param_versions(self):
        return tuple(p._version for p in self.parameters())

    def refresh_native(self):
        """Drop the HBM weight snapshot, every plan and the cached AdaLN rows: the next forward()/sample() re-uploads the CURRENT
        parameter values.  load_state_dict() does this by itself, and so does native() when a parameter was modified in place
        (p.copy_, EMA swap: torch bumps the tensor's version counter); call it explicitly after edits torch cannot see (.data)."""
        self._drop_native()

    def native(self):
        """Upload the current parameter values to HBM in the kernels' layouts (once; redone after load_state_dict or an in-place
        parameter update)."""
        if self._native is not None and self._native_versions != self._param_versions():
            self._drop_native()
        if self._native is not None:
            return self._native
        _lib.require_gpu()
        lib = _lib.load()
        cfg = _lib.DitConfig(dim=self.dim, depth=self.depth, heads=self.heads, dim_head=self.dim_head, ff_inner=self.ff_inner,
                             mel_dim=self.mel_dim, text_num_embeds=self.text_num_embeds, text_dim=self.text_dim,
                             conv_layers=self.conv_layers, text_mask_padding=int(self.text_mask_padding),
                             pe_attn_head=self.pe_attn_head or 0, qk_norm=int(self.qk_norm == "rms_norm"),
                             long_skip=int(getattr(self, "long_skip", False)), precision=self.precision,
                             rope_layout=self.rope_layout, backbone=self.BACKBONE, skip_connect=_lib.F5_SKIP[self.skip_connect_type])
        h = C.c_void_p()
        _lib.check(lib.f5_model_create(C.byref(cfg), C.byref(h)), "model_create")
        try:
            _lib.set_tensors(h, "f5_model_set_tensor", "f5_model_has_tensor", self.state_dict())
            _lib.check(lib.f5_model_finalize(h), "model_finalize")
        except Exception:
            lib.f5_model_destroy(h)
            raise
        self._native = h
        self._native_versions = self._param_versions()
        return h

    def plan(self, batch, seq, evals=1):
        """Workspace for (batch, seq) problems ON THE CURRENT STREAM; reused while it is large enough.  A plan's buffers are ordered only by the
        stream its calls run on, so every stream gets plans of its own (F5TTSWrapper.generate samples its text chunks on several streams)."""
        self.native()  # (drops stale plans when a parameter changed in place)
        stream = torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else 0
        for (st, b, n, e), h in self._plans:
            if st == stream and b >= batch and n >= seq and e >= evals:
                self._finish_plan(h)  # a deferred sample() still owns this plan's staged inputs and flag words: complete it first
                return h
        lib = _lib.load()
        # no plan of this stream fits: the new one covers the old ones' shapes too (elementwise maximum), so that a caller that alternates between
        # shapes -- batch inference over length buckets: few long or many short utterances -- converges on ONE plan per stream instead of
        # allocating and freeing gigabytes (and dropping captured graphs) at every change of bucket
        for (st_, b, n, e), _h in self._plans:
            if st_ == stream:
                batch, seq, evals = max(batch, b), max(seq, n), max(evals, e)
        seq_cap = min(4096, -(-seq // 64) * 64)
        h = C.c_void_p()
        _lib.check(lib.f5_plan_create(self.native(), batch, seq_cap, max(evals, 1), C.byref(h)), "plan_create")
        mine = [i for i, (k, _) in enumerate(self._plans) if k[0] == stream]
        while len(mine) >= 2 or (len(self._plans) >= 12 and mine):  # keep HBM use bounded: at most two buckets per stream
            key, old = self._plans.pop(mine.pop(0))
            self._finish_plan(old)  # never destroy a plan whose deferred sample() is still in flight
            lib.f5_plan_destroy(old)
            mine = [i for i, (k, _) in enumerate(self._plans) if k[0] == stream]
        self._plans.append(((stream, batch, seq_cap, max(evals, 1)), h))
        return h

    def set_kernels(self, gemm=None, attn=None):
        """A/B switch between the reference tile kernels (0) and the tuned kernels (1) for existing plans."""
        lib = _lib.load()
        for _, h in self._plans:
            if gemm is not None:
                _lib.check(lib.f5_plan_set_option(h, b"gemm_kernel", int(gemm)))
            if attn is not None:
                _lib.check(lib.f5_plan_set_option(h, b"attn_kernel", int(attn)))

    # ------------------------------------------------------------------ reference API
    def clear_cache(self):
        self.text_cond, self.text_uncond = None, None

    def _text_embed(self, plan, text, seq_len, drop_text):
        lib = _lib.load()
        b = text.shape[0]
        ids = text.to(device="cuda", dtype=torch.int32).contiguous()
        out = torch.empty(b, seq_len, self.text_dim, device="cuda", dtype=torch.float32)
        _lib.check(lib.f5_text_embed(plan, b, seq_len, _lib.ptr(ids), ids.shape[1], int(bool(drop_text)), _lib.ptr(out), _lib.stream_ptr()),
                   "text_embed")
        return out

    def forward(self, x, cond, text, time, drop_audio_cond, drop_text, mask=None, cache=False):
        lib = _lib.load()
        batch, seq_len = x.shape[0], x.shape[1]
        plan = self.plan(batch, self._plan_seq(seq_len, text.shape[1]), 1)
        if time.ndim == 0:
            time = time.repeat(batch)
        if cache:
            if drop_text:
                if self.text_uncond is None:
                    self.text_uncond = self._text_embed(plan, text, seq_len, True)
                text_embed = self.text_uncond
            else:
                if self.text_cond is None:
                    self.text_cond = self._text_embed(plan, text, seq_len, False)
                text_embed = self.text_cond
        else:
            text_embed = self._text_embed(plan, text, seq_len, drop_text)
        xf = x.to(device="cuda", dtype=torch.float32).contiguous()
        cf = cond.to(device="cuda", dtype=torch.float32).contiguous()
        tf = time.to(device="cuda", dtype=torch.float32).contiguous()
        mk = None if mask is None else mask.to(device="cuda", dtype=torch.uint8).contiguous()
        out = torch.empty(batch, seq_len, self.mel_dim, device="cuda", dtype=torch.float32)
        self._native_forward(lib, plan, batch, seq_len, xf, cf, text_embed, tf, int(bool(drop_audio_cond)), mk, out)
        return out.to(x.dtype) if x.dtype != torch.float32 else out

    def _plan_seq(self, seq_len, text_len):
        return seq_len  # (text beyond the frame count is curtailed, dit.py:51)

    def _native_forward(self, lib, plan, batch, seq_len, xf, cf, text_embed, tf, drop_audio_cond, mk, out):
        _lib.check(lib.f5_dit_forward(plan, batch, seq_len, _lib.ptr(xf), _lib.ptr(cf), _lib.ptr(text_embed), _lib.ptr(tf),
                                      drop_audio_cond, _lib.ptr(mk), _lib.ptr(out), _lib.stream_ptr()), "dit_forward")

    # ------------------------------------------------------------------ whole-loop entry used by CFM.sample
    def native_sample(self, cond, text, lens, durations, y0, tgrid, steps, cfg_strength, method="euler", use_mask=True,
                      return_trajectory=False, use_graph=True, defer_guard=False):
        """cond/y0 f32 [B,N,mel] on the GPU, text int [B,nt] (-1 padded), lens/durations int [B], tgrid f32 [steps+1] (any device)."""
        lib = _lib.load()
        B, N = cond.shape[0], cond.shape[1]
        evals = steps * (2 if method == "midpoint" else 1)
        # capturing + instantiating a ~5000-node graph costs about as much as one small sample(): only replay shapes that recur
        # (serving with fixed buckets, batch inference); one-off shapes (free-form generate()) run eagerly on the stream
        key = (B, N, int(text.shape[1]), steps, method, float(cfg_strength), bool(use_mask))
        seen = self._seen_shapes.get(key, 0)
        self._seen_shapes[key] = seen + 1
        if use_graph == "auto":
            use_graph = seen >= 1
        plan = self.plan(B, N, evals)
        dev = "cuda"
        cond = cond.to(device=dev, dtype=torch.float32).contiguous()
        y0 = y0.to(device=dev, dtype=torch.float32).contiguous()
        ids = text.to(device=dev, dtype=torch.int32).contiguous()
        lens32 = lens.to(device=dev, dtype=torch.int32).contiguous()
        dur32 = durations.to(device=dev, dtype=torch.int32).contiguous() if use_mask else None
        tg = tgrid.detach().to("cpu", torch.float32).contiguous()
        out = torch.empty_like(cond)
        traj = torch.empty(steps + 1, B, N, self.mel_dim, device=dev, dtype=torch.float32) if return_trajectory else None
        meth = {"euler": _lib.F5_ODE_EULER, "midpoint": _lib.F5_ODE_MIDPOINT}[method]
        # defer_guard: f5_sample enqueues and returns without its one synchronisation (the fp16 range-guard read); finish_pending() does it
        _lib.check(lib.f5_plan_set_option(plan, b"residual_guard", 2 if defer_guard else 1), "plan_set_option")
        _lib.check(lib.f5_sample(plan, B, N, _lib.ptr(cond), _lib.ptr(ids), ids.shape[1], _lib.ptr(lens32), _lib.ptr(dur32), _lib.ptr(y0),
                                 C.c_void_p(tg.data_ptr()), steps, float(cfg_strength), meth, _lib.ptr(out), _lib.ptr(traj),
                                 int(bool(use_graph)), _lib.stream_ptr()), "sample")
        if defer_guard:
            self._pending[plan.value] = (plan, torch.cuda.current_stream())
            return out, traj
        self._report_fallback(lib, plan)
        return out, traj

    def native_sample_ragged(self, cond_cat, text, lens, frames, y0_cat, tgrid, steps, cfg_strength, method="euler", use_graph="auto"):
        """Utterances of different frame counts in ONE set of launches (include/f5hip.h: f5_sample_ragged).  cond_cat / y0_cat f32
        [sum(frames), mel] (the utterances one after the other), text int [B, nt] (-1 padded), lens int [B], frames list of ints.
        Returns out_cat [sum(frames), mel]; every utterance's rows equal its own batch-1 native_sample()."""
        lib = _lib.load()
        if self.BACKBONE != _lib.F5_BACKBONE_DIT:
            raise NotImplementedError("ragged sampling is built for the DiT backbone")
        B = len(frames)
        evals = steps * (2 if method == "midpoint" else 1)
        # the plan must hold T = sum(round_up(n_i + 16, 16)) rows per CFG half and (B + 1) text rows: size it by total rows, not by the longest
        # utterance (seq is capped at 4096 by plan(), so utterances near that length need a larger batch dimension)
        seq = min(4096, -(-(max(max(frames), int(text.shape[1])) + 32) // 64) * 64)
        rows = sum(-(-(int(f) + 16) // 16) * 16 for f in frames)
        plan = self.plan(max(B, -(-rows // seq), -(-(B + 1) * int(text.shape[1]) // seq)), seq, evals)
        dev = "cuda"
        cond_cat = cond_cat.to(device=dev, dtype=torch.float32).contiguous()
        y0_cat = y0_cat.to(device=dev, dtype=torch.float32).contiguous()
        ids = text.to(device=dev, dtype=torch.int32).contiguous()
        lens32 = lens.to(device=dev, dtype=torch.int32).contiguous()
        fr = torch.tensor([int(f) for f in frames], dtype=torch.int32)
        assert cond_cat.shape[0] == int(fr.sum()) == y0_cat.shape[0]
        tg = tgrid.detach().to("cpu", torch.float32).contiguous()
        out = torch.empty_like(cond_cat)
        meth = {"euler": _lib.F5_ODE_EULER, "midpoint": _lib.F5_ODE_MIDPOINT}[method]
        _lib.check(lib.f5_plan_set_option(plan, b"residual_guard", 1), "plan_set_option")
        # a list of frame counts that recurs (batch inference over fixed buckets, a server's chunk pattern) replays its hipGraph from the second
        # call on; one-off shapes run eagerly (capturing costs about one small sample())
        key = ("ragged", tuple(int(f) for f in frames), int(text.shape[1]), steps, method, float(cfg_strength))
        seen = self._seen_shapes.get(key, 0)
        self._seen_shapes[key] = seen + 1
        graph = seen >= 1 if use_graph == "auto" else bool(use_graph)
        _lib.check(lib.f5_plan_set_option(plan, b"ragged_graph", int(graph)), "plan_set_option")
        _lib.check(lib.f5_sample_ragged(plan, B, C.c_void_p(fr.data_ptr()), _lib.ptr(cond_cat), _lib.ptr(ids), ids.shape[1], _lib.ptr(lens32),
                                        _lib.ptr(y0_cat), C.c_void_p(tg.data_ptr()), steps, float(cfg_strength), meth, _lib.ptr(out),
                                        _lib.stream_ptr()), "sample_ragged")
        self._report_fallback(lib, plan)
        return out

    def _finish_plan(self, plan):
        """Complete the deferred sample() of ONE plan, if it has one (plan reuse, eviction)."""
        ent = self._pending.pop(plan.value, None)
        if ent is None:
            return
        lib = _lib.load()
        _, stream = ent
        with torch.cuda.stream(stream):
            _lib.check(lib.f5_sample_finish(plan, C.c_void_p(stream.cuda_stream)), "sample_finish")
        self._report_fallback(lib, plan)

    def finish_pending(self):
        """Complete every sample() issued with defer_guard=True: synchronise its stream, read the range-guard flag and let the library repeat
        the loop with fp32 residual storage if it was raised (the outputs are rewritten in place).  Returns the number of calls finished."""
        lib = _lib.load()
        pending, self._pending = self._pending, {}
        for plan, stream in pending.values():
            with torch.cuda.stream(stream):
                _lib.check(lib.f5_sample_finish(plan, C.c_void_p(stream.cuda_stream)), "sample_finish")
            self._report_fallback(lib, plan)
        return len(pending)

    def _report_fallback(self, lib, plan):
        if self.precision == _lib.F5_PREC_BF16:
            # fp16 residual-stream range guard (include/f5hip.h, plan option "residual_guard"): the library repeated the loop with fp32
            # residual storage when an activation reached fp16's range, and keeps fp32 storage for this plan; say so once per event
            n = C.c_int(0)
            _lib.check(lib.f5_plan_get_option(plan, b"residual_fallbacks", C.byref(n)), "plan_get_option")
            if n.value > self._fallbacks_seen.get(plan.value, 0):
                self._fallbacks_seen[plan.value] = n.value
                import struct
                import warnings
                bits, nan = C.c_int(0), C.c_int(0)
                _lib.check(lib.f5_plan_get_option(plan, b"residual_guard_amax_bits", C.byref(bits)), "plan_get_option")
                _lib.check(lib.f5_plan_get_option(plan, b"residual_guard_nan", C.byref(nan)), "plan_get_option")
                amax = struct.unpack("f", struct.pack("I", bits.value & 0xffffffff))[0]
                diag = {}
                for k in ("pass", "blocks", "row"):
                    v = C.c_int(0)
                    _lib.check(lib.f5_plan_get_option(plan, f"residual_guard_{k}".encode(), C.byref(v)), "plan_get_option")
                    diag[k] = v.value & 0xffffffff
                warnings.warn(f"libf5hip: the residual stream left the fp16 range (largest |element| seen {amax:.6g}, NaN seen: {bool(nan.value)}, "
                              f"passes 0x{diag['pass']:x}, blocks 0x{diag['blocks']:x}, first row {diag['row']}); "
                              "sample() was repeated with fp32 residual storage, which this plan keeps from now on", RuntimeWarning, stacklevel=3)

    def residual_fallbacks(self):
        """Number of sample() calls (over the live plans) that the library repeated with fp32 residual storage."""
        lib, total = _lib.load(), 0
        for _, h in self._plans:
            n = C.c_int(0)
            _lib.check(lib.f5_plan_get_option(h, b"residual_fallbacks", C.byref(n)), "plan_get_option")
            total += n.value
        return total

    def set_tap(self, plan, name, dst):
        _lib.check(_lib.load().f5_plan_set_tap(plan, None if name is None else name.encode(), _lib.ptr(dst)))
