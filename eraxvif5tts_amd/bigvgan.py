"""BigVGAN-v2 vocoder (plug point B, the ``vocoder_name="bigvgan"`` branch) on MI355X: stands in for the object the reference builds with
``third_party.BigVGAN.bigvgan.BigVGAN.from_pretrained(local_path, use_cuda_kernel=False)`` + ``remove_weight_norm()`` + ``.eval().to(device)``
(infer/utils_infer.py:125-138) and calls as ``vocoder(mel[b, 100, T]) -> wave[b, 1, T * 256]`` (infer/f5tts_wrapper.py:526,
eval/eval_infer_batch.py:189).

PARITY UNPINNED.  The BigVGAN source is a git submodule that is absent from the reference tree, and no checkpoint exists offline: the
generator (``csrc/bigvgan.hip``) is restated from the published BigVGAN-v2 code as recalled -- ``nvidia/bigvgan_v2_24khz_100band_256x``: 112.4 M
parameters, which the restated shapes reproduce -- and checked against the CPU oracle's restatement of the same text (``oracle/cpu_ref.py``), not against
NVIDIA's implementation.  Tensor names follow the published checkpoint (``bigvgan_generator.pt``: ``{"generator": state_dict}``); weight-normed
convolutions (``weight_g`` / ``weight_v``, or torch's ``parametrizations.weight.original0/1``) are folded at load time, so ``remove_weight_norm()``
is a no-op kept for the reference's call.  All arithmetic runs in libf5hip (fp32-input MFMA); there is no PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import json
import os

import torch
from torch import nn

from . import _lib

BIGVGAN_V2_24KHZ_100BAND_256X = dict(num_mels=100, upsample_initial_channel=1536, upsample_rates=[4, 4, 2, 2, 2, 2],
                                      upsample_kernel_sizes=[8, 8, 4, 4, 4, 4], resblock="1", resblock_kernel_sizes=[3, 7, 11],
                                      resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], activation="snakebeta", snake_logscale=True,
                                      use_tanh_at_final=False, use_bias_at_final=False)


def _state_spec(h):
    ch = h["upsample_initial_channel"]
    spec = {"conv_pre.weight": (ch, h["num_mels"], 7), "conv_pre.bias": (ch,)}
    nk = len(h["resblock_kernel_sizes"])
    for i, k in enumerate(h["upsample_kernel_sizes"]):
        spec[f"ups.{i}.0.weight"] = (ch, ch // 2, k)
        spec[f"ups.{i}.0.bias"] = (ch // 2,)
        ch //= 2
        for j, ks in enumerate(h["resblock_kernel_sizes"]):
            p = f"resblocks.{i * nk + j}."
            for t in range(3):
                for cv in ("convs1", "convs2"):
                    spec[p + f"{cv}.{t}.weight"] = (ch, ch, ks)
                    spec[p + f"{cv}.{t}.bias"] = (ch,)
            for a in range(6):
                spec[p + f"activations.{a}.act.alpha"] = (ch,)
                spec[p + f"activations.{a}.act.beta"] = (ch,)
    spec["activation_post.act.alpha"] = (ch,)
    spec["activation_post.act.beta"] = (ch,)
    spec["conv_post.weight"] = (1, ch, 7)
    if h.get("use_bias_at_final", True):
        spec["conv_post.bias"] = (1,)
    return spec


def fold_weight_norm(state_dict):
    """checkpoint names -> plain names: ``x.weight_g`` + ``x.weight_v`` (torch.nn.utils.weight_norm, dim 0) or
    ``x.parametrizations.weight.original0`` + ``original1`` become ``x.weight = g * v / ||v||`` (norm over every axis but the first)."""
    out = {}
    for k, v in state_dict.items():
        if k.endswith(".weight_v") or k.endswith(".parametrizations.weight.original1"):
            base = k[: -len(".weight_v")] if k.endswith(".weight_v") else k[: -len(".parametrizations.weight.original1")]
            g = state_dict.get(base + ".weight_g", state_dict.get(base + ".parametrizations.weight.original0"))
            vf = v.float()
            norm = vf.reshape(vf.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (vf.ndim - 1)))
            out[base + ".weight"] = g.float() * vf / norm
        elif k.endswith(".weight_g") or k.endswith(".parametrizations.weight.original0"):
            continue
        else:
            out[k] = v
    return out


class BigVGAN(nn.Module):
    def __init__(self, h=None, use_cuda_kernel=False):
        super().__init__()
        if use_cuda_kernel:
            raise NotImplementedError("use_cuda_kernel: the fused CUDA activation of the BigVGAN repository does not exist here; the HIP path is always fused")
        self.h = {**BIGVGAN_V2_24KHZ_100BAND_256X, **(dict(h) if h else {})}
        if str(self.h.get("resblock", "1")) != "1" or self.h.get("activation", "snakebeta") != "snakebeta":
            raise NotImplementedError("only AMPBlock1 with the snakebeta activation (every bigvgan_v2 configuration) is on the MI355X path")
        if any(len(d) != 3 for d in self.h["resblock_dilation_sizes"]):
            raise NotImplementedError("AMPBlock1 with three dilations per block")
        from .model.backbones.dit import _register
        for name, shape in _state_spec(self.h).items():
            t = torch.zeros(shape)
            if name.endswith("weight"):
                nn.init.normal_(t, std=0.01)  # (the published init_weights)
            _register(self, name, t)
        self._native = None
        self.register_load_state_dict_post_hook(lambda module, _k: module._drop_native())

    # ------------------------------------------------------------------ the reference's calls
    @classmethod
    def from_pretrained(cls, model_id, use_cuda_kernel=False, **_):
        """A LOCAL directory with ``config.json`` and ``bigvgan_generator.pt`` (what huggingface nvidia/bigvgan_v2_24khz_100band_256x holds;
        the reference's is_local branch, utils_infer.py:131-133).  Hub names cannot be resolved offline."""
        if not os.path.isdir(model_id):
            raise RuntimeError(f"{model_id}: only a local BigVGAN directory can be loaded (no network); download nvidia/bigvgan_v2_24khz_100band_256x by hand")
        with open(os.path.join(model_id, "config.json"), "r") as f:
            h = json.load(f)
        model = cls(h, use_cuda_kernel=use_cuda_kernel)
        ckpt = torch.load(os.path.join(model_id, "bigvgan_generator.pt"), map_location="cpu", weights_only=True)
        model.load_state_dict(ckpt["generator"] if "generator" in ckpt else ckpt)
        return model

    def load_state_dict(self, state_dict, strict=True, **kw):
        sd = fold_weight_norm(dict(state_dict))
        own = self.state_dict()
        filt = {}
        for k in list(sd):  # the 12-tap anti-aliasing buffers of every Activation1d are identical: keep one pair, drop the rest
            if k.endswith("upsample.filter"):
                filt["up"] = sd.pop(k)
            elif k.endswith("downsample.lowpass.filter"):
                filt["down"] = sd.pop(k)
        self._filters = {k: v.reshape(-1).float() for k, v in filt.items()}
        return super().load_state_dict({k: v for k, v in sd.items() if k in own or strict}, strict=strict, **kw)

    def remove_weight_norm(self):
        """Kept for the reference's call (utils_infer.py:136): the weights were folded when they were loaded."""
        return self

    # ------------------------------------------------------------------ native handle
    def _drop_native(self):
        if self._native is not None:
            _lib.load().f5_bigvgan_destroy(self._native)
            self._native = None

    def __del__(self):
        try:
            self._drop_native()
        except Exception:  # noqa: BLE001
            pass

    def native(self):
        if self._native is not None:
            return self._native
        _lib.require_gpu()
        lib = _lib.load()
        h = self.h
        cfg = _lib.BigVGANConfig(num_mels=h["num_mels"], upsample_initial_channel=h["upsample_initial_channel"], num_upsamples=len(h["upsample_rates"]),
                                 num_kernels=len(h["resblock_kernel_sizes"]), snake_logscale=int(bool(h.get("snake_logscale", True))),
                                 use_tanh_at_final=int(bool(h.get("use_tanh_at_final", True))), use_bias_at_final=int(bool(h.get("use_bias_at_final", True))))
        for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
            cfg.upsample_rates[i], cfg.upsample_kernel_sizes[i] = u, k
        for j, (ks, dil) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
            cfg.resblock_kernel_sizes[j] = ks
            for t in range(3):
                cfg.resblock_dilations[j][t] = dil[t]
        hd = C.c_void_p()
        _lib.check(lib.f5_bigvgan_create(C.byref(cfg), C.byref(hd)), "bigvgan_create")
        try:
            tensors = dict(self.state_dict())
            for key, name in (("up", "aa_up_filter"), ("down", "aa_down_filter")):
                if key in getattr(self, "_filters", {}):
                    tensors[name] = self._filters[key]
            _lib.set_tensors(hd, "f5_bigvgan_set_tensor", "f5_bigvgan_has_tensor", tensors)
            _lib.check(lib.f5_bigvgan_finalize(hd), "bigvgan_finalize")
        except Exception:
            lib.f5_bigvgan_destroy(hd)
            raise
        self._native = hd
        return hd

    @torch.no_grad()
    def forward(self, x):
        """mel [b, num_mels, T] on the GPU -> wave [b, 1, T * prod(upsample_rates)] (BigVGAN.forward)."""
        lib = _lib.load()
        mel = x.to(device="cuda", dtype=torch.float32).contiguous()
        b, _, T = mel.shape
        up = 1
        for u in self.h["upsample_rates"]:
            up *= u
        out = torch.empty(b, 1, T * up, device=mel.device, dtype=torch.float32)
        _lib.check(lib.f5_bigvgan_forward(self.native(), b, T, _lib.ptr(mel), _lib.ptr(out), _lib.stream_ptr()), "bigvgan_forward")
        return out
