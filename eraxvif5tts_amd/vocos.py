"""Vocos vocoder (plug point B) on MI355X: drop-in for the object ``load_vocoder`` returns in the reference
(``Vocos.from_hparams(config.yaml)`` + ``load_state_dict(pytorch_model.bin)``, utils_infer.py:101-124) and for its use
``vocoder.decode(mel[b, 100, T]) -> wave[b, (T-1)*256]`` (f5tts_wrapper.py:524, eval_infer_batch.py:188).

The ``vocos`` package itself is a third-party dependency that is absent from the reference tree; its model
(charactr/vocos-mel-24khz: ConvNeXt backbone + ISTFT head) is restated in ``csrc/vocoder.hip``.  All arithmetic runs in
libf5hip (fp32-input MFMA); there is no PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from . import _lib

DEFAULT_HPARAMS = dict(n_mels=100, dim=512, intermediate_dim=1536, num_layers=8, n_fft=1024, hop_length=256)


def _state_spec(hp):
    D, I, C_, L, F = hp["dim"], hp["intermediate_dim"], hp["n_mels"], hp["num_layers"], hp["n_fft"] // 2 + 1
    spec = {"backbone.embed.weight": (D, C_, 7), "backbone.embed.bias": (D,), "backbone.norm.weight": (D,), "backbone.norm.bias": (D,),
            "backbone.final_layer_norm.weight": (D,), "backbone.final_layer_norm.bias": (D,),
            "head.out.weight": (2 * F, D), "head.out.bias": (2 * F,)}
    for i in range(L):
        p = f"backbone.convnext.{i}."
        spec.update({p + "dwconv.weight": (D, 1, 7), p + "dwconv.bias": (D,), p + "norm.weight": (D,), p + "norm.bias": (D,),
                     p + "pwconv1.weight": (I, D), p + "pwconv1.bias": (I,), p + "pwconv2.weight": (D, I), p + "pwconv2.bias": (D,),
                     p + "gamma": (D,)})
    return spec


class Vocos(nn.Module):
    def __init__(self, **hparams):
        super().__init__()
        self.hp = {**DEFAULT_HPARAMS, **hparams}
        from .model.backbones.dit import _register
        for name, shape in _state_spec(self.hp).items():
            t = torch.zeros(shape)
            if name.endswith("weight") and len(shape) > 1:
                nn.init.trunc_normal_(t, std=0.02)
            elif name.endswith("norm.weight") or name.endswith("layer_norm.weight"):
                t.fill_(1.0)
            elif name.endswith("gamma"):
                t.fill_(1.0 / self.hp["num_layers"])
            _register(self, name, t)
        _register(self, "head.istft.window", torch.hann_window(self.hp["n_fft"]), buffer=True)
        self._native = None
        self.register_load_state_dict_post_hook(lambda module, _k: module._drop_native())

    @classmethod
    def from_hparams(cls, config_path):
        """Reads the keys this model needs from a vocos ``config.yaml`` (feature_extractor / backbone / head init_args)."""
        import yaml
        with open(config_path, "r") as f:
            cfg = yaml.safe_load(f)
        hp = {}
        fe = (cfg.get("feature_extractor") or {}).get("init_args", {})
        bb = (cfg.get("backbone") or {}).get("init_args", {})
        hd = (cfg.get("head") or {}).get("init_args", {})
        for src, dst in (("n_mels", "n_mels"), ("n_fft", "n_fft"), ("hop_length", "hop_length")):
            if src in fe:
                hp[dst] = fe[src]
        for src, dst in (("input_channels", "n_mels"), ("dim", "dim"), ("intermediate_dim", "intermediate_dim"), ("num_layers", "num_layers")):
            if src in bb:
                hp[dst] = bb[src]
        for src, dst in (("n_fft", "n_fft"), ("hop_length", "hop_length")):
            if src in hd:
                hp[dst] = hd[src]
        return cls(**hp)

    def _drop_native(self):
        if self._native is not None:
            _lib.load().f5_vocoder_destroy(self._native)
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
        hp = self.hp
        cfg = _lib.VocosConfig(n_mels=hp["n_mels"], dim=hp["dim"], inter_dim=hp["intermediate_dim"], layers=hp["num_layers"],
                               n_fft=hp["n_fft"], hop=hp["hop_length"])
        h = C.c_void_p()
        _lib.check(lib.f5_vocoder_create(C.byref(cfg), C.byref(h)), "vocoder_create")
        try:
            _lib.set_tensors(h, "f5_vocoder_set_tensor", "f5_vocoder_has_tensor", self.state_dict())
            _lib.check(lib.f5_vocoder_finalize(h), "vocoder_finalize")
        except Exception:
            lib.f5_vocoder_destroy(h)
            raise
        self._native = h
        return h

    @torch.no_grad()
    def decode(self, features_input: torch.Tensor, **_kwargs) -> torch.Tensor:
        """mel [b, n_mels, T] -> wave [b, (T-1)*hop]"""
        lib = _lib.load()
        mel = features_input.to(device="cuda", dtype=torch.float32).contiguous()
        B, Cm, T = mel.shape
        assert Cm == self.hp["n_mels"], f"expected {self.hp['n_mels']} mel channels, got {Cm}"
        wave = torch.empty(B, (T - 1) * self.hp["hop_length"], device="cuda", dtype=torch.float32)
        _lib.check(lib.f5_vocoder_decode(self.native(), B, T, _lib.ptr(mel), _lib.ptr(wave), _lib.stream_ptr()), "vocoder_decode")
        return wave

    @torch.no_grad()
    def istft_head(self, head_out: torch.Tensor) -> torch.Tensor:
        """head.out activations [b, T, n_fft+2] (log-magnitude | phase) -> wave [b, (T-1)*hop] (the ISTFT head alone)"""
        lib = _lib.load()
        x = head_out.to(device="cuda", dtype=torch.float32).contiguous()
        B, T, F2 = x.shape
        assert F2 == self.hp["n_fft"] + 2
        wave = torch.empty(B, (T - 1) * self.hp["hop_length"], device="cuda", dtype=torch.float32)
        _lib.check(lib.f5_vocoder_istft_head(self.native(), B, T, _lib.ptr(x), _lib.ptr(wave), _lib.stream_ptr()), "istft_head")
        return wave

    def forward(self, mel):
        return self.decode(mel)
