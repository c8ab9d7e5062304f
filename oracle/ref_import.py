"""Container-only loader for the *real* reference model code (test infrastructure, never shipped).

ORACLE / TEST INFRASTRUCTURE ONLY.  Nothing under ``eraxvif5tts_amd/`` may import this file.
Only ``oracle/make_golden.py`` (run by hand in the build container, where ``/root/reference``
exists) uses it, to turn the reference's own ``cfm.py`` / ``dit.py`` / ``modules.py`` into golden
input/output vectors under ``tests/golden/``.  The reference source itself never travels.

Recipe: the three reference files are imported *unmodified, by path*.
``f5_tts/model/__init__.py`` is bypassed by pre-seeding bare package objects, and the third-party
packages that are absent from this image get stand-ins:

* no-arithmetic shells: ``torchaudio``, ``librosa``, ``jieba``, ``pypinyin`` (never called when
  ``cond`` is a mel and ``text`` an id tensor);
* arithmetic restated here from the published algorithms (=> parity at these three boundaries is
  "unpinned"):
    - ``x_transformers.x_transformers.RotaryEmbedding / apply_rotary_pos_emb`` (x_transformers>=1.31.14,
      reference pyproject.toml:42; call sites dit.py:16,134,215 and modules.py:20,476-480),
    - ``torchdiffeq.odeint`` fixed-grid ``euler`` / ``midpoint`` (reference pyproject.toml:36; call cfm.py:19,197).

The single deliberate deviation from the shipped reference: ``force_no_attn_dropout()`` wraps
``F.scaled_dot_product_attention`` so the hard-coded ``dropout_p=0.1`` of modules.py:490 becomes 0.0
(the functional dropout is live at inference in the reference and makes it non-deterministic).
"""
from __future__ import annotations

import importlib
import math
import sys
import types

import torch

REF_ROOT = "/root/reference/src"


# --------------------------------------------------------------------------- stand-ins with arithmetic
class RotaryEmbedding(torch.nn.Module):
    """x_transformers RotaryEmbedding (no xpos, no interpolation): interleaved duplicate layout."""

    def __init__(self, dim, base=10000.0):
        super().__init__()
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
        self.register_buffer("inv_freq", inv_freq)

    def forward_from_seq_len(self, seq_len):
        t = torch.arange(seq_len, device=self.inv_freq.device)
        return self.forward(t)

    def forward(self, t):
        if t.ndim == 1:
            t = t[None, :]
        freqs = torch.einsum("bi,j->bij", t.type_as(self.inv_freq), self.inv_freq)
        freqs = torch.stack((freqs, freqs), dim=-1).flatten(-2)  # '... d r -> ... (d r)'
        return freqs, 1.0


def _rotate_half(x):
    x = x.unflatten(-1, (-1, 2))
    x1, x2 = x.unbind(dim=-1)
    return torch.stack((-x2, x1), dim=-1).flatten(-2)


def apply_rotary_pos_emb(t, freqs, scale=1):
    rot_dim, seq_len, orig_dtype = freqs.shape[-1], t.shape[-2], t.dtype
    freqs = freqs[:, -seq_len:, :]
    if t.ndim == 4 and freqs.ndim == 3:
        freqs = freqs[:, None]
    t, t_unrotated = t[..., :rot_dim], t[..., rot_dim:]
    t = (t * freqs.cos() * scale) + (_rotate_half(t) * freqs.sin() * scale)
    return torch.cat((t, t_unrotated), dim=-1).type(orig_dtype)


class RMSNorm(torch.nn.Module):
    """x_transformers.RMSNorm (imported by backbones/unett.py:17; third-party, restated as published for the pinned >= 1.31 releases:
    ``F.normalize(x, dim=-1) * dim ** 0.5 * g`` with a learned per-channel gain ``g`` initialised to 1).  Parity at this boundary is unpinned."""

    def __init__(self, dim, unit_offset=False):
        super().__init__()
        self.unit_offset = unit_offset
        self.scale = dim ** 0.5
        self.g = torch.nn.Parameter(torch.zeros(dim))
        torch.nn.init.constant_(self.g, 1.0 - float(unit_offset))

    def forward(self, x):
        gamma = self.g + float(self.unit_offset)
        return torch.nn.functional.normalize(x, dim=-1) * self.scale * gamma


def odeint(func, y0, t, *, method="euler", **_unused):
    """torchdiffeq fixed-grid solvers on the caller's grid: returns the stacked states at every t."""
    ys = [y0]
    y = y0
    for t0, t1 in zip(t[:-1], t[1:]):
        dt = t1 - t0
        if method == "euler":
            dy = dt * func(t0, y)
        elif method == "midpoint":
            half = 0.5 * dt
            y_mid = y + func(t0, y) * half
            dy = dt * func(t0 + half, y_mid)
        else:
            raise ValueError(f"unsupported fixed-grid method {method}")
        y = y + dy
        ys.append(y)
    return torch.stack(ys, dim=0)


# --------------------------------------------------------------------------- loader
def _shell(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def load_reference():
    """Returns (modules, dit, cfm) = the reference's own python modules, imported by path."""
    sys.dont_write_bytecode = True  # the reference tree is read-only
    if "f5_tts.model.cfm" in sys.modules:
        return (sys.modules["f5_tts.model.modules"], sys.modules["f5_tts.model.backbones.dit"],
                sys.modules["f5_tts.model.cfm"])
    for name in ("torchaudio", "librosa", "jieba", "pypinyin"):
        _shell(name)
    lf = _shell("librosa.filters")

    def _librosa_mel(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, **_):  # librosa is absent: the oracle's restatement of its published algorithm
        import cpu_ref
        return cpu_ref.librosa_mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    lf.mel = _librosa_mel
    sys.modules["librosa"].filters = lf
    sys.modules["pypinyin"].lazy_pinyin = None
    sys.modules["pypinyin"].Style = None
    xt = _shell("x_transformers")
    xtx = _shell("x_transformers.x_transformers")
    xtx.RotaryEmbedding = RotaryEmbedding
    xtx.apply_rotary_pos_emb = apply_rotary_pos_emb
    xt.x_transformers = xtx
    xt.RMSNorm = RMSNorm
    td = _shell("torchdiffeq")
    td.odeint = odeint
    for pkg, path in (("f5_tts", f"{REF_ROOT}/f5_tts"), ("f5_tts.model", f"{REF_ROOT}/f5_tts/model"),
                      ("f5_tts.model.backbones", f"{REF_ROOT}/f5_tts/model/backbones")):
        m = types.ModuleType(pkg)
        m.__path__ = [path]
        sys.modules[pkg] = m
    modules = importlib.import_module("f5_tts.model.modules")
    dit = importlib.import_module("f5_tts.model.backbones.dit")
    cfm = importlib.import_module("f5_tts.model.cfm")
    return modules, dit, cfm


def load_unett():
    """the reference's own backbones/unett.py (UNetT: the E2-TTS backbone, plug point A of SURVEY 8f-4), imported by path after load_reference()."""
    load_reference()
    return importlib.import_module("f5_tts.model.backbones.unett")


def load_mmdit():
    """the reference's own backbones/mmdit.py (MMDiT, plug point A of SURVEY 8f-4), imported by path after load_reference()."""
    load_reference()
    return importlib.import_module("f5_tts.model.backbones.mmdit")


def force_no_attn_dropout(modules):
    """Deterministic oracle: dropout_p of modules.py:490 forced to 0.0 (no edit to reference files)."""
    real = torch.nn.functional.scaled_dot_product_attention

    def sdpa(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, **kw):
        return real(q, k, v, attn_mask=attn_mask, dropout_p=0.0, is_causal=is_causal, **kw)

    ns = types.SimpleNamespace(**{k: getattr(modules.F, k) for k in dir(modules.F) if not k.startswith("__")})
    ns.scaled_dot_product_attention = sdpa
    modules.F = ns


def rerandomize_zero_init(model, std=0.02, seed=1234):
    """DiT.initialize_weights zeroes AdaLN / output layers (dit.py:162-172): a fresh model outputs 0.
    Golden vectors re-randomise every all-zero parameter (and GRN gamma/beta) so all paths are live."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for _, p in model.named_parameters():
            if p.numel() > 0 and torch.count_nonzero(p) == 0:
                p.copy_(torch.randn(p.shape, generator=g) * std)
    return model
