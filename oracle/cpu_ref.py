"""CPU restatement of the F5-TTS flow-matching inference path (ORACLE -- test infrastructure only).

This file is the *checker*, never the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  ``eraxvif5tts_amd`` must not (and does not).

It restates, as plain functional fp32 PyTorch over a flat ``{name: tensor}`` weight dict (names =
the reference ``DiT.state_dict()`` keys), every arithmetic step of SURVEY.md section 8(a).  Each function
cites the reference lines it follows (paths relative to /root/reference/src/f5_tts).

Pinning: the reference ships no tests/golden vectors, so this restatement is pinned against outputs
of the reference's own ``cfm.py`` / ``dit.py`` / ``modules.py`` run in the build container
(``oracle/make_golden.py`` -> ``tests/golden/*.npz``; checked by ``tests/test_oracle_golden.py``).
Three third-party boundaries are "parity unpinned" (restated from the published algorithms, source
absent from the container): x_transformers RoPE, torchdiffeq fixed-grid solvers, vocos.
Deviation from the shipped reference: attention dropout (modules.py:490, live at inference) is 0.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

LN_EPS = 1e-6


# ----------------------------------------------------------------------------- small helpers
def _lin(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


def _layernorm(x, eps=LN_EPS):
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc * torch.rsqrt(var + eps)


def mish(x):
    return x * torch.tanh(F.softplus(x))


def gelu_tanh(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def silu(x):
    return x * torch.sigmoid(x)


# ----------------------------------------------------------------------------- a6: time grid
def time_grid(steps, sway_sampling_coef=None, t_start=0.0, dtype=torch.float32):
    """cfm.py:193-195: linspace(t_start,1,steps+1) then t += s*(cos(pi/2 t) - 1 + t)."""
    t = torch.linspace(t_start, 1, steps + 1, dtype=dtype)
    if sway_sampling_coef is not None:
        t = t + sway_sampling_coef * (torch.cos(torch.pi / 2 * t) - 1 + t)
    return t


# ----------------------------------------------------------------------------- a10: time embedding
def timestep_embedding(W, time):
    """modules.py:149-161,721-731. time: [b] -> [b, dim].  sin half first, scale 1000, 256 freqs."""
    half = 128
    k = math.log(10000) / (half - 1)
    freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -k)
    arg = 1000.0 * time[:, None].float() * freqs[None, :]
    emb = torch.cat([arg.sin(), arg.cos()], dim=-1).to(time.dtype)
    h = silu(_lin(emb, W["time_embed.time_mlp.0.weight"], W["time_embed.time_mlp.0.bias"]))
    return _lin(h, W["time_embed.time_mlp.2.weight"], W["time_embed.time_mlp.2.bias"])


# ----------------------------------------------------------------------------- a11/a12: text embedding
def text_pos_table(dim, n_pos, theta=10000.0):
    """modules.py:196-207 precompute_freqs_cis: [cos | sin] (cos first), theta_j = theta^(-2j/dim)."""
    inv = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
    ang = torch.outer(torch.arange(n_pos).float(), inv)
    return torch.cat([ang.cos(), ang.sin()], dim=-1)


def grn(x, gamma, beta):
    """modules.py:225-234: L2 norm over the *sequence* axis, divided by its channel mean."""
    gx = torch.sqrt((x * x).sum(dim=1, keepdim=True))
    nx = gx / (gx.mean(dim=-1, keepdim=True) + 1e-6)
    return gamma * (x * nx) + beta + x


def convnext_v2_block(W, pre, x):
    """modules.py:241-269. x: [b, n, d]."""
    d = x.shape[-1]
    y = F.conv1d(x.transpose(1, 2), W[pre + "dwconv.weight"], W[pre + "dwconv.bias"], padding=3, groups=d)
    y = y.transpose(1, 2)
    y = _layernorm(y) * W[pre + "norm.weight"] + W[pre + "norm.bias"]
    y = gelu_erf(_lin(y, W[pre + "pwconv1.weight"], W[pre + "pwconv1.bias"]))
    y = grn(y, W[pre + "grn.gamma"], W[pre + "grn.beta"])
    y = _lin(y, W[pre + "pwconv2.weight"], W[pre + "pwconv2.bias"])
    return x + y


def text_embedding(W, cfg, text, seq_len, drop_text=False):
    """dit.py:49-79. text: int [b, nt] with -1 padding -> [b, seq_len, text_dim]."""
    ids = text + 1
    ids = ids[:, :seq_len]
    ids = F.pad(ids, (0, seq_len - ids.shape[1]), value=0)
    filler = ids == 0  # computed BEFORE drop_text zeroing (dit.py:54-58)
    if drop_text:
        ids = torch.zeros_like(ids)
    x = W["text_embed.text_embed.weight"][ids]
    n_conv = cfg.get("conv_layers", 0)
    if n_conv > 0:
        pos = torch.arange(seq_len).clamp(max=4095)  # modules.py:210-219, batch_start = 0
        x = x + text_pos_table(x.shape[-1], 4096)[pos][None]
        mp = cfg.get("text_mask_padding", True)
        if mp:
            x = x.masked_fill(filler[..., None], 0.0)
        for i in range(n_conv):
            x = convnext_v2_block(W, f"text_embed.text_blocks.{i}.", x)
            if mp:
                x = x.masked_fill(filler[..., None], 0.0)
    return x


# ----------------------------------------------------------------------------- a13: input embedding
def conv_pos_embed(W, x):
    """modules.py:167-190 (called with mask=None from dit.py:96): 2x [grouped conv k=31 g=16 -> Mish]."""
    y = x.transpose(1, 2)
    for i in (0, 2):
        y = mish(F.conv1d(y, W[f"input_embed.conv_pos_embed.conv1d.{i}.weight"],
                          W[f"input_embed.conv_pos_embed.conv1d.{i}.bias"], padding=15, groups=16))
    return y.transpose(1, 2)


def input_embedding(W, x, cond, text_embed, drop_audio_cond=False):
    """dit.py:91-97: proj(cat(x, cond, text)) then x + conv_pos_embed(x)."""
    if drop_audio_cond:
        cond = torch.zeros_like(cond)
    h = _lin(torch.cat([x, cond, text_embed], dim=-1), W["input_embed.proj.weight"], W["input_embed.proj.bias"])
    return conv_pos_embed(W, h) + h


# ----------------------------------------------------------------------------- a14/a18: RoPE
def rope_angles(seq_len, dim_head=64, base=10000.0):
    """x_transformers RotaryEmbedding.forward_from_seq_len (dit.py:134,215): angle[p, j] = p * base^(-2j/dim)."""
    inv = 1.0 / (base ** (torch.arange(0, dim_head, 2).float() / dim_head))
    return torch.outer(torch.arange(seq_len).float(), inv)  # [n, dim_head/2]


def apply_rope(t, ang, half_split=False):
    """x_transformers apply_rotary_pos_emb on [b, h, n, d]: adjacent pairs (x0,x1)->(x0 c - x1 s, x1 c + x0 s).
    ``half_split=True`` is the other published rotary form (freqs = cat(freqs, freqs), rotate_half on the two halves of the head:
    frequency j turns features (j, j + d/2)); x_transformers is not vendored in the reference tree, so both are kept behind ONE switch
    (cfg["rope_layout"] == "half_split"; the adjacent form is what the pinned >= 1.31 releases compute)."""
    c, s = ang.cos()[None, None], ang.sin()[None, None]
    if half_split:
        d2 = t.shape[-1] // 2
        x0, x1 = t[..., :d2].float(), t[..., d2:].float()
        return torch.cat([x0 * c - x1 * s, x1 * c + x0 * s], dim=-1).to(t.dtype)
    x0, x1 = t[..., 0::2].float(), t[..., 1::2].float()
    out = torch.stack([x0 * c - x1 * s, x1 * c + x0 * s], dim=-1).flatten(-2)
    return out.to(t.dtype)


# ----------------------------------------------------------------------------- a15-a19: DiT block
def attention(W, pre, cfg, x, mask, ang):
    """modules.py:442-503 with dropout_p forced to 0 (deviation stated in the module docstring)."""
    b, n, _ = x.shape
    h, dh = cfg["heads"], cfg.get("dim_head", 64)
    q = _lin(x, W[pre + "to_q.weight"], W[pre + "to_q.bias"]).view(b, n, h, dh).transpose(1, 2)
    k = _lin(x, W[pre + "to_k.weight"], W[pre + "to_k.bias"]).view(b, n, h, dh).transpose(1, 2)
    v = _lin(x, W[pre + "to_v.weight"], W[pre + "to_v.bias"]).view(b, n, h, dh).transpose(1, 2)
    if cfg.get("qk_norm") == "rms_norm" and (pre + "q_norm.weight") in W:  # modules.py:275-294,394-396,463-467: RMSNorm(dim_head, eps 1e-6) per head, before RoPE
        q = F.rms_norm(q, (dh,), weight=W[pre + "q_norm.weight"], eps=1e-6)
        k = F.rms_norm(k, (dh,), weight=W[pre + "k_norm.weight"], eps=1e-6)
    pn = cfg.get("pe_attn_head", None)
    pn = h if pn is None else pn
    hs = cfg.get("rope_layout", "adjacent") == "half_split"
    q = torch.cat([apply_rope(q[:, :pn], ang, hs), q[:, pn:]], dim=1)
    k = torch.cat([apply_rope(k[:, :pn], ang, hs), k[:, pn:]], dim=1)
    s = torch.einsum("bhid,bhjd->bhij", q, k) / math.sqrt(dh)
    if mask is not None:
        s = s.masked_fill(~mask[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = torch.einsum("bhij,bhjd->bhid", p, v).transpose(1, 2).reshape(b, n, h * dh)
    o = _lin(o, W[pre + "to_out.0.weight"], W[pre + "to_out.0.bias"])
    if mask is not None:
        o = o.masked_fill(~mask[..., None], 0.0)
    return o


def dit_block(W, i, cfg, x, t_emb, mask, ang, trace=None):
    """modules.py:627-641 + AdaLayerNorm modules.py:310-315 (chunk order shift,scale,gate x {msa,mlp})."""
    pre = f"transformer_blocks.{i}."
    emb = _lin(silu(t_emb), W[pre + "attn_norm.linear.weight"], W[pre + "attn_norm.linear.bias"])
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = emb.chunk(6, dim=1)
    n1 = _layernorm(x) * (1 + sc_a[:, None]) + sh_a[:, None]
    a = attention(W, pre + "attn.", cfg, n1, mask, ang)
    x = x + g_a[:, None] * a
    n2 = _layernorm(x) * (1 + sc_m[:, None]) + sh_m[:, None]
    f = _lin(gelu_tanh(_lin(n2, W[pre + "ff.ff.0.0.weight"], W[pre + "ff.ff.0.0.bias"])),
             W[pre + "ff.ff.2.weight"], W[pre + "ff.ff.2.bias"])
    y = x + g_m[:, None] * f
    if trace is not None:
        trace.update({f"blk{i}.n1": n1, f"blk{i}.attn": a, f"blk{i}.x_mid": x, f"blk{i}.n2": n2, f"blk{i}.out": y})
    return y


# ----------------------------------------------------------------------------- a9/a20: DiT forward
def dit_forward(W, cfg, x, cond, text, time, drop_audio_cond, drop_text, mask=None, text_embed=None, trace=None):
    """dit.py:185-233.  ``text_embed`` may be passed to mimic the cache (dit.py:202-210)."""
    b, n, _ = x.shape
    if time.ndim == 0:
        time = time.repeat(b)
    t_emb = timestep_embedding(W, time)
    if text_embed is None:
        text_embed = text_embedding(W, cfg, text, n, drop_text=drop_text)
    h = input_embedding(W, x, cond, text_embed, drop_audio_cond=drop_audio_cond)
    ang = rope_angles(n, cfg.get("dim_head", 64))
    if trace is not None:
        trace.update({"t_emb": t_emb, "text_embed": text_embed, "input_embed": h})
    residual = h  # dit.py:217-218
    for i in range(cfg["depth"]):
        h = dit_block(W, i, cfg, h, t_emb, mask, ang, trace=trace)
    if cfg.get("long_skip_connection", False):  # dit.py:153,227-228: Linear(2 dim -> dim, no bias) on cat(x, residual)
        h = _lin(torch.cat([h, residual], dim=-1), W["long_skip_connection.weight"])
    emb = _lin(silu(t_emb), W["norm_out.linear.weight"], W["norm_out.linear.bias"])
    scale, shift = emb.chunk(2, dim=1)  # modules.py:333: (scale, shift) order
    h = _layernorm(h) * (1 + scale)[:, None] + shift[:, None]
    out = _lin(h, W["proj_out.weight"], W["proj_out.bias"])
    if trace is not None:
        trace.update({"final_norm": h, "out": out})
    return out


# ----------------------------------------------------------------------------- 8f-4: UNetT (flat U-Net transformer, the E2-TTS backbone)
def rmsnorm_xt(x, g):
    """x_transformers.RMSNorm as backbones/unett.py:17,146,156,175 uses it (third-party, unpinned): F.normalize(x, dim=-1) * sqrt(dim) * g."""
    return F.normalize(x, dim=-1) * math.sqrt(x.shape[-1]) * g


def unett_forward(W, cfg, x, cond, text, time, drop_audio_cond, drop_text, mask=None, text_embed=None):
    """backbones/unett.py:185-253.  The time embedding is PREPENDED as one more token (:211-213: [b, n, d] -> [b, n+1, d]; the key mask gets a
    leading 1), RoPE spans n+1 positions (:215); the first depth/2 layers push their input on a stack, the last depth/2 pop it and mix it in
    (skip_connect_type "concat": Linear(2d -> d, no bias) on cat(x, skip); "add"; "none") BEFORE the layer's attention (:225-238); every layer is
    x = attn(RMSNorm(x)) + x; x = ff(RMSNorm(x)) + x (:241-242, no AdaLN, no gates); output = proj_out(RMSNorm(x)[:, 1:]) (:246-248)."""
    b, n, _ = x.shape
    if time.ndim == 0:
        time = time.repeat(b)
    t_emb = timestep_embedding(W, time)
    if text_embed is None:
        text_embed = text_embedding(W, cfg, text, n, drop_text=drop_text)
    h = input_embedding(W, x, cond, text_embed, drop_audio_cond=drop_audio_cond)
    h = torch.cat([t_emb[:, None, :], h], dim=1)
    if mask is not None:
        mask = F.pad(mask, (1, 0), value=True)
    ang = rope_angles(n + 1, cfg.get("dim_head", 64))
    depth, sct = cfg["depth"], cfg.get("skip_connect_type", "concat")
    skips = []
    for i in range(depth):
        pre = f"layers.{i}."
        if i < depth // 2:
            skips.append(h)
        else:
            skip = skips.pop()
            if sct == "concat":
                h = _lin(torch.cat([h, skip], dim=-1), W[pre + "0.weight"])
            elif sct == "add":
                h = h + skip
        h = attention(W, pre + "2.", cfg, rmsnorm_xt(h, W[pre + "1.g"]), mask, ang) + h
        n2 = rmsnorm_xt(h, W[pre + "3.g"])
        h = _lin(gelu_tanh(_lin(n2, W[pre + "4.ff.0.0.weight"], W[pre + "4.ff.0.0.bias"])), W[pre + "4.ff.2.weight"], W[pre + "4.ff.2.bias"]) + h
    assert not skips
    h = rmsnorm_xt(h, W["norm_out.g"])[:, 1:, :]
    return _lin(h, W["proj_out.weight"], W["proj_out.bias"])


def unett_param_shapes(cfg, vocab_size, mel_dim=100):
    """Names/shapes of the reference UNetT.state_dict() (unett.py:104-183) for an arch dict (rotary_embed.inv_freq excluded)."""
    D, L = cfg["dim"], cfg["depth"]
    td = cfg.get("text_dim") or mel_dim
    inner, ff = cfg["heads"] * cfg.get("dim_head", 64), int(D * cfg.get("ff_mult", 4))
    sh = {"time_embed.time_mlp.0.weight": (D, 256), "time_embed.time_mlp.0.bias": (D,), "time_embed.time_mlp.2.weight": (D, D),
          "time_embed.time_mlp.2.bias": (D,), "text_embed.text_embed.weight": (vocab_size + 1, td)}
    for i in range(cfg.get("conv_layers", 0)):
        p = f"text_embed.text_blocks.{i}."
        sh.update({p + "dwconv.weight": (td, 1, 7), p + "dwconv.bias": (td,), p + "norm.weight": (td,), p + "norm.bias": (td,),
                   p + "pwconv1.weight": (2 * td, td), p + "pwconv1.bias": (2 * td,), p + "grn.gamma": (1, 1, 2 * td), p + "grn.beta": (1, 1, 2 * td),
                   p + "pwconv2.weight": (td, 2 * td), p + "pwconv2.bias": (td,)})
    sh.update({"input_embed.proj.weight": (D, 2 * mel_dim + td), "input_embed.proj.bias": (D,)})
    for i in (0, 2):
        sh.update({f"input_embed.conv_pos_embed.conv1d.{i}.weight": (D, D // 16, 31), f"input_embed.conv_pos_embed.conv1d.{i}.bias": (D,)})
    for i in range(L):
        p = f"layers.{i}."
        if i >= L // 2 and cfg.get("skip_connect_type", "concat") == "concat":
            sh[p + "0.weight"] = (D, 2 * D)
        sh.update({p + "1.g": (D,), p + "3.g": (D,)})
        for nm in ("to_q", "to_k", "to_v"):
            sh.update({p + f"2.{nm}.weight": (inner, D), p + f"2.{nm}.bias": (inner,)})
        sh.update({p + "2.to_out.0.weight": (D, inner), p + "2.to_out.0.bias": (D,), p + "4.ff.0.0.weight": (ff, D), p + "4.ff.0.0.bias": (ff,),
                   p + "4.ff.2.weight": (D, ff), p + "4.ff.2.bias": (D,)})
    sh.update({"norm_out.g": (D,), "proj_out.weight": (mel_dim, D), "proj_out.bias": (mel_dim,)})
    return sh


def random_unett_weights(cfg, vocab_size, seed=0, mel_dim=100):
    """Deterministic (CPU generator) random UNetT weights: N(0, 1/fan_in) matrices, small biases, norm gains around 1."""
    g = torch.Generator().manual_seed(seed)
    W = {}
    for name, shape in unett_param_shapes(cfg, vocab_size, mel_dim).items():
        if name.endswith(".g") or name.endswith("norm.weight"):
            W[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("text_embed.weight"):
            W[name] = torch.randn(shape, generator=g)
        elif len(shape) == 1 or name.endswith("grn.gamma") or name.endswith("grn.beta"):
            W[name] = 0.05 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            W[name] = torch.randn(shape, generator=g) / math.sqrt(fan_in)
    return W


# ----------------------------------------------------------------------------- 8f-4: MMDiT (joint text / audio attention, SD3-style blocks)
def mmdit_text_embedding(W, cfg, text, drop_text=False):
    """backbones/mmdit.py:30-61: ids + 1 (0 = filler), NOT padded to the frame count; + sinusoidal table (1024 positions, index clamped
    there, modules.py:210-219); filler rows (decided before drop_text zeroes the ids) are zeroed when text_mask_padding."""
    ids = text + 1
    filler = ids == 0
    if drop_text:
        ids = torch.zeros_like(ids)
    x = W["text_embed.text_embed.weight"][ids]
    pos = torch.arange(ids.shape[1]).clamp(max=1023)
    x = x + text_pos_table(x.shape[-1], 1024)[pos][None]
    if cfg.get("text_mask_padding", True):
        x = x.masked_fill(filler[..., None], 0.0)
    return x


def _ada6(W, pre, t_emb):
    return _lin(silu(t_emb), W[pre + "linear.weight"], W[pre + "linear.bias"]).chunk(6, dim=1)


def _ff(W, pre, x):
    return _lin(gelu_tanh(_lin(x, W[pre + "ff.0.0.weight"], W[pre + "ff.0.0.bias"])), W[pre + "ff.2.weight"], W[pre + "ff.2.bias"])


def joint_attention(W, pre, cfg, x, c, mask, ang_x, ang_c, context_pre_only):
    """modules.py:509-606 (JointAttnProcessor, dropout_p forced to 0): x and c have their own q/k/v projections; RoPE is applied to each
    stream with its own positions 0.. (all heads); queries / keys / values are concatenated [x | c] along the sequence, the key mask is the
    frame mask padded with True over the text; outputs are split back, to_out / to_out_c applied, padded x rows zeroed."""
    b, n, _ = x.shape
    nt = c.shape[1]
    h, dh = cfg["heads"], cfg.get("dim_head", 64)
    hs = cfg.get("rope_layout", "adjacent") == "half_split"

    def proj(t, sfx, ang):
        q = _lin(t, W[pre + f"to_q{sfx}.weight"], W[pre + f"to_q{sfx}.bias"]).view(b, -1, h, dh).transpose(1, 2)
        k = _lin(t, W[pre + f"to_k{sfx}.weight"], W[pre + f"to_k{sfx}.bias"]).view(b, -1, h, dh).transpose(1, 2)
        v = _lin(t, W[pre + f"to_v{sfx}.weight"], W[pre + f"to_v{sfx}.bias"]).view(b, -1, h, dh).transpose(1, 2)
        return apply_rope(q, ang, hs), apply_rope(k, ang, hs), v

    qx, kx, vx = proj(x, "", ang_x)
    qc, kc, vc = proj(c, "_c", ang_c)
    q, k, v = torch.cat([qx, qc], dim=2), torch.cat([kx, kc], dim=2), torch.cat([vx, vc], dim=2)
    s = torch.einsum("bhid,bhjd->bhij", q, k) / math.sqrt(dh)
    if mask is not None:
        km = F.pad(mask, (0, nt), value=True)
        s = s.masked_fill(~km[:, None, None, :], float("-inf"))
    o = torch.einsum("bhij,bhjd->bhid", torch.softmax(s, dim=-1), v).transpose(1, 2).reshape(b, n + nt, h * dh)
    ox, oc = o[:, :n], o[:, n:]
    ox = _lin(ox, W[pre + "to_out.0.weight"], W[pre + "to_out.0.bias"])
    oc = None if context_pre_only else _lin(oc, W[pre + "to_out_c.weight"], W[pre + "to_out_c.bias"])
    if mask is not None:
        ox = ox.masked_fill(~mask[..., None], 0.0)
    return ox, oc


def mmdit_forward(W, cfg, x, cond, text, time, drop_audio_cond, drop_text, mask=None, text_embed=None):
    """backbones/mmdit.py:146-190 + MMDiTBlock (modules.py:646-707).  c = text embedding [b, nt, d]; x = Linear(cat(x, cond)) + its position
    conv (:72-79).  Every block: AdaLN of both streams from t, joint attention, gated residual + gated FF on both; the LAST block is
    context_pre_only (c only feeds the attention: AdaLayerNorm_Final (scale, shift), no to_out_c, no FF)."""
    b, n, _ = x.shape
    if time.ndim == 0:
        time = time.repeat(b)
    t_emb = timestep_embedding(W, time)
    c = mmdit_text_embedding(W, cfg, text, drop_text=drop_text) if text_embed is None else text_embed
    if drop_audio_cond:
        cond = torch.zeros_like(cond)
    h = _lin(torch.cat([x, cond], dim=-1), W["audio_embed.linear.weight"], W["audio_embed.linear.bias"])
    y = h.transpose(1, 2)
    for i in (0, 2):
        y = mish(F.conv1d(y, W[f"audio_embed.conv_pos_embed.conv1d.{i}.weight"], W[f"audio_embed.conv_pos_embed.conv1d.{i}.bias"], padding=15, groups=16))
    h = y.transpose(1, 2) + h
    dh = cfg.get("dim_head", 64)
    ang_x, ang_c = rope_angles(n, dh), rope_angles(c.shape[1], dh)
    depth = cfg["depth"]
    for i in range(depth):
        pre = f"transformer_blocks.{i}."
        last = i == depth - 1
        if last:
            sc, sh = _lin(silu(t_emb), W[pre + "attn_norm_c.linear.weight"], W[pre + "attn_norm_c.linear.bias"]).chunk(2, dim=1)
            norm_c = _layernorm(c) * (1 + sc)[:, None] + sh[:, None]
        else:
            c_sh_a, c_sc_a, c_g_a, c_sh_m, c_sc_m, c_g_m = _ada6(W, pre + "attn_norm_c.", t_emb)
            norm_c = _layernorm(c) * (1 + c_sc_a[:, None]) + c_sh_a[:, None]
        x_sh_a, x_sc_a, x_g_a, x_sh_m, x_sc_m, x_g_m = _ada6(W, pre + "attn_norm_x.", t_emb)
        norm_x = _layernorm(h) * (1 + x_sc_a[:, None]) + x_sh_a[:, None]
        ax, ac = joint_attention(W, pre + "attn.", cfg, norm_x, norm_c, mask, ang_x, ang_c, last)
        if last:
            c = None
        else:
            c = c + c_g_a[:, None] * ac
            c = c + c_g_m[:, None] * _ff(W, pre + "ff_c.", _layernorm(c) * (1 + c_sc_m[:, None]) + c_sh_m[:, None])
        h = h + x_g_a[:, None] * ax
        h = h + x_g_m[:, None] * _ff(W, pre + "ff_x.", _layernorm(h) * (1 + x_sc_m[:, None]) + x_sh_m[:, None])
    scale, shift = _lin(silu(t_emb), W["norm_out.linear.weight"], W["norm_out.linear.bias"]).chunk(2, dim=1)
    h = _layernorm(h) * (1 + scale)[:, None] + shift[:, None]
    return _lin(h, W["proj_out.weight"], W["proj_out.bias"])


def mmdit_param_shapes(cfg, vocab_size, mel_dim=100):
    """Names/shapes of the reference MMDiT.state_dict() (mmdit.py:85-130) for an arch dict (rotary_embed.inv_freq excluded)."""
    D, L = cfg["dim"], cfg["depth"]
    inner, ff = cfg["heads"] * cfg.get("dim_head", 64), int(D * cfg.get("ff_mult", 4))
    sh = {"time_embed.time_mlp.0.weight": (D, 256), "time_embed.time_mlp.0.bias": (D,), "time_embed.time_mlp.2.weight": (D, D),
          "time_embed.time_mlp.2.bias": (D,), "text_embed.text_embed.weight": (vocab_size + 1, D),
          "audio_embed.linear.weight": (D, 2 * mel_dim), "audio_embed.linear.bias": (D,)}
    for i in (0, 2):
        sh.update({f"audio_embed.conv_pos_embed.conv1d.{i}.weight": (D, D // 16, 31), f"audio_embed.conv_pos_embed.conv1d.{i}.bias": (D,)})
    for i in range(L):
        p, last = f"transformer_blocks.{i}.", i == L - 1
        sh.update({p + "attn_norm_c.linear.weight": ((2 if last else 6) * D, D), p + "attn_norm_c.linear.bias": ((2 if last else 6) * D,),
                   p + "attn_norm_x.linear.weight": (6 * D, D), p + "attn_norm_x.linear.bias": (6 * D,)})
        for nm in ("to_q", "to_k", "to_v", "to_q_c", "to_k_c", "to_v_c"):
            sh.update({p + f"attn.{nm}.weight": (inner, D), p + f"attn.{nm}.bias": (inner,)})
        sh.update({p + "attn.to_out.0.weight": (D, inner), p + "attn.to_out.0.bias": (D,)})
        for s in (("x",) if last else ("c", "x")):
            sh.update({p + f"ff_{s}.ff.0.0.weight": (ff, D), p + f"ff_{s}.ff.0.0.bias": (ff,), p + f"ff_{s}.ff.2.weight": (D, ff), p + f"ff_{s}.ff.2.bias": (D,)})
        if not last:
            sh.update({p + "attn.to_out_c.weight": (D, inner), p + "attn.to_out_c.bias": (D,)})
    sh.update({"norm_out.linear.weight": (2 * D, D), "norm_out.linear.bias": (2 * D,), "proj_out.weight": (mel_dim, D), "proj_out.bias": (mel_dim,)})
    return sh


def random_mmdit_weights(cfg, vocab_size, seed=0, mel_dim=100):
    """Deterministic (CPU generator) random MMDiT weights: N(0, 1/fan_in) matrices, small biases (AdaLN / output layers live, unlike the
    reference's zero init)."""
    g = torch.Generator().manual_seed(seed)
    W = {}
    for name, shape in mmdit_param_shapes(cfg, vocab_size, mel_dim).items():
        if name.endswith("text_embed.weight"):
            W[name] = torch.randn(shape, generator=g)
        elif len(shape) == 1:
            W[name] = 0.05 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            W[name] = torch.randn(shape, generator=g) / math.sqrt(fan_in)
    return W


# ----------------------------------------------------------------------------- a4-a8, a21: sampler
def lens_to_mask(lens, length=None):
    """utils.py:42-47."""
    length = int(lens.max()) if length is None else length
    return torch.arange(length)[None, :] < lens[:, None]


def sample(W, cfg, cond, text, duration, *, lens=None, steps=32, cfg_strength=1.0, sway_sampling_coef=None,
           seed=None, max_duration=4096, y0=None, method="euler", no_ref_audio=False, return_trajectory=True,
           duplicate_test=False, t_inter=0.1, edit_mask=None):
    """cfm.py:82-208 with cond given as mel [b, nc, 100] and text as id tensor [b, nt] (-1 padded).

    ``y0`` (zero-padded [b, N, 100]) overrides the per-sample ``manual_seed(seed); randn`` of cfm.py:178-183 (the duplicate-test blend of
    cfm.py:187-190 is applied to it either way).  ``edit_mask`` [b, nc] bool: cfm.py:123-125 (speech editing);
    ``duplicate_test`` / ``t_inter``: cfm.py:137-139, 185-191.
    """
    cond = cond.float()
    b, nc, nmel = cond.shape
    if lens is None:
        lens = torch.full((b,), nc, dtype=torch.long)
    cond_mask = lens_to_mask(lens)
    if edit_mask is not None:
        cond_mask = cond_mask & edit_mask
    if isinstance(duration, int):
        duration = torch.full((b,), duration, dtype=torch.long)
    duration = torch.maximum(torch.maximum((text != -1).sum(dim=-1), lens) + 1, duration).clamp(max=max_duration)
    N = int(duration.max())
    if duplicate_test:
        test_cond = F.pad(cond, (0, 0, nc, N - 2 * nc), value=0.0)
    cond = F.pad(cond, (0, 0, 0, N - nc), value=0.0)
    if no_ref_audio:
        cond = torch.zeros_like(cond)
    cond_mask = F.pad(cond_mask, (0, N - cond_mask.shape[-1]), value=False)[..., None]
    step_cond = torch.where(cond_mask, cond, torch.zeros_like(cond))
    mask = lens_to_mask(duration) if b > 1 else None  # cfm.py:152-155

    if y0 is None:
        rows = []
        for d in duration.tolist():
            if seed is not None:
                torch.manual_seed(seed)
            rows.append(F.pad(torch.randn(d, nmel), (0, 0, 0, N - d)))
        y0 = torch.stack(rows)

    fwd = {"UNetT": unett_forward, "MMDiT": mmdit_forward}.get(cfg.get("backbone"), dit_forward)  # plug point A: cfm.py only calls transformer(...)
    if cfg.get("backbone") == "MMDiT":
        te_c = mmdit_text_embedding(W, cfg, text, drop_text=False)
        te_u = mmdit_text_embedding(W, cfg, text, drop_text=True)
    else:
        te_c = text_embedding(W, cfg, text, N, drop_text=False)
        te_u = text_embedding(W, cfg, text, N, drop_text=True)

    def fn(t, x):
        pred = fwd(W, cfg, x, step_cond, text, t, False, False, mask=mask, text_embed=te_c)
        if cfg_strength < 1e-5:
            return pred
        null = fwd(W, cfg, x, step_cond, text, t, True, True, mask=mask, text_embed=te_u)
        return pred + (pred - null) * cfg_strength

    t_start = 0.0
    if duplicate_test:
        t_start = t_inter
        y0 = (1 - t_start) * y0 + t_start * test_cond
        steps = int(steps * (1 - t_start))
    t = time_grid(steps, sway_sampling_coef, t_start=t_start)
    y = y0
    traj = [y0]
    for t0, t1 in zip(t[:-1], t[1:]):  # torchdiffeq fixed grid (a7)
        dt = t1 - t0
        if method == "euler":
            y = y + dt * fn(t0, y)
        elif method == "midpoint":
            half = 0.5 * dt
            y = y + dt * fn(t0 + half, y + fn(t0, y) * half)
        else:
            raise ValueError(method)
        if return_trajectory:
            traj.append(y)
    out = torch.where(cond_mask, cond, y)  # cfm.py:200-202 (rows past duration are NOT zeroed)
    return out, (torch.stack(traj) if return_trajectory else None)


# ----------------------------------------------------------------------------- a3: mel extraction
def hz_to_mel_htk(f):
    return 2595.0 * math.log10(1.0 + f / 700.0)


def mel_filterbank(n_freqs=513, n_mels=100, sr=24000, f_min=0.0, f_max=None):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk') restated (unpinned: torchaudio absent)."""
    f_max = sr / 2 if f_max is None else f_max
    all_freqs = torch.linspace(0, sr // 2, n_freqs)
    m_pts = torch.linspace(hz_to_mel_htk(f_min), hz_to_mel_htk(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.minimum(down, up), min=0.0)  # [n_freqs, n_mels]


def mel_spectrogram(wave, n_fft=1024, hop=256, win=1024, n_mels=100, sr=24000):
    """modules.py:75-101: MelSpectrogram(power=1, center=True, reflect pad, periodic hann) -> clamp(1e-5).log()."""
    window = torch.hann_window(win, periodic=True)
    spec = torch.stft(wave, n_fft, hop_length=hop, win_length=win, window=window, center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True).abs()
    mel = torch.matmul(spec.transpose(-1, -2), mel_filterbank(n_fft // 2 + 1, n_mels, sr)).transpose(-1, -2)
    return mel.clamp(min=1e-5).log()  # [b, n_mels, nw//hop + 1]


# ----------------------------------------------------------------------------- f4: the BigVGAN mel front-end (modules.py:29-72)
def librosa_mel_filterbank(sr=24000, n_fft=1024, n_mels=100, fmin=0.0, fmax=None):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with its defaults htk=False, norm="slaney" -- the call of modules.py:45 -- restated from
    the published algorithm (librosa is absent from the reference tree: parity unpinned at this boundary): n_mels + 2 band edges equally spaced on
    the Slaney / Auditory-toolbox mel scale (linear below 1 kHz at 200/3 Hz per mel, log above with log(6.4)/27 per mel), triangular weights over
    the rfft bin frequencies, each filter scaled by 2 / (its band width in Hz).  float64 arithmetic, float32 result [n_mels, n_fft // 2 + 1]."""
    import numpy as np
    fmax = sr / 2.0 if fmax is None else fmax
    f_sp, min_log_hz, logstep = 200.0 / 3.0, 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float64)
    for i in range(n_mels):
        weights[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    weights *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return weights.astype(np.float32)


def bigvgan_mel_spectrogram(wave, n_fft=1024, n_mels=100, sr=24000, hop=256, win=1024, fmin=0, fmax=None):
    """modules.py:29-72 (get_bigvgan_mel_spectrogram): reflect padding of (n_fft - hop) / 2 on both sides, torch.stft(center=False, periodic Hann),
    sqrt(re^2 + im^2 + 1e-9), the librosa filterbank, log(clamp(min=1e-5)).  wave [b, nw] -> [b, n_mels, (nw + 2 pad - n_fft) // hop + 1]."""
    mel_basis = torch.from_numpy(librosa_mel_filterbank(sr, n_fft, n_mels, fmin, fmax)).float()
    pad = (n_fft - hop) // 2
    w = F.pad(wave.float().unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    spec = torch.stft(w, n_fft, hop_length=hop, win_length=win, window=torch.hann_window(win), center=False, pad_mode="reflect", normalized=False,
                      onesided=True, return_complex=True)
    spec = torch.sqrt(torch.view_as_real(spec).pow(2).sum(-1) + 1e-9)
    return torch.log(torch.clamp(torch.matmul(mel_basis, spec), min=1e-5))


# ----------------------------------------------------------------------------- f3: sample-rate conversion (unpinned: torchaudio absent)
def resample(wave, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """torchaudio.transforms.Resample(orig, new) with its defaults (resampling_method "sinc_interp_hann", width 6, rolloff 0.99), the call
    of infer/f5tts_wrapper.py:338-341 / utils_infer.py:431-433, restated from the published algorithm (torchaudio.functional.functional
    ``_get_sinc_resample_kernel`` + ``_apply_sinc_resample_kernel``) as a DIRECT polyphase sum in float64 numpy -- on purpose not the
    strided conv1d the product's host branch uses, so that the two agree only if both are right:

        g = gcd; orig, new = orig/g, new/g;  base = min(orig, new) * rolloff;  width = ceil(lpw * orig / base)
        out[i * new + j] = sum_{k = 0}^{2 width + orig - 1}  xpad[i * orig + k] * h_j[k]          xpad = x padded (width, width + orig)
        h_j[k] = sinc(pi * t) * cos(pi * t / (2 lpw))^2 * base / orig,   t = clamp(((k - width) / orig - j / new) * base, -lpw, lpw)
        length = ceil(new * n / orig)

    wave: float tensor [..., n] -> float32 tensor [..., length]."""
    import numpy as np
    if int(orig_freq) == int(new_freq):
        return wave
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    lpw = float(lowpass_filter_width)
    base = min(orig, new) * rolloff
    width = int(math.ceil(lpw * orig / base))
    taps = 2 * width + orig
    k = np.arange(taps, dtype=np.float64)
    h = np.empty((new, taps), dtype=np.float64)
    for j in range(new):
        t = np.clip(((k - width) / orig - j / new) * base, -lpw, lpw)
        win = np.cos(t * math.pi / lpw / 2.0) ** 2
        tp = t * math.pi
        with np.errstate(invalid="ignore", divide="ignore"):
            sinc = np.where(tp == 0.0, 1.0, np.sin(tp) / tp)
        h[j] = sinc * win * (base / orig)
    h = h.astype(np.float32).astype(np.float64)  # torchaudio casts the kernel to the waveform dtype (float32) before convolving
    x = wave.detach().cpu().to(torch.float64).numpy()
    shape = x.shape
    x = x.reshape(-1, shape[-1])
    n = x.shape[-1]
    xp = np.pad(x, ((0, 0), (width, width + orig)))
    steps = (xp.shape[-1] - taps) // orig + 1
    target = int(math.ceil(new * n / orig))
    out = np.zeros((x.shape[0], steps * new), dtype=np.float64)
    for i in range(steps):
        seg = xp[:, i * orig: i * orig + taps]  # [rows, taps]
        out[:, i * new: (i + 1) * new] = seg @ h.T
    out = out[:, :target].reshape(shape[:-1] + (target,))
    return torch.from_numpy(out.astype(np.float32))


# ----------------------------------------------------------------------------- a22: Vocos (unpinned)
def vocos_backbone(V, mel):
    """vocos VocosBackbone (source absent; restated): conv k7 -> LN -> 8x ConvNeXt(layer-scale) -> LN."""
    x = F.conv1d(mel, V["backbone.embed.weight"], V["backbone.embed.bias"], padding=3).transpose(1, 2)
    x = _layernorm(x) * V["backbone.norm.weight"] + V["backbone.norm.bias"]
    i = 0
    while f"backbone.convnext.{i}.dwconv.weight" in V:
        p = f"backbone.convnext.{i}."
        d = x.shape[-1]
        y = F.conv1d(x.transpose(1, 2), V[p + "dwconv.weight"], V[p + "dwconv.bias"], padding=3, groups=d).transpose(1, 2)
        y = _layernorm(y) * V[p + "norm.weight"] + V[p + "norm.bias"]
        y = _lin(gelu_erf(_lin(y, V[p + "pwconv1.weight"], V[p + "pwconv1.bias"])), V[p + "pwconv2.weight"], V[p + "pwconv2.bias"])
        x = x + V[p + "gamma"] * y
        i += 1
    return _layernorm(x) * V["backbone.final_layer_norm.weight"] + V["backbone.final_layer_norm.bias"]


def istft_center(spec_re, spec_im, n_fft=1024, hop=256):
    """torch.istft(center=True, hann periodic window) written out: irfft -> window -> overlap-add ->
    divide by overlap-added window^2 -> trim n_fft/2 each side.  spec: [b, n_fft/2+1, T] -> [b, (T-1)*hop]."""
    b, _, T = spec_re.shape
    window = torch.hann_window(n_fft, periodic=True)
    frames = torch.fft.irfft(torch.complex(spec_re, spec_im), n=n_fft, dim=1) * window[None, :, None]
    out_len = n_fft + hop * (T - 1)
    y = torch.zeros(b, out_len)
    env = torch.zeros(out_len)
    w2 = window * window
    for t in range(T):
        y[:, t * hop: t * hop + n_fft] += frames[:, :, t]
        env[t * hop: t * hop + n_fft] += w2
    half = n_fft // 2
    return y[:, half: out_len - half] / env[half: out_len - half]


def vocos_decode(V, mel):
    """vocos Vocos.decode (wrapper.py:524): backbone -> Linear(512->1026) -> exp/clip(1e2), cos/sin -> ISTFT."""
    h = _lin(vocos_backbone(V, mel), V["head.out.weight"], V["head.out.bias"]).transpose(1, 2)
    mag, ph = h.chunk(2, dim=1)
    mag = torch.exp(mag).clamp(max=1e2)
    return istft_center(mag * torch.cos(ph), mag * torch.sin(ph))


# ----------------------------------------------------------------------------- a1: the composed generate() chain
def cross_fade_concat(waves, cross_fade_duration, sr=24000):
    """f5tts_wrapper.py:541-575: linear cross-fade of consecutive chunk waveforms (numpy, float64 ramps as np.linspace gives them)."""
    import numpy as np
    if cross_fade_duration <= 0:
        return np.concatenate(waves)
    final = waves[0]
    for nxt in waves[1:]:
        n = min(int(cross_fade_duration * sr), len(final), len(nxt))
        if n <= 0:
            final = np.concatenate([final, nxt])
            continue
        final = np.concatenate([final[:-n], final[-n:] * np.linspace(1, 0, n) + nxt[:n] * np.linspace(0, 1, n), nxt[n:]])
    return final


def generate_chain(W, cfg, V, ref_audio, ref_text_nbytes, chunks, *, nfe_step=32, cfg_strength=2.0, sway_sampling_coef=-1.0, speed=1.0,
                   cross_fade_duration=0.15, target_rms=0.1, hop=256, sr=24000, method="euler"):
    """F5TTSWrapper.generate (f5tts_wrapper.py:476-575) as the reference's CPU path runs it, from the stored prompt waveform on:
    prompt mel (modules.py:75-101) -> per text chunk: duration rule (:500-504), CFM.sample (cfm.py:82-208; the initial noise is drawn from
    torch's GLOBAL CPU generator, one randn(duration, 100) per chunk in chunk order, cfm.py:178-183 without a seed) -> cut at ref_audio_len
    (:519) -> Vocos.decode (:524) -> rms rule on the stored prompt (:529-531) -> cross-fade (:541-575).
    ref_audio [1, nw] f32 at 24 kHz (the wrapper's ref_audio_processed); chunks = [(token ids [1, nt] incl. the reference text, number of utf-8
    bytes of the chunk's own text)].  Returns (wave, [mel [100, frames] per chunk])."""
    ref_audio_len = ref_audio.shape[-1] // hop
    cond = mel_spectrogram(ref_audio).permute(0, 2, 1)  # cfm.py:110-112: a raw-wave prompt becomes [1, nw // hop + 1, 100]
    waves, mels = [], []
    for ids, gen_nbytes in chunks:
        local_speed = 0.3 if gen_nbytes < 10 else speed
        duration = ref_audio_len + int(ref_audio_len / ref_text_nbytes * gen_nbytes / local_speed)
        out, _ = sample(W, cfg, cond, ids, duration, steps=nfe_step, cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef,
                        method=method, return_trajectory=False)
        gen = out.float()[:, ref_audio_len:, :].permute(0, 2, 1)
        wave = vocos_decode(V, gen)
        rms = torch.sqrt(torch.mean(torch.square(ref_audio)))
        if rms < target_rms:
            wave = wave * rms / target_rms
        waves.append(wave.squeeze().numpy())
        mels.append(gen.squeeze(0).numpy())
    return cross_fade_concat(waves, cross_fade_duration, sr), mels


# ----------------------------------------------------------------------------- f4: the BigVGAN vocoder (parity UNPINNED)
# The reference loads it from a third-party checkout (infer/utils_infer.py:125-138: `from third_party.BigVGAN import bigvgan`,
# BigVGAN.from_pretrained("nvidia/bigvgan_v2_24khz_100band_256x"), remove_weight_norm()) that is ABSENT from /root/reference, and calls
# `vocoder(mel)` (f5tts_wrapper.py:526, eval_infer_batch.py:189).  What follows restates the PUBLISHED BigVGAN-v2 generator (NVIDIA/BigVGAN
# bigvgan.py, activations.py, alias_free_activation/torch/{act,resample,filter}.py) as recalled: conv_pre k7 -> per stage ConvTranspose1d(k, u,
# padding (k - u) / 2) then the mean of three AMPBlock1 (kernel 3 / 7 / 11, dilations 1, 3, 5; anti-aliased SnakeBeta in front of every conv) ->
# anti-aliased SnakeBeta -> conv_post k7 -> clamp(-1, 1) (tanh when use_tanh_at_final).  Nothing pins it: no source, no checkpoint, no reference test.
BIGVGAN_V2_24K_100BAND_256X = dict(num_mels=100, upsample_initial_channel=1536, upsample_rates=[4, 4, 2, 2, 2, 2], upsample_kernel_sizes=[8, 8, 4, 4, 4, 4],
                                   resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], snake_logscale=True,
                                   use_tanh_at_final=False, use_bias_at_final=False)


def kaiser_sinc_filter1d(cutoff, half_width, kernel_size):
    """alias_free_activation/torch/filter.py: Kaiser-windowed sinc low-pass, normalised to unit sum -> [kernel_size]."""
    even = kernel_size % 2 == 0
    half_size = kernel_size // 2
    delta_f = 4 * half_width
    A = 2.285 * (half_size - 1) * math.pi * delta_f + 7.95
    beta = 0.1102 * (A - 8.7) if A > 50.0 else (0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0) if A >= 21.0 else 0.0)
    window = torch.kaiser_window(kernel_size, beta=beta, periodic=False)
    time = (torch.arange(-half_size, half_size) + 0.5) if even else (torch.arange(kernel_size) - half_size)
    filt = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    return filt / filt.sum()


def _aa_snake(x, alpha, beta, up_f, dn_f, logscale):
    """Activation1d(SnakeBeta): UpSample1d(2, 12) -> x + sin^2(alpha x) / (beta + 1e-9) -> DownSample1d(2, 12).  x [b, C, T]."""
    C = x.shape[1]
    ratio, k = 2, up_f.numel()
    pad = k // ratio - 1
    pad_left, pad_right = pad * ratio + (k - ratio) // 2, pad * ratio + (k - ratio + 1) // 2
    u = F.pad(x, (pad, pad), mode="replicate")
    u = ratio * F.conv_transpose1d(u, up_f.view(1, 1, -1).expand(C, -1, -1), stride=ratio, groups=C)[..., pad_left:-pad_right]
    a, b = alpha.view(1, -1, 1), beta.view(1, -1, 1)
    if logscale:
        a, b = torch.exp(a), torch.exp(b)
    u = u + (1.0 / (b + 1e-9)) * torch.sin(u * a) ** 2
    kd = dn_f.numel()
    u = F.pad(u, (kd // 2 - int(kd % 2 == 0), kd // 2), mode="replicate")
    return F.conv1d(u, dn_f.view(1, 1, -1).expand(C, -1, -1), stride=ratio, groups=C)


def bigvgan_forward(W, hp, mel):
    """mel [b, num_mels, T] -> wave [b, 1, T * prod(upsample_rates)].  W: state dict with weight norm REMOVED (`...weight`, `...bias`)."""
    up_f = W.get("aa_up_filter", kaiser_sinc_filter1d(0.25, 0.3, 12))
    dn_f = W.get("aa_down_filter", kaiser_sinc_filter1d(0.25, 0.3, 12))
    ls = bool(hp.get("snake_logscale", True))
    nk = len(hp["resblock_kernel_sizes"])
    x = F.conv1d(mel.float(), W["conv_pre.weight"], W["conv_pre.bias"], padding=3)
    for i, (u, k) in enumerate(zip(hp["upsample_rates"], hp["upsample_kernel_sizes"])):
        x = F.conv_transpose1d(x, W[f"ups.{i}.0.weight"], W[f"ups.{i}.0.bias"], stride=u, padding=(k - u) // 2)
        xs = None
        for j, (ks, dil) in enumerate(zip(hp["resblock_kernel_sizes"], hp["resblock_dilation_sizes"])):
            p = f"resblocks.{i * nk + j}."
            y = x
            for t, d in enumerate(dil):
                xt = _aa_snake(y, W[p + f"activations.{2 * t}.act.alpha"], W[p + f"activations.{2 * t}.act.beta"], up_f, dn_f, ls)
                xt = F.conv1d(xt, W[p + f"convs1.{t}.weight"], W[p + f"convs1.{t}.bias"], dilation=d, padding=(ks * d - d) // 2)
                xt = _aa_snake(xt, W[p + f"activations.{2 * t + 1}.act.alpha"], W[p + f"activations.{2 * t + 1}.act.beta"], up_f, dn_f, ls)
                xt = F.conv1d(xt, W[p + f"convs2.{t}.weight"], W[p + f"convs2.{t}.bias"], padding=(ks - 1) // 2)
                y = xt + y
            xs = y if xs is None else xs + y
        x = xs / nk
    x = _aa_snake(x, W["activation_post.act.alpha"], W["activation_post.act.beta"], up_f, dn_f, ls)
    x = F.conv1d(x, W["conv_post.weight"], W.get("conv_post.bias"), padding=3)
    return torch.tanh(x) if hp.get("use_tanh_at_final", True) else torch.clamp(x, -1.0, 1.0)


def bigvgan_param_shapes(hp):
    C0 = hp["upsample_initial_channel"]
    shapes = {"conv_pre.weight": (C0, hp["num_mels"], 7), "conv_pre.bias": (C0,)}
    nk = len(hp["resblock_kernel_sizes"])
    ch = C0
    for i, (u, k) in enumerate(zip(hp["upsample_rates"], hp["upsample_kernel_sizes"])):
        shapes[f"ups.{i}.0.weight"] = (ch, ch // 2, k)
        shapes[f"ups.{i}.0.bias"] = (ch // 2,)
        ch //= 2
        for j, (ks, dil) in enumerate(zip(hp["resblock_kernel_sizes"], hp["resblock_dilation_sizes"])):
            p = f"resblocks.{i * nk + j}."
            for t in range(len(dil)):
                for cv in ("convs1", "convs2"):
                    shapes[p + f"{cv}.{t}.weight"] = (ch, ch, ks)
                    shapes[p + f"{cv}.{t}.bias"] = (ch,)
            for a in range(2 * len(dil)):
                shapes[p + f"activations.{a}.act.alpha"] = (ch,)
                shapes[p + f"activations.{a}.act.beta"] = (ch,)
    shapes["activation_post.act.alpha"] = (ch,)
    shapes["activation_post.act.beta"] = (ch,)
    shapes["conv_post.weight"] = (1, ch, 7)
    if hp.get("use_bias_at_final", True):
        shapes["conv_post.bias"] = (1,)
    return shapes


def random_bigvgan_weights(hp, seed=0):
    g = torch.Generator().manual_seed(seed)
    W = {}
    for name, shape in bigvgan_param_shapes(hp).items():
        if name.endswith(".alpha") or name.endswith(".beta"):
            W[name] = torch.randn(shape, generator=g) * 0.3  # (log scale: exp(.) around 1)
        elif name.endswith(".bias"):
            W[name] = torch.randn(shape, generator=g) * 0.05
        else:
            fan_in = shape[1] * shape[2] if not name.startswith("ups.") else shape[0] * shape[2] / 2
            W[name] = torch.randn(shape, generator=g) * (1.0 / math.sqrt(max(fan_in, 1.0)))
    return W


# ----------------------------------------------------------------------------- duration predictor
def duration_predictor(W, tokens, mask, add_one=True, prefix="", g_cond=None):
    """DurationPredictor.forward (model/duration_predictor.py:28-46; phoneme_forward :48-68 with add_one=False):
    Embedding(tokens+1) -> [Conv1d(k, pad k//2)(x*mask) -> relu -> GroupNorm(1 group: over all channels AND positions, eps 1e-5)] x2
    -> Conv1d(F->1, 1)(x*mask) * mask.  Dropout is the identity at inference.  tokens [b, nt] (pad -1), mask [b, nt] -> [b, 1, nt].
    ``g_cond`` [b, gin, 1] or [b, gin, nt]: the optional speaker conditioning of a net built with gin_channels != 0."""
    F_ = torch.nn.functional
    g = lambda n: W[prefix + n].float()
    ids = tokens.long() + (1 if add_one else 0)
    x = g("text_embed.weight")[ids].transpose(1, 2)  # [b, C, nt]
    if g_cond is not None:  # speaker conditioning (:25-26, :33-35): x + Conv1d(gin -> C, 1)(g), g [b, gin, 1 | nt]
        x = x + F_.conv1d(g_cond.float(), g("cond.weight"), g("cond.bias"))
    m = mask.float().unsqueeze(1)  # [b, 1, nt]
    k = g("conv_1.weight").shape[-1]
    for i in ("1", "2"):
        x = F_.conv1d(x * m, g(f"conv_{i}.weight"), g(f"conv_{i}.bias"), padding=k // 2)
        x = F_.group_norm(torch.relu(x), 1, g(f"norm_{i}.weight"), g(f"norm_{i}.bias"), eps=1e-5)
    x = F_.conv1d(x * m, g("proj.weight"), g("proj.bias"))
    return x * m


# ----------------------------------------------------------------------------- synthetic weights
def dit_param_shapes(cfg, vocab_size, mel_dim=100):
    """Names/shapes of DiT.state_dict() for a given arch dict."""
    D, L, td = cfg["dim"], cfg["depth"], cfg.get("text_dim") or mel_dim
    inner = cfg["heads"] * cfg.get("dim_head", 64)
    ff = int(D * cfg.get("ff_mult", 4))
    s = {"time_embed.time_mlp.0.weight": (D, 256), "time_embed.time_mlp.0.bias": (D,),
         "time_embed.time_mlp.2.weight": (D, D), "time_embed.time_mlp.2.bias": (D,),
         "text_embed.text_embed.weight": (vocab_size + 1, td)}
    for i in range(cfg.get("conv_layers", 0)):
        p = f"text_embed.text_blocks.{i}."
        s.update({p + "dwconv.weight": (td, 1, 7), p + "dwconv.bias": (td,), p + "norm.weight": (td,), p + "norm.bias": (td,),
                  p + "pwconv1.weight": (2 * td, td), p + "pwconv1.bias": (2 * td,), p + "grn.gamma": (1, 1, 2 * td),
                  p + "grn.beta": (1, 1, 2 * td), p + "pwconv2.weight": (td, 2 * td), p + "pwconv2.bias": (td,)})
    s.update({"input_embed.proj.weight": (D, 2 * mel_dim + td), "input_embed.proj.bias": (D,)})
    for i in (0, 2):
        s.update({f"input_embed.conv_pos_embed.conv1d.{i}.weight": (D, D // 16, 31), f"input_embed.conv_pos_embed.conv1d.{i}.bias": (D,)})
    for i in range(L):
        p = f"transformer_blocks.{i}."
        s.update({p + "attn_norm.linear.weight": (6 * D, D), p + "attn_norm.linear.bias": (6 * D,)})
        for nm in ("to_q", "to_k", "to_v"):
            s.update({p + f"attn.{nm}.weight": (inner, D), p + f"attn.{nm}.bias": (inner,)})
        s.update({p + "attn.to_out.0.weight": (D, inner), p + "attn.to_out.0.bias": (D,),
                  p + "ff.ff.0.0.weight": (ff, D), p + "ff.ff.0.0.bias": (ff,), p + "ff.ff.2.weight": (D, ff), p + "ff.ff.2.bias": (D,)})
        if cfg.get("qk_norm") == "rms_norm":  # modules.py:394-396
            s.update({p + "attn.q_norm.weight": (cfg.get("dim_head", 64),), p + "attn.k_norm.weight": (cfg.get("dim_head", 64),)})
    if cfg.get("long_skip_connection", False):  # dit.py:153
        s["long_skip_connection.weight"] = (D, 2 * D)
    s.update({"norm_out.linear.weight": (2 * D, D), "norm_out.linear.bias": (2 * D,),
              "proj_out.weight": (mel_dim, D), "proj_out.bias": (mel_dim,)})
    return s


def random_dit_weights(cfg, vocab_size, seed=0, mel_dim=100):
    """Seeded synthetic weights (no checkpoint exists offline): N(0, 1/fan_in)-style for matrices, small biases,
    zero-init tensors of the reference re-randomised so every path is live."""
    g = torch.Generator().manual_seed(seed)
    W = {}
    for name, shape in dit_param_shapes(cfg, vocab_size, mel_dim).items():
        if name.endswith("norm.weight"):
            W[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("bias") or "grn." in name:
            W[name] = 0.05 * torch.randn(shape, generator=g)
        elif name == "text_embed.text_embed.weight":
            W[name] = torch.randn(shape, generator=g)
        elif "attn_norm.linear.weight" in name or "norm_out.linear.weight" in name:
            W[name] = torch.randn(shape, generator=g) * (0.5 / math.sqrt(shape[1]))
        else:
            fan_in = math.prod(shape[1:])
            W[name] = torch.randn(shape, generator=g) / math.sqrt(fan_in)
    return W


def fwd_1024_inputs(which):
    """Inputs of base_fwd_1024.npz, regenerated from seeds by the tests (CPU generator: deterministic), so only outputs are stored.
    which = "b1": B = 1, N = 1024, no key mask (cfm.py:152-155: batch 1 builds none) -- the single-utterance production shape;
    which = "b2": B = 2, N = 1000, durations 1000 / 870 (key mask; ragged 256-row tiles, partial attention key tiles)."""
    V = 2545
    g = torch.Generator().manual_seed({"b1": 1024, "b2": 1000}[which])
    B, N, dur = (1, 1024, [1024]) if which == "b1" else (2, 1000, [1000, 870])
    nc = N // 3
    x = torch.randn(B, N, 100, generator=g)
    cond = (torch.randn(B, N, 100, generator=g) * 2 - 3).clamp(float(math.log(1e-5)), 3.0)
    cond[:, nc:] = 0
    text = torch.randint(0, V, (B, N // 6), generator=g)
    if B > 1:
        text[1, 140:] = -1
        x[1, dur[1]:] = 0
    mask = None if B == 1 else lens_to_mask(torch.tensor(dur))
    t = torch.tensor({"b1": 0.6180, "b2": 0.1415}[which])
    return x, cond, text, mask, t, dur


def random_vocos_weights(seed=0, dim=512, inter=1536, layers=8, n_mels=100, n_fft=1024):
    g = torch.Generator().manual_seed(seed)
    V = {"backbone.embed.weight": torch.randn(dim, n_mels, 7, generator=g) / math.sqrt(n_mels * 7),
         "backbone.embed.bias": 0.05 * torch.randn(dim, generator=g),
         "backbone.norm.weight": 1 + 0.1 * torch.randn(dim, generator=g), "backbone.norm.bias": 0.05 * torch.randn(dim, generator=g),
         "backbone.final_layer_norm.weight": 1 + 0.1 * torch.randn(dim, generator=g),
         "backbone.final_layer_norm.bias": 0.05 * torch.randn(dim, generator=g),
         "head.out.weight": torch.randn(n_fft + 2, dim, generator=g) / math.sqrt(dim), "head.out.bias": 0.05 * torch.randn(n_fft + 2, generator=g)}
    for i in range(layers):
        p = f"backbone.convnext.{i}."
        V.update({p + "dwconv.weight": torch.randn(dim, 1, 7, generator=g) / math.sqrt(7), p + "dwconv.bias": 0.05 * torch.randn(dim, generator=g),
                  p + "norm.weight": 1 + 0.1 * torch.randn(dim, generator=g), p + "norm.bias": 0.05 * torch.randn(dim, generator=g),
                  p + "pwconv1.weight": torch.randn(inter, dim, generator=g) / math.sqrt(dim), p + "pwconv1.bias": 0.05 * torch.randn(inter, generator=g),
                  p + "pwconv2.weight": torch.randn(dim, inter, generator=g) / math.sqrt(inter), p + "pwconv2.bias": 0.05 * torch.randn(dim, generator=g),
                  p + "gamma": 0.125 + 0.02 * torch.randn(dim, generator=g)})
    return V
