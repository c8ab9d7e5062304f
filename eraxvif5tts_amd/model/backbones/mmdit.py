"""MMDiT backbone (SD3-style joint text / audio transformer; plug point A of the reference) executed by libf5hip on MI355X.

Drop-in for ``f5_tts.model.backbones.mmdit.MMDiT`` (reference mmdit.py:85-190): same constructor kwargs, the same ``state_dict()`` names and
shapes (``audio_embed.linear`` / ``.conv_pos_embed``, ``transformer_blocks.<i>.attn_norm_x|attn_norm_c.linear``, ``.attn.to_q|to_k|to_v[_c]``,
``.attn.to_out.0``, ``.attn.to_out_c``, ``.ff_x.ff`` / ``.ff_c.ff`` -- the last block is ``context_pre_only``: a 2-chunk ``attn_norm_c`` and
no ``to_out_c`` / ``ff_c``), the same ``forward(x, cond, text, time, drop_audio_cond, drop_text, mask=None, cache=False)``, ``clear_cache()``
and ``.dim`` -- what ``CFM`` touches (cfm.py:65,164-172,198).

All arithmetic runs in the HIP library (``include/f5hip.h``, ``F5_BACKBONE_MMDIT``): the text is a residual stream of its own length nt
(embedding + sinusoidal table of 1024 positions, mmdit.py:30-61), every block modulates both streams from the time embedding, projects each with
its own weights (RoPE per stream from position 0), runs ONE attention over the joint [frames | text] sequence of every utterance (text keys are
never masked, modules.py:573) and applies the gated updates to both streams.  The whole ODE loop of ``CFM.sample`` (CFG doubling, hipGraph
replay) is shared with the DiT backbone: ``native_sample`` is inherited.
"""
from __future__ import annotations

import torch
from torch import nn

from ... import _lib
from .dit import DiT


def _mmdit_spec(dim, depth, heads, dim_head, ff_inner, mel_dim, vocab):
    """(name, shape, init) for every tensor of the reference MMDiT.state_dict() (mmdit.py:99-130; zero-initialised AdaLN / output layers :132-144)."""
    inner = heads * dim_head
    spec = [("time_embed.time_mlp.0.weight", (dim, 256), "linear"), ("time_embed.time_mlp.0.bias", (dim,), ("bias", 256)),
            ("time_embed.time_mlp.2.weight", (dim, dim), "linear"), ("time_embed.time_mlp.2.bias", (dim,), ("bias", dim)),
            ("text_embed.text_embed.weight", (vocab + 1, dim), "normal"),
            ("audio_embed.linear.weight", (dim, 2 * mel_dim), "linear"), ("audio_embed.linear.bias", (dim,), ("bias", 2 * mel_dim))]
    for i in (0, 2):
        spec += [(f"audio_embed.conv_pos_embed.conv1d.{i}.weight", (dim, dim // 16, 31), "linear"),
                 (f"audio_embed.conv_pos_embed.conv1d.{i}.bias", (dim,), ("bias", dim // 16 * 31))]
    for i in range(depth):
        p, last = f"transformer_blocks.{i}.", i == depth - 1
        nc = 2 if last else 6
        spec += [(p + "attn_norm_c.linear.weight", (nc * dim, dim), "zeros"), (p + "attn_norm_c.linear.bias", (nc * dim,), "zeros"),
                 (p + "attn_norm_x.linear.weight", (6 * dim, dim), "zeros"), (p + "attn_norm_x.linear.bias", (6 * dim,), "zeros")]
        for nm in ("to_q", "to_k", "to_v", "to_q_c", "to_k_c", "to_v_c"):
            spec += [(p + f"attn.{nm}.weight", (inner, dim), "linear"), (p + f"attn.{nm}.bias", (inner,), ("bias", dim))]
        spec += [(p + "attn.to_out.0.weight", (dim, inner), "linear"), (p + "attn.to_out.0.bias", (dim,), ("bias", inner))]
        if not last:
            spec += [(p + "attn.to_out_c.weight", (dim, inner), "linear"), (p + "attn.to_out_c.bias", (dim,), ("bias", inner))]
        for s in (("x",) if last else ("c", "x")):
            spec += [(p + f"ff_{s}.ff.0.0.weight", (ff_inner, dim), "linear"), (p + f"ff_{s}.ff.0.0.bias", (ff_inner,), ("bias", dim)),
                     (p + f"ff_{s}.ff.2.weight", (dim, ff_inner), "linear"), (p + f"ff_{s}.ff.2.bias", (dim,), ("bias", ff_inner))]
    spec += [("norm_out.linear.weight", (2 * dim, dim), "zeros"), ("norm_out.linear.bias", (2 * dim,), "zeros"),
             ("proj_out.weight", (mel_dim, dim), "zeros"), ("proj_out.bias", (mel_dim,), "zeros")]
    return spec


class MMDiT(DiT):
    BACKBONE = _lib.F5_BACKBONE_MMDIT

    def __init__(self, *, dim, depth=8, heads=8, dim_head=64, dropout=0.1, ff_mult=4, mel_dim=100, text_num_embeds=256, text_mask_padding=True,
                 qk_norm=None, precision=None, rope_layout=None):
        nn.Module.__init__(self)
        self.checkpoint_activations = False
        # text_dim = dim: TextEmbedding(dim, ...) (mmdit.py:101); no ConvNeXt text blocks; RoPE on every head (JointAttnProcessor)
        self._setup(dim=dim, depth=depth, heads=heads, dim_head=dim_head, ff_mult=ff_mult, mel_dim=mel_dim, text_num_embeds=text_num_embeds,
                    text_dim=dim, text_mask_padding=text_mask_padding, qk_norm=qk_norm, conv_layers=0, pe_attn_head=None, precision=precision,
                    rope_layout=rope_layout)

    def _spec(self):
        return _mmdit_spec(self.dim, self.depth, self.heads, self.dim_head, self.ff_inner, self.mel_dim, self.text_num_embeds)

    def _plan_seq(self, seq_len, text_len):
        return max(seq_len, text_len)  # the text stream's buffers are sized by the plan's sequence bound (CFM.sample always has text_len < frames, cfm.py:131)

    def _text_embed(self, plan, text, seq_len, drop_text):
        """mmdit.py:40-61: [b, nt, dim] -- the text keeps its own length (``seq_len`` only names the plan)."""
        lib = _lib.load()
        b, nt = text.shape
        ids = text.to(device="cuda", dtype=torch.int32).contiguous()
        out = torch.empty(b, nt, self.dim, device="cuda", dtype=torch.float32)
        _lib.check(lib.f5_text_embed(plan, b, seq_len, _lib.ptr(ids), nt, int(bool(drop_text)), _lib.ptr(out), _lib.stream_ptr()), "text_embed")
        return out

    def _native_forward(self, lib, plan, batch, seq_len, xf, cf, text_embed, tf, drop_audio_cond, mk, out):
        _lib.check(lib.f5_mmdit_forward(plan, batch, seq_len, text_embed.shape[1], _lib.ptr(xf), _lib.ptr(cf), _lib.ptr(text_embed), _lib.ptr(tf),
                                        drop_audio_cond, _lib.ptr(mk), _lib.ptr(out), _lib.stream_ptr()), "mmdit_forward")
