"""UNetT backbone (the E2-TTS flat U-Net transformer; plug point A of the reference) executed by libf5hip on MI355X.

Drop-in for ``f5_tts.model.backbones.unett.UNetT`` (reference unett.py:103-253): same constructor kwargs, the same ``state_dict()`` names and
shapes (``layers.<i>.0.weight`` skip projections of the later half, ``layers.<i>.1.g`` / ``.3.g`` / ``norm_out.g`` RMSNorm gains,
``layers.<i>.2.*`` attention, ``layers.<i>.4.ff.*`` feed-forward), the same ``forward(x, cond, text, time, drop_audio_cond, drop_text, mask=None,
cache=False)``, ``clear_cache()`` and ``.dim`` -- what ``CFM`` touches (cfm.py:65,164-172,198).  ``configs/E2TTS_Base.yaml`` / ``E2TTS_Small.yaml``
select it with ``model.backbone: UNetT``.

All arithmetic runs in the HIP library (``include/f5hip.h``, ``F5_BACKBONE_UNETT``): the time embedding is prepended as one more token, every layer
is RMSNorm -> attention -> residual, RMSNorm -> feed-forward -> residual, the first depth/2 layers push their input on a stack and the last
depth/2 mix it back in (concat + Linear, add, or none).  The whole ODE loop of ``CFM.sample`` (CFG doubling, hipGraph replay) is shared with the
DiT backbone: ``native_sample`` is inherited.  Third-party arithmetic restated from the published algorithm, as for the DiT: x_transformers'
``RMSNorm`` (``F.normalize(x) * sqrt(dim) * g``) and ``RotaryEmbedding`` -- parity unpinned at those two boundaries.
"""
from __future__ import annotations

from torch import nn

from ... import _lib
from .dit import DiT


def _unett_spec(dim, depth, heads, dim_head, ff_inner, mel_dim, vocab, text_dim, conv_layers, skip_connect_type):
    """(name, shape, init) for every tensor of the reference UNetT.state_dict() (unett.py:104-183)."""
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
        p = f"layers.{i}."
        if i >= depth // 2 and skip_connect_type == "concat":
            spec += [(p + "0.weight", (dim, 2 * dim), "linear")]
        spec += [(p + "1.g", (dim,), "ones")]
        for nm in ("to_q", "to_k", "to_v"):
            spec += [(p + f"2.{nm}.weight", (inner, dim), "linear"), (p + f"2.{nm}.bias", (inner,), ("bias", dim))]
        spec += [(p + "2.to_out.0.weight", (dim, inner), "linear"), (p + "2.to_out.0.bias", (dim,), ("bias", inner)),
                 (p + "3.g", (dim,), "ones"),
                 (p + "4.ff.0.0.weight", (ff_inner, dim), "linear"), (p + "4.ff.0.0.bias", (ff_inner,), ("bias", dim)),
                 (p + "4.ff.2.weight", (dim, ff_inner), "linear"), (p + "4.ff.2.bias", (dim,), ("bias", ff_inner))]
    spec += [("norm_out.g", (dim,), "ones"), ("proj_out.weight", (mel_dim, dim), "linear"), ("proj_out.bias", (mel_dim,), ("bias", dim))]
    return spec


class UNetT(DiT):
    BACKBONE = _lib.F5_BACKBONE_UNETT

    def __init__(self, *, dim, depth=8, heads=8, dim_head=64, dropout=0.1, ff_mult=4, mel_dim=100, text_num_embeds=256, text_dim=None,
                 text_mask_padding=True, qk_norm=None, conv_layers=0, pe_attn_head=None, skip_connect_type="concat", precision=None, rope_layout=None):
        nn.Module.__init__(self)
        assert depth % 2 == 0, "UNet-Transformer's depth should be even."  # unett.py:120
        if skip_connect_type not in ("add", "concat", "none"):
            raise ValueError(f"skip_connect_type={skip_connect_type!r}")
        self.checkpoint_activations = False
        self._setup(dim=dim, depth=depth, heads=heads, dim_head=dim_head, ff_mult=ff_mult, mel_dim=mel_dim, text_num_embeds=text_num_embeds,
                    text_dim=text_dim, text_mask_padding=text_mask_padding, qk_norm=qk_norm, conv_layers=conv_layers, pe_attn_head=pe_attn_head,
                    precision=precision, rope_layout=rope_layout, skip_connect_type=skip_connect_type)

    def _spec(self):
        return _unett_spec(self.dim, self.depth, self.heads, self.dim_head, self.ff_inner, self.mel_dim, self.text_num_embeds, self.text_dim,
                           self.conv_layers, self.skip_connect_type)
