"""Per-op parity of the HIP kernels against the CPU oracle / plain fp32 torch math (through the C ABI).

Tolerances (stated): fp32-input MFMA path rel-L2 <= 2e-6 (accumulation order only); bf16 path, with operands pre-rounded
to bf16 so only the accumulation order and the output rounding differ, rel-L2 <= 2e-5 for f32 outputs."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2
from oracle import cpu_ref

pytestmark = pytest.mark.gpu
P_BF16, P_FP32 = 0, 1


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    from eraxvif5tts_amd import _lib
    _lib.require_gpu()


@pytest.mark.parametrize("prec", [P_FP32, P_BF16])
@pytest.mark.parametrize("shape", [(64, 64, 64), (100, 72, 96), (300, 1024, 512), (513, 100, 1024)])
@pytest.mark.parametrize("act", ["none", "gelu_tanh", "gelu_erf", "mish"])
def test_linear_tile_kernel(prec, shape, act):
    import gpu_helpers as G
    M, N, K = shape
    g = torch.Generator().manual_seed(M * 7 + N)
    A, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    if prec == P_BF16:
        A, W = G.bf16_round(A), G.bf16_round(W)
    ref = A.double() @ W.double().t() + b.double()
    ref = {"none": lambda v: v, "gelu_tanh": lambda v: F.gelu(v, approximate="tanh"), "gelu_erf": F.gelu, "mish": F.mish}[act](ref).float()
    out = G.op_linear(prec, 0, A, W, b, act)
    assert rel_l2(out, ref) < (2e-6 if prec == P_FP32 else 2e-5)


def test_layernorm_modulate():
    import gpu_helpers as G
    g = torch.Generator().manual_seed(3)
    for rows, dim in ((37, 1024), (5, 128), (9, 768)):
        x = torch.randn(rows, dim, generator=g) * 3 + 1
        sc, sh = torch.randn(dim, generator=g) * 0.3, torch.randn(dim, generator=g)
        ref = cpu_ref._layernorm(x) * (1 + sc) + sh
        assert rel_l2(G.op_ln_mod(x, sc, sh), ref) < 2e-6


def _attn_ref(qkv, mask):
    q, k, v = [qkv[:, :, i].transpose(1, 2).double() for i in range(3)]  # [B,H,N,64]
    s = q @ k.transpose(-1, -2) / 8.0
    if mask is not None:
        s = s.masked_fill(~mask[:, None, None, :], float("-inf"))
    o = torch.softmax(s, dim=-1) @ v
    return o.transpose(1, 2).reshape(qkv.shape[0], qkv.shape[1], -1).float()


def _fused_ref(epi, A, W, b, act, gate, rowmask, rope, rope_heads, seq):
    """fp64 restatement of the three fused store epilogues (oracle/cpu_ref.py: dit_block's linears, apply_rope)."""
    import gpu_helpers as G
    v = A.double() @ W.double().t() + b.double()
    if act == "gelu_tanh":
        v = F.gelu(v, approximate="tanh")
    if epi == G.EPI_GATE_T:
        if gate is not None:
            v = v * gate.double()
        if rowmask is not None:
            v = v * rowmask.double()[:, None]
    if epi == G.EPI_ROPE_T:
        M, N = v.shape
        inner = N // 3
        pos = torch.arange(M) % seq
        cs = rope.double()[pos]  # [M, 32, 2]
        cos, sin = cs[..., 0], cs[..., 1]
        v = v.clone()
        for part in range(2):  # q, k
            for h in range(rope_heads):
                c0 = part * inner + h * 64
                x = v[:, c0:c0 + 64].reshape(M, 32, 2)
                x0, x1 = x[..., 0], x[..., 1]
                v[:, c0:c0 + 64] = torch.stack([x0 * cos - x1 * sin, x1 * cos + x0 * sin], dim=-1).reshape(M, 64)
    return v.float()


# whole-tile shapes (lean epilogue), ragged rows / narrow tiles (generic epilogue), every schedule the launcher can pick
@pytest.mark.parametrize("knobs", [{}, {"gemm_persist": 0}, {"gemm_persist": 0, "gemm_lean": 0}, {"gemm_variant": 0}],
                         ids=["default_persistent_grid", "one_tile_per_workgroup", "generic_epilogue", "plain_ring"])
@pytest.mark.parametrize("epi_name,shape,seq", [("store", (512, 1024, 256), 0), ("store", (10240, 1024, 128), 0), ("store", (10300, 2048, 192), 0),
                                                ("gate", (768, 512, 128), 0), ("gate", (10240, 1024, 256), 0), ("gate", (10301, 1024, 128), 0),
                                                ("rope", (1024, 768, 128), 256), ("rope", (4096, 3072, 128), 1024), ("rope", (4120, 3072, 128), 1030)])
def test_linear_fused_epilogues(knobs, epi_name, shape, seq):
    import gpu_helpers as G
    from eraxvif5tts_amd import _lib
    lib = _lib.load()
    M, N, K = shape
    epi = {"store": G.EPI_STORE_T, "gate": G.EPI_GATE_T, "rope": G.EPI_ROPE_T}[epi_name]
    g = torch.Generator().manual_seed(M + 3 * N + K)
    A, W, b = G.bf16_round(torch.randn(M, K, generator=g)), G.bf16_round(torch.randn(N, K, generator=g) / math.sqrt(K)), torch.randn(N, generator=g)
    act = "gelu_tanh" if epi_name == "store" else "none"
    gate = torch.randn(N, generator=g) if epi_name == "gate" else None
    rowmask = (torch.rand(M, generator=g) > 0.2) if epi_name == "gate" else None
    rope, heads = None, 0
    if epi_name == "rope":
        ang = torch.rand(seq, 32, generator=g) * 6.28
        rope, heads = torch.stack([ang.cos(), ang.sin()], dim=-1), (1 if N == 3072 else N // 3 // 64)
    ref = _fused_ref(epi, A, W, b, act, gate, rowmask, rope, heads, seq)
    try:
        for k, v in knobs.items():
            _lib.check(lib.f5_tuning_set(k.encode(), v))
        out = G.op_linear_fused(1, epi, A, W, b, act, gate, rowmask, rope, heads, seq)
    finally:
        for k in knobs:
            _lib.check(lib.f5_tuning_set(k.encode(), {"gemm_lean": 1, "gemm_variant": 1, "gemm_persist": 1}[k]))
    base = G.op_linear_fused(0, epi, A, W, b, act, gate, rowmask, rope, heads, seq)
    assert rel_l2(base, ref) < 3e-3   # bf16 output rounding: 2^-9 relative per element
    assert rel_l2(out, ref) < 3e-3
    assert rel_l2(out, base) < 1e-3   # same contraction up to fp32 summation order and the exp2/rcp form of GELU
    if rowmask is not None:
        assert torch.count_nonzero(out[~rowmask]) == 0  # masked rows are exact zeros (modules.py:499-501)


@pytest.mark.parametrize("knobs", [{}, {"gemm_persist": 0}, {"gemm_persist": 0, "gemm_lean": 0}, {"gemm_variant": 0}],
                         ids=["default_persistent_grid", "one_tile_per_workgroup", "generic_epilogue", "plain_ring"])
@pytest.mark.parametrize("shape", [(768, 512, 128), (10240, 1024, 1024), (10240, 1024, 2048), (10301, 1024, 128), (1000, 100, 256)])
@pytest.mark.parametrize("masked", [False, True], ids=["all_rows", "row_mask"])
def test_linear_in_place_residual_epilogue(knobs, shape, masked):
    """x += gate * (A W^T + b) IN PLACE on the fp16 residual stream (how the attention out-projection and the second FF linear update the
    stream in the bf16 production mode, reference model/modules.py:635,639; masked query rows keep their value, :499-501): the tuned kernel in
    every schedule the launcher can pick and the reference tile kernel against fp64 on the same fp16-rounded stream; the tuned schedules
    against each other bit for bit (the fp32 sums and the update are the same in every variant)."""
    import gpu_helpers as G
    from eraxvif5tts_amd import _lib
    lib = _lib.load()
    M, N, K = shape
    g = torch.Generator().manual_seed(M + 5 * N + K)
    A, W, b = G.bf16_round(torch.randn(M, K, generator=g)), G.bf16_round(torch.randn(N, K, generator=g) / math.sqrt(K)), torch.randn(N, generator=g)
    gate = torch.randn(N, generator=g)
    rowmask = (torch.rand(M, generator=g) > 0.2) if masked else None
    x = (torch.randn(M, N, generator=g) * 3).half().float()  # the stream as stored
    upd = gate.double() * (A.double() @ W.double().t() + b.double())
    if masked:
        upd = upd * rowmask[:, None].double()
    ref = (x.double() + upd).float()
    try:
        for k, v in knobs.items():
            _lib.check(lib.f5_tuning_set(k.encode(), v))
        out = G.op_linear_fused(1, G.EPI_RESID, A, W, b, "none", gate, rowmask, stream_in=x)
    finally:
        for k in knobs:
            _lib.check(lib.f5_tuning_set(k.encode(), {"gemm_lean": 1, "gemm_variant": 1, "gemm_persist": 1}[k]))
    base = G.op_linear_fused(0, G.EPI_RESID, A, W, b, "none", gate, rowmask, stream_in=x)
    dflt = G.op_linear_fused(1, G.EPI_RESID, A, W, b, "none", gate, rowmask, stream_in=x)
    assert rel_l2(out, ref) < 6e-4 and rel_l2(base, ref) < 6e-4   # one fp16 rounding of the result: 2^-11 relative
    assert torch.equal(out, dflt)                                  # schedule-independent
    if masked:
        assert torch.equal(out[~rowmask], x[~rowmask])             # masked rows untouched
    assert torch.equal(out, out.half().float())                    # the values are fp16 numbers


@pytest.mark.parametrize("prec", [P_FP32, P_BF16])
@pytest.mark.parametrize("B,N,H,masked", [(2, 56, 2, True), (1, 41, 2, False), (2, 200, 3, True), (1, 64, 16, False)])
def test_attention_reference_kernel(prec, B, N, H, masked):
    import gpu_helpers as G
    g = torch.Generator().manual_seed(N)
    qkv = torch.randn(B, N, 3, H, 64, generator=g)
    if prec == P_BF16:
        qkv = G.bf16_round(qkv)
    mask = None
    if masked:
        lens = torch.tensor([N, max(1, N - 13)][:B])
        mask = torch.arange(N)[None, :] < lens[:, None]
    ref = _attn_ref(qkv, mask)
    out = G.op_attention(prec, 0, qkv, mask)
    tol = 3e-6 if prec == P_FP32 else 4e-3  # bf16: only the final output rounding (8-bit mantissa)
    assert rel_l2(out, ref) < tol


@pytest.mark.parametrize("prec", [P_FP32, P_BF16])
@pytest.mark.parametrize("dim,B,N", [(128, 2, 50), (1024, 2, 70), (768, 1, 40)])
def test_conv_pos_embed(prec, dim, B, N):
    import gpu_helpers as G
    g = torch.Generator().manual_seed(dim + N)
    cg = dim // 16
    x = torch.randn(B, N, dim, generator=g)
    w0, w1 = [torch.randn(dim, cg, 31, generator=g) / math.sqrt(cg * 31) for _ in range(2)]
    b0, b1 = torch.randn(dim, generator=g) * 0.1, torch.randn(dim, generator=g) * 0.1
    if prec == P_BF16:
        x, w0, w1 = G.bf16_round(x), G.bf16_round(w0), G.bf16_round(w1)
    W = {"input_embed.conv_pos_embed.conv1d.0.weight": w0, "input_embed.conv_pos_embed.conv1d.0.bias": b0,
         "input_embed.conv_pos_embed.conv1d.2.weight": w1, "input_embed.conv_pos_embed.conv1d.2.bias": b1}
    ref = cpu_ref.conv_pos_embed(W, x)
    out = G.op_conv_pos(prec, x, w0, b0, w1, b1)
    # bf16: the intermediate activation between the two convolutions is rounded to bf16 (2^-9 relative)
    assert rel_l2(out, ref) < (3e-6 if prec == P_FP32 else 4e-3)


@pytest.mark.parametrize("conv31", [1, 0], ids=["halo_tile_kernel", "implicit_gemm"])
@pytest.mark.parametrize("dim,B,N", [(1024, 2, 70), (1024, 3, 256), (1024, 1, 700), (1024, 2, 1000), (128, 2, 300)])
def test_conv_pos_embed_tuned_kernels(conv31, dim, B, N):
    """The two tuned forms of the grouped Conv1d(k=31)+Mish pair: conv31.hip (dim 1024: 64 channels per group) and the implicit GEMM of
    gemm_fast.hip, against the oracle and against the reference tile kernel.  Ragged last tiles, utterance edges (zero padding) and
    several utterances per launch are all in these shapes."""
    import gpu_helpers as G
    from eraxvif5tts_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(dim + N)
    x = G.bf16_round(torch.randn(B, N, dim, generator=g))
    cg = dim // 16
    w0, w1 = [G.bf16_round(torch.randn(dim, cg, 31, generator=g) / math.sqrt(cg * 31)) for _ in range(2)]
    b0, b1 = torch.randn(dim, generator=g) * 0.1, torch.randn(dim, generator=g) * 0.1
    W = {"input_embed.conv_pos_embed.conv1d.0.weight": w0, "input_embed.conv_pos_embed.conv1d.0.bias": b0,
         "input_embed.conv_pos_embed.conv1d.2.weight": w1, "input_embed.conv_pos_embed.conv1d.2.bias": b1}
    ref = cpu_ref.conv_pos_embed(W, x)
    base = G.op_conv_pos(P_BF16, x, w0, b0, w1, b1)
    try:
        _lib.check(lib.f5_tuning_set(b"op_conv_kernel", 1))
        _lib.check(lib.f5_tuning_set(b"conv31", conv31))
        out = G.op_conv_pos(P_BF16, x, w0, b0, w1, b1)
    finally:
        _lib.check(lib.f5_tuning_set(b"op_conv_kernel", 0))
        _lib.check(lib.f5_tuning_set(b"conv31", 1))
    assert rel_l2(base, ref) < 4e-3  # the intermediate activation between the two convolutions is rounded to bf16
    assert rel_l2(out, ref) < 6e-3   # ... and so is the stored branch of the tuned path
    assert rel_l2(out, base) < 4e-3


# ----------------------------------------------------------------------------- tuned kernels (bf16) vs the same references
@pytest.mark.parametrize("shape", [(256, 256, 64), (512, 1024, 1024), (300, 3072, 128), (1000, 100, 1024), (2048, 2048, 2048), (77, 512, 640)])
@pytest.mark.parametrize("act", ["none", "gelu_tanh"])
def test_linear_tuned_kernel(shape, act):
    import gpu_helpers as G
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    A, W, b = G.bf16_round(torch.randn(M, K, generator=g)), G.bf16_round(torch.randn(N, K, generator=g) / math.sqrt(K)), torch.randn(N, generator=g)
    ref = A.double() @ W.double().t() + b.double()
    ref = (F.gelu(ref, approximate="tanh") if act == "gelu_tanh" else ref).float()
    out = G.op_linear(P_BF16, 1, A, W, b, act)
    assert rel_l2(out, ref) < 2e-5
    # and bit-for-bit the same contraction as the reference tile kernel up to fp32 summation order
    assert rel_l2(out, G.op_linear(P_BF16, 0, A, W, b, act)) < 2e-5


@pytest.mark.parametrize("B,N,H,masked", [(2, 56, 2, True), (1, 41, 2, False), (2, 200, 3, True), (1, 128, 16, False), (2, 1024, 4, True),
                                          (1, 1024, 2, False), (3, 333, 1, True)])
@pytest.mark.parametrize("variant", [0, 2, 5], ids=["by_grid_size", "64_queries_per_wave", "pipelined_32_queries_per_wave"])
def test_attention_tuned_kernel(B, N, H, masked, variant):
    import gpu_helpers as G
    from eraxvif5tts_amd import _lib
    _lib.check(_lib.load().f5_tuning_set(b"attn_variant", variant))
    g = torch.Generator().manual_seed(N + H)
    qkv = G.bf16_round(torch.randn(B, N, 3, H, 64, generator=g) * 1.5)
    mask = None
    if masked:
        lens = torch.tensor([N, max(1, N - 13), max(1, N // 2)][:B])
        mask = torch.arange(N)[None, :] < lens[:, None]
    ref = _attn_ref(qkv, mask)
    try:
        out = G.op_attention(P_BF16, 1, qkv, mask)
    finally:
        _lib.check(_lib.load().f5_tuning_set(b"attn_variant", 0))
    # P is rounded to bf16 before the PV product and the output is bf16: 2^-9 relative each
    assert rel_l2(out, ref) < 6e-3
    valid = slice(None) if mask is None else mask
    assert torch.isfinite(out).all()
    assert (out[valid] - ref[valid]).abs().max() < 0.05


@pytest.mark.parametrize("variant", [0, 2, 5], ids=["by_grid_size", "64_queries_per_wave", "pipelined_32_queries_per_wave"])
@pytest.mark.parametrize("N", [2048, 2050, 4000, 4096])
@pytest.mark.parametrize("masked", [False, True], ids=["unmasked", "ragged_lens"])
def test_attention_tuned_kernel_long_sequences(variant, N, masked):
    """The long-form (C4) sequence lengths up to the reference's hard cap of 4096 frames (cfm.py:93,135): every attention schedule the
    launcher can select there -- the default one included -- against the fp64 softmax, whole tiles (2048, 4096), ragged last tiles
    (2050, 4000) and ragged per-utterance lengths behind the key-padding mask (modules.py:483-501)."""
    import gpu_helpers as G
    from eraxvif5tts_amd import _lib
    B, H = (2, 2) if masked else (1, 2)
    g = torch.Generator().manual_seed(N + 7 * variant)
    qkv = G.bf16_round(torch.randn(B, N, 3, H, 64, generator=g) * 1.5)
    mask = None
    if masked:
        lens = torch.tensor([N, N - 1037])
        mask = torch.arange(N)[None, :] < lens[:, None]
    ref = _attn_ref(qkv, mask)
    _lib.check(_lib.load().f5_tuning_set(b"attn_variant", variant))
    try:
        out = G.op_attention(P_BF16, 1, qkv, mask)
    finally:
        _lib.check(_lib.load().f5_tuning_set(b"attn_variant", 0))
    valid = slice(None) if mask is None else mask
    assert torch.isfinite(out).all()
    assert rel_l2(out[valid], ref[valid]) < 6e-3
    assert (out[valid] - ref[valid]).abs().max() < 0.05


@pytest.mark.parametrize("B,N,H", [(1, 1024, 2), (3, 512, 8), (2, 768, 1), (20, 768, 16), (1, 4096, 2), (9, 256, 16)])
def test_attention_persistent_grid(B, N, H):
    """attn_persist_kernel (attn_variant 6: two workgroups per CU walk the (batch, head, 256-query block) items; the K/V ring, the Q prefetch by
    LDS-DMA and the staged output stores run on across item boundaries) against the fp64 softmax (modules.py:483-497): fewer items than
    workgroups, exactly one each, an uneven one-or-two split (20 x 16 heads x 3 blocks = 960 items on 512 workgroups), a (batch x head) count
    that is not a multiple of the 8 XCDs, the shortest eligible sequence (256 = 4 key tiles), N = 4096; and bit-identical to the one-item-per-
    workgroup kernel (same arithmetic, different schedule)."""
    import gpu_helpers as G
    from eraxvif5tts_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(N + H + B)
    qkv = G.bf16_round(torch.randn(B, N, 3, H, 64, generator=g) * 1.5)
    ref = _attn_ref(qkv, None)
    outs = {}
    for variant in (6, 2):
        _lib.check(lib.f5_tuning_set(b"attn_variant", variant))
        try:
            outs[variant] = G.op_attention(P_BF16, 1, qkv, None)
        finally:
            _lib.check(lib.f5_tuning_set(b"attn_variant", 0))
    assert torch.isfinite(outs[6]).all()
    assert rel_l2(outs[6], ref) < 6e-3 and (outs[6] - ref).abs().max() < 0.05
    assert torch.equal(outs[6], outs[2])


@pytest.mark.parametrize("masked", [False, True], ids=["unmasked", "ragged_lens"])
def test_attention_beyond_the_validity_table(masked):
    """N = 8320 = 130 key tiles: past the 128-tile table of key validity bits the wide kernel keeps in LDS.  Unmasked (whole tiles) it still
    runs; with a mask the launcher must hand the call to the pipelined kernel.  One head, against the fp64 softmax."""
    import gpu_helpers as G
    from eraxvif5tts_amd import _lib
    B, N, H = 1, 8320, 1
    g = torch.Generator().manual_seed(4)
    qkv = G.bf16_round(torch.randn(B, N, 3, H, 64, generator=g) * 1.5)
    mask = (torch.arange(N)[None, :] < torch.tensor([N - 301])[:, None]) if masked else None
    ref = _attn_ref(qkv, mask)
    _lib.check(_lib.load().f5_tuning_set(b"attn_variant", 2))
    try:
        out = G.op_attention(P_BF16, 1, qkv, mask)
    finally:
        _lib.check(_lib.load().f5_tuning_set(b"attn_variant", 0))
    valid = slice(None) if mask is None else mask
    assert torch.isfinite(out).all()
    assert rel_l2(out[valid], ref[valid]) < 6e-3


@pytest.mark.parametrize("variant", [0, 2, 5, 6])
def test_attention_long_sequence_spiked_scores(variant):
    """online-softmax rescale path of every schedule at N = 4096: the running max jumps late (key 3900) and in the first tile."""
    import gpu_helpers as G
    from eraxvif5tts_amd import _lib
    g = torch.Generator().manual_seed(11)
    B, N, H = 1, 4096, 2
    qkv = torch.randn(B, N, 3, H, 64, generator=g)
    qkv[0, 3900, 1] = qkv[0, 70, 0] * 8.0   # key 3900 aligned with query 70
    qkv[0, 3, 1] = qkv[0, 2049, 0] * 6.0    # key 3 aligned with query 2049
    qkv = G.bf16_round(qkv)
    ref = _attn_ref(qkv, None)
    _lib.check(_lib.load().f5_tuning_set(b"attn_variant", variant))
    try:
        out = G.op_attention(P_BF16, 1, qkv, None)
    finally:
        _lib.check(_lib.load().f5_tuning_set(b"attn_variant", 0))
    assert rel_l2(out, ref) < 6e-3


@pytest.mark.parametrize("variant", [2, 5, 6])
def test_attention_deferred_rescale_thresholds(variant):
    """Exponent-reference handling of both schedules (pipelined kernel: the reference moves only when a row maximum outgrew it by more
    than 2^16; 64-queries-per-wave kernel: no maximum unless a row sum leaves the guarded range): score jumps just below and far above
    the 2^16 threshold, early and late, and a row whose maximum keeps creeping up tile by tile -- against the fp64 softmax."""
    import gpu_helpers as G
    from eraxvif5tts_amd import _lib
    g = torch.Generator().manual_seed(9)
    B, N, H = 1, 1024, 2
    qkv = torch.randn(B, N, 3, H, 64, generator=g) * 0.5
    qn = qkv[0, :, 0] / qkv[0, :, 0].norm(dim=-1, keepdim=True)      # unit queries
    for q, (key, nat) in {5: (700, 10.0), 37: (130, 12.5), 90: (1000, 40.0), 200: (3, 60.0)}.items():
        qkv[0, q, 0] = qn[q] * 8.0                                   # |q| = 8: q.k / 8 = |k| cos
        qkv[0, key, 1] = qn[q] * nat                                 # score of (q, key) = nat nats (16 log2 units = 11.1 nats)
    qkv[0, 300, 0] = qn[300] * 8.0
    for i, key in enumerate(range(10, 1024, 64)):                    # one key per tile, each 1.5 nats above the previous
        qkv[0, key, 1] = qn[300] * (1.5 * (i + 1))
    qkv = G.bf16_round(qkv)
    ref = _attn_ref(qkv, None)
    _lib.check(_lib.load().f5_tuning_set(b"attn_variant", variant))
    try:
        out = G.op_attention(P_BF16, 1, qkv, None)
    finally:
        _lib.check(_lib.load().f5_tuning_set(b"attn_variant", 0))
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) < 6e-3
    for q in (5, 37, 90, 200, 300):
        assert (out[0, q] - ref[0, q]).abs().max() < 0.03 * max(1.0, float(ref[0, q].abs().max())), q


@pytest.mark.parametrize("masked", [False, True], ids=["unmasked", "leading_keys_masked"])
def test_attention_wide_kernel_range_guard(masked):
    """The 64-queries-per-wave kernel takes no row maximum on its common path: numerators are formed against the reference a query already
    has and the per-tile row sums are checked against 2^64; past that (or inf / NaN) the wave redoes the tile the classic way.  Score
    jumps just below the guard (62 log2 units: P up to 2^62 stays on the common path), just above it, far above it (exp2 overflows to
    inf), early and late in the sequence, a query whose maximum creeps up tile by tile, and -- masked -- an utterance whose whole first
    tile is masked out (its queries enter the second tile with the finite start reference), against the fp64 softmax."""
    import gpu_helpers as G
    from eraxvif5tts_amd import _lib
    g = torch.Generator().manual_seed(21)
    B, N, H = 2, 1024, 2
    qkv = torch.randn(B, N, 3, H, 64, generator=g) * 0.5
    LOG2E = 1.4426950408889634
    jumps = {5: (700, 62.0 / LOG2E), 37: (130, 65.5 / LOG2E), 90: (1000, 144.0 / LOG2E), 200: (3, 430.0 / LOG2E), 411: (200, 90.0 / LOG2E),
             600: (90, 63.5 / LOG2E), 777: (960, 64.5 / LOG2E)}
    for bb in range(B):
        qn = qkv[bb, :, 0] / qkv[bb, :, 0].norm(dim=-1, keepdim=True)
        for q, (key, nat) in jumps.items():
            qkv[bb, q, 0] = qn[q] * 8.0          # |q| = 8: q.k / 8 = |k| cos
            qkv[bb, key, 1] = qn[q] * nat        # score of (q, key) = nat nats
        qkv[bb, 300, 0] = qn[300] * 8.0
        for i, key in enumerate(range(74, 1024, 64)):   # one key per tile, each 6 nats above the previous
            qkv[bb, key, 1] = qn[300] * (6.0 * (i + 1))
    qkv = G.bf16_round(qkv)
    mask = None
    if masked:
        mask = torch.ones(B, N, dtype=torch.bool)
        mask[1, :70] = False      # the first tile (and 6 keys of the second) of utterance 1
        mask[1, 1000:] = False
        mask[0, 990:] = False
    ref = _attn_ref(qkv, mask)
    _lib.check(_lib.load().f5_tuning_set(b"attn_variant", 2))
    try:
        out = G.op_attention(P_BF16, 1, qkv, mask)
    finally:
        _lib.check(_lib.load().f5_tuning_set(b"attn_variant", 0))
    assert torch.isfinite(out).all()
    if mask is None:
        assert rel_l2(out, ref) < 6e-3
    else:  # masked QUERIES are computed and dropped downstream; compare the valid ones
        assert rel_l2(out[mask], ref[mask]) < 6e-3
    for bb in range(B):
        for q in list(jumps) + [300]:
            assert (out[bb, q] - ref[bb, q]).abs().max() < 0.03 * max(1.0, float(ref[bb, q].abs().max())), (bb, q)


def test_attention_tuned_kernel_spiked_scores():
    """online-softmax rescale path: one key dominates late in the sequence (running max jumps by > 60)."""
    import gpu_helpers as G
    g = torch.Generator().manual_seed(5)
    B, N, H = 1, 512, 2
    qkv = G.bf16_round(torch.randn(B, N, 3, H, 64, generator=g))
    qkv[0, 400, 1] = qkv[0, 7, 0] * 8.0  # key 400 aligned with query 7
    qkv = G.bf16_round(qkv)
    ref = _attn_ref(qkv, None)
    out = G.op_attention(P_BF16, 1, qkv, None)
    assert rel_l2(out, ref) < 6e-3


@pytest.mark.parametrize("M,D,N,Kb,epi", [
    (300, 128, 384, 128, 4),     # tiny-arch QKV: ragged token tile, 128-wide feature tiles
    (300, 128, 256, 256, 0),     # tiny-arch FF1 (GELU)
    (1024, 1024, 3072, 1024, 4),  # F5TTS_Base QKV: persistent 256 x 256 tiles (stats from the out-projection's persistent in-place epilogue)
    (1024, 1024, 2048, 2048, 0),  # F5TTS_Base FF1 behind an FF2-shaped producer
    (1100, 1024, 2048, 1024, 0),  # M % 256 != 0: whole tiles on the persistent schedule + a tail launch
    (512, 1024, 3072, 1024, 4),   # fewer tiles than CUs
    (4096, 1024, 3072, 1024, -16),  # persistent 256 x 256 tiles with RoPE on ALL 16 heads (F5TTS_v1_Base: pe_attn_head = null)
])
@pytest.mark.parametrize("offset", [0.0, 40.0])
def test_layernorm_fold_site_against_fp64(M, D, N, Kb, epi, offset):
    """The LayerNorm fold of one call site (modules.py:301-317 / 637-638 folded into the projection behind it; gemm.h) against fp64 math:
    (1) the in-place residual epilogue updates the fp16 stream, (2) its partial row sums give (mean, rstd) of the STORED rows -- also with a
    common offset of 40 standard deviations on every row, where a plain sum-of-squares form would lose its digits: the pivot is the row's
    previous mean --, (3) Linear(LN(x) (1 + scale) + shift) comes out of the fp16 GEMM + fold epilogue with bf16 output rounding only."""
    import gpu_helpers as G
    rope_heads = 1
    if epi < 0:
        rope_heads, epi = -epi, 4
    g = torch.Generator().manual_seed(M + N + int(offset))
    x = torch.randn(M, D, generator=g) * 1.7 + offset + torch.randn(M, 1, generator=g) * 0.5
    A = G.bf16_round(torch.randn(M, Kb, generator=g))
    Wo = G.bf16_round(torch.randn(D, Kb, generator=g) / Kb ** 0.5)
    bo = torch.randn(D, generator=g) * 0.1
    gate = torch.randn(D, generator=g) * 0.3
    W = torch.randn(N, D, generator=g) / D ** 0.5
    bias = torch.randn(N, generator=g) * 0.1
    scale, shift = torch.randn(D, generator=g) * 0.2, torch.randn(D, generator=g) * 0.3
    seq = M // 2 if epi == 4 and M % 2 == 0 else M
    rope = None
    if epi == 4:
        ang = torch.arange(seq)[:, None] * (1.0 / 10000.0 ** (torch.arange(0, 64, 2) / 64.0))[None, :]
        rope = torch.stack([ang.cos(), ang.sin()], dim=-1).reshape(seq, 64).float()
    xh = x.half().float()
    pivot = torch.stack([xh.mean(dim=1) + 0.01, torch.ones(M)], dim=1) if offset else None  # "previous mean": near, not equal to, the new one
    xs, stats, out = G.op_ln_fold(epi, x, A, Wo, bo, gate, W, bias, scale, shift, pivot=pivot, act="gelu_tanh" if epi == 0 else "none", rope=rope,
                                  rope_heads=rope_heads, seq=seq)
    # (1) the stream: x + gate * (A Wo^T + bo), stored as fp16
    want = xh.double() + gate.double() * (A.double() @ Wo.double().t() + bo.double())
    assert (xs.double() - want).abs().max() <= want.abs().max() * 2.0 ** -10  # within one fp16 ulp of the exact sum
    # (2) statistics of the STORED rows
    mean, var = xs.double().mean(dim=1), xs.double().var(dim=1, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-6)
    assert (stats[:, 0].double() - mean).abs().max() < 2e-5 * (1 + mean.abs().max())
    assert ((stats[:, 1].double() - rstd) / rstd).abs().max() < 2e-4
    # (3) the folded projection
    h = (xs.double() - mean[:, None]) * rstd[:, None] * (1 + scale.double()) + shift.double()
    ref = h @ W.double().t() + bias.double()
    if epi == 0:
        ref = 0.5 * ref * (1 + torch.tanh(0.7978845608028654 * (ref + 0.044715 * ref ** 3)))
    else:
        inner = N // 3
        r = ref.clone()
        pos = torch.arange(M) % seq
        cs = rope.double()[pos].reshape(M, 32, 2)
        for part in (0, 1):
            for hd in range(rope_heads):
                c0 = part * inner + hd * 64
                blk = ref[:, c0: c0 + 64].reshape(M, 32, 2)
                rot = torch.stack([blk[..., 0] * cs[..., 0] - blk[..., 1] * cs[..., 1], blk[..., 1] * cs[..., 0] + blk[..., 0] * cs[..., 1]], dim=-1)
                r[:, c0: c0 + 64] = rot.reshape(M, 64)
        ref = r
    err = rel_l2(out, ref)
    print(f"ln_fold M={M} D={D} N={N} epi={epi} offset={offset}: rel-L2 {err:.2e}")
    assert err < 3e-3  # bf16 output rounding (2^-9 relative per element) + fp16 rounding of W'


def _with_knob(lib, key, value, fn):
    from eraxvif5tts_amd import _lib
    _lib.check(lib.f5_tuning_set(key, value))
    try:
        return fn()
    finally:
        _lib.check(lib.f5_tuning_set(key, 1))


@pytest.mark.parametrize("epi_name,shape,seq", [("rope", (2048, 3072, 1024), 1024), ("store", (2048, 2048, 1024), 0), ("resid", (6144, 1024, 2048), 0),
                                                ("store", (6144, 2048, 1024), 0)])
def test_w4_kernel_on_partly_filled_grids(epi_name, shape, seq):
    """Small batches: the launcher gives a launch to the one-wave-per-SIMD kernel from three quarters of the CUs on (256-row tiles) or from half
    (128-row tiles) -- single-utterance projections (192 / 128 short tiles), three utterances' FF1 (192 tall tiles) and FF2 (192 short ones).  A
    workgroup then walks one tile and the request front re-walks it past the end; outputs bit for bit as the 8-wave kernel's."""
    import gpu_helpers as G
    from eraxvif5tts_amd import _lib
    lib = _lib.load()
    M, N, K = shape
    g = torch.Generator().manual_seed(M + 11 * N + K)
    A, W, b = G.bf16_round(torch.randn(M, K, generator=g)), G.bf16_round(torch.randn(N, K, generator=g) / math.sqrt(K)), torch.randn(N, generator=g)
    if epi_name == "resid":
        gate = torch.randn(N, generator=g)
        x = (torch.randn(M, N, generator=g) * 3).half().float()
        run = lambda: G.op_linear_fused(1, G.EPI_RESID, A, W, b, "none", gate, None, stream_in=x)
    else:
        epi = {"store": G.EPI_STORE_T, "rope": G.EPI_ROPE_T}[epi_name]
        rope, heads = None, 0
        if epi_name == "rope":
            ang = torch.rand(seq, 32, generator=g) * 6.28
            rope, heads = torch.stack([ang.cos(), ang.sin()], dim=-1), 1
        run = lambda: G.op_linear_fused(1, epi, A, W, b, "gelu_tanh" if epi_name == "store" else "none", None, None, rope, heads, seq)
    new = run()
    old = _with_knob(lib, b"gemm_w4", 0, run)
    assert torch.isfinite(new).all() and torch.equal(new, old)


@pytest.mark.parametrize("bm", [256, 128], ids=["256_row_tiles", "128_row_tiles"])
@pytest.mark.parametrize("epi_name,shape,seq", [("store", (8192, 2048, 256), 0), ("store", (8192, 2048, 1024), 0), ("rope", (8192, 3072, 384), 1024),
                                                ("rope", (16384, 1536, 1024), 2048), ("resid", (16384, 1024, 512), 0), ("resid_masked", (16384, 1024, 2048), 0),
                                                ("resid_masked", (8320, 1024, 256), 0)])
def test_w4_kernel_equals_the_8_wave_kernel(bm, epi_name, shape, seq):
    """Round 4, late: whole-tile block linears with at least one 256 x 256 tile per CU run on the one-wave-per-SIMD kernel (csrc/gemm_w4.hip: four
    waves, 128 x 128 outputs per wave, 64-deep stages refilled in place, hand-placed main loop).  Same MFMA, K order, start value and epilogue
    arithmetic as the 8-wave persistent kernel, so the outputs must agree BIT FOR BIT (knob gemm_w4 = 0 selects the 8-wave kernel) -- store + GELU,
    QKV + RoPE (first head / all heads), the in-place fp16 residual update with and without a row mask -- and both agree with fp64.  Both tile
    heights of the kernel (knob gemm_w4_bm: 256 x 256 with 128 x 128 per wave, 128 x 256 with 64 x 128 per wave; a shape the forced height cannot
    fill the CUs with falls back to the 8-wave kernel, M = 8320 is a multiple of 128 only)."""
    import gpu_helpers as G
    from eraxvif5tts_amd import _lib
    lib = _lib.load()
    M, N, K = shape
    g = torch.Generator().manual_seed(M + 7 * N + K)
    A, W, b = G.bf16_round(torch.randn(M, K, generator=g)), G.bf16_round(torch.randn(N, K, generator=g) / math.sqrt(K)), torch.randn(N, generator=g)
    if epi_name.startswith("resid"):
        gate = torch.randn(N, generator=g)
        rowmask = (torch.rand(M, generator=g) > 0.2) if epi_name == "resid_masked" else None
        x = (torch.randn(M, N, generator=g) * 3).half().float()
        run = lambda: G.op_linear_fused(1, G.EPI_RESID, A, W, b, "none", gate, rowmask, stream_in=x)
        upd = gate.double() * (A.double() @ W.double().t() + b.double())
        if rowmask is not None:
            upd = upd * rowmask[:, None].double()
        ref, tol = (x.double() + upd).float(), 6e-4
    else:
        epi = {"store": G.EPI_STORE_T, "rope": G.EPI_ROPE_T}[epi_name]
        act = "gelu_tanh" if epi_name == "store" else "none"
        rope, heads = None, 0
        if epi_name == "rope":
            ang = torch.rand(seq, 32, generator=g) * 6.28
            rope, heads = torch.stack([ang.cos(), ang.sin()], dim=-1), (1 if N == 3072 else N // 3 // 64)
        run = lambda: G.op_linear_fused(1, epi, A, W, b, act, None, None, rope, heads, seq)
        ref, tol = _fused_ref(epi, A, W, b, act, None, None, rope, heads, seq), 3e-3
    _lib.check(lib.f5_tuning_set(b"gemm_w4_bm", bm))
    try:
        new = run()
    finally:
        _lib.check(lib.f5_tuning_set(b"gemm_w4_bm", 0))
    old = _with_knob(lib, b"gemm_w4", 0, run)
    assert rel_l2(new, ref) < tol and rel_l2(old, ref) < tol
    assert torch.equal(new, old)


@pytest.mark.parametrize("bm", [256, 128], ids=["256_row_tiles", "128_row_tiles"])
@pytest.mark.parametrize("N,Kb,epi", [(3072, 1024, 4), (2048, 2048, 0), (3072, 1024, -16)])
def test_w4_kernel_layernorm_fold_site_equals_the_8_wave_kernel(bm, N, Kb, epi):
    """The LayerNorm fold of one call site at a size the one-wave-per-SIMD kernel takes (M = 16 384 token rows: producer 64 x 4, consumer 64 x 8 / 12
    tiles): in-place residual epilogue with partial row statistics -> statistics -> fp16-operand projection with the fold epilogue (+ RoPE / GELU).
    Stream, statistics and output bit for bit against the 8-wave kernel (gemm_w4 = 0)."""
    import gpu_helpers as G
    from eraxvif5tts_amd import _lib
    lib = _lib.load()
    M, D = 16384, 1024
    rope_heads = 1
    if epi < 0:
        rope_heads, epi = -epi, 4
    g = torch.Generator().manual_seed(N + Kb + epi)
    x = torch.randn(M, D, generator=g) * 1.7 + torch.randn(M, 1, generator=g) * 0.5
    A = G.bf16_round(torch.randn(M, Kb, generator=g))
    Wo = G.bf16_round(torch.randn(D, Kb, generator=g) / Kb ** 0.5)
    bo, gate = torch.randn(D, generator=g) * 0.1, torch.randn(D, generator=g) * 0.3
    W, bias = torch.randn(N, D, generator=g) / D ** 0.5, torch.randn(N, generator=g) * 0.1
    scale, shift = torch.randn(D, generator=g) * 0.2, torch.randn(D, generator=g) * 0.3
    seq = 2048
    rope = None
    if epi == 4:
        ang = torch.arange(seq)[:, None] * (1.0 / 10000.0 ** (torch.arange(0, 64, 2) / 64.0))[None, :]
        rope = torch.stack([ang.cos(), ang.sin()], dim=-1).reshape(seq, 64).float()
    run = lambda: G.op_ln_fold(epi, x, A, Wo, bo, gate, W, bias, scale, shift, pivot=None, act="gelu_tanh" if epi == 0 else "none", rope=rope,
                               rope_heads=rope_heads, seq=seq)
    _lib.check(lib.f5_tuning_set(b"gemm_w4_bm", bm))
    try:
        new = run()
    finally:
        _lib.check(lib.f5_tuning_set(b"gemm_w4_bm", 0))
    old = _with_knob(lib, b"gemm_w4", 0, run)
    for a, c in zip(new, old):
        assert torch.isfinite(a).all() and torch.equal(a, c)
