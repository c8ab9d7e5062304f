"""Shared helpers of the GPU parity tests (everything goes through the C ABI of libf5hip.so)."""
import ctypes as C

import torch

from eraxvif5tts_amd import _lib
from eraxvif5tts_amd.model import CFM, DiT


def make_dit(arch, vocab, weights, precision):
    m = DiT(**arch, text_num_embeds=vocab, mel_dim=100, precision=precision)
    sd = m.state_dict()
    missing = [k for k in sd if k not in weights and k != "rotary_embed.inv_freq"]
    assert not missing, missing
    m.load_state_dict({k: v for k, v in weights.items() if k in sd}, strict=False)
    return m.cuda()


def make_cfm(arch, vocab, weights, precision, method="euler"):
    m = make_dit(arch, vocab, weights, precision)
    return CFM(transformer=m, mel_spec_kwargs={"mel_spec_type": "vocos"}, odeint_kwargs={"method": method}).cuda()


def op_linear(precision, kernel, A, W, bias=None, act="none"):
    lib = _lib.load()
    M, K = A.shape
    N = W.shape[0]
    A = A.cuda().float().contiguous()
    W = W.cuda().float().contiguous()
    b = None if bias is None else bias.cuda().float().contiguous()
    out = torch.empty(M, N, device="cuda")
    _lib.check(lib.f5_op_linear(precision, kernel, M, N, K, _lib.ptr(A), _lib.ptr(W), _lib.ptr(b), _lib.ACT[act], _lib.ptr(out), _lib.stream_ptr()))
    return out.cpu()


EPI_STORE_T, EPI_RESID, EPI_ROPE_T, EPI_GATE_T = 0, 2, 4, 5


def op_linear_fused(kernel, epi, A, W, bias, act="none", gate=None, rowmask=None, rope=None, rope_heads=0, seq=0, stream_in=None):
    """One DiT block linear with its fused epilogue (include/f5hip.h: f5_op_linear_fused).  epi 2 updates `stream_in` (the fp16 residual
    stream, given as f32) in place and returns it."""
    lib = _lib.load()
    M, K = A.shape
    N = W.shape[0]
    dev = [None if t is None else t.cuda().float().contiguous() for t in (A, W, bias, gate, rope)]
    mk = None if rowmask is None else rowmask.cuda().to(torch.uint8).contiguous()
    out = torch.empty(M, N, device="cuda") if stream_in is None else stream_in.cuda().float().contiguous().clone()
    _lib.check(lib.f5_op_linear_fused(kernel, epi, M, N, K, _lib.ptr(dev[0]), _lib.ptr(dev[1]), _lib.ptr(dev[2]), _lib.ACT[act],
                                      _lib.ptr(dev[3]), _lib.ptr(mk), _lib.ptr(dev[4]), rope_heads, seq, _lib.ptr(out), _lib.stream_ptr()))
    return out.cpu()


def op_attention(precision, kernel, qkv, mask=None):
    lib = _lib.load()
    B, N, three, H, dh = qkv.shape
    assert three == 3 and dh == 64
    q = qkv.cuda().float().contiguous()
    mk = None if mask is None else mask.cuda().to(torch.uint8).contiguous()
    out = torch.empty(B, N, H * 64, device="cuda")
    _lib.check(lib.f5_op_attention(precision, kernel, B, N, H, _lib.ptr(q), _lib.ptr(mk), _lib.ptr(out), _lib.stream_ptr()))
    return out.cpu()


def op_conv_pos(precision, x, w0, b0, w1, b1):
    lib = _lib.load()
    B, N, D = x.shape
    xs = [t.cuda().float().contiguous() for t in (x, w0, b0, w1, b1)]
    out = torch.empty(B, N, D, device="cuda")
    _lib.check(lib.f5_op_conv_pos_embed(precision, B, N, D, *[_lib.ptr(t) for t in xs], _lib.ptr(out), _lib.stream_ptr()))
    return out.cpu()


def op_ln_mod(x, scale, shift):
    lib = _lib.load()
    rows, dim = x.shape
    xs = [t.cuda().float().contiguous() for t in (x, scale, shift)]
    out = torch.empty(rows, dim, device="cuda")
    _lib.check(lib.f5_op_layernorm_modulate(rows, dim, *[_lib.ptr(t) for t in xs], _lib.ptr(out), _lib.stream_ptr()))
    torch.cuda.synchronize()
    return out.cpu()


def bf16_round(t):
    return t.to(torch.bfloat16).float()


def op_ln_fold(epi, x, A, Wo, bo, gate, W, bias, scale, shift, pivot=None, act="none", rope=None, rope_heads=0, seq=0):
    """include/f5hip.h: f5_op_ln_fold.  Returns (updated fp16 stream as f32 [M, D], stats [M, 2] = (mean, rstd), out [M, N])."""
    lib = _lib.load()
    M, D = x.shape
    N, Kb = W.shape[0], A.shape[1]
    dev = [None if t is None else t.cuda().float().contiguous() for t in (A, Wo, bo, gate, pivot, W, bias, scale, shift, rope)]
    xs = x.cuda().float().contiguous().clone()
    stats = torch.empty(M, 2, device="cuda")
    out = torch.empty(M, N, device="cuda")
    _lib.check(lib.f5_op_ln_fold(epi, M, D, N, Kb, _lib.ptr(xs), *[_lib.ptr(t) for t in dev[:9]], _lib.ACT[act], _lib.ptr(dev[9]), rope_heads, seq,
                                 _lib.ptr(stats), _lib.ptr(out), _lib.stream_ptr()))
    return xs.cpu(), stats.cpu(), out.cpu()
