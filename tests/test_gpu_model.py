"""End-to-end parity of the HIP path (DiT.forward stage taps, CFM.sample) against the golden vectors captured from the
reference and against the CPU oracle.

Stated tolerances (SURVEY.md 8c calibration: the reference model itself in plain bf16 deviates 1.1e-2..1.2e-2 rel-L2 from
its own fp32 run): fp32 mode rel-L2 <= 2e-4 end to end; bf16 production mode rel-L2 <= 2e-2 on the generated frames."""
import numpy as np
import pytest
import torch

from conftest import golden_arch, golden_weights, load_golden, rel_l2
from oracle import cpu_ref

pytestmark = pytest.mark.gpu
TOL = {"fp32": 2e-4, "bf16": 2e-2}
STAGE_TOL = {"fp32": 1e-4, "bf16": 1.5e-2}


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    from eraxvif5tts_amd import _lib
    _lib.require_gpu()


@pytest.fixture(params=["reference-kernels", "tuned-kernels"], autouse=True)
def kernels(request, monkeypatch):
    """every model-level test runs twice: reference tile kernels only / tuned kernels forced wherever they support the
    problem (fp32 mode always uses the fp32-input MFMA tile kernels)."""
    v = "0" if request.param == "reference-kernels" else "1"
    monkeypatch.setenv("F5HIP_GEMM_KERNEL", v)
    monkeypatch.setenv("F5HIP_ATTN_KERNEL", v)
    return request.param


def _gen_rows(t, dur):
    """concatenate the rows < duration of every batch element (what callers slice out)."""
    return torch.cat([t[..., b, : int(d), :].reshape(-1, t.shape[-1]) for b, d in enumerate(dur)])


@pytest.mark.parametrize("name", ["tiny_base", "tiny_v1"])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_forward_stage_taps(name, prec):
    import gpu_helpers as G
    z = load_golden(name)
    arch, W = golden_arch(z), golden_weights(z)
    m = G.make_dit(arch, int(z["vocab"]), W, prec)
    x, cond, text = [torch.from_numpy(z[k]).cuda() for k in ("trace_x", "trace_cond", "text")]
    dur = torch.from_numpy(z["duration"])
    B, N, D = x.shape[0], x.shape[1], arch["dim"]
    mask = cpu_ref.lens_to_mask(dur).cuda()
    t = torch.from_numpy(z["trace_t"]).cuda()
    for drop, tag in ((False, "trc"), (True, "tru")):
        plan = m.plan(B, N, 1)
        taps = {"t_emb": torch.zeros(B, D), "input_embed": torch.zeros(B, N, D), "final_norm": torch.zeros(B, N, D)}
        for i in range(arch["depth"]):
            for s in ("n1", "attn", "out"):
                taps[f"blk{i}.{s}"] = torch.zeros(B, N, D)
        taps = {k: v.cuda() for k, v in taps.items()}
        for k, v in taps.items():
            m.set_tap(plan, k, v)
        out = m(x=x, cond=cond, text=text, time=t, mask=mask, drop_audio_cond=drop, drop_text=drop, cache=False)
        torch.cuda.synchronize()
        m.set_tap(plan, None, None)
        valid = mask.cpu()
        for k, v in taps.items():
            ref = torch.from_numpy(z[f"{tag}.{k}"])
            got = v.cpu()
            if k.endswith(".attn"):  # the module zero-fills padded query rows (modules.py:499-501); the fused kernel skips them
                got, ref = got[valid], ref[valid]
            assert rel_l2(got, ref) < STAGE_TOL[prec], (k, rel_l2(got, ref))
        assert rel_l2(out.cpu(), z[f"{tag}.out"]) < STAGE_TOL[prec]


@pytest.mark.parametrize("name", ["tiny_base", "tiny_v1"])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("graph", [False, True])
def test_sample_matches_reference_golden(name, prec, graph):
    import gpu_helpers as G
    z = load_golden(name)
    arch, W = golden_arch(z), golden_weights(z)
    c = G.make_cfm(arch, int(z["vocab"]), W, prec)
    out, traj = c.sample(cond=torch.from_numpy(z["cond"]).cuda(), text=torch.from_numpy(z["text"]).cuda(),
                         duration=torch.from_numpy(z["duration"]).cuda(), lens=torch.from_numpy(z["lens"]).cuda(), steps=int(z["steps"]),
                         cfg_strength=float(z["cfg_strength"]), sway_sampling_coef=float(z["sway"]), y0=torch.from_numpy(z["y0"]),
                         use_graph=graph)
    dur = z["duration"]
    assert out.shape == z["out"].shape and traj.shape == z["traj"].shape
    assert rel_l2(_gen_rows(out.cpu(), dur), _gen_rows(torch.from_numpy(z["out"]), dur)) < TOL[prec]
    assert rel_l2(_gen_rows(traj.cpu(), dur), _gen_rows(torch.from_numpy(z["traj"]), dur)) < TOL[prec]
    # prompt frames are the conditioning itself (cfm.py:200-202)
    lens = z["lens"]
    for b in range(out.shape[0]):
        assert torch.equal(out[b, : lens[b]].cpu(), torch.from_numpy(z["cond"])[b, : lens[b]])


def test_graph_replay_is_bit_identical_to_eager():
    import gpu_helpers as G
    z = load_golden("tiny_base")
    arch, W = golden_arch(z), golden_weights(z)
    c = G.make_cfm(arch, int(z["vocab"]), W, "bf16")
    kw = dict(cond=torch.from_numpy(z["cond"]).cuda(), text=torch.from_numpy(z["text"]).cuda(), duration=torch.from_numpy(z["duration"]).cuda(),
              lens=torch.from_numpy(z["lens"]).cuda(), steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, y0=torch.from_numpy(z["y0"]))
    a, _ = c.sample(**kw, use_graph=False)
    b, _ = c.sample(**kw, use_graph=True)   # capture + first replay
    d, _ = c.sample(**kw, use_graph=True)   # cached replay
    assert torch.equal(a, b) and torch.equal(a, d)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_b1_midpoint_and_cfg0(prec):
    import gpu_helpers as G
    z = load_golden("tiny_b1_midpoint")
    arch, W = golden_arch(z), golden_weights(z)
    c = G.make_cfm(arch, int(z["vocab"]), W, prec, method="midpoint")
    for tag, cs in (("cfg0", 0.0), ("cfg2", 2.0)):
        out, traj = c.sample(cond=torch.from_numpy(z["cond"]).cuda(), text=torch.from_numpy(z["text"]).cuda(), duration=int(z["duration"]),
                             steps=int(z["steps"]), cfg_strength=cs, sway_sampling_coef=float(z["sway"]), y0=torch.from_numpy(z["traj_" + tag][0]))
        assert out.shape[1] == 41  # duration rule of cfm.py:132-135
        assert rel_l2(out.cpu(), z["out_" + tag]) < TOL[prec]
        assert rel_l2(traj.cpu(), z["traj_" + tag]) < TOL[prec]


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_true_size_base_forward(prec):
    """F5TTS_Base dimensions (1024 x 22 layers x 16 heads, pe_attn_head=1): one network evaluation vs the reference golden."""
    import gpu_helpers as G
    z = load_golden("base_fwd")
    cfg = dict(dim=1024, depth=22, heads=16, ff_mult=2, text_dim=512, text_mask_padding=False, conv_layers=4, pe_attn_head=1)
    W = cpu_ref.random_dit_weights(cfg, int(z["vocab"]), seed=int(z["seed"]))
    m = G.make_dit(cfg, int(z["vocab"]), W, prec)
    x, cond, text, mask, t = [torch.from_numpy(z[k]).cuda() for k in ("x", "cond", "text", "mask", "t")]
    for drop, key in ((False, "out_c"), (True, "out_u")):
        out = m(x=x, cond=cond, text=text, time=t, mask=mask, drop_audio_cond=drop, drop_text=drop)
        assert rel_l2(out.cpu(), z[key]) < STAGE_TOL[prec], key


# ----------------------------------------------------------------------------- duration predictor
def _golden_duration_net():
    from eraxvif5tts_amd.model import DurationPredictor
    g = load_golden("duration_predictor")
    net = DurationPredictor(text_num_embeds=int(g["text_num_embeds"]), in_channels=int(g["in_channels"]), filter_channels=int(g["filter_channels"]),
                            kernel_size=int(g["kernel_size"]), p_dropout=0.5)
    net.load_state_dict({k[2:]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith("w.")}, strict=True)
    return g, net.eval().cuda()


def test_duration_predictor_hip_matches_reference():
    g, net = _golden_duration_net()
    tokens, mask = torch.from_numpy(g["tokens"]).cuda(), torch.from_numpy(g["mask"]).cuda()
    out = net(tokens, mask).cpu()
    assert out.shape == g["out"].shape
    assert rel_l2(out, torch.from_numpy(g["out"])) < 1e-5          # fp32 kernels vs the reference module's fp32 output
    assert torch.count_nonzero(out[torch.from_numpy(g["mask"]).unsqueeze(1) == 0]) == 0
    assert rel_l2(net.phoneme_forward(tokens.clamp(min=0), mask).cpu(), torch.from_numpy(g["out_phoneme"])) < 1e-5
    assert rel_l2(net(tokens[2:3, :1], mask[2:3, :1]).cpu(), torch.from_numpy(g["out_one_token"])) < 1e-5
    with pytest.raises(NotImplementedError):
        net(tokens, mask, g=torch.zeros(4, 8, 1).cuda())


def test_duration_predictor_larger_shapes_vs_oracle():
    """Sizes a real predictor would have (text_dim 512, 256 filters, k = 5, 300 tokens) against the CPU oracle on seeded weights."""
    from eraxvif5tts_amd.model import DurationPredictor
    torch.manual_seed(11)
    net = DurationPredictor(text_num_embeds=2545, in_channels=512, filter_channels=256, kernel_size=5, p_dropout=0.1).eval()
    lens = torch.tensor([300, 151, 77])
    tokens = torch.randint(0, 2545, (3, 300))
    tokens = torch.where(torch.arange(300)[None, :] < lens[:, None], tokens, torch.full_like(tokens, -1))
    mask = (torch.arange(300)[None, :] < lens[:, None]).int()
    ref = cpu_ref.duration_predictor({k: v.detach() for k, v in net.state_dict().items()}, tokens, mask)
    out = net.cuda()(tokens.cuda(), mask.cuda()).cpu()
    assert rel_l2(out, ref) < 1e-5


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("pe_heads", [1, None])
def test_rope_layout_switch_half_split(prec, pe_heads):
    """x_transformers is not vendored, so the rotary pairing sits behind ONE switch (f5_dit_config.rope_layout / DiT(rope_layout=),
    oracle cfg["rope_layout"]).  The half-split form (frequency j turns features j and j+32) on the HIP path against the oracle with
    the same switch; and it must differ from the default adjacent-pair form (otherwise the switch would be dead)."""
    import gpu_helpers as G
    arch = dict(dim=128, depth=2, heads=2, ff_mult=2, text_dim=64, conv_layers=1, pe_attn_head=pe_heads, text_mask_padding=False)
    V = 40
    W = cpu_ref.random_dit_weights(arch, V, seed=31)
    g = torch.Generator().manual_seed(32)
    B, N = 2, 70
    x, cond = torch.randn(B, N, 100, generator=g), torch.randn(B, N, 100, generator=g)
    text = torch.randint(0, V, (B, 20), generator=g)
    t = torch.tensor([0.3, 0.7])
    dur = torch.tensor([N, N - 9])
    mask = cpu_ref.lens_to_mask(dur)
    ref = {lay: cpu_ref.dit_forward(W, {**arch, "rope_layout": lay}, x, cond, text, t, False, False, mask=mask) for lay in ("adjacent", "half_split")}
    assert rel_l2(ref["half_split"], ref["adjacent"]) > 1e-2
    for lay in ("adjacent", "half_split"):
        m = G.make_dit({**arch, "rope_layout": lay}, V, W, prec)
        out = m(x=x.cuda(), cond=cond.cuda(), text=text.cuda(), time=t.cuda(), mask=mask.cuda(), drop_audio_cond=False, drop_text=False, cache=False)
        v = mask
        assert rel_l2(out.cpu()[v], ref[lay][v]) < STAGE_TOL[prec], lay
