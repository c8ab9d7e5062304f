"""The CPU oracle (oracle/cpu_ref.py) against golden vectors produced by the REAL reference code
(oracle/make_golden.py, run in the build container).  fp32 vs fp32: tolerance 2e-5 rel-L2 (op order only)."""
import numpy as np
import pytest
import torch

from conftest import golden_arch, golden_weights, load_golden, rel_l2
from oracle import cpu_ref

TOL = 2e-5


@pytest.mark.parametrize("name", ["tiny_base", "tiny_v1"])
def test_sample_end_to_end(name):
    z = load_golden(name)
    cfg, W = golden_arch(z), golden_weights(z)
    out, traj = cpu_ref.sample(W, cfg, torch.from_numpy(z["cond"]), torch.from_numpy(z["text"]), torch.from_numpy(z["duration"]),
                               lens=torch.from_numpy(z["lens"]), steps=int(z["steps"]), cfg_strength=float(z["cfg_strength"]),
                               sway_sampling_coef=float(z["sway"]), seed=int(z["seed"]))
    assert torch.equal(traj[0], torch.from_numpy(z["y0"]))  # CPU mt19937 stream reproduces cfm.py:178-183
    assert rel_l2(traj, z["traj"]) < TOL
    assert rel_l2(out, z["out"]) < TOL
    # rows past each sample's duration are not zeroed by the reference (cfm.py:200-202)
    assert out.shape == z["out"].shape


@pytest.mark.parametrize("name", ["tiny_base", "tiny_v1"])
@pytest.mark.parametrize("branch", ["trc", "tru"])
def test_forward_stage_traces(name, branch):
    z = load_golden(name)
    cfg, W = golden_arch(z), golden_weights(z)
    drop = branch == "tru"
    mask = cpu_ref.lens_to_mask(torch.from_numpy(z["duration"]))
    trace = {}
    cpu_ref.dit_forward(W, cfg, torch.from_numpy(z["trace_x"]), torch.from_numpy(z["trace_cond"]), torch.from_numpy(z["text"]),
                        torch.from_numpy(z["trace_t"]), drop, drop, mask=mask, trace=trace)
    for key in ["t_emb", "text_embed", "input_embed", "blk0.n1", "blk0.attn", "blk0.out", "blk1.n1", "blk1.attn", "blk1.out",
                "final_norm", "out"]:
        assert rel_l2(trace[key], z[f"{branch}.{key}"]) < TOL, key


OPTION_CASES = {"no_ref": dict(no_ref_audio=True), "dup": dict(duplicate_test=True), "edit": dict(use_edit_mask=True), "nolens": dict(int_duration=True)}


def option_kwargs(z, tag, as_cuda=False):
    """keyword arguments of one tiny_options.npz case for cpu_ref.sample / CFM.sample (same names in both)."""
    dev = (lambda t: t.cuda()) if as_cuda else (lambda t: t)
    o = OPTION_CASES[tag]
    kw = dict(steps=int(z["steps"]), cfg_strength=float(z["cfg_strength"]), sway_sampling_coef=float(z["sway"]))
    if o.get("int_duration"):
        kw["duration"] = int(z["int_duration"])
    else:
        kw["duration"], kw["lens"] = dev(torch.from_numpy(z["duration"])), dev(torch.from_numpy(z["lens"]))
    if o.get("no_ref_audio"):
        kw["no_ref_audio"] = True
    if o.get("duplicate_test"):
        kw["duplicate_test"], kw["t_inter"] = True, float(z["t_inter"])
    if o.get("use_edit_mask"):
        kw["edit_mask"] = dev(torch.from_numpy(z["edit_mask"]))
    return kw


@pytest.mark.parametrize("tag", list(OPTION_CASES))
def test_sample_options(tag):
    """The remaining switches of CFM.sample (cfm.py:123-125 edit_mask, 137-143 duplicate_test / no_ref_audio, 185-191 t_inter, lens=None with
    an integer duration) against the reference's own outputs on the tiny_base network."""
    z, zb = load_golden("tiny_options"), load_golden("tiny_base")
    cfg, W = golden_arch(zb), golden_weights(zb)
    out, traj = cpu_ref.sample(W, cfg, torch.from_numpy(z["cond"]), torch.from_numpy(z["text"]), seed=int(z["seed"]), **option_kwargs(z, tag))
    assert traj.shape == z["traj_" + tag].shape
    assert rel_l2(traj, z["traj_" + tag]) < TOL
    assert rel_l2(out, z["out_" + tag]) < TOL


def test_b1_midpoint_and_cfg0():
    z = load_golden("tiny_b1_midpoint")
    cfg, W = golden_arch(z), golden_weights(z)
    for tag, cs in (("cfg0", 0.0), ("cfg2", 2.0)):
        out, traj = cpu_ref.sample(W, cfg, torch.from_numpy(z["cond"]), torch.from_numpy(z["text"]), int(z["duration"]),
                                   steps=int(z["steps"]), cfg_strength=cs, sway_sampling_coef=float(z["sway"]), seed=int(z["seed"]),
                                   method="midpoint")
        assert out.shape[1] == 41  # duration rule: max(text_len, lens) + 1 beats the requested 30 (cfm.py:132-135)
        assert rel_l2(traj, z["traj_" + tag]) < TOL
        assert rel_l2(out, z["out_" + tag]) < TOL


@pytest.mark.parametrize("fixture,v1", [("base_fwd", False), ("v1_fwd", True)])
def test_true_size_base_forward(fixture, v1):
    """F5TTS_Base (pe_attn_head=1) and F5TTS_v1_Base (all-head RoPE, text mask padding) dims, 1024 x 22 layers: weights regenerated from the
    seed on this machine, outputs from the reference's own DiT.forward."""
    z = load_golden(fixture)
    cfg = dict(dim=1024, depth=22, heads=16, ff_mult=2, text_dim=512, text_mask_padding=v1, conv_layers=4, pe_attn_head=None if v1 else 1)
    W = cpu_ref.random_dit_weights(cfg, int(z["vocab"]), seed=int(z["seed"]))
    for drop, key in ((False, "out_c"), (True, "out_u")):
        out = cpu_ref.dit_forward(W, cfg, torch.from_numpy(z["x"]), torch.from_numpy(z["cond"]), torch.from_numpy(z["text"]),
                                  torch.from_numpy(z["t"]), drop, drop, mask=torch.from_numpy(z["mask"]))
        assert rel_l2(out, z[key]) < 5e-5, key


@pytest.mark.parametrize("which", ["b1", "b2"])
def test_true_size_base_forward_at_production_length(which):
    """base_fwd_1024.npz: the reference's own true-size F5TTS_Base evaluated once at N = 1024 (B = 1, no mask) and N = 1000 (B = 2, key mask);
    inputs regenerated from seeds (checksums stored), outputs from the reference."""
    z = load_golden("base_fwd_1024")
    cfg = dict(dim=1024, depth=22, heads=16, ff_mult=2, text_dim=512, text_mask_padding=False, conv_layers=4, pe_attn_head=1)
    W = cpu_ref.random_dit_weights(cfg, int(z["vocab"]), seed=int(z["seed"]))
    x, cond, text, mask, t, _ = cpu_ref.fwd_1024_inputs(which)
    assert float(x.double().sum()) == float(z[which + ".x_sum"]) and float(cond.double().sum()) == float(z[which + ".cond_sum"])
    assert int(text.sum()) == int(z[which + ".text_sum"])
    for drop, key in ((False, "out_c"), (True, "out_u")):
        out = cpu_ref.dit_forward(W, cfg, x, cond, text, t, drop, drop, mask=mask)
        assert rel_l2(out, z[f"{which}.{key}"]) < 5e-5, key


def _unett_case(z, tag):
    import ast
    arch = ast.literal_eval(str(z[f"{tag}.arch"]))
    W = cpu_ref.random_unett_weights(arch, int(z[f"{tag}.vocab"]), seed=int(z[f"{tag}.seed"]))
    return arch, W


@pytest.mark.parametrize("tag", ["a", "b"])
def test_unett_forward_and_sample_match_reference(tag):
    """the oracle's restatement of the reference's UNetT (backbones/unett.py:185-253: time token prepended, RMSNorm, skip
    stack with concat projections) -- one forward of both CFG branches with a key mask, and CFM.sample driving it -- against vectors the
    reference's own unett.py + cfm.py produced (E2TTS-like arch without text blocks; a variant with text blocks, text mask padding, all-head RoPE)."""
    z = load_golden("tiny_unett")
    arch, W = _unett_case(z, tag)
    g = lambda k: torch.from_numpy(z[f"{tag}.{k}"])
    for drop, key in ((False, "out_c"), (True, "out_u")):
        out = cpu_ref.unett_forward(W, arch, g("x"), g("cond"), g("text"), g("t"), drop, drop, mask=g("mask"))
        assert rel_l2(out, z[f"{tag}.{key}"]) < 2e-5, key
    out, traj = cpu_ref.sample(W, dict(arch, backbone="UNetT"), g("cond")[:, :16], g("text"), g("duration"), lens=g("lens"), steps=4, cfg_strength=2.0,
                               sway_sampling_coef=-1.0, seed=5)
    assert rel_l2(traj, z[f"{tag}.sample_traj"]) < 2e-5 and rel_l2(out, z[f"{tag}.sample_out"]) < 2e-5


def test_true_size_e2tts_forward():
    """E2TTS_Base dims (configs/E2TTS_Base.yaml:25-31: UNetT, 1024 x 24 layers, ff_mult 4): weights regenerated from the seed on this machine."""
    z = load_golden("e2tts_fwd")
    cfg = dict(dim=1024, depth=24, heads=16, ff_mult=4, text_mask_padding=False, pe_attn_head=1)
    W = cpu_ref.random_unett_weights(cfg, int(z["vocab"]), seed=int(z["seed"]))
    for drop, key in ((False, "out_c"), (True, "out_u")):
        out = cpu_ref.unett_forward(W, cfg, torch.from_numpy(z["x"]), torch.from_numpy(z["cond"]), torch.from_numpy(z["text"]),
                                    torch.from_numpy(z["t"]), drop, drop, mask=torch.from_numpy(z["mask"]))
        assert rel_l2(out, z[key]) < 5e-5, key


@pytest.mark.parametrize("tag", ["qk", "ls", "both"])
def test_dit_constructor_switches_match_reference(tag):
    """qk_norm = "rms_norm" (RMSNorm(dim_head, eps 1e-6) on q and k of every head before RoPE, modules.py:275-294,463-467) and
    long_skip_connection = True (Linear(2 dim -> dim) on cat(x, input embedding) after the blocks, dit.py:153,217-228): the oracle's restatement
    against vectors the reference's own DiT produced with the switch(es) on -- forward of both CFG branches and CFM.sample."""
    import ast
    z = load_golden("tiny_switches")
    arch = ast.literal_eval(str(z[f"{tag}.arch"]))
    W = cpu_ref.random_dit_weights(arch, int(z[f"{tag}.vocab"]), seed=int(z[f"{tag}.seed"]))
    g = lambda k: torch.from_numpy(z[f"{tag}.{k}"])
    for drop, key in ((False, "out_c"), (True, "out_u")):
        out = cpu_ref.dit_forward(W, arch, g("x"), g("cond"), g("text"), g("t"), drop, drop, mask=g("mask"))
        assert rel_l2(out, z[f"{tag}.{key}"]) < 2e-5, key
    out, traj = cpu_ref.sample(W, arch, g("cond")[:, :16], g("text"), g("duration"), lens=g("lens"), steps=4, cfg_strength=2.0,
                               sway_sampling_coef=-1.0, seed=5)
    assert rel_l2(traj, z[f"{tag}.sample_traj"]) < 2e-5 and rel_l2(out, z[f"{tag}.sample_out"]) < 2e-5


def test_true_size_sampler_matches_reference():
    """CFM.sample of the reference over its true-size F5TTS_Base DiT (B = 2, unequal durations -> key mask, NFE 3, CFG 2, sway -1, seeded noise)
    vs the oracle's sample() on the same weights (regenerated from the seed) and the same initial noise (trajectory row 0)."""
    z = load_golden("base_sample")
    cfg = dict(dim=1024, depth=22, heads=16, ff_mult=2, text_dim=512, text_mask_padding=False, conv_layers=4, pe_attn_head=1)
    W = cpu_ref.random_dit_weights(cfg, int(z["vocab"]), seed=int(z["seed"]))
    g = lambda k: torch.from_numpy(z[k])
    out, traj = cpu_ref.sample(W, cfg, g("cond"), g("text"), g("duration"), lens=g("lens"), steps=3, cfg_strength=2.0, sway_sampling_coef=-1.0,
                               y0=g("traj")[0])
    assert rel_l2(traj, z["traj"]) < 5e-5 and rel_l2(out, z["out"]) < 5e-5
    # the reference's own seeded noise: manual_seed(11) then randn(duration, 100) per sample (cfm.py:178-183)
    out2, _ = cpu_ref.sample(W, cfg, g("cond"), g("text"), g("duration"), lens=g("lens"), steps=1, cfg_strength=0.0, seed=11)
    torch.manual_seed(11)
    assert torch.equal(g("traj")[0, 0], torch.randn(96, 100))


def _mmdit_case(z, tag):
    import ast
    arch = ast.literal_eval(str(z[f"{tag}.arch"]))
    W = cpu_ref.random_mmdit_weights(arch, int(z[f"{tag}.vocab"]), seed=int(z[f"{tag}.seed"]))
    return arch, W


@pytest.mark.parametrize("tag", ["a", "b"])
def test_mmdit_forward_and_sample_match_reference(tag):
    """the oracle's restatement of the reference's MMDiT (backbones/mmdit.py:146-190, MMDiTBlock / JointAttnProcessor
    modules.py:509-707: text stream of its own length, joint attention over [frames | text], context_pre_only last block) -- one forward
    of both CFG branches with a key mask, CFM.sample driving it at B = 2 (mask) and B = 1 (no mask, midpoint) -- against vectors the
    reference's own mmdit.py + cfm.py produced."""
    z = load_golden("tiny_mmdit")
    arch, W = _mmdit_case(z, tag)
    g = lambda k: torch.from_numpy(z[f"{tag}.{k}"])
    for drop, key in ((False, "out_c"), (True, "out_u")):
        out = cpu_ref.mmdit_forward(W, arch, g("x"), g("cond"), g("text"), g("t"), drop, drop, mask=g("mask"))
        assert rel_l2(out, z[f"{tag}.{key}"]) < 2e-5, key
    cfg = dict(arch, backbone="MMDiT")
    out, traj = cpu_ref.sample(W, cfg, g("cond")[:, :16], g("text"), g("duration"), lens=g("lens"), steps=4, cfg_strength=2.0,
                               sway_sampling_coef=-1.0, seed=5)
    assert rel_l2(traj, z[f"{tag}.sample_traj"]) < 2e-5 and rel_l2(out, z[f"{tag}.sample_out"]) < 2e-5
    out, traj = cpu_ref.sample(W, cfg, g("cond")[:1, :16], g("text")[:1], 36, lens=g("lens")[:1], steps=3, cfg_strength=1.5, seed=9, method="midpoint")
    assert rel_l2(traj, z[f"{tag}.b1_traj"]) < 2e-5 and rel_l2(out, z[f"{tag}.b1_out"]) < 2e-5


def test_mmdit_forward_at_the_quoted_size():
    """MMDiT at the size the reference quotes for it (scripts/count_params_gflops.py:18: dim 512, depth 16, heads 16, ff_mult 2): weights
    regenerated from the seed on this machine."""
    z = load_golden("mmdit_fwd")
    cfg = dict(dim=512, depth=16, heads=16, ff_mult=2, text_mask_padding=True)
    W = cpu_ref.random_mmdit_weights(cfg, int(z["vocab"]), seed=int(z["seed"]))
    for drop, key in ((False, "out_c"), (True, "out_u")):
        out = cpu_ref.mmdit_forward(W, cfg, torch.from_numpy(z["x"]), torch.from_numpy(z["cond"]), torch.from_numpy(z["text"]),
                                    torch.from_numpy(z["t"]), drop, drop, mask=torch.from_numpy(z["mask"]))
        assert rel_l2(out, z[key]) < 5e-5, key


def test_time_grid_sway():
    t = cpu_ref.time_grid(32, -1.0)
    ref = 1 - torch.cos(torch.pi / 2 * torch.linspace(0, 1, 33))
    assert torch.allclose(t, ref, atol=1e-6)
    assert float(t[0]) == 0.0 and abs(float(t[-1]) - 1.0) < 1e-6


def test_mel_and_istft_consistency():
    """a3/a22 are 'parity unpinned' (torchaudio / vocos absent): cross-check the restatements against
    independent formulations (scipy STFT; torch.istft)."""
    import scipy.signal
    g = torch.Generator().manual_seed(0)
    wav = torch.randn(1, 24000 // 4, generator=g) * 0.1
    mel = cpu_ref.mel_spectrogram(wav)
    assert mel.shape == (1, 100, wav.shape[1] // 256 + 1)
    x = np.pad(wav[0].numpy().astype(np.float64), 512, mode="reflect")
    _, _, Z = scipy.signal.stft(x, window=scipy.signal.get_window("hann", 1024, fftbins=True), nperseg=1024, noverlap=768,
                                nfft=1024, boundary=None, padded=False, scaling="spectrum")
    mag = np.abs(Z) * scipy.signal.get_window("hann", 1024, fftbins=True).sum()
    fb = cpu_ref.mel_filterbank().double().numpy()
    ref = np.log(np.clip(fb.T @ mag, 1e-5, None))
    assert np.abs(mel[0].numpy() - ref).max() < 2e-3
    # ISTFT restatement vs torch.istft
    T = 9
    re, im = torch.randn(2, 513, T, generator=g), torch.randn(2, 513, T, generator=g)
    im[:, 0] = 0
    im[:, -1] = 0
    mine = cpu_ref.istft_center(re, im)
    ref = torch.istft(torch.complex(re, im), 1024, hop_length=256, win_length=1024, window=torch.hann_window(1024), center=True)
    assert mine.shape == ref.shape == (2, (T - 1) * 256)
    assert torch.allclose(mine, ref, atol=1e-5)


def test_duration_predictor_speaker_conditioning_matches_reference_fixture():
    """the g path of oracle/cpu_ref.duration_predictor vs the reference DurationPredictor(gin_channels = 12) (tests/golden/duration_predictor_g.npz)."""
    g = load_golden("duration_predictor_g")
    W = {k[2:]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith("w.")}
    tokens, mask = torch.from_numpy(g["tokens"]), torch.from_numpy(g["mask"])
    g1, gt = torch.from_numpy(g["g1"]), torch.from_numpy(g["gt"])
    assert rel_l2(cpu_ref.duration_predictor(W, tokens, mask, g_cond=g1), torch.from_numpy(g["out_g1"])) < 2e-6
    assert rel_l2(cpu_ref.duration_predictor(W, tokens, mask, g_cond=gt), torch.from_numpy(g["out_gt"])) < 2e-6
    assert rel_l2(cpu_ref.duration_predictor(W, tokens.clamp(min=0), mask, add_one=False, g_cond=g1), torch.from_numpy(g["out_phoneme_g1"])) < 2e-6


def test_duration_predictor_matches_reference_fixture():
    """oracle/cpu_ref.duration_predictor vs the reference DurationPredictor's own outputs (tests/golden/duration_predictor.npz)."""
    g = load_golden("duration_predictor")
    W = {k[2:]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith("w.")}
    tokens, mask = torch.from_numpy(g["tokens"]), torch.from_numpy(g["mask"])
    assert rel_l2(cpu_ref.duration_predictor(W, tokens, mask), torch.from_numpy(g["out"])) < 2e-6
    assert rel_l2(cpu_ref.duration_predictor(W, tokens.clamp(min=0), mask, add_one=False), torch.from_numpy(g["out_phoneme"])) < 2e-6
    assert rel_l2(cpu_ref.duration_predictor(W, tokens[2:3, :1], mask[2:3, :1]), torch.from_numpy(g["out_one_token"])) < 2e-6
    assert torch.count_nonzero(torch.from_numpy(g["out"])[mask.unsqueeze(1) == 0]) == 0  # padded tokens carry no duration


def test_rope_layout_switch_of_the_oracle():
    """The half-split rotary form behind cfg["rope_layout"]: permuting the head features (new 2j <- j, new 2j+1 <- j+32) turns it
    into the adjacent-pair form exactly -- the identity the HIP path uses to serve both layouts with one kernel (csrc/model.hip)."""
    g = torch.Generator().manual_seed(0)
    t = torch.randn(2, 3, 11, 64, generator=g)
    ang = cpu_ref.rope_angles(11)
    perm = torch.stack([torch.arange(32), torch.arange(32) + 32], dim=1).flatten()  # new feature index -> old feature index
    a = cpu_ref.apply_rope(t, ang, half_split=True)[..., perm]
    b = cpu_ref.apply_rope(t[..., perm], ang, half_split=False)
    assert torch.equal(a, b)
    assert not torch.allclose(cpu_ref.apply_rope(t, ang, True), cpu_ref.apply_rope(t, ang, False))


def test_bigvgan_mel_front_end_matches_reference():
    """the other vocoder's mel front-end.  tests/golden/bigvgan_mel.npz holds what the reference's own get_bigvgan_mel_spectrogram
    (modules.py:29-72) returned for seeded waveforms; the oracle restates that function.  The one third-party piece, librosa's filterbank (absent
    from the reference tree), is restated from the published algorithm and checked through its defining properties: Slaney normalisation = unit
    area in Hz for every filter that lies inside the spectrum, one peak per filter, band edges equally spaced on the Slaney mel scale."""
    import numpy as np
    z = load_golden("bigvgan_mel")
    wave = torch.from_numpy(z["wave"])
    mel = cpu_ref.bigvgan_mel_spectrogram(wave)
    assert mel.shape == z["mel"].shape == (2, 100, 93) and (mel - torch.from_numpy(z["mel"])).abs().max() < 1e-5
    odd = cpu_ref.bigvgan_mel_spectrogram(wave[:1, :7777])
    assert odd.shape == z["mel_7777"].shape == (1, 100, 30) and (odd - torch.from_numpy(z["mel_7777"])).abs().max() < 1e-5
    fb = cpu_ref.librosa_mel_filterbank(24000, 1024, 100)
    assert fb.shape == (100, 513) and np.allclose(fb.sum(axis=1), z["fb_rowsum"]) and (fb >= 0).all()
    df = 24000 / 1024
    area = fb.astype(np.float64).sum(axis=1) * df  # integral over Hz of a filter sampled at the FFT bins
    # (the filters below ~1.5 kHz are about two bins wide: up to 12 % sampling error of the integral, not of the normalisation; the wide ones are exact)
    assert np.abs(area - 1.0).max() < 0.15 and np.abs(area[60:] - 1.0).max() < 5e-3
    peaks = fb.argmax(axis=1)
    assert (np.diff(peaks) > 0).all() and peaks[0] >= 1 and peaks[-1] <= 511
    # below 1 kHz the scale is linear: equal spacing of the band centres in Hz
    low = peaks[:20] * df
    assert np.abs(np.diff(low) - np.diff(low).mean()).max() <= df + 1e-9
