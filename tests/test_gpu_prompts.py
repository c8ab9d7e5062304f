"""The batched-inference front (eval/prompts.py: the reference's utils_eval.get_inference_prompt + eval_infer_batch loop) over the HIP sampler:
bucketed batches built on the device, every utterance of a RAGGED bucket equal to its own batch-1 sample() bit for bit, the padded form within
tolerance of it, and the buckets through the utterance-sharded sampler."""
import pytest
import torch

from conftest import rel_l2
from oracle import cpu_ref

pytestmark = pytest.mark.gpu
CHARS = " abcdefghijklmnopqrstuvwxyz."


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    from eraxvif5tts_amd import _lib
    _lib.require_gpu()


def _cfm(prec):
    from eraxvif5tts_amd.model import CFM, DiT
    arch = dict(dim=128, depth=2, heads=2, ff_mult=2, text_dim=64, conv_layers=2, pe_attn_head=1, text_mask_padding=False)
    W = cpu_ref.random_dit_weights(arch, len(CHARS), seed=61)
    m = DiT(**arch, text_num_embeds=len(CHARS), mel_dim=100, precision=prec)
    m.load_state_dict({k: v for k, v in W.items() if k in m.state_dict()}, strict=False)
    return CFM(transformer=m.cuda(), mel_spec_kwargs={"mel_spec_type": "vocos"}, vocab_char_map={c: i for i, c in enumerate(CHARS)}).cuda()


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_ragged_buckets_equal_batch1_samples_bit_for_bit(prec, monkeypatch):
    from eraxvif5tts_amd.eval import prompts as P
    from eraxvif5tts_amd.eval.sharded import sample_sharded
    monkeypatch.setenv("F5HIP_GEMM_KERNEL", "1")  # tuned kernels wherever they support the problem, in the batch-1 calls too
    monkeypatch.setenv("F5HIP_ATTN_KERNEL", "1")
    cfm = _cfm(prec)
    meta = P.synthetic_metainfo(14, seed=5, min_secs=3.2, max_secs=9.0)
    buckets = P.get_inference_prompt(meta, tokenizer="char", infer_batch_size=1400, num_buckets=8, min_secs=3, max_secs=40, device="cuda")
    assert sum(len(b[0]) for b in buckets) == 14 and max(len(b[0]) for b in buckets) >= 2
    assert all(P.ragged_ok(cfm, b) for b in buckets)
    kw = dict(nfe_step=3, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=11)
    got = {u: mel for u, mel, _ in P.infer_prompts(cfm, buckets, mode="ragged", **kw)}
    assert len(got) == 14
    for utts, _, ref_mels, ref_lens, totals, texts in buckets:
        for i, u in enumerate(utts):
            one, _ = cfm.sample(cond=ref_mels[i:i + 1].cuda(), text=[texts[i]], duration=int(totals[i]), lens=torch.tensor([ref_lens[i]]).cuda(), steps=3,
                                cfg_strength=2.0, sway_sampling_coef=-1.0, seed=11, return_trajectory=False, use_graph=False)
            want = one[0, ref_lens[i]: totals[i]].t()[None].float()
            assert got[u].shape == want.shape == (1, 100, totals[i] - ref_lens[i])
            assert torch.equal(got[u], want), u  # the ragged bucket gives every utterance the arithmetic of its own batch-1 call
    # mode "padded" is the reference's form: ONE padded + key-masked batch per bucket (eval_infer_batch.py:163-183).  It is NOT the batch-1
    # arithmetic -- the text embedding of a padded batch runs its ConvNeXt blocks (GRN: a norm over the sequence axis, modules.py:232-234)
    # over the bucket's longest length -- so it is checked against the oracle's padded batch, noise drawn as the CPU path draws it.
    from eraxvif5tts_amd.model.utils import list_str_to_idx
    cfm.noise_device = "cpu"
    bucket = max(buckets, key=lambda b: len(b[0]))
    padded = {u: mel for u, mel, _ in P.infer_prompts(cfm, [bucket], mode="padded", **kw)}
    cfm.noise_device = None
    utts, _, ref_mels, ref_lens, totals, texts = bucket
    W = cpu_ref.random_dit_weights(dict(dim=128, depth=2, heads=2, ff_mult=2, text_dim=64, conv_layers=2, pe_attn_head=1, text_mask_padding=False), len(CHARS), seed=61)
    arch = dict(dim=128, depth=2, heads=2, ff_mult=2, text_dim=64, conv_layers=2, pe_attn_head=1, text_mask_padding=False)
    ref, _ = cpu_ref.sample(W, arch, ref_mels.cpu().float(), list_str_to_idx(texts, cfm.vocab_char_map), torch.tensor(totals), lens=torch.tensor(ref_lens),
                            steps=3, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=11, return_trajectory=False)
    for i, u in enumerate(utts):
        assert rel_l2(padded[u][0].t().cpu(), ref[i, ref_lens[i]: totals[i]]) < {"fp32": 2e-4, "bf16": 2e-2}[prec], u
    # bucket shapes that recur replay a hipGraph (second call captures, third replays): same bits as the eager first pass
    for _ in range(2):
        again = {u: mel for u, mel, _ in P.infer_prompts(cfm, buckets, mode="ragged", **kw)}
        assert all(torch.equal(again[u], got[u]) for u in got)
    # the same buckets through the utterance-sharded sampler (one process: no collective), ragged sample_fn
    outs = sample_sharded(P.ragged_sample_fn(cfm), [P.sample_kwargs(b, seed=11, nfe_step=3) for b in buckets])
    flat = [u for b in buckets for u in b[0]]
    assert len(outs) == 14 and all(torch.equal(o.t()[None], got[u]) for o, u in zip(outs, flat))
