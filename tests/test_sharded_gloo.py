"""world_size-2 CPU (gloo) test of the utterance-sharded inference path: contiguous split, no collective inside sampling,
one all_gather of the finished mels, identical result and order on every rank."""
import os
import tempfile

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from eraxvif5tts_amd.eval.sharded import gather_utterances, sample_sharded, split_between_processes


def _fake_sample(cond, text, duration, lens, steps=2, **_):
    """deterministic stand-in for CFM.sample: depends on every input so ordering mistakes show."""
    b, n = cond.shape[0], int(duration.max())
    out = torch.zeros(b, n, 100)
    for i in range(b):
        out[i] = (torch.arange(n)[:, None] * 0.01 + cond[i].sum() + text[i].sum() * 1e-3 + steps)
    return out, None


def _batches():
    g = torch.Generator().manual_seed(0)
    bs = []
    for k in range(5):  # 5 batches over 2 ranks: uneven split (3 + 2)
        b = 1 + k % 3
        lens = torch.randint(3, 6, (b,), generator=g)
        dur = lens + torch.randint(2, 9, (b,), generator=g)
        bs.append(dict(cond=torch.randn(b, int(lens.max()), 100, generator=g), text=torch.randint(0, 50, (b, 7), generator=g), duration=dur,
                       lens=lens, steps=2 + k))
    return bs


def _worker(rank, world, rendezvous, q):
    # file rendezvous: no port to race for with other processes of the host
    dist.init_process_group("gloo", init_method="file://" + rendezvous, rank=rank, world_size=world)
    outs = sample_sharded(_fake_sample, _batches(), device="cpu")
    # a bare gather_utterances (no batch list to derive the ragged metadata from): ONE all_gather_object + the payload all_gather
    local = sample_sharded(_fake_sample, _batches(), device="cpu", gather=False)
    again = gather_utterances(local, [t.shape[0] for t in local], 100, "cpu")
    assert len(again) == len(outs) and all(torch.equal(a, b) for a, b in zip(again, outs))
    # plain numpy arrays through the queue: a torch tensor travels as a shared-memory handle that dies with this process if the parent is
    # slow to pick it up
    q.put((rank, [o.numpy().copy() for o in outs]))
    dist.barrier()
    dist.destroy_process_group()


def test_split_is_contiguous_and_complete():
    items = list(range(11))
    parts = [split_between_processes(items, r, 4) for r in range(4)]
    assert sum(parts, []) == items and [len(p) for p in parts] == [3, 3, 3, 2]
    assert split_between_processes(items, 0, 1) == items
    assert split_between_processes([1], 3, 4) == []


def test_two_rank_gather_matches_single_process():
    single = sample_sharded(_fake_sample, _batches(), device="cpu")
    rendezvous = os.path.join(tempfile.mkdtemp(prefix="f5_gloo_"), "store")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, rendezvous, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        assert len(results[rank]) == len(single)
        for a, b in zip(results[rank], single):
            a = torch.from_numpy(a)
            assert a.shape == b.shape and torch.equal(a, b)


def test_gather_metadata_follows_the_sampler_clamping():
    """the locally derived frame counts must equal len(out[i, lens_i:duration_i]) for what CFM.sample returns (cfm.py:127-135: at least
    max(n_text, lens) + 1 rows, at most max_duration; the requested duration may exceed the rows that exist, lens may exceed the duration)."""
    from eraxvif5tts_amd.eval.sharded import _generated_frames
    text = torch.full((3, 12), -1)
    text[0, :5] = 1
    text[1, :12] = 1
    text[2, :3] = 1
    kw = dict(text=text, lens=torch.tensor([4, 6, 9]), duration=torch.tensor([5000, 8, 7]), max_duration=40)
    eff = torch.maximum(torch.maximum((text != -1).sum(-1), kw["lens"]) + 1, kw["duration"]).clamp(max=40)
    out = torch.zeros(3, int(eff.max()), 100)
    want = [out[i, int(kw["lens"][i]): int(kw["duration"][i])].shape[0] for i in range(3)]
    assert _generated_frames(kw) == want == [36, 2, 0]
