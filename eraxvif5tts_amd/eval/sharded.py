"""Utterance-sharded batch inference over the GPUs of one node (one process per GPU, RCCL over xGMI).

Model: the reference's only multi-GPU inference path, ``f5_tts/eval/eval_infer_batch.py:26-27,160-196``: the list of
pre-bucketed prompt batches is split contiguously over the processes (``accelerator.split_between_processes``), every rank
runs ``CFM.sample`` on its own batches with a full weight replica, and the only synchronisation is a barrier before and
after.  There is no collective inside the ODE loop.  New here (asked for by the north star): the finished mels are
gathered to every rank with ONE all_gather per call (``torch.distributed``; backend ``nccl`` is RCCL on ROCm, ``gloo`` on
CPU for the tests) instead of being written to per-rank files.
"""
from __future__ import annotations

from typing import Callable, List, Sequence

import torch
import torch.distributed as dist


def split_between_processes(items: Sequence, rank: int, world_size: int):
    """Contiguous split with the remainder spread over the first ranks (accelerate's PartialState.split_between_processes)."""
    n = len(items)
    base, extra = divmod(n, world_size)
    start = rank * base + min(rank, extra)
    end = start + base + (1 if rank < extra else 0)
    return items[start:end]


def gather_utterances(local: List[torch.Tensor], frame_counts: List[int], mel_dim: int, device) -> List[torch.Tensor]:
    """all_gather of a ragged list of [n_i, mel] tensors: one collective for the lengths, one for the padded payload.
    Returns the utterances of every rank in global (rank-major = original) order, on every rank."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return list(local)
    counts = torch.tensor([len(local)], device=device, dtype=torch.long)
    all_counts = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts)
    max_items = int(max(int(c) for c in all_counts))
    lens = torch.zeros(max_items, device=device, dtype=torch.long)
    if local:
        lens[: len(local)] = torch.tensor(frame_counts, device=device, dtype=torch.long)
    all_lens = [torch.zeros_like(lens) for _ in range(world)]
    dist.all_gather(all_lens, lens)
    max_frames = int(max(int(l.max()) for l in all_lens)) if max_items else 0
    payload = torch.zeros(max_items, max_frames, mel_dim, device=device, dtype=torch.float32)
    for i, t in enumerate(local):
        payload[i, : t.shape[0]] = t
    gathered = [torch.zeros_like(payload) for _ in range(world)]
    dist.all_gather(gathered, payload)  # the single data collective of the whole path
    out = []
    for r in range(world):
        for i in range(int(all_counts[r])):
            out.append(gathered[r][i, : int(all_lens[r][i])])
    return out


@torch.no_grad()
def sample_sharded(sample_fn: Callable, batches: Sequence[dict], mel_dim: int = 100, device="cuda", gather: bool = True):
    """batches: list of dicts with the keyword arguments of ``CFM.sample`` for one padded batch (cond [b, nc, mel], text,
    duration [b], lens [b], steps, ...).  Each rank runs its contiguous share; returns the generated part
    ``out[i, lens_i:duration_i]`` of every utterance of every batch, in the original order (on every rank if gather)."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    mine = split_between_processes(list(batches), rank, world)
    local, counts = [], []
    for kw in mine:
        out, _ = sample_fn(**kw)
        lens, dur = kw["lens"], kw["duration"]
        for i in range(out.shape[0]):
            gen = out[i, int(lens[i]): int(dur[i]), :].to(torch.float32)  # eval_infer_batch.py:185
            local.append(gen)
            counts.append(gen.shape[0])
    if not gather:
        return local
    return gather_utterances(local, counts, mel_dim, device)
