"""Utterance-sharded batch inference over the GPUs of one node (one process per GPU, RCCL over xGMI).

Model: the reference's only multi-GPU inference path, ``f5_tts/eval/eval_infer_batch.py:26-27,160-196``: the list of
pre-bucketed prompt batches is split contiguously over the processes (``accelerator.split_between_processes``), every rank
runs ``CFM.sample`` on its own batches with a full weight replica, and the only synchronisation is a barrier before and
after.  There is no collective inside the ODE loop.  New here (asked for by the north star): the finished mels are
gathered to every rank with ONE data all_gather per call (the ragged metadata is derived locally from the batch list) (``torch.distributed``; backend ``nccl`` is RCCL on ROCm, ``gloo`` on
CPU for the tests) instead of being written to per-rank files.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist


def split_between_processes(items: Sequence, rank: int, world_size: int):
    """Contiguous split with the remainder spread over the first ranks (accelerate's PartialState.split_between_processes)."""
    n = len(items)
    base, extra = divmod(n, world_size)
    start = rank * base + min(rank, extra)
    end = start + base + (1 if rank < extra else 0)
    return items[start:end]


def gather_utterances(local: List[torch.Tensor], frame_counts: List[int], mel_dim: int, device,
                      all_frame_counts: Optional[List[List[int]]] = None, force_collective: bool = False) -> List[torch.Tensor]:
    """all_gather of a ragged list of [n_i, mel] tensors.  ONE data collective (the padded payload); the metadata -- how many utterances
    each rank holds and how many frames each has -- is either handed in (``all_frame_counts[r]`` = frame counts of rank r: sample_sharded
    derives it from the batch list every rank already has, no collective) or exchanged with ONE ``all_gather_object``.
    Returns the utterances of every rank in global (rank-major = original) order, on every rank."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1 and not (force_collective and dist.is_initialized()):  # (force_collective: rehearse the RCCL calls on a one-GPU box)
        return list(local)
    if all_frame_counts is None:
        all_frame_counts = [None] * world
        dist.all_gather_object(all_frame_counts, [int(c) for c in frame_counts])
    rank = dist.get_rank()
    # a rank whose own counts disagree with the agreed metadata must not skip the collective (the other ranks would wait in it for ever):
    # it sends a payload of the agreed shape and raises afterwards
    agreed = [int(c) for c in all_frame_counts[rank]] == [int(c) for c in frame_counts]
    max_items = max(len(c) for c in all_frame_counts)
    max_frames = max((max(c) for c in all_frame_counts if len(c)), default=0)
    payload = torch.zeros(max_items, max_frames, mel_dim, device=device, dtype=torch.float32)
    for i, t in enumerate(local[:max_items]):
        n = min(t.shape[0], max_frames)
        payload[i, :n] = t[:n]
    gathered = [torch.zeros_like(payload) for _ in range(world)]
    dist.all_gather(gathered, payload)  # the single data collective of the whole path (RCCL over xGMI under backend "nccl")
    if not agreed:
        raise RuntimeError(f"rank {rank}: frame counts {list(map(int, frame_counts))} disagree with the metadata {list(map(int, all_frame_counts[rank]))}")
    out = []
    for r in range(world):
        for i, n in enumerate(all_frame_counts[r]):
            out.append(gathered[r][i, : int(n)])
    return out


def _generated_frames(kw) -> List[int]:
    """frames of ``out[i, lens_i:duration_i]`` (eval_infer_batch.py:185) for each utterance of one batch, with the clamping CFM.sample applies
    (cfm.py:127-135): out has max_i(min(max(max(n_text_i, lens_i) + 1, duration_i), max_duration)) rows, and a slice never has fewer than 0."""
    lens, dur, text = kw["lens"], kw["duration"], kw.get("text")
    max_duration = int(kw.get("max_duration", 4096))
    b = len(lens)
    if torch.is_tensor(text):
        ntext = [int((text[i] != -1).sum()) for i in range(b)]
    elif text is not None:
        ntext = [len(t) for t in text]
    else:
        ntext = [0] * b
    rows = max(min(max(max(ntext[i], int(lens[i])) + 1, int(dur[i])), max_duration) for i in range(b))
    return [max(0, min(int(dur[i]), rows) - min(int(lens[i]), rows)) for i in range(b)]


@torch.no_grad()
def sample_sharded(sample_fn: Callable, batches: Sequence[dict], mel_dim: int = 100, device="cuda", gather: bool = True,
                   force_collective: bool = False):
    """batches: list of dicts with the keyword arguments of ``CFM.sample`` for one padded batch (cond [b, nc, mel], text,
    duration [b], lens [b], steps, ...).  Each rank runs its contiguous share; returns the generated part
    ``out[i, lens_i:duration_i]`` of every utterance of every batch, in the original order (on every rank if gather).
    Every rank holds the whole batch list (as every rank of the reference builds the whole prompt list before splitting it,
    eval_infer_batch.py:160-163), so the gather's metadata needs no collective: one all_gather of the payload is the only exchange."""
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
    all_counts = [[n for kw in split_between_processes(list(batches), r, world) for n in _generated_frames(kw)] for r in range(world)]
    return gather_utterances(local, counts, mel_dim, device, all_frame_counts=all_counts, force_collective=force_collective)
