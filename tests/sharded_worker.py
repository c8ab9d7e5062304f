"""Rank program of tests/test_gpu_sharded.py (not a test): one process per rank, started by torch.distributed.run, every rank on the
GPU it is given (LOCAL_RANK modulo the visible devices -- two ranks share the one GPU of a test box), `gloo` for the gather; with
F5_TEST_BACKEND=nccl (world size 1 on a one-GPU box: RCCL refuses two ranks on one device) the SAME collectives run over RCCL: the payload
all_gather on device tensors, an all_gather_object, the MAX all_reduce and the barriers bench.py issues.
Runs the REAL CFM.sample of a tiny DiT through eraxvif5tts_amd.eval.sharded.sample_sharded and saves rank 0's gathered list."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ARCH = dict(dim=128, depth=2, heads=2, ff_mult=2, text_dim=64, conv_layers=1, pe_attn_head=1, text_mask_padding=False)
VOCAB = 40


def make_cfm(device="cuda"):
    from eraxvif5tts_amd.model import CFM, DiT
    from oracle import cpu_ref
    W = cpu_ref.random_dit_weights(ARCH, VOCAB, seed=41)
    m = DiT(**ARCH, text_num_embeds=VOCAB, mel_dim=100, precision="bf16")
    m.load_state_dict({k: v for k, v in W.items() if k in m.state_dict()}, strict=False)
    return CFM(transformer=m.to(device), mel_spec_kwargs={"mel_spec_type": "vocos"}).to(device)


def make_batches(device="cuda"):
    """5 prompt batches of unequal size over 2 ranks: uneven contiguous split (3 + 2), ragged lengths inside a batch."""
    g = torch.Generator().manual_seed(42)
    out = []
    for k in range(5):
        b = 1 + k % 3
        lens = torch.randint(20, 40, (b,), generator=g)
        dur = lens + torch.randint(30, 90, (b,), generator=g)
        out.append(dict(cond=torch.randn(b, int(lens.max()), 100, generator=g).to(device), text=torch.randint(0, VOCAB, (b, 12), generator=g).to(device),
                        duration=dur.to(device), lens=lens.to(device), steps=3, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=7 + k,
                        return_trajectory=False))
    return out


def main():
    from eraxvif5tts_amd.eval.sharded import sample_sharded
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    backend = os.environ.get("F5_TEST_BACKEND", "gloo")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
    else:
        dist.init_process_group("gloo")
    cfm = make_cfm()
    dev = "cpu" if dist.get_backend() == "gloo" else "cuda"
    outs = sample_sharded(cfm.sample, make_batches(), device=dev, force_collective=True)
    extra = {}
    if backend == "nccl":  # the other collectives of bench.py's multi-GPU leg, and the metadata exchange of a bare gather_utterances
        from eraxvif5tts_amd.eval.sharded import gather_utterances
        t = torch.tensor([1.5 + dist.get_rank()], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        names = [None] * dist.get_world_size()
        dist.all_gather_object(names, f"rank{dist.get_rank()}:{torch.cuda.get_device_name()}")
        again = gather_utterances([o.cuda() for o in outs], [o.shape[0] for o in outs], 100, "cuda", force_collective=True)  # metadata by all_gather_object
        extra = {"max": float(t.item()), "names": names, "regather_equal": all(torch.equal(a.cpu(), b.cpu()) for a, b in zip(again, outs)),
                 "backend": dist.get_backend()}
    if dist.get_rank() == 0:
        torch.save({"world": dist.get_world_size(), "outs": [o.cpu() for o in outs], **extra}, sys.argv[1])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
