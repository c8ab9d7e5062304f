"""The multi-GPU path on real kernels: two ranks (one process each, started by torch.distributed.run exactly as the driver starts
bench.py) share the test box's one GPU, `gloo` carries the gather.  The gathered mels of the two-rank run must equal the
single-process run bit for bit (utterances are independent; the split is over whole prompt batches, eval_infer_batch.py:163),
and `bench.py --gpus 2` must really run two ranks and say so in its JSON line."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("WORLD_SIZE", None)
    return env


def test_two_real_ranks_equal_one_process(tmp_path):
    from eraxvif5tts_amd import _lib
    from eraxvif5tts_amd.eval.sharded import sample_sharded
    _lib.require_gpu()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import sharded_worker as SW
    out_file = str(tmp_path / "gathered.pt")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "tests", "sharded_worker.py"), out_file]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = torch.load(out_file, weights_only=True)
    assert got["world"] == 2
    single = sample_sharded(SW.make_cfm().sample, SW.make_batches(), device="cuda")
    assert len(got["outs"]) == len(single) == 9
    for a, b in zip(got["outs"], single):
        assert a.shape == b.shape and torch.isfinite(a).all() and torch.equal(a, b.cpu())


def test_rccl_backend_runs_the_collectives_at_world_size_one(tmp_path):
    """RCCL itself (torch.distributed backend "nccl"), on the one GPU a test box has: process-group init with a device id, the payload
    all_gather of the finished mels on DEVICE tensors, all_gather_object, the MAX all_reduce and the barriers -- every call the 8-GPU run
    makes -- at world size 1 (RCCL refuses two ranks on one device).  The gathered list must equal the plain single-process result."""
    from eraxvif5tts_amd import _lib
    from eraxvif5tts_amd.eval.sharded import sample_sharded
    _lib.require_gpu()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import sharded_worker as SW
    out_file = str(tmp_path / "rccl.pt")
    env = _env()
    env["F5_TEST_BACKEND"] = "nccl"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "tests", "sharded_worker.py"), out_file]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = torch.load(out_file, weights_only=True)
    assert got["world"] == 1 and got["backend"] == "nccl" and got["max"] == 1.5 and got["regather_equal"] and len(got["names"]) == 1
    single = sample_sharded(SW.make_cfm().sample, SW.make_batches(), device="cuda")
    assert len(got["outs"]) == len(single) == 9
    for a, b in zip(got["outs"], single):
        assert a.shape == b.shape and torch.equal(a, b.cpu())


def test_bench_gpus_flag_launches_the_ranks():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts two ranks (before touching the GPU itself) and relays
    rank 0's line; strong scaling splits the SAME utterances.  (gloo stands in for RCCL: both ranks share this box's one GPU.)"""
    env = _env()
    env["F5_BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scaling", "strong", "--batch", "4", "--seq-len", "256", "--nfe", "2",
           "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(line) == 1
    j = json.loads(line[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["distributed"]["world_size"] == 2 and j["distributed"]["backend"] == "gloo"
    assert len(j["distributed"]["devices"]) == 2 and j["config"]["global_batch"] == 4 and j["config"]["per_gpu_batch"] == 2
    assert j["value"] > 0 and any(k["kernel"] == "attention" for k in j["roofline"]["kernels"])
    # per-rank record: own wall time, HIP-event time of sample() and of the gather, the collective alone (separates batch efficiency, skew, gather)
    ranks = j["distributed"]["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and all(r["utterances"] == 2 and r["sample_ms"] > 0 and r["gather_only_ms"] >= 0 and
                                                       r["elapsed_s"] * 1e3 >= r["sample_ms"] for r in ranks)
    assert j["distributed"]["gather_payload_bytes"] == 2 * 256 * 100 * 4
    # a WORLD_SIZE that contradicts --gpus is an error, not a silent single-GPU run
    env["WORLD_SIZE"] = "4"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "contradicts" in (r.stderr + r.stdout)
