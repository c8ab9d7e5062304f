import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _memoize_weight_generators():
    """The seeded weight generators of the oracle take ~5 s for a true-size network and are called with the same (arch, vocab, seed) by many
    parametrizations: keep the last few results for the session (suite wall time).  Callers get a shallow copy; no test
    mutates the tensors in place."""
    import functools
    import json

    from oracle import cpu_ref

    def memo(fn):
        cache = {}

        @functools.wraps(fn)
        def wrapped(cfg, vocab_size, seed=0, **kw):
            key = (json.dumps(cfg, sort_keys=True, default=str), int(vocab_size), int(seed), tuple(sorted(kw.items())))
            if key not in cache:
                if len(cache) >= 4:
                    cache.pop(next(iter(cache)))
                cache[key] = fn(cfg, vocab_size, seed=seed, **kw)
            return dict(cache[key])
        return wrapped
    for name in ("random_dit_weights", "random_unett_weights", "random_mmdit_weights"):
        if not hasattr(getattr(cpu_ref, name), "__wrapped__"):
            setattr(cpu_ref, name, memo(getattr(cpu_ref, name)))


def pytest_configure(config):
    _memoize_weight_generators()
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def golden_arch(z):
    def conv(v):
        if v == "None":
            return None
        if v in ("True", "False"):
            return v == "True"
        return int(v)
    return {str(k): conv(str(v)) for k, v in zip(z["arch_keys"], z["arch_vals"])}


def golden_weights(z, prefix="W."):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in z.items() if k.startswith(prefix)}


def rel_l2(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="session")
def has_gpu():
    return torch.cuda.is_available()
