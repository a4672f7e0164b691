import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "semi-pd_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # The CPU oracle runs on torch's intra-op pool.  On a GPU box with hundreds of hardware threads the default (one thread
    # per hardware thread) collapses -- bench.py measured 3.2 s for ONE layer's decode step with 256 threads -- and the oracle
    # side of the engine tests was most of the GPU suite's 20 minutes.  At most 32 threads, like bench.py's cpu_baseline.
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with `-m gpu`; if they are collected on a box without a GPU they
    # fail loudly instead of being skipped (no silent fallback).
    pass


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def from_bits(arr: np.ndarray, dtype: torch.dtype) -> torch.Tensor:
    """Inverse of make_golden.bits(): uint16 arrays hold bf16 bit patterns."""
    if dtype == torch.bfloat16:
        return torch.from_numpy(arr.view(np.int16).copy()).view(torch.bfloat16)
    return torch.from_numpy(arr.copy())


DTYPES = {"torch.bfloat16": torch.bfloat16, "torch.float16": torch.float16, "torch.float32": torch.float32}


@pytest.fixture(scope="session")
def device():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")
