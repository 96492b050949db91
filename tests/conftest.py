import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when collected on a machine without a GPU and without -m gpu
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def from_bits(arr, is_bf16):
    """uint16 bit patterns -> torch half tensor"""
    t = torch.from_numpy(arr.astype(np.int16) if arr.dtype == np.uint16 else arr)
    return t.view(torch.bfloat16 if is_bf16 else torch.float16)


def to_bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def ulp_diff(a, b):
    """|a-b| in units of 16-bit ulps via the monotone integer order of the bit patterns"""
    def key(t):
        x = t.detach().cpu().contiguous().view(torch.int16).to(torch.int32) & 0xFFFF
        return torch.where(x >= 0x8000, 0x8000 - (x - 0x8000) - 1, x + 0x8000)
    return (key(a) - key(b)).abs()
