import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gfe-mamba_amd")          # source root mirroring the reference's import paths
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, SRC):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def golden(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}


def tt(a, dtype=None, device="cpu"):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(device)


def sub_sd(fx, prefix, dtype=torch.float32, device="cpu"):
    """{key: tensor} for fixture entries starting with `prefix` (prefix stripped)."""
    out = {}
    for k, v in fx.items():
        if k.startswith(prefix):
            t = torch.from_numpy(np.ascontiguousarray(v))
            if t.dtype.is_floating_point:
                t = t.to(dtype)
            out[k[len(prefix):]] = t.to(device)
    return out


def rel_err(a, b):
    """max |a-b| / max(|b|) -- the 'rel' of BASELINE.json's tolerance (1e-3 fp32 / 1e-2 bf16)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
