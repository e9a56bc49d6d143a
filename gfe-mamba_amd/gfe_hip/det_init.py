"""Deterministic, platform-independent weight initialiser (numpy Philox keyed by parameter name).

There is no network for checkpoints, so benchmarks, parity tests and the golden-fixture generator all
regenerate the same weights from (seed, state-dict key, shape).  Values follow the magnitudes of the
reference's own initialisers (cross_atten/mamba.py:141-165 for dt/A_log/D; torch defaults elsewhere).
"""
import math
import zlib

import numpy as np
import torch


def _rng(seed, key):
    return np.random.Generator(np.random.Philox(key=[zlib.crc32(key.encode()), seed & 0xFFFFFFFF]))


def det_tensor(key, shape, seed=0):
    """Returns a float32 CPU tensor for state-dict entry `key` of shape `shape`."""
    g = _rng(seed, key)
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if shape else 1
    leaf = key.split(".")[-1]

    def normal(scale=1.0):
        return (g.standard_normal(n, dtype=np.float32) * scale).reshape(shape)

    def uniform(bound):
        return ((g.random(n, dtype=np.float32) * 2 - 1) * bound).reshape(shape)

    if leaf == "A_log":                                     # mamba.py:160-161 S4D-real, lightly perturbed
        N = shape[-1]
        base = np.log(np.arange(1, N + 1, dtype=np.float32))
        arr = base + normal(0.05)
    elif leaf == "D":                                       # mamba.py:164
        arr = 1.0 + normal(0.1)
    elif key.endswith("dt_proj.bias"):                      # mamba.py:150-155 inverse-softplus of dt in [1e-3, 1e-1]
        dt = np.exp(g.random(n, dtype=np.float32) * (math.log(0.1) - math.log(0.001)) + math.log(0.001)).clip(min=1e-4)
        arr = (dt + np.log(-np.expm1(-dt))).astype(np.float32).reshape(shape)
    elif key.endswith("dt_proj.weight"):                    # mamba.py:141-145
        arr = uniform(shape[1] ** -0.5)
    elif leaf in ("cls_token", "pos_embedding", "weights", "biases") or key.endswith("categorical_embeds.weight"):
        arr = normal(1.0)                                   # torch.randn parameters / nn.Embedding
    elif len(shape) >= 2:                                   # Linear / Conv weights: U(+-1/sqrt(fan_in))
        arr = uniform(1.0 / math.sqrt(float(np.prod(shape[1:]))))
    elif leaf == "weight":                                  # norm scales
        arr = 1.0 + normal(0.1)
    else:                                                   # biases
        arr = normal(0.05)
    return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))


def det_state_dict(template, seed=0, prefix=""):
    """New state dict with the keys/shapes of `template` (a state_dict or {key: shape}); integer buffers are copied."""
    out = {}
    for k, v in template.items():
        if isinstance(v, torch.Tensor):
            if not v.dtype.is_floating_point:
                out[k] = v.clone()
                continue
            shape = v.shape
        else:
            shape = v
        out[k] = det_tensor(prefix + k, shape, seed)
    return out


def det_inputs(batch, vol=(96, 96, 96), cards=(11, 2, 2, 4, 4, 3, 3), n_cont=25, seed=0):
    """Synthetic batch of SURVEY.md 8-d config 1: x ~ clip(N(0,1), -1, 1), x_cat ~ U{0..card-1}, x_num ~ N(0,1), y ~ Bern(.5)."""
    g = _rng(seed, "inputs")
    x = np.clip(g.standard_normal((batch, 1) + tuple(vol), dtype=np.float32), -1, 1)
    x_cat = np.stack([g.integers(0, c, size=batch) for c in cards], axis=1).astype(np.int64)
    x_num = g.standard_normal((batch, n_cont), dtype=np.float32)
    y = (g.random(batch) < 0.5).astype(np.int64)
    return torch.from_numpy(x), torch.from_numpy(x_cat), torch.from_numpy(x_num), torch.from_numpy(y)
