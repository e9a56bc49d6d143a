"""Drop-in for the reference's cross_atten/pscan.py (npo2 :13, pad_npo2 :20, PScan :35, pscan :226) -- MI355X build.

The reference evaluates H[t] = A[t]*H[t-1] + X[t] with 2*log2(L) in-place Blelloch sweeps over power-of-two padded
copies; here `pscan` is one streaming HIP kernel pair (gfe-mamba_amd/csrc/pscan.hip) with its own autograd Function.
npo2 / pad_npo2 are kept for API compatibility (callers of the reference import them); the kernels need no padding.
"""
import math

import torch.nn.functional as F

from gfe_hip.scan_ops import _PScan as PScan  # noqa: F401  (autograd.Function, same role as the reference's PScan)
from gfe_hip.scan_ops import pscan            # noqa: F401


def npo2(len):
    """Next power of two >= len (pscan.py:13-18)."""
    return 2 ** math.ceil(math.log2(len))


def pad_npo2(X):
    """(B, L, D, N) -> (B, npo2(L), D, N), zero padded (pscan.py:20-33)."""
    len_npo2 = npo2(X.size(1))
    return F.pad(X, (0, 0, 0, 0, 0, len_npo2 - X.size(1)), "constant", 0)
