"""ORACLE -- test infrastructure.  ctypes access to oracle/libgfe_oracle.so (scan_ref.c)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(os.path.join(_HERE, "libgfe_oracle.so"))
        _lib.gfe_oracle_selective_scan.restype = ctypes.c_int
        _lib.gfe_oracle_selective_scan.argtypes = [ctypes.c_void_p] * 9 + [ctypes.c_int64] * 4 + [ctypes.c_int]
    return _lib


def selective_scan(u, delta, A, Bm, Cm, D=None, z=None, bias=None, softplus=False):
    """numpy float32 arrays, token-major; returns y (B, L, ED)."""
    f = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
    u, delta, A, Bm, Cm, D, z, bias = map(f, (u, delta, A, Bm, Cm, D, z, bias))
    y = np.empty_like(u)
    p = lambda a: None if a is None else a.ctypes.data
    B, L, ED = u.shape
    rc = lib().gfe_oracle_selective_scan(p(u), p(delta), p(A), p(Bm), p(Cm), p(D), p(z), p(bias), p(y), B, L, ED, A.shape[1], int(softplus))
    assert rc == 0
    return y
