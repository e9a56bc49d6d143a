"""ctypes binding of libgfe_hip.so (C-ABI declared in include/gfe_hip.h).

This is the only place the shared library is loaded.  There is no CPU fallback: if the library is
missing or an entry point reports an error the call raises.  Signatures are parsed from the header
so `include/gfe_hip.h` stays the single source of truth for the boundary.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
LIB_PATH = os.environ.get("GFE_HIP_LIB", os.path.join(_HERE, "libgfe_hip.so"))   # override: kernel A/B experiments only
HEADER_PATH = os.path.join(_ROOT, "include", "gfe_hip.h")

GFE_F32, GFE_BF16 = 0, 1
_ERR = {-1: "GFE_ERR_NULL", -2: "GFE_ERR_SHAPE", -3: "GFE_ERR_DTYPE", -4: "GFE_ERR_HIP"}


class GfeError(RuntimeError):
    pass


def parse_header(path=HEADER_PATH):
    """Returns {name: (restype, [(ctype_str, argname), ...])} for every `gfe_*` prototype."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    protos = {}
    for m in re.finditer(r"\b(int|const char\*)\s+(gfe_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        alist = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?)(\w+)$", a)
                alist.append((mm.group(1).strip().replace(" *", "*"), mm.group(2)))
        protos[name] = (ret, alist)
    return protos


_CT = {
    "int": ctypes.c_int, "int64_t": ctypes.c_int64, "float": ctypes.c_float, "double": ctypes.c_double,
    "int*": ctypes.POINTER(ctypes.c_int),
}


def _ctype(t):
    if t in _CT:
        return _CT[t]
    if t.endswith("*"):
        return ctypes.c_void_p
    raise GfeError(f"unmapped C type in gfe_hip.h: {t!r}")


_lib = None
_protos = None


def lib():
    """Loads the library once (after torch, so both share torch's HIP runtime)."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  -- torch's bundled libamdhip64 (same SONAME) must be resident first
    if not os.path.exists(LIB_PATH):
        raise GfeError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C gfe-mamba_amd/csrc`). There is no CPU fallback for the product path.")
    L = ctypes.CDLL(LIB_PATH)
    _protos = parse_header()
    for name, (ret, args) in _protos.items():
        fn = getattr(L, name)  # AttributeError if the header declares a symbol the library lacks
        fn.restype = ctypes.c_char_p if ret.startswith("const char") else ctypes.c_int
        fn.argtypes = [_ctype(t) for t, _ in args]
    if L.gfe_abi_version() != _abi_from_header():
        raise GfeError("libgfe_hip.so ABI version differs from include/gfe_hip.h: rebuild")
    _lib = L
    return L


def _abi_from_header():
    m = re.search(r"#define\s+GFE_ABI_VERSION\s+(\d+)", open(HEADER_PATH).read())
    return int(m.group(1))


def protos():
    lib()
    return _protos


def ptr(t):
    """Device (or host) pointer of a tensor, None -> NULL."""
    return None if t is None else t.data_ptr()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def dtype_code(dt):
    import torch
    if dt == torch.float32:
        return GFE_F32
    if dt == torch.bfloat16:
        return GFE_BF16
    raise GfeError(f"unsupported activation dtype {dt}")


def call(name, *args):
    """Calls an entry point; raises on a non-zero status."""
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise GfeError(f"{name} failed: {_ERR.get(rc, rc)}")
    return rc
