"""ctypes binding of libff_hip.so.  Prototypes are parsed from include/ff_kernels.h, so the header is
the single source of truth for the ABI (tests/test_abi.py checks every declared symbol is exported)."""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.normpath(os.path.join(HERE, "..", "include", "ff_kernels.h"))
LIB_PATH = os.path.join(HERE, "libff_hip.so")

_CT = {
    "const float*": ctypes.c_void_p, "float*": ctypes.c_void_p, "void*": ctypes.c_void_p, "const void*": ctypes.c_void_p,
    "const unsigned char*": ctypes.c_void_p, "unsigned char*": ctypes.c_void_p, "double*": ctypes.c_void_p, "const double*": ctypes.c_void_p,
    "double": ctypes.c_double, "void**": ctypes.c_void_p, "int*": ctypes.c_void_p,
    "int": ctypes.c_int, "long long": ctypes.c_longlong, "unsigned long long": ctypes.c_ulonglong, "float": ctypes.c_float, "const char*": ctypes.c_char_p,
}


def parse_header(path: str = HEADER) -> Dict[str, Tuple[str, List[str]]]:
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"^\s*#.*$", "", txt, flags=re.M)
    protos = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(ff_\w+)\s*\(([^)]*)\)\s*;", txt):
        ret = " ".join(m.group(1).split())
        args = []
        a = m.group(3).strip()
        if a and a != "void":
            for part in a.split(","):
                part = " ".join(part.split())
                ty = re.sub(r"\s*\w+$", "", part) if not part.endswith("*") else part
                ty = ty.replace(" *", "*")
                args.append(ty)
        protos[m.group(2)] = (ret.replace(" *", "*"), args)
    return protos


class FFError(RuntimeError):
    pass


_lib = None


def load(path: str = LIB_PATH) -> ctypes.CDLL:
    """Load the kernel library; fails loudly when it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise FFError(f"{path} is missing: build it with `python -m isr2_amd.build` (hipcc, gfx950). "
                      "The product path has no CPU/PyTorch fallback.")
    lib = ctypes.CDLL(path)
    for name, (ret, args) in parse_header().items():
        fn = getattr(lib, name)
        fn.restype = _CT[ret]
        fn.argtypes = [_CT[a] for a in args]
    _lib = lib
    return lib


def check(status: int):
    if status != 0:
        raise FFError(load().ff_last_error().decode())
