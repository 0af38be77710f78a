"""The C-ABI library builds for gfx950, loads, and exports every symbol include/ff_kernels.h declares
(no compute calls: this runs without a GPU)."""
import ctypes
import os

import pytest


def test_library_builds_and_exports_header_symbols():
    from isr2_amd.build import build
    from isr2_amd import lib as L
    path = build()
    assert os.path.exists(path)
    protos = L.parse_header()
    assert len(protos) >= 25
    handle = ctypes.CDLL(path)
    for name in protos:
        assert hasattr(handle, name), f"{name} declared in include/ff_kernels.h but not exported"
    lib = L.load()
    assert lib.ff_abi_version() == 6
    assert lib.ff_last_error() is not None


def test_argument_validation_needs_no_gpu():
    from isr2_amd import lib as L
    lib = L.load()
    # null pointers are rejected before any launch
    rc = lib.ff_layernorm(None, 180, None, 180, 10, 180, None, None, 1e-5, None)
    assert rc != 0 and b"ff_layernorm" in lib.ff_last_error()
    rc = lib.ff_window_attn(1, 540, 0, 180, 360, 1, 180, 0, 1, 1, 32, 32, 32, 32, 16, 15, 16, 15, 0, 0, 0, 6, 30, 0.18, None)
    assert rc != 0 and b"256 tokens" in lib.ff_last_error()


def test_product_path_refuses_cpu():
    import torch
    from isr2_amd.lib import FFError
    from isr2_amd.model import FreqFusionHIP
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(FFError):
        FreqFusionHIP({}, "cpu")
    from isr2_amd import ops
    with pytest.raises(FFError):
        ops.layernorm(torch.zeros(4, 180), torch.ones(180), torch.zeros(180))


def test_product_package_never_imports_the_oracle():
    import re
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "image-super-resolution-2_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
    plug = os.path.join(os.path.dirname(pkg), "models", "team29_FreqFusion", "io.py")
    assert not re.search(r"^\s*(from|import)\s+oracle", open(plug).read(), flags=re.M)
