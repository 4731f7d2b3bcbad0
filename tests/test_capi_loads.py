"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol include/tmvb.h declares."""
import ctypes
import os

import pytest


def test_library_loads_and_exports_header_symbols(tmvb):
    tmvb.build()
    assert os.path.exists(tmvb.LIB_PATH)
    L = ctypes.CDLL(tmvb.LIB_PATH)
    syms = tmvb.exported_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), f"libtmvb_hip.so does not export {s}"
    assert tmvb.lib().tmvb_abi_version() == 2


def test_no_cpu_fallback_without_device(tmvb):
    """The product path must fail loudly when there is no gfx950 device (no silent CPU path)."""
    if tmvb.lib().tmvb_device_count() > 0:
        pytest.skip("a GPU is visible")
    pc = tmvb.PackedCorpus([0, 2], [0, 1], [1, 1], 3)
    with pytest.raises(tmvb.EngineError):
        tmvb.gpuLDA(pc, 2)


def test_product_package_does_not_import_the_oracle():
    """oracle/ is test infrastructure: nothing under the product package may import, link or load it."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "topicmodelsvb.jl_amd")
    banned = ("import oracle", "from oracle", "libtmvb_oracle", "tmvb_oracle.h", "oracle_np", "orc_")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".jl")):
                src = open(os.path.join(dp, f), errors="replace").read()
                for b in banned:
                    assert b not in src, f"{f} references the oracle ({b})"
