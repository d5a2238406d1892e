"""CPU tier: the C-ABI libraries load and export every symbol include/lvi_hotpath.h declares;
the product path fails loudly without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest


def _declared_symbols():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "include", "lvi_hotpath.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lvi_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree(pkg):
    assert _declared_symbols() == sorted(pkg._abi.SIGNATURES.keys())


def test_hip_library_exports_every_symbol(pkg):
    assert os.path.exists(pkg.HIP_LIB_PATH), "liblvi_hip.so not built: run __graft_entry__.build()"
    dll = ctypes.CDLL(pkg.HIP_LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(dll, name), f"{name} missing from liblvi_hip.so"
    lib = pkg.load_hip()
    assert lib.backend == "hip-gfx950"
    assert lib.dll.lvi_abi_version() == 6


def test_oracle_exports_every_symbol(pkg, oracle):
    for name in _declared_symbols():
        assert hasattr(oracle.dll, name)
    assert oracle.backend == "cpu-oracle"


def test_struct_sizes(pkg):
    A = pkg._abi
    assert ctypes.sizeof(A.LidarParams) == 108
    assert ctypes.sizeof(A.IcpResult) == 6 * 4 + 64 * 4 + 24
    assert ctypes.sizeof(A.KernelStat) == 48 + 8 + 8 + 8
    assert A.PT_DTYPE.itemsize == 16 and A.LIVOX_DTYPE.itemsize == 20


def test_product_fails_loudly_without_gpu(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = pkg.load_hip()
    with pytest.raises(pkg.LviError) as e:
        pkg.LidarHotpath(lib)
    assert e.value.code == pkg._abi.LVI_ERR_NO_DEVICE


def test_product_package_never_references_the_oracle(pkg):
    """the oracle is test infrastructure: nothing under the product package may import or link it"""
    bad = []
    for dirpath, _, files in os.walk(pkg.PKG_DIR):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"liblvi_oracle|from oracle|import oracle|oracle/", txt):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
