"""The C-ABI libraries export every symbol include/ceed.h declares (no compute without a GPU)."""
import os
import re

import pytest

from ceedpetscsolid_amd import ceed as cd
from conftest import ROOT


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "ceed.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    funcs = re.findall(r"CEED_EXTERN\s+(?:const\s+)?[A-Za-z_]+\s*\**\s*(Ceed[A-Za-z0-9_]+)\s*\(", txt)
    data = re.findall(r"CEED_EXTERN\s+(?:const\s+)?[A-Za-z_ ]+?\*?\s*(?:const\s+)?(CEED_[A-Z_]+|CeedMemTypes)\s*(?:\[\d*\])?;", txt)
    return sorted(set(funcs)), sorted(set(data))


def test_header_parse_finds_the_api():
    funcs, data = header_symbols()
    assert "CeedOperatorApply" in funcs and "CeedXOperatorSetDirichletMaskMode" in funcs and len(funcs) >= 55
    assert "CEED_VECTOR_ACTIVE" in data and "CEED_STRIDES_BACKEND" in data and "CeedMemTypes" in data


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_library_exports_every_declared_symbol(which, oracle_lib):
    if which == "product" and not os.path.exists(cd.PRODUCT_LIB):
        pytest.fail("product library not built: run __graft_entry__.build()")
    lib = oracle_lib if which == "oracle" else cd.CeedLib(cd.PRODUCT_LIB)
    funcs, data = header_symbols()
    missing = [s for s in funcs + data if not hasattr(lib.lib, s)]
    assert not missing, missing


def test_product_refuses_cpu_resources_and_missing_gpu():
    """No fallback path: a CPU resource string is an error, and without a device CeedInit fails loudly."""
    import torch
    lib = cd.CeedLib(cd.PRODUCT_LIB)
    with pytest.raises(cd.CeedError):
        cd.Ceed(lib, "/cpu/self")
    if not torch.cuda.is_available():
        with pytest.raises(cd.CeedError):
            cd.Ceed(lib, "/gpu/hip/mi355x")


def test_oracle_refuses_device_memory(oracle):
    v = oracle.vector(3)
    with pytest.raises(cd.CeedError):
        v.set_device_pointer(0x1000)
