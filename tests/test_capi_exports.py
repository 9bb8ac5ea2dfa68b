"""CPU-side checks of the drop-in boundary: libchgpu.so loads and exports every symbol include/chgpu.h declares
(no compute calls — there is no GPU here), and the product package never touches oracle/."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    so = os.path.join(REPO, "clickhouse_amd", "libchgpu.so")
    if not os.path.exists(so):
        g.build()
    return so


def test_library_exports_every_declared_symbol(built):
    from clickhouse_amd import _capi
    L = ctypes.CDLL(built)
    declared = _capi.declared_symbols()
    assert len(declared) >= 40
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, f"libchgpu.so does not export: {missing}"
    # and the ctypes signature table covers exactly the header
    assert sorted(_capi.SIGNATURES) == declared


def test_abi_version_and_error_channel(built):
    from clickhouse_amd import _capi
    L = _capi.lib()
    assert L.chgpu_abi_version() == 1
    # NULL arguments are rejected with BAD_ARGUMENTS and a message, never a crash
    rc = L.chgpu_ctx_synchronize(None)
    assert rc == _capi.ERR_BAD_ARGUMENTS
    assert b"NULL" in L.chgpu_last_error()


def test_no_gpu_means_loud_failure_not_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import clickhouse_amd as ch
    with pytest.raises(ch.ChgpuError) as e:
        ch.Context(0)
    assert e.value.code == ch._capi.ERR_DEVICE


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(REPO, "clickhouse_amd")
    offenders = []
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", "Makefile")):
                text = open(os.path.join(root, f), errors="replace").read()
                if re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M) or "ch_oracle" in text or "libchoracle" in text:
                    offenders.append(os.path.join(root, f))
    assert not offenders, offenders
