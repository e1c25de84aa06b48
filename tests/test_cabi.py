"""CPU-side checks of the C-ABI library: it loads, and exports every symbol include/presight_hip.h declares.
No compute call is made (no GPU in the build container)."""
import ctypes
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge

    ge.build()
    from presight_amd import _lib

    return _lib


def test_header_symbols_are_exported(built):
    protos = built.parse_header()
    assert len(protos) >= 20
    h = ctypes.CDLL(built.LIB_PATH)
    missing = [n for n in protos if not hasattr(h, n)]
    assert not missing, missing
    out = subprocess.run(["nm", "-D", "--defined-only", built.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("ps_")}
    undeclared = exported - set(protos) - {"ps_set_error"}
    assert not undeclared, f"exported but not declared in the header: {undeclared}"


def test_abi_version_and_shape_query(built):
    h = built.lib()
    assert h.ps_abi_version() == 1
    # every MLP shape on the PreSight path + cfg-1 variants has a kernel
    for dims in ([32, 64, 80], [40, 64, 80], [64, 64, 64, 64], [47, 64, 64, 3], [8, 64, 1], [32, 32, 32, 3], [16, 32, 32, 64],
                 [4, 32, 80], [47, 32, 32, 3], [2, 32, 1]):
        assert h.ps_mlp_shape_supported(dims[0], dims[1], dims[-1], len(dims) - 1) == 1, dims
    assert h.ps_mlp_shape_supported(7, 24, 3, 2) == 0


def test_product_never_imports_oracle():
    """The product path must not route through the oracle (or any CPU fallback)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "presight_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src, f"{f} mentions the oracle"


def test_mlp_spec_matches_library(built):
    from presight_amd.ops import MlpSpec

    h = built.lib()
    for dims in ([32, 64, 80], [64, 64, 64, 64], [47, 64, 64, 3], [8, 64, 1], [16, 32, 32, 64]):
        pf, gf, npart = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        assert h.ps_mlp_sizes(dims[0], dims[1], dims[-1], len(dims) - 1, 1000, ctypes.byref(pf), ctypes.byref(gf), ctypes.byref(npart)) == 0
        s = MlpSpec(dims)
        assert (s.packed, s.g_total) == (pf.value, gf.value), dims
