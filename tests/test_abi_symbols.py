"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/dfusion_hip.h declares.  No compute entry point is called (no GPU here)."""
import ctypes
import os

import pytest

from dynamicfusion_body_amd import _lib, build


@pytest.fixture(scope="module")
def lib_path():
    return build.build_library()


def test_library_builds_and_loads(lib_path):
    assert os.path.exists(lib_path)
    lib = _lib.load()
    assert lib.dfh_version() == _lib.ABI_VERSION
    assert lib.dfh_last_error() == b""


def test_every_declared_symbol_is_exported(lib_path):
    names = _lib.declared_symbols()
    assert "dfh_integrate_depth" in names and "dfh_version" in names
    lib = ctypes.CDLL(lib_path)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, "declared in include/dfusion_hip.h but not exported: %s" % missing


def test_every_declared_symbol_has_a_binding(lib_path):
    assert sorted(_lib._SIGNATURES) == _lib.declared_symbols()


def test_library_has_gfx950_code_object(lib_path):
    blob = open(lib_path, "rb").read()
    assert b"gfx950" in blob
    assert b"integrate_depth_kernel" in blob


def test_no_oracle_in_product():
    """The product path must never import the oracle (it is the checker)."""
    pkg = os.path.dirname(os.path.abspath(_lib.__file__))
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
