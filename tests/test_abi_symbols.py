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
    """The product path must never import the oracle (it is the checker), must not bring a CPU nearest-neighbour structure
    of its own (scipy's KD-trees: the reference's host-side search, restated only in oracle/graph_np.py), and the library's
    call paths must not read the environment (one getenv: DFH_OPTIONS, at the first call)."""
    import re
    pkg = os.path.dirname(os.path.abspath(_lib.__file__))
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert not re.search(r"import\s+c?KDTree|scipy\.spatial", txt), f
    # the Python layer's A/B switches are options of the library too (py_*): its call paths read no environment variable
    for f in ("solve.py", "pipeline.py", "device.py", "kernels.py", "fusion.py", "fusion_dm.py", "graph.py", "mesh.py"):
        assert "os.environ" not in open(os.path.join(pkg, f)).read(), f
    n_getenv = 0
    csrc = os.path.join(pkg, "csrc")
    for f in os.listdir(csrc):
        n_getenv += len(re.findall(r"\bgetenv\s*\(", open(os.path.join(csrc, f)).read()))
    assert n_getenv <= 2, n_getenv


def test_options_round_trip():
    """dfh_set_option / dfh_get_option: known names round-trip, unknown names are refused (no GPU needed)."""
    import pytest
    _lib.set_option("k1_no_bricks", 1)
    assert _lib.get_option("k1_no_bricks") == 1
    _lib.set_option("k1_no_bricks", None)
    assert _lib.get_option("k1_no_bricks") == -1
    with pytest.raises(ValueError):
        _lib.set_option("no_such_switch", 1)
    assert _lib.get_option("no_such_switch") < -(2 ** 62)
