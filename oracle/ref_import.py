"""Import the reference's hot-path modules IN THIS CONTAINER ONLY (test infrastructure).

The reference (/root/reference) never travels to the GPU box; this helper exists so that
tests/golden/make_golden.py can run the reference's own functions on seeded inputs and
commit the resulting vectors under tests/golden/.  It is never imported by the product
(dynamicfusion_body_amd/*) nor by anything that runs on the GPU box.

`import core` pulls tensorflow / OpenGL / skimage / pyopencl (core/__init__.py:2-4,
core/fusion_dm.py:38,42, core/sdf.py:8) which are absent here and are never touched by
the hot path, so inert empty modules are registered for them.
Run python with -B so no bytecode is written into the read-only reference tree.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("DFUSION_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "core", "fusion_dm.py"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load():
    """Returns (util_module, FusionDM, Fusion) from the reference."""
    if not available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
    sys.dont_write_bytecode = True
    if "pyopencl" not in sys.modules:
        _stub("pyopencl")
        sk = _stub("skimage")
        sk.measure = _stub("skimage.measure")
        tf = _stub("tensorflow")
        tf.nn = types.SimpleNamespace(elu=None, relu=None)
        tf.contrib = _stub("tensorflow.contrib")
        tf.contrib.slim = _stub("tensorflow.contrib.slim")
        gl = _stub("OpenGL")
        gl.GLUT = _stub("OpenGL.GLUT")
        gl.GLU = _stub("OpenGL.GLU")
        gl.GL = _stub("OpenGL.GL")
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import core.util as util  # noqa: E402
    from core.fusion_dm import FusionDM  # noqa: E402
    from core.fusion import Fusion  # noqa: E402
    return util, FusionDM, Fusion
