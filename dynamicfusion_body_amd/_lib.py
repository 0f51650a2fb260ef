"""ctypes binding of libdfusion_hip.so (include/dfusion_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails, the caller
gets an exception.  Build with `python -m dynamicfusion_body_amd.build`.
"""
import ctypes
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
# DFH_LIB_PATH: an alternative build of the SAME library (kernel experiments, tools/build_variant.sh) -- not a fallback
LIB_PATH = os.environ.get("DFH_LIB_PATH") or os.path.join(_PKG, "libdfusion_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "dfusion_hip.h")

F32, F64 = 0, 1
ABI_VERSION = 4

_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_int_p = ctypes.POINTER(ctypes.c_int)
_vp = ctypes.c_void_p
_int = ctypes.c_int
_dbl = ctypes.c_double

_SIGNATURES = {
    "dfh_version": (_int, []),
    "dfh_last_error": (ctypes.c_char_p, []),
    "dfh_stream_synchronize": (_int, [_vp]),
    "dfh_set_option": (_int, [ctypes.c_char_p, ctypes.c_long]),
    "dfh_get_option": (ctypes.c_long, [ctypes.c_char_p]),
    "dfh_integrate_workspace_bytes": (ctypes.c_size_t, [_int, _int, _int, _c_int_p, _int, _int]),
    "dfh_integrate_depth": (_int, [_vp, _vp, _int, _c_int_p, _int, _int, _int, _vp, _int, _int, _int,
                                   _c_double_p, _c_double_p, _c_double_p, _dbl, _c_double_p, _dbl, _dbl, _vp, ctypes.c_size_t, _vp]),
    "dfh_integrate_depth_path": (_int, [_int, _c_int_p, _int, _int, _int, _int, _int]),
    "dfh_integrate_depth_ocl": (_int, [_vp, _vp, _c_int_p, _int, _int, _vp, _int, _int, ctypes.POINTER(ctypes.c_float),
                                       ctypes.POINTER(ctypes.c_float), ctypes.c_float, ctypes.c_float, _vp]),
    "dfh_integrate_multi_workspace_bytes": (ctypes.c_size_t, [_int]),
    "dfh_integrate_depth_multi": (_int, [_vp, _vp, _int, _c_int_p, _int, _int, _int, _int, ctypes.POINTER(ctypes.c_void_p), _int, _int,
                                         _int, _c_double_p, _c_double_p, _c_double_p, _dbl, _c_double_p, _dbl, _dbl, _vp,
                                         ctypes.c_size_t, _vp]),
    "dfh_integrate_depth_multi_fresh": (_int, [_vp, _vp, _int, _c_int_p, _int, _int, _int, _dbl, _int, ctypes.POINTER(ctypes.c_void_p), _int, _int,
                                         _int, _c_double_p, _c_double_p, _c_double_p, _dbl, _c_double_p, _dbl, _dbl, _vp,
                                         ctypes.c_size_t, _vp]),
    "dfh_fuse_volume_rigid": (_int, [_vp, _vp, _int, _c_int_p, _int, _int, _vp, _int, _c_int_p,
                                     _c_double_p, _dbl, _dbl, _vp]),
    "dfh_dqb_workspace_bytes": (ctypes.c_size_t, [_c_int_p, _int, _int]),
    "dfh_dqb_workspace_bytes_cached": (ctypes.c_size_t, [_c_int_p, _int, _int, _int, _int, _int]),
    "dfh_fuse_volume_dqb": (_int, [_vp, _vp, _int, _c_int_p, _int, _int, _vp, _int, _c_int_p, _vp, _vp, _vp, _int,
                                   _int, _c_double_p, _dbl, _dbl, _vp, ctypes.c_size_t, _int, _vp]),
    "dfh_residual_rigid": (_int, [_vp, _vp, _vp, _int, _c_double_p, _vp, _vp]),
    "dfh_gn_build_rigid": (_int, [_vp, _vp, _vp, _vp, _int, _c_double_p, _vp, _vp]),
    "dfh_residual_data": (_int, [_vp, _vp, _vp, _vp, _int, _int, _vp, _vp, _vp, _int, _c_double_p, _vp, _vp]),
    "dfh_residual_reg": (_int, [_vp, _int, _int, _vp, _vp, _vp, _dbl, _vp, _vp]),
    "dfh_warp_points": (_int, [_vp, _vp, _vp, _int, _int, _vp, _vp, _vp, _int, _c_double_p, _vp, _vp, _vp]),
    "dfh_closest_correspondences": (_int, [_vp, _vp, _int, _vp, _int, _int, _dbl, _vp, _vp, _vp, _vp]),
    "dfh_nearest_points": (_int, [_vp, _int, _vp, _int, _vp, _vp, _vp]),
    "dfh_graph_unsupported": (_int, [_vp, _int, _vp, _int, _vp, _vp, _int, _vp, _vp]),
    "dfh_dq_blend_points": (_int, [_vp, _int, _vp, _int, _vp, _vp, _vp, _int, _vp, _vp]),
    "dfh_sample_knn": (_int, [_vp, _int, _vp, _vp, _int, _int, _vp, _vp, _vp]),
    "dfh_dqb_skip_layout": (_int, [_c_int_p, _int, _int, _c_int_p, _int, _int, ctypes.POINTER(ctypes.c_size_t)]),
    "dfh_dqb_build_candidates": (_int, [_c_int_p, _int, _int, _vp, _int, _int, _vp, ctypes.c_size_t, _vp]),
    "dfh_sample_knn_bricks": (_int, [_vp, _int, _vp, _vp, _int, _int, _c_int_p, _int, _int, _vp, ctypes.c_size_t, _vp, _vp, _vp]),
    "dfh_gn_associate": (_int, [_vp, _vp, _vp, _int, _int, _vp, _c_double_p, _vp, _int, _int, _int, _c_double_p,
                                _c_double_p, _c_double_p, _dbl, _c_double_p, _dbl, _dbl, _vp, _vp, _vp]),
    "dfh_gn_build": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _vp, _vp, _vp, _vp, _int, _c_double_p, _dbl,
                            _vp, _vp, _int, _vp, _vp, _vp, _vp]),
    "dfh_permute_samples": (_int, [_vp, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dfh_gn_partial_doubles": (ctypes.c_size_t, [_int]),
    "dfh_gn_build_planned": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _vp, _vp, _vp, _vp, _int, _c_double_p, _dbl,
                                    _vp, _vp, _int, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                    _dbl, _vp]),
    "dfh_gn_build_planned_assoc": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _vp, _vp, _vp, _vp, _int, _c_double_p, _dbl,
                                          _vp, _vp, _int, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                          _dbl, _vp, _int, _int, _c_double_p, _c_double_p, _c_double_p, _dbl, _c_double_p, _dbl, _dbl, _vp]),
    "dfh_gn_iteration": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _vp, _vp, _vp, _vp, _int, _c_double_p, _dbl,
                                _vp, _vp, _int, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                _dbl, _vp, _int, _int, _c_double_p, _c_double_p, _c_double_p, _dbl, _c_double_p, _dbl, _dbl,
                                _int, _dbl, _dbl, _vp, _vp, ctypes.c_size_t, _dbl, _vp]),
    "dfh_gn_views_bytes": (ctypes.c_size_t, [_int]),
    "dfh_gn_pack_views": (_int, [_vp, _int, ctypes.POINTER(ctypes.c_void_p), _c_double_p, _vp]),
    "dfh_gn_views_bytes_cells": (ctypes.c_size_t, [_int, _int, _int]),
    "dfh_gn_pack_views_cells": (_int, [_vp, _int, ctypes.POINTER(ctypes.c_void_p), _int, _int, _c_double_p, _vp]),
    "dfh_gn_associate_views": (_int, [_vp, _vp, _vp, _int, _int, _vp, _c_double_p, _vp, _int, _int, _int, _int, _c_double_p,
                                      _c_double_p, _dbl, _c_double_p, _dbl, _dbl, _vp, _vp, _vp]),
    "dfh_gn_build_planned_assoc_views": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _vp, _vp, _vp, _vp, _int, _c_double_p, _dbl,
                                                _vp, _vp, _int, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                                _dbl, _vp, _int, _int, _int, _c_double_p, _c_double_p, _dbl, _c_double_p, _dbl, _dbl, _vp, _int, _vp]),
    "dfh_gn_iteration_views": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _vp, _vp, _vp, _vp, _int, _c_double_p, _dbl,
                                      _vp, _vp, _int, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                      _dbl, _vp, _int, _int, _int, _c_double_p, _c_double_p, _dbl, _c_double_p, _dbl, _dbl,
                                      _int, _dbl, _dbl, _vp, _vp, ctypes.c_size_t, _dbl, _int, _vp, _int, _vp]),
    "dfh_gn_frame_solve_views": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _vp, _vp, _vp, _vp, _int, _c_double_p, _dbl,
                                      _vp, _vp, _int, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                      _dbl, _vp, _int, _int, _int, _c_double_p, _c_double_p, _dbl, _c_double_p, _dbl, _dbl,
                                      _int, _dbl, _dbl, _vp, _vp, ctypes.c_size_t, _dbl, _int, _vp, _int,
                                      _int, _dbl, _vp, _vp, ctypes.c_size_t, _vp]),
    "dfh_gn_pack_upper": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _vp, _vp]),
    "dfh_gn_unpack_upper": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _vp, _vp]),
    "dfh_gn_sort_workspace_bytes": (ctypes.c_size_t, [_int]),
    "dfh_gn_sort_samples": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "dfh_gn_tile_samples": (_int, []),
    "dfh_gn_plan_count": (_int, [_vp, _int, _int, _vp, _vp, _vp]),
    "dfh_gn_plan_workspace_bytes": (ctypes.c_size_t, [_int, _int]),
    "dfh_gn_plan_build": (_int, [_vp, _int, _int, _int, _vp, _int, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                 ctypes.c_size_t, _vp]),
    "dfh_pcg_workspace_bytes": (ctypes.c_size_t, [_int, _int]),
    "dfh_pcg_solve": (_int, [_vp, _vp, _vp, _vp, _int, _int, _dbl, _dbl, _vp, _vp, ctypes.c_size_t, _vp]),
    "dfh_pcg_solve_update": (_int, [_vp, _vp, _vp, _vp, _int, _int, _dbl, _dbl, _vp, _vp, ctypes.c_size_t, _vp, _dbl, _vp]),
    "dfh_pcg_set_mode": (_int, [_int]),
    "dfh_pcg_path": (_int, [_int]),
    "dfh_pcg_status": (_int, [_vp, ctypes.POINTER(ctypes.c_long)]),
    "dfh_pcg_status_peek": (_int, [_vp, ctypes.POINTER(ctypes.c_long)]),
    "dfh_apply_twist": (_int, [_vp, _vp, _int, _dbl, _vp]),
    "dfh_relax_twists": (_int, [_vp, _int, _dbl, _vp]),
    "dfh_gn_global_sampled_bytes": (ctypes.c_size_t, [_int, _int]),
    "dfh_gn_global_sampled_views": (_int, [_vp, _vp, _vp, _vp, _int, _int, _vp, _int, _c_double_p, _dbl, _vp, _int, _int, _int, _c_double_p,
                                           _c_double_p, _dbl, _c_double_p, _dbl, _dbl, _int, _dbl, _int, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "dfh_gn_global_apply": (_int, [_vp, _dbl, _int, _vp, _vp, _vp]),
    "dfh_gn_global_step_bytes": (ctypes.c_size_t, []),
    "dfh_gn_global_step": (_int, [_vp, _int, _vp, _int, _dbl, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "dfh_surface_workspace_bytes": (ctypes.c_size_t, [_c_int_p]),
    "dfh_surface_count": (_int, [_vp, _vp, _int, _c_int_p, _dbl, _vp, ctypes.c_size_t, _vp, _vp]),
    "dfh_surface_emit": (_int, [_vp, _vp, _int, _c_int_p, _int, _dbl, _vp, _vp, _vp, ctypes.c_long, _vp]),
    "dfh_mc_workspace_bytes": (ctypes.c_size_t, [_c_int_p, _int]),
    "dfh_mc_count": (_int, [_vp, _int, _c_int_p, _int, _dbl, _vp, ctypes.c_size_t, _vp, _vp]),
    "dfh_mc_emit": (_int, [_vp, _int, _c_int_p, _int, _dbl, _vp, ctypes.c_size_t, _vp, _vp, _vp, _vp, ctypes.c_long, ctypes.c_long,
                           ctypes.c_long, _vp]),
    "dfh_mc_reorder_workspace_bytes": (ctypes.c_size_t, [ctypes.c_long, ctypes.c_long]),
    "dfh_mc_reorder": (_int, [_vp, _vp, _vp, _vp, ctypes.c_long, ctypes.c_long, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
}

_lib = None


class DfhError(RuntimeError):
    pass


class DfhTimeout(DfhError):
    """A persistent kernel gave up waiting in a grid barrier (DFH_E_TIMEOUT)."""


def declared_symbols(header=HEADER_PATH):
    """Every function the public header declares (used by the symbol-export test)."""
    txt = open(header).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dfh_[a-z0-9_]+)\s*\(", txt)))


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DfhError("%s not found: the HIP library is the only implementation of this path "
                       "(no CPU fallback). Build it with `python -m dynamicfusion_body_amd.build`." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    v = lib.dfh_version()
    if v != ABI_VERSION:
        raise DfhError("libdfusion_hip.so ABI version %d != expected %d (rebuild)" % (v, ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().dfh_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError("%s: %s" % (what, msg))
        if rc == -4:
            raise DfhTimeout("%s: %s" % (what, msg))
        raise DfhError("%s failed (%d): %s" % (what, rc, msg))


_options_touched = set()


def set_option(name, value):
    """A development switch of the library (include/dfusion_hip.h: dfh_set_option); value None = unset (-1)."""
    check(load().dfh_set_option(name.encode(), -1 if value is None else int(value)), "dfh_set_option")
    _options_touched.add(name)


def reset_options():
    """Every switch changed through set_option() back to unset (tests call this between cases)."""
    for name in sorted(_options_touched):
        check(load().dfh_set_option(name.encode(), -1), "dfh_set_option")
    _options_touched.clear()


def get_option(name):
    return int(load().dfh_get_option(name.encode()))


def opt_on(name):
    """True when a development switch is set (> 0).  The Python layer's own A/B switches (py_*) live in the library's option
    table too: DFH_OPTIONS="py_no_side_stream=1" or set_option(); no call path reads the environment."""
    return load().dfh_get_option(name.encode()) > 0


_darr_cache = {}


def darr(values, n):
    """n doubles as a ctypes array.  The same small numpy arrays (intrinsics, poses) are passed on every launch of a frame:
    conversions are remembered by content (a few entries; the returned arrays are read-only by convention)."""
    import numpy as np
    a = np.ascontiguousarray(np.asarray(values, dtype=np.float64).reshape(-1))
    if a.size != n:
        raise ValueError("expected %d values, got %d" % (n, a.size))
    key = a.tobytes()
    hit = _darr_cache.get(key)
    if hit is None:
        if len(_darr_cache) > 256:
            _darr_cache.clear()
        hit = _darr_cache[key] = (ctypes.c_double * n).from_buffer_copy(key)
    return hit


def iarr(values):
    vals = [int(v) for v in values]
    return (ctypes.c_int * len(vals))(*vals)
