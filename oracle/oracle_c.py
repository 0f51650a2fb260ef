"""ctypes loader of oracle/liboracle_c.so (the C restatement) -- TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle_c.so")
_lib = None


def build(force=False):
    src = os.path.join(HERE, "oracle_c.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "-B", "liboracle_c.so"], check=True, capture_output=True)
    return LIB


def load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(LIB)
        _lib.oracle_c_fuse_depths.restype = ctypes.c_long
        _lib.oracle_c_threads.restype = ctypes.c_int
    return _lib


def threads():
    return int(load().oracle_c_threads())


def fuse_depths(dm, lw, K, Kinv, tsdf, tsdf_w, tdist, tsdf_res=None, scale=1.0, center=np.zeros(3), wmax=100.0,
                x_range=None, n_threads=0):
    """Same contract as oracle_np.fuse_depths (in place on float64 C-contiguous volumes)."""
    lib = load()
    assert tsdf.dtype == np.float64 and tsdf_w.dtype == np.float64 and tsdf.flags.c_contiguous and tsdf_w.flags.c_contiguous
    dm = np.ascontiguousarray(dm)
    if dm.dtype not in (np.float32, np.float64):
        dm = dm.astype(np.float64)
    X, Y, Z = tsdf.shape
    a, b = (0, X) if x_range is None else x_range
    dp = lambda arr: np.ascontiguousarray(np.asarray(arr, dtype=np.float64)).ctypes.data_as(ctypes.c_void_p)
    Kc, Kic, lwc, cc = (np.ascontiguousarray(np.asarray(v, dtype=np.float64)) for v in (K, Kinv, lw, center))
    n = lib.oracle_c_fuse_depths(tsdf.ctypes.data_as(ctypes.c_void_p), tsdf_w.ctypes.data_as(ctypes.c_void_p),
                                 ctypes.c_int(X), ctypes.c_int(Y), ctypes.c_int(Z),
                                 ctypes.c_int(X if tsdf_res is None else int(tsdf_res)), ctypes.c_int(a), ctypes.c_int(b),
                                 dm.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(1 if dm.dtype == np.float32 else 0),
                                 ctypes.c_int(dm.shape[0]), ctypes.c_int(dm.shape[1]),
                                 Kc.ctypes.data_as(ctypes.c_void_p), Kic.ctypes.data_as(ctypes.c_void_p),
                                 lwc.ctypes.data_as(ctypes.c_void_p), ctypes.c_double(scale),
                                 cc.ctypes.data_as(ctypes.c_void_p), ctypes.c_double(tdist), ctypes.c_double(wmax),
                                 ctypes.c_int(n_threads))
    return int(n)
