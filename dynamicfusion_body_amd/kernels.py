"""Tensor-level entry points: device tensors in, updated in place, asynchronous on the
current HIP stream.  Each function is one call through the C ABI (include/dfusion_hip.h);
the reference-shaped classes in fusion_dm.py / fusion.py are built on these."""
import numpy as np
import torch

from . import _lib
from .device import current_stream_ptr, dtype_code, require_gpu


def _check_volume_pair(T, Wt, res, x_range):
    if not (isinstance(T, torch.Tensor) and isinstance(Wt, torch.Tensor)):
        raise ValueError("volumes must be torch tensors on the GPU")
    if not (T.is_cuda and Wt.is_cuda):
        raise ValueError("volumes must live on the GPU")
    if T.dtype != Wt.dtype:
        raise ValueError("tsdf and weight volumes must share a dtype")
    if not (T.is_contiguous() and Wt.is_contiguous()):
        raise ValueError("volumes must be contiguous [x][y][z]")
    x0, x1 = x_range
    want = (x1 - x0, res[1], res[2])
    if tuple(T.shape) != want or tuple(Wt.shape) != want:
        raise ValueError("volume shape %s / %s does not match slab %s of grid %s"
                         % (tuple(T.shape), tuple(Wt.shape), want, tuple(res)))


def integrate_depth(T, Wt, depth, K, Kinv, lw, scale, center, tdist, wmax=100.0, tsdf_res=None,
                    res=None, x_range=None):
    """K1 = FusionDM.fuseDepths (reference core/fusion_dm.py:180-217) on device tensors.

    T, Wt : (x1-x0, Y, Z) float32/float64 CUDA tensors holding planes [x0,x1) of a `res`
            grid (default: the whole grid).  depth: (H, W) float32/float64 CUDA tensor of
            negative depths.  Runs asynchronously on the current stream."""
    require_gpu()
    lib = _lib.load()
    if res is None:
        res = tuple(T.shape)
    if x_range is None:
        x_range = (0, res[0])
    if tsdf_res is None:
        tsdf_res = res[0]
    _check_volume_pair(T, Wt, res, x_range)
    if x_range[1] == x_range[0]:
        return T, Wt
    if not (isinstance(depth, torch.Tensor) and depth.is_cuda and depth.dim() == 2 and depth.is_contiguous()):
        raise ValueError("depth must be a contiguous 2-D CUDA tensor")
    H, W = depth.shape
    rc = lib.dfh_integrate_depth(T.data_ptr(), Wt.data_ptr(), dtype_code(T), _lib.iarr(res), int(tsdf_res),
                                 int(x_range[0]), int(x_range[1]), depth.data_ptr(), dtype_code(depth),
                                 int(H), int(W), _lib.darr(K, 9), _lib.darr(Kinv, 9), _lib.darr(lw, 12),
                                 float(scale), _lib.darr(np.asarray(center, dtype=np.float64), 3),
                                 float(tdist), float(wmax), current_stream_ptr())
    _lib.check(rc, "dfh_integrate_depth")
    return T, Wt
