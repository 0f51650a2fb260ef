"""Band-voxel extraction (K9: dfh_surface_count / _emit) on the bench's canonical volume: HIP-event times of the count (+ scan)
and the emit pass, and of the count with a band so narrow that no voxel lies in it (the pure streaming read).
python3 tools/kbench_extract.py [--res 256] [--reps 20]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicfusion_body_amd import _lib, kernels, scene
from dynamicfusion_body_amd.device import current_stream_ptr, dtype_code

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
R = a.res
H, W, fx, cx, cy = scene.CAMERAS["C2" if R <= 256 else "C5"]
K = scene.intrinsics(fx, cx, cy)
Kinv = np.linalg.inv(K)
scale, center, tdist = scene.grid_params(R)
T = torch.full((R, R, R), tdist / scale, dtype=torch.float32, device="cuda")
Wt = torch.zeros_like(T)
for ang in (0.0, 40.0, -40.0):
    lw = scene.view_extrinsic(ang)
    d = torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda()
    kernels.integrate_depth(T, Wt, d, K, Kinv, lw, scale, center, tdist)
lib = _lib.load()
res = _lib.iarr((R, R, R))
ws = torch.empty((lib.dfh_surface_workspace_bytes(res) + 7) // 8, dtype=torch.int64, device="cuda")
total = torch.zeros(1, dtype=torch.int64, device="cuda")


def count(band):
    _lib.check(lib.dfh_surface_count(T.data_ptr(), Wt.data_ptr(), dtype_code(T), res, float(band), ws.data_ptr(), ws.numel() * 8,
                                     total.data_ptr(), current_stream_ptr()), "count")


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps * 1e3


t_empty = timeit(lambda: count(1e-9))
t_count = timeit(lambda: count(4.0))
S = int(total.item())
pos = torch.empty((S, 3), dtype=torch.float64, device="cuda")
nrm = torch.empty((S, 3), dtype=torch.float64, device="cuda")


def emit():
    _lib.check(lib.dfh_surface_emit(T.data_ptr(), Wt.data_ptr(), dtype_code(T), res, 0, 4.0, ws.data_ptr(), pos.data_ptr(), nrm.data_ptr(), S,
                                    current_stream_ptr()), "emit")


t_emit = timeit(emit)
nb = (R ** 3 + 1023) // 1024
print("%d^3: %d band samples (%.1f %% of the voxels); count + scan %.1f us (no voxel in the band: %.1f us = %.0f GB/s), emit %.1f us"
      % (R, S, 100.0 * S / R ** 3, t_count, t_empty, 8.0 * R ** 3 / t_empty / 1e3, t_emit))
