#!/usr/bin/env python3
"""Timing of the marching-cubes passes at R^3 (HIP events): fused-sphere TSDF, level 0."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamicfusion_body_amd import kernels, mesh, scene

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
R = a.res
H, W, fx, cx, cy = scene.CAMERAS["C2" if R <= 256 else "C5"]
K = scene.intrinsics(fx, cx, cy); Kinv = np.linalg.inv(K)
scale, center, tdist = scene.grid_params(R)
T = torch.full((R, R, R), tdist, dtype=torch.float32, device="cuda")
Wt = torch.zeros((R, R, R), dtype=torch.float32, device="cuda")
for ang in (0.0, 40.0, -40.0):
    lw = scene.view_extrinsic(ang)
    d = torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda()
    kernels.integrate_depth(T, Wt, d, K, Kinv, lw, scale, center, tdist)

def timeit(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

v, f, n, val = mesh.marching_cubes(T, 0.0, 1, order="lattice")
own = torch.floor(v[:, :2] + 1e-4).to(torch.int64)
rows = torch.unique(own[:, 0] * R + own[:, 1]).numel()
print("z rows owning a vertex: %d of %d (%.1f%%)" % (rows, R * R, 100.0 * rows / (R * R)))
for order in ("lattice", "reference"):
    v, f, n, val = mesh.marching_cubes(T, 0.0, 1, order=order)
    ms = timeit(lambda: mesh.marching_cubes(T, 0.0, 1, order=order), a.reps)
    vol_bytes = R ** 3 * 4
    print("marching cubes %d^3 order=%-9s: %8.1f us  (%d vertices, %d faces)  %.0f Mvox/s  volume read 2x + code 1x = %.0f GB/s"
          % (R, order, ms * 1e3, v.shape[0], f.shape[0], R ** 3 / ms / 1e3, (3 * vol_bytes + 2 * vol_bytes) / ms / 1e6))
