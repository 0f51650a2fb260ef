#!/usr/bin/env python3
"""K1 with V views: V consecutive sweeps (dfh_integrate_depth) against one fused sweep (dfh_integrate_depth_multi)."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamicfusion_body_amd import kernels, scene
ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--views", type=int, default=3)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--orbit", action="store_true", help="BASELINE config 5's views: 45 degrees apart around the grid (default: a fan of +-50 degrees)")
a = ap.parse_args()
R, V = a.res, a.views
H, W, fx, cx, cy = scene.CAMERAS["C2" if R <= 256 else "C5"]
K = scene.intrinsics(fx, cx, cy); Kinv = np.linalg.inv(K)
scale, center, tdist = scene.grid_params(R)
lws = [scene.view_extrinsic(ang) for ang in (45.0 * np.arange(V) if a.orbit else np.linspace(-50, 50, V))]
dms = [torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32)).cuda() for lw in lws]
T = torch.full((R, R, R), tdist / scale, dtype=torch.float32, device="cuda"); Wt = torch.zeros_like(T)
ws = kernels.integrate_workspace(min(V, 16), H, W, (R, R, R))
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(a.reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps * 1e3
def seq():
    for d, lw in zip(dms, lws):
        kernels.integrate_depth(T, Wt, d, K, Kinv, lw, scale, center, tdist)
def fused():
    kernels.integrate_depth_views(T, Wt, dms, K, Kinv, lws, scale, center, tdist, workspace=ws)
us_s, us_f = timeit(seq), timeit(fused)
print("%d^3, %d views of %dx%d: consecutive sweeps %.1f us (%.1f us/view, %.0f Mvox/s per view) | fused sweep %.1f us (%.1f us/view, %.0f Mvox/s per view)"
      % (R, V, W, H, us_s, us_s / V, R ** 3 * V / us_s, us_f, us_f / V, R ** 3 * V / us_f))
