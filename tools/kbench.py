#!/usr/bin/env python3
"""Per-view timing of K1 on the GPU box (HIP events, many reps per view).
usage: python tools/kbench.py [--res 256] [--reps 50] [--angles 0,30,-45,60]"""
import argparse, os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamicfusion_body_amd import kernels, scene

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--angles", default="0,30,-45,60,135")
ap.add_argument("--dtype", default="f32")
a = ap.parse_args()
R = a.res
cam = "C2" if R <= 256 else "C5"
H, W, fx, cx, cy = scene.CAMERAS[cam]
K = scene.intrinsics(fx, cx, cy); Kinv = np.linalg.inv(K)
scale, center, tdist = scene.grid_params(R)
vt = torch.float32 if a.dtype == "f32" else torch.float64
T = torch.full((R, R, R), tdist, dtype=vt, device="cuda")
Wt = torch.zeros((R, R, R), dtype=vt, device="cuda")
# HBM copy ceiling for reference (float4 copy via torch)
src = torch.empty(R * R * R * 2, dtype=torch.float32, device="cuda"); dst = torch.empty_like(src)
for _ in range(3): dst.copy_(src)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): dst.copy_(src)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print("copy ceiling: %.1f us for %d MB r+w -> %.0f GB/s" % (ms * 1e3, src.numel() * 8 / 1e6, src.numel() * 8 / ms / 1e6))
for ang in [float(x) for x in a.angles.split(",")]:
    lw = scene.view_extrinsic(ang)
    d = torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32)).cuda()
    for _ in range(5):
        kernels.integrate_depth(T, Wt, d, K, Kinv, lw, scale, center, tdist)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.reps):
        kernels.integrate_depth(T, Wt, d, K, Kinv, lw, scale, center, tdist)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    # updated fraction for this view alone
    T2 = torch.full((R, R, R), tdist, dtype=vt, device="cuda"); W2 = torch.zeros((R, R, R), dtype=vt, device="cuda")
    kernels.integrate_depth(T2, W2, d, K, Kinv, lw, scale, center, tdist)
    frac = float((W2 > 0).float().mean())
    alg = 16.0 * R ** 3 + 4 * H * W
    print("view %6.1f: %8.1f us  updated %.3f  alg %.0f GB/s (%.1f%% of 8TB/s)  touched-bytes est %.0f GB/s"
          % (ang, ms * 1e3, frac, alg / ms / 1e6, alg / ms / 1e6 / 80, (16.0 * R ** 3 * frac) / ms / 1e6))
