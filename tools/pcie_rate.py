#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in numpy path (FusionDM.fuseDepths with float64 numpy volumes:
H2D of T,w + kernel + D2H of T,w) next to the resident-tensor rate, 256^3."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamicfusion_body_amd import FusionDM, scene
R = 256
H, W, fx, cx, cy = scene.CAMERAS["C2"]
K = scene.intrinsics(fx, cx, cy)
scale, center, tdist = scene.grid_params(R)
f = FusionDM(tdist, K, tsdf_res=R)
lw = scene.view_extrinsic(0.0)
dm = scene.render_depth(K, lw, H, W, dtype=np.float32)
T = np.zeros((R, R, R)) + tdist; Wt = np.zeros((R, R, R))
f.fuseDepths(dm, lw, T, Wt, scale=scale, center=center)
t0 = time.perf_counter()
n = 3
for _ in range(n):
    f.fuseDepths(dm, lw, T, Wt, scale=scale, center=center)
dt = (time.perf_counter() - t0) / n
print("numpy-in/numpy-out (float64 host volumes, PCIe both ways): %.1f ms per 256^3 view = %.0f Mvox/s" % (dt * 1e3, R ** 3 / dt / 1e6))
Td, Wd = f._new_volume_pair(); d = torch.from_numpy(dm).cuda()
f.fuseDepths(d, lw, Td, Wd, scale=scale, center=center); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    f.fuseDepths(d, lw, Td, Wd, scale=scale, center=center)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 50
print("resident CUDA tensors: %.3f ms per view = %.0f Mvox/s" % (dt * 1e3, R ** 3 / dt / 1e6))
