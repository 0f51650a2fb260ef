#!/usr/bin/env python3
"""Timing of K2 (rigid) and K3 (DQB) TSDF->TSDF fusion at R^3 (HIP events)."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamicfusion_body_amd import kernels, scene
from dynamicfusion_body_amd.dq import twist_exp_dq

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--nodes", type=int, default=512)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--lib", default=None, help="alternative libdfusion_hip.so (kernel experiments)")
ap.add_argument("--options", default="", help="library switches, e.g. k3_exact=1 (the fp64 chain on float32 volumes)")
a = ap.parse_args()
from dynamicfusion_body_amd import _lib
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)
for kv in filter(None, a.options.split(",")):
    _lib.set_option(kv.split("=")[0], int(kv.split("=")[1]))
R, N, k = a.res, a.nodes, 4
tdist = 4.0
g = torch.arange(R, device="cuda", dtype=torch.float32)
d = torch.sqrt((g[:, None, None] - R / 2) ** 2 + (g[None, :, None] - R / 2) ** 2 + (g[None, None, :] - R / 2) ** 2)
live = torch.clamp(d - 0.3125 * R + 0.7, -1.5 * tdist, 1.5 * tdist).contiguous()
T = torch.clamp(d - 0.3125 * R, -tdist, tdist).contiguous(); W = torch.ones_like(T)
lw = twist_exp_dq(np.array([0.01, -0.02, 0.015, 0.3, -0.2, 0.1]))
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = timeit(lambda: kernels.fuse_volume_rigid(T, W, live, lw, tdist), a.reps)
alg = 20.0 * R ** 3
print("K2 rigid  %d^3: %8.1f us  %8.0f Mvox/s  alg %.0f GB/s (%.1f%% of 8 TB/s)" % (R, ms * 1e3, R ** 3 / ms / 1e3, alg / ms / 1e6, alg / ms / 1e6 / 80))
rng = np.random.default_rng(0)
node_pos, node_w = scene.fibonacci_nodes(N, R)
node_dq = twist_exp_dq(rng.normal(size=(N, 6)) * np.array([.01, .01, .01, .3, .3, .3]))
ws = kernels.dqb_workspace((R, R, R))
kernels.fuse_volume_dqb(T, W, live, node_pos, node_dq, node_w, k, lw, tdist, workspace=ws, rebuild_candidates=True)
P = torch.from_numpy(node_pos).cuda(); Q = torch.from_numpy(node_dq).cuda(); Wn = torch.from_numpy(node_w).cuda()
ms_c = timeit(lambda: kernels.fuse_volume_dqb(T, W, live, P, Q, Wn, k, lw, tdist, workspace=ws, rebuild_candidates=True), 3)
ms = timeit(lambda: kernels.fuse_volume_dqb(T, W, live, P, Q, Wn, k, lw, tdist, workspace=ws, rebuild_candidates=False), a.reps)
cnt = ws.view(-1, 257)[:, 0]          # kCap + 1 ints per brick (csrc/dfh_fuse_volume.hip)
print("K3 dqb    %d^3, %d nodes: %8.1f us (+%.1f us candidate rebuild)  %8.0f Mvox/s  alg %.0f GB/s (%.1f%% of 8 TB/s)"
      % (R, N, ms * 1e3, (ms_c - ms) * 1e3, R ** 3 / ms / 1e3, alg / ms / 1e6, alg / ms / 1e6 / 80))
for level, what in ((1, "node indices"), (2, "indices + weights")):
    wsc = kernels.dqb_workspace((R, R, R), knn=k, n_nodes=N, level=level)
    ms_c = timeit(lambda: kernels.fuse_volume_dqb(T, W, live, P, Q, Wn, k, lw, tdist, workspace=wsc, rebuild_candidates=True), 3)
    ms = timeit(lambda: kernels.fuse_volume_dqb(T, W, live, P, Q, Wn, k, lw, tdist, workspace=wsc, rebuild_candidates=False), a.reps)
    alg_c = (20.0 + 2 * k + (8 * (k + 1) if level == 2 else 0)) * R ** 3
    print("K3 dqb, stored %s: %8.1f us (first call, search + store: %.1f us)  %8.0f Mvox/s  alg %.0f GB/s (%.1f%% of 8 TB/s)"
          % (what, ms * 1e3, ms_c * 1e3, R ** 3 / ms / 1e3, alg_c / ms / 1e6, alg_c / ms / 1e6 / 80))
    del wsc
print("   candidates per brick: mean %.1f max %d, overflow bricks %d of %d" % (float(cnt.clamp(min=0).float().mean()), int(cnt.max()), int((cnt < 0).sum()), cnt.numel()))

