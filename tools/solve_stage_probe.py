#!/usr/bin/env python3
"""Where does the solve stage of a composed frame go?  First GN iteration (pattern check + plan) against the other nine."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamicfusion_body_amd import scene
from dynamicfusion_body_amd.pipeline import SlabFrame
R, N = 256, 512
H, W, fx, cx, cy = scene.CAMERAS["C2"]
K = scene.intrinsics(fx, cx, cy)
scale, center, tdist = scene.grid_params(R)
node_pos, node_w = scene.fibonacci_nodes(N, R)
lws = [scene.view_extrinsic(a) for a in (0.0, 40.0, -40.0)]
sf = SlabFrame(K, scale, center, R, tdist / scale, node_pos, node_w, knn=4, pcg_iters=10, band=4.0, distributed=False)
for lw in lws:
    sf.integrate(torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda(), lw)
sf.refresh_samples()
ds = []
for f in range(8):
    off = np.array([0.10, -0.07, 0.05]) * (f + 1) * scale
    ds.append([torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0, sphere_offset=off)).cuda() for lw in lws])
for f in range(3):
    sf.step(ds[f], lws)
torch.cuda.synchronize()
sv = sf.fs.solver
acc = {"pattern+plan (host+device, 2 read-backs)": 0.0, "first iteration after it": 0.0, "nine iterations": 0.0, "refresh_samples": 0.0}
for f in range(3, 8):
    d, lw = ds[f][0], lws[0]
    sf.refresh_samples(); torch.cuda.synchronize()
    t0 = time.perf_counter(); sv._build_pattern(); torch.cuda.synchronize(); t1 = time.perf_counter()
    sf.fs.gn_iteration(d, lw, huber=0.5); torch.cuda.synchronize(); t2 = time.perf_counter()
    for _ in range(9):
        sf.fs.gn_iteration(d, lw, huber=0.5)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    sf.refresh_samples(); torch.cuda.synchronize(); t4 = time.perf_counter()
    for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
        acc[k] += v * 1e3 / 5
print({k: round(v, 3) for k, v in acc.items()}, "samples", sv.S, "rows", sv.n_rows, "blocks", sv.B)
