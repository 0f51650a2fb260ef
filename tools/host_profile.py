#!/usr/bin/env python3
"""Where the HOST spends its time in the composed frame loop: cProfile over a few steady-state frames (no device
synchronisation of its own).  The frame is ~100 launches; wherever the host has just waited for a read-back the
device idles until the next launch is issued, so Python time right after a read-back is frame time."""
import cProfile, io, os, pstats, sys
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
sf = SlabFrame(K, scale, center, R, tdist / scale, node_pos, node_w, knn=4, pcg_iters=10, band=4.0, distributed=False)
views = [scene.view_extrinsic(a) for a in (0.0, 40.0, -40.0)]
def depth(lw, off, r):
    return torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0, sphere_offset=off * scale, sphere_r=scene.SPHERE_R * r)).cuda()
for lw in views:
    sf.integrate(depth(lw, np.zeros(3), 1.0), lw)
sf.refresh_samples()
frames = []
for f in range(14):
    ph = 2 * np.pi * f / 21.0
    off = np.array([0.5 * np.sin(ph), -0.3 * np.sin(ph), 0.2 * np.sin(ph)])
    frames.append([depth(lw, off, 1.0 + 0.004 * np.sin(ph)) for lw in views])
for f in range(6):
    sf.step(frames[f], views, gn_iters=10)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for f in range(6, 14):
    sf.step(frames[f], views, gn_iters=10)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print("8 frames; times below are totals over them (divide by 8)")
print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:6000])
