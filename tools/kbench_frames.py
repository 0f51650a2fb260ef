#!/usr/bin/env python3
"""Per-frame wall time of the composed frame loop (bench scene, 256^3, 512 nodes), one line per frame:
shows warm-up effects (allocations, block-pattern growth) that the bench's average hides."""
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
sf = SlabFrame(K, scale, center, R, tdist / scale, node_pos, node_w, knn=4, pcg_iters=10, band=4.0, distributed=False)
views = [scene.view_extrinsic(a) for a in (0.0, 40.0, -40.0)]
def depth(lw, off, r):
    return torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0, sphere_offset=off * scale, sphere_r=scene.SPHERE_R * r)).cuda()
for lw in views:
    sf.integrate(depth(lw, np.zeros(3), 1.0), lw)
sf.refresh_samples()
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 16
frames = []
for f in range(nf):
    ph = 2 * np.pi * f / 21.0
    off = np.array([0.5 * np.sin(ph), -0.3 * np.sin(ph), 0.2 * np.sin(ph)])
    frames.append([depth(lw, off, 1.0 + 0.004 * np.sin(ph)) for lw in views])
torch.cuda.synchronize()
for f in range(nf):
    b0 = sf.fs.solver.B if getattr(sf.fs.solver, "_pattern", None) else 0
    t0 = time.perf_counter()
    n = sf.step(frames[f], views, gn_iters=10)
    torch.cuda.synchronize()
    print("frame %2d: %7.3f ms  samples %d  blocks %d -> %d" % (f, (time.perf_counter() - t0) * 1e3, n, b0, sf.fs.solver.B))
