#!/usr/bin/env python3
"""Stage times of the composed frame (bench.py's frame leg, 256^3 / 512 nodes / 3 views) with the data term on the first view
only (round 2) and on all views; and the valid-sample counts of both."""
import os, sys
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
angles = (0.0, 40.0, -40.0)
lws = [scene.view_extrinsic(a) for a in angles]
depths = []
for f in range(8):
    off = np.array([0.10, -0.07, 0.05]) * (f + 1) * scale
    depths.append([torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0, sphere_offset=off,
                                                       sphere_r=scene.SPHERE_R * (1.0 + 0.004 * (f + 1)))).cuda() for lw in lws])
for dv in (1, None):
    sf = SlabFrame(K, scale, center, R, tdist / scale, node_pos, node_w, knn=4, pcg_iters=10, band=4.0, distributed=False)
    for lw in lws:
        sf.integrate(torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda(), lw)
    sf.refresh_samples()
    for f in range(2):
        sf.step(depths[f], lws, gn_iters=10, data_views=dv)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for f in range(2, 8):
        sf.step(depths[f], lws, gn_iters=10, data_views=dv)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 6 * 1e3
    st = {}
    for f in range(2, 8):
        sf.step(depths[f], lws, gn_iters=10, data_views=dv, stage_ms=st)
    cost, cnt = sf.fs.solver.cost()
    print("data term on %s: frame %.3f ms; stages (with syncs) %s; samples %d, valid %d, rows %d, blocks %d" %
          ("view 0 only" if dv == 1 else "all %d views" % len(lws), ms, {k: round(v / 6, 3) for k, v in st.items()}, sf.fs.solver.S, cnt,
           sf.fs.solver.n_rows, sf.fs.solver.B))
