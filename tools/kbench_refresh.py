#!/usr/bin/env python3
"""Where the per-frame sample refresh spends its time (256^3, band 4): extraction, k-NN, sort, pattern, plan."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamicfusion_body_amd import scene
from dynamicfusion_body_amd.pipeline import SlabFrame, extract_surface_samples
from dynamicfusion_body_amd.solve import sample_knn
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--nodes", type=int, default=512)
a = ap.parse_args()
R = a.res
H, W, fx, cx, cy = scene.CAMERAS["C2" if R <= 256 else "C5"]
K = scene.intrinsics(fx, cx, cy)
scale, center, tdist = scene.grid_params(R)
node_pos, node_w = scene.fibonacci_nodes(a.nodes, R)
sf = SlabFrame(K, scale, center, R, tdist / scale, node_pos, node_w, knn=4, pcg_iters=10, band=4.0)
for ang in (0.0, 40.0, -40.0):
    lw = scene.view_extrinsic(ang)
    sf.integrate(torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda(), lw)
sf.refresh_samples()
lw_cam = scene.view_extrinsic(0.0)
d = torch.from_numpy(scene.render_depth(K, lw_cam, H, W, dtype=np.float32, sphere_offset=np.array([0.1, -0.07, 0.05]) * scale, sphere_r=scene.SPHERE_R * 1.004)).cuda()
sf.step(d, lw_cam, gn_iters=2)
sv = sf.fs.solver
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r
ms, (pos, nrm) = t(lambda: extract_surface_samples(sf.T, sf.Wt, 4.0)); print("extract        %.3f ms  (%d samples)" % (ms, pos.shape[0]))
ms, (nbr, wts) = t(lambda: sample_knn(pos, sv.node_pos, sv.node_w, 4)); print("sample_knn     %.3f ms" % ms)
ms, _ = t(lambda: sample_knn(pos, sv.node_pos, sv.node_w, 4, bricks=sf.knn_bricks)); print("sample_knn via brick lists %.3f ms" % ms)
ms, _ = t(lambda: sv.set_samples(pos, nrm, nbr=nbr, weights=wts)); print("set_samples    %.3f ms (sort by tuple + gathers)" % ms)
ms, _ = t(lambda: sv._build_pattern()); print("pattern+plan   %.3f ms" % ms)
ms, _ = t(lambda: sf.refresh_samples()); print("refresh total  %.3f ms" % ms)
keys = sv._pattern_keys
ms, _ = t(lambda: sv._build_plan(keys, reg=False)); print("  data plan (+ coverage flag) %.3f ms (rows=%d)" % (ms, sv.n_rows))
ms, _ = t(lambda: sv._build_plan(keys)); print("  data + regulariser plan %.3f ms" % ms)
