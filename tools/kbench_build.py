#!/usr/bin/env python3
"""HIP-event time of one dfh_gn_build (data rows + regulariser) and one associate at the frame leg's
sample count (256^3, band 4 -> ~1 M samples, 512 nodes)."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--nodes", type=int, default=512)
ap.add_argument("--band", type=float, default=4.0)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--lib", default=None)
a = ap.parse_args()
if a.lib:
    from dynamicfusion_body_amd import _lib
    _lib.LIB_PATH = os.path.abspath(a.lib)
import torch
from dynamicfusion_body_amd import scene
from dynamicfusion_body_amd.pipeline import SlabFrame
R = a.res
H, W, fx, cx, cy = scene.CAMERAS["C2" if R <= 256 else "C5"]
K = scene.intrinsics(fx, cx, cy)
scale, center, tdist = scene.grid_params(R)
node_pos, node_w = scene.fibonacci_nodes(a.nodes, R)
sf = SlabFrame(K, scale, center, R, tdist / scale, node_pos, node_w, knn=4, pcg_iters=10, band=a.band)
for ang in (0.0, 40.0, -40.0):
    lw = scene.view_extrinsic(ang)
    sf.integrate(torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda(), lw)
S = sf.refresh_samples()
lw_cam = scene.view_extrinsic(0.0)
d = torch.from_numpy(scene.render_depth(K, lw_cam, H, W, dtype=np.float32, sphere_offset=np.array([0.1, -0.07, 0.05]) * scale, sphere_r=scene.SPHERE_R * 1.004)).cuda()
sf.step(d, lw_cam, gn_iters=2)
sv = sf.fs.solver
fs = sf.fs
def timeit(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
t_as = timeit(lambda: sv.associate_depth(d, fs.K, fs.Kinv, lw_cam, fs.scale, fs.center, fs.half, fs.lw, 4.0), a.reps)
t_b = timeit(lambda: sv.build(fs.lw, 0.05), a.reps)
t_b0 = timeit(lambda: sv.build(fs.lw, 0.0), a.reps)
c, n = sv.cost()
print("samples %d valid %d blocks %d: associate %.1f us, build %.1f us (data only %.1f us)" % (sv.S, n, sv.B, t_as, t_b, t_b0))
nb = sv.snbr.view(sv.S, -1)
v = sv.valid.bool()
nbv = nb[v]
chg = (nbv[1:] != nbv[:-1]).any(dim=1)
tile = (torch.nonzero(v).flatten() // 256)
runs_total = int(chg.sum()) + 1 + int((tile[1:] != tile[:-1]).sum())
ntiles = (sv.S + 255) // 256
print("tiles %d, valid per tile %.1f, runs (valid samples) total ~%d = %.1f per tile, distinct tuples %d" % (ntiles, float(v.sum()) / ntiles, runs_total, runs_total / ntiles, torch.unique(nbv, dim=0).shape[0]))
bp = sv.blk_ptr.long(); ll = bp[1:] - bp[:-1]
npn = sv.node_ptr.long(); nl = npn[1:] - npn[:-1]
print("plan: rows %d, block-list entries %d (max %d, mean %.1f, >256: %d), node-list max %d mean %.1f" % (sv.n_rows, int(ll.sum()), int(ll.max()), float(ll.float().mean()), int((ll > 256).sum()), int(nl.max()), float(nl.float().mean())))
