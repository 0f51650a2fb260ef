#!/usr/bin/env python3
"""PCG timing on the bench's GN system (256^3 canonical volume, N-node warp field): HIP-event time of
dfh_pcg_solve for several iteration counts -> start-up cost and per-iteration cost."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamicfusion_body_amd import kernels, scene
from dynamicfusion_body_amd.pipeline import FrameSolver

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--nodes", type=int, default=512)
ap.add_argument("--iters", type=int, nargs="+", default=[1, 2, 10, 40])
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
R, N, k = a.res, a.nodes, 4
H, W, fx, cx, cy = scene.CAMERAS["C2" if R <= 256 else "C5"]
K = scene.intrinsics(fx, cx, cy); Kinv = np.linalg.inv(K)
scale, center, tdist = scene.grid_params(R)
T = torch.full((R, R, R), tdist, dtype=torch.float32, device="cuda")
Wt = torch.zeros((R, R, R), dtype=torch.float32, device="cuda")
for ang in (0.0, 40.0, -40.0):
    lw = scene.view_extrinsic(ang)
    d = torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda()
    kernels.integrate_depth(T, Wt, d, K, Kinv, lw, scale, center, tdist)
fs = FrameSolver(K, scale, center, R / 2, knn=k, pcg_iters=10)
node_pos, node_w = scene.fibonacci_nodes(N, R)
ident = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0, 0]), (N, 1))
fs.set_graph(node_pos, ident, node_w)
fs.set_canonical(T, Wt, band=4.0)
lw_cam = scene.view_extrinsic(0.0)
live = scene.render_depth(K, lw_cam, H, W, dtype=np.float32, sphere_offset=np.array([0.6, -0.4, 0.3]) * scale, sphere_r=scene.SPHERE_R * 1.02)
depth = torch.from_numpy(live).cuda()
fs.gn_iteration(depth, lw_cam, rw=0.05, lm_abs=1e-2, lm_rel=1e-2, max_dist=4.0)     # builds pattern + system
sv = fs.solver
vals0 = sv.vals.clone()
print("nodes %d blocks %d mode %s wpb %s" % (N, sv.B, "multilaunch" if os.environ.get("DFH_PCG_MULTILAUNCH") else "persistent", os.environ.get("DFH_PCG_WPB", "16")))
for it in a.iters:
    sv.pcg_iters = it
    nbytes = sv.lib.dfh_pcg_workspace_bytes(N, it)
    sv.pcg_ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device="cuda")
    for _ in range(3):
        sv.vals.copy_(vals0); sv.solve_linear(1e-2, 1e-2)
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(a.reps):
        sv.vals.copy_(vals0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); sv.solve_linear(1e-2, 1e-2); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    print("  pcg_iters %3d: %8.1f us per solve   |dx| %.6e" % (it, tot / a.reps * 1e3, float(sv.dx.norm())))
