#!/usr/bin/env python3
"""How much of a known translation the solve recovers in 10 GN iterations, over solver settings (the metric of
tests/test_gpu_configs.py::test_solve_recovers_a_known_translation), at R^3 with N nodes.  usage: recovery_probe.py [R] [N]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamicfusion_body_amd import kernels, scene
from dynamicfusion_body_amd.pipeline import FrameSolver
from dynamicfusion_body_amd.solve import warp_points
R = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
H, W, fx, cx, cy = scene.CAMERAS["C2"]
K = scene.intrinsics(fx, cx, cy); Kinv = np.linalg.inv(K)
scale, center, tdist = scene.grid_params(R)
T = torch.full((R, R, R), tdist, dtype=torch.float32, device="cuda"); Wt = torch.zeros_like(T)
angles = (0.0, 40.0, -40.0)
lws = [scene.view_extrinsic(a) for a in angles]
for lw in lws:
    d = torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda()
    kernels.integrate_depth(T, Wt, d, K, Kinv, lw, scale, center, tdist)
truth = np.array([0.6, -0.4, 0.3])
depths = [torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0, sphere_offset=truth * scale)).cuda() for lw in lws]
node_pos, node_w = scene.fibonacci_nodes(N, R)
ident = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0, 0]), (N, 1))
tt = torch.from_numpy(truth).cuda()
for pcg in (10, 20, 40):
    for lm_abs in (10.0, 3.0, 1.0, 0.3):
        for rw in (5.0, 2.0):
            fs = FrameSolver(K, scale, center, R / 2, knn=4, pcg_iters=pcg, distributed=False)
            fs.set_graph(node_pos, ident, node_w)
            fs.set_canonical(T, Wt, band=2.0)
            sv = fs.solver
            out = []
            for it in range(10):
                fs.gn_iteration(depths, lws, rw=rw, lm_abs=lm_abs, lm_rel=1e-2, max_dist=2.0, huber=0.5)
                if it in (4, 9):
                    wp, _ = warp_points(sv.spos, None, fs.lw, nbr=sv.snbr, node_dq=sv.node_dq, node_pos=sv.node_pos, node_w=sv.node_w)
                    disp = wp - sv.spos
                    tn = sv.snrm @ tt
                    sel = (sv.valid > 0) & (tn.abs() >= 0.2)
                    out.append(float(((disp[sel] * sv.snrm[sel]).sum(dim=1) / tn[sel]).mean()))
            c, n = sv.cost()
            print("pcg %2d lm_abs %4.1f rw %3.1f: normal share after 5 / 10 iterations %.3f / %.3f, objective %.1f on %d valid" % (pcg, lm_abs, rw, out[0], out[1], c, n), flush=True)
