#!/usr/bin/env python3
"""Prototype: a GLOBAL rigid step (all nodes share one twist: the 6x6 system is the sum of all blocks / all J^T r) before the
node iterations.  usage: recovery_probe2.py [R] [N]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamicfusion_body_amd import _lib, kernels, scene
from dynamicfusion_body_amd.device import current_stream_ptr
from dynamicfusion_body_amd.pipeline import FrameSolver
from dynamicfusion_body_amd.solve import warp_points
R = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
H, W, fx, cx, cy = scene.CAMERAS["C2"]
K = scene.intrinsics(fx, cx, cy); Kinv = np.linalg.inv(K)
scale, center, tdist = scene.grid_params(R)
T = torch.full((R, R, R), tdist, dtype=torch.float32, device="cuda"); Wt = torch.zeros_like(T)
lws = [scene.view_extrinsic(a) for a in (0.0, 40.0, -40.0)]
for lw in lws:
    d = torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda()
    kernels.integrate_depth(T, Wt, d, K, Kinv, lw, scale, center, tdist)
node_pos, node_w = scene.fibonacci_nodes(N, R)
ident = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0, 0]), (N, 1))
lib = _lib.load()
for name, truth, infl in (("translation", np.array([0.6, -0.4, 0.3]), 1.0), ("translation + 1 % inflation", np.array([0.6, -0.4, 0.3]), 1.01),
                          ("inflation only 1 %", np.zeros(3), 1.01)):
    depths = [torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0, sphere_offset=truth * scale,
                                                  sphere_r=scene.SPHERE_R * infl)).cuda() for lw in lws]
    for G in (0, 1, 2, 3):
      for LM in ((0.0,) if G == 0 else (1e-3, 0.1)):
          fs = FrameSolver(K, scale, center, R / 2, knn=4, pcg_iters=10, distributed=False)
          fs.set_graph(node_pos, ident, node_w)
          fs.set_canonical(T, Wt, band=2.0)
          sv = fs.solver
          for g in range(G):
              sv.build_associated(depths, K, Kinv, lws, scale, center, R / 2, fs.lw, 5.0, 2.0, 0.5)
              A = sv.vals.view(-1, 6, 6).sum(dim=0)
              b = sv.rhs.view(-1, 6).sum(dim=0)
              xi = -torch.linalg.solve(A + LM * torch.diag(A.diagonal()), b)
              cg, ng = sv.cost()
              print("      global step %d: objective %.1f on %d valid, xi = %s" % (g, cg, ng, " ".join("%.4g" % v for v in xi.tolist())), flush=True)
              xr = xi.repeat(N).contiguous()
              _lib.check(lib.dfh_apply_twist(sv.node_dq.data_ptr(), xr.data_ptr(), N, 1.0, current_stream_ptr()), "apply")
          costs = []
          for it in range(10):
              fs.gn_iteration(depths, lws, rw=5.0, lm_abs=10.0, lm_rel=1e-2, max_dist=2.0, huber=0.5)
          c, n = sv.cost()
          wp, _ = warp_points(sv.spos, None, fs.lw, nbr=sv.snbr, node_dq=sv.node_dq, node_pos=sv.node_pos, node_w=sv.node_w)
          disp = wp - sv.spos
          cpos = sv.spos - R / 2
          true_disp = torch.from_numpy(truth).cuda() + (infl - 1.0) * cpos          # (the sphere is centred on the grid)
          tn = (sv.snrm * true_disp).sum(dim=1)
          sel = (sv.valid > 0) & (tn.abs() >= 0.2)
          share = float(((disp[sel] * sv.snrm[sel]).sum(dim=1) / tn[sel]).mean())
          left = float((((disp[sel] - true_disp[sel]) * sv.snrm[sel]).sum(dim=1) ** 2).mean().sqrt())
          print("%-28s global steps %d: normal share %.3f, rms point-to-plane left %.3f voxel, objective per valid %.4f (%d valid)" % (name + " lm %g" % LM, G, share, left, c / max(n, 1), n), flush=True)
