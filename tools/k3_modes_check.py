#!/usr/bin/env python3
"""K3 on float32 volumes: search every call (mode 0) against search + store (1) and stored (3), LDS / plain kernels: same bits?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamicfusion_body_amd import kernels, scene, _lib
from dynamicfusion_body_amd.dq import twist_exp_dq
R, N, k = int(sys.argv[1]) if len(sys.argv) > 1 else 128, 300, 4
tdist = 4.0
g = torch.arange(R, device="cuda", dtype=torch.float32)
d = torch.sqrt((g[:, None, None] - R / 2) ** 2 + (g[None, :, None] - R / 2) ** 2 + (g[None, None, :] - R / 2) ** 2)
live = torch.clamp(d - 0.3125 * R + 0.7, -1.5 * tdist, 1.5 * tdist).contiguous()
T0 = torch.clamp(d - 0.3125 * R, -tdist, tdist).contiguous(); W0 = torch.ones_like(T0)
rng = np.random.default_rng(0)
node_pos, node_w = scene.fibonacci_nodes(N, R)
for name, dqs, lw in (("random", twist_exp_dq(rng.normal(size=(N, 6)) * np.array([.01, .01, .01, .3, .3, .3])), twist_exp_dq(np.array([0.01, -0.02, 0.015, 0.3, -0.2, 0.1]))),
                      ("identity", np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0, 0]), (N, 1)), np.array([1.0, 0, 0, 0, 0, 0, 0, 0]))):
    def run(ws, rebuild, opts=()):
        for o in opts: _lib.set_option(o, 1)
        T, W = T0.clone(), W0.clone()
        kernels.fuse_volume_dqb(T, W, live, node_pos, dqs, node_w, k, lw, tdist, workspace=ws, rebuild_candidates=rebuild)
        torch.cuda.synchronize()
        for o in opts: _lib.set_option(o, None)
        return T, W
    ref = run(kernels.dqb_workspace((R, R, R)), True)
    ws2 = kernels.dqb_workspace((R, R, R), knn=k, n_nodes=N, level=2)
    outs = {"mode1": run(ws2, True), "mode3 lds": run(ws2, False), "mode3 lds again": run(ws2, False), "mode3 plain": run(ws2, False, ("k3_no_lds",)),
            "exact": run(kernels.dqb_workspace((R, R, R)), True, ("k3_exact",))}
    for n, (T, W) in outs.items():
        bad = (T != ref[0]) | (W != ref[1])
        print("%-9s %-16s mismatching voxels %8d of %d   max|dT| %.3g  max|dW| %.3g   mask differs at %d" %
              (name, n, int(bad.sum()), bad.numel(), float((T - ref[0]).abs().max()), float((W - ref[1]).abs().max()),
               int(((T != T0) != (ref[0] != T0)).sum())))
