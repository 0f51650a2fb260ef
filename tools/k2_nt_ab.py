import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dynamicfusion_body_amd import _lib, kernels
from dynamicfusion_body_amd.dq import twist_exp_dq
for R in (256, 512):
    g = torch.arange(R, device="cuda", dtype=torch.float32)
    d = torch.sqrt((g[:, None, None] - R / 2) ** 2 + (g[None, :, None] - R / 2) ** 2 + (g[None, None, :] - R / 2) ** 2)
    live = torch.clamp(d - 0.3125 * R + 0.7, -4.0, 4.0).contiguous()
    T0 = torch.clamp(d - 0.3125 * R, -4.0, 4.0).contiguous(); W0 = torch.ones_like(T0)
    lw = twist_exp_dq(np.array([0.01, -0.02, 0.015, 0.3, -0.2, 0.1]))
    outs = {}
    for nt in (0, 1):
        _lib.set_option("k2_nt", nt)
        T, W = T0.clone(), W0.clone()
        kernels.fuse_volume_rigid(T, W, live, lw, 4.0)
        outs[nt] = (T.clone(), W.clone())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): kernels.fuse_volume_rigid(T, W, live, lw, 4.0)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("K2 %d^3 nt=%d: %.1f us = %.0f GB/s algorithmic" % (R, nt, ms * 1e3, 20.0 * R ** 3 / ms / 1e6))
    print("same bits:", torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]))
