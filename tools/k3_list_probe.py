import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dynamicfusion_body_amd import _lib, kernels, scene
from dynamicfusion_body_amd.dq import twist_exp_dq
R, N, k, tdist = 256, 512, 4, 4.0
g = torch.arange(R, device="cuda", dtype=torch.float32)
d = torch.sqrt((g[:, None, None] - R / 2) ** 2 + (g[None, :, None] - R / 2) ** 2 + (g[None, None, :] - R / 2) ** 2)
live = torch.clamp(d - 0.3125 * R + 0.7, -tdist, tdist).contiguous()
T = torch.clamp(d - 0.3125 * R, -tdist, tdist).contiguous(); W = torch.ones_like(T)
rng = np.random.default_rng(0)
node_pos, node_w = scene.fibonacci_nodes(N, R)
dqs = twist_exp_dq(rng.normal(size=(N, 6)) * np.array([.001, .001, .001, .1, .1, .1]))
ident = np.array([1.0, 0, 0, 0, 0, 0, 0, 0])
ws = kernels.dqb_workspace((R, R, R), knn=k, n_nodes=N, level=2)
P = torch.from_numpy(node_pos).cuda(); Q = torch.from_numpy(dqs).cuda(); Wn = torch.from_numpy(node_w).cuda()
kernels.fuse_volume_dqb(T, W, live, P, Q, Wn, k, ident, tdist, workspace=ws, rebuild_candidates=True)
for skip in (0, 1, 3, 0, 1, 3):
    _lib.set_option("k3_skip", skip)
    kernels.fuse_volume_dqb(T, W, live, P, Q, Wn, k, ident, tdist, workspace=ws, rebuild_candidates=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        kernels.fuse_volume_dqb(T, W, live, P, Q, Wn, k, ident, tdist, workspace=ws, rebuild_candidates=False)
    e1.record(); torch.cuda.synchronize()
    tabs = kernels.dqb_skip_tables(ws, (R, R, R), (R, R, R), N)
    print("k3_skip=%d: %.1f us per call; bricks streaming %.3f" % (skip, e0.elapsed_time(e1) * 100, float(tabs["S"].float().mean()) if skip else 0.0))
