#!/usr/bin/env python3
"""The constant-live skip of K3 (csrc/dfh_fuse_volume.hip) on the bench's frame scene: what the live volume's cells hold, how
many 64-voxel runs still take the warp kernel, the per-brick displacement bounds against the displacements the warp really
produces (every voxel, the reference's chain through dfh_warp_points), skip on / off equality and timing.
Usage: python tools/k3_skip_probe.py [--res 256] [--nodes 512] [--frames 3] [--check-bound]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dynamicfusion_body_amd import _lib, kernels, scene                  # noqa: E402
from dynamicfusion_body_amd.pipeline import SlabFrame                   # noqa: E402
from dynamicfusion_body_amd.solve import sample_knn, warp_points        # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--nodes", type=int, default=512)
ap.add_argument("--frames", type=int, default=3)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--check-bound", action="store_true")
a = ap.parse_args()
R, N = a.res, a.nodes
H, W, fx, cx, cy = scene.CAMERAS["C2" if R <= 256 else "C5"]
K = scene.intrinsics(fx, cx, cy)
scale, center, tdist = scene.grid_params(R)
tvox = tdist / scale
node_pos, node_w = scene.fibonacci_nodes(N, R)
angles = (0.0, 40.0, -40.0) if R <= 256 else tuple(45.0 * v for v in range(8))
lws = [scene.view_extrinsic(x) for x in angles]
sf = SlabFrame(K, scale, center, R, tvox, node_pos, node_w, knn=4, pcg_iters=10, band=4.0, distributed=False)
for lw in lws:
    sf.integrate(torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda(), lw)
sf.refresh_samples()
for f in range(a.frames):
    off = np.array([0.10, -0.07, 0.05]) * (f + 1) * scale
    depths = [torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0, sphere_offset=off,
                                                  sphere_r=scene.SPHERE_R * (1.0 + 0.004 * (f + 1)))).cuda() for lw in lws]
    sf.step(depths, lws, gn_iters=10)
torch.cuda.synchronize()
live = sf.live
print("tvox = %r; live volume: min %.7g max %.7g" % (tvox, float(live.min()), float(live.max())))
c = live.view(R // 4, 4, R // 4, 4, R // 4, 4).permute(0, 2, 4, 1, 3, 5).reshape(-1, 64)
cmin, cmax = c.min(dim=1).values, c.max(dim=1).values
uni = cmin == cmax
vals, cnts = torch.unique(cmin[uni], return_counts=True)
order = torch.argsort(cnts, descending=True)[:6]
print("cells: %d, uniform %.4f; values of uniform cells (value: share of all cells): %s" %
      (c.shape[0], float(uni.float().mean()), ", ".join("%.9g: %.4f" % (float(vals[i]), float(cnts[i]) / c.shape[0]) for i in order)))
print("cells with all voxels >= tvox: %.4f; == tvox: %.4f" % (float((cmin >= tvox).float().mean()), float(((cmin == tvox) & uni).float().mean())))

sv = sf.fs.solver
dq = sv.node_dq.clone()
tr = 2.0 * torch.stack([dq[:, 4:] .norm(dim=1)]).flatten()
print("node DQs: |r - 1| max %.3g median %.3g; 2|d| max %.3g median %.3g" % (float((dq[:, :4] - torch.tensor([1.0, 0, 0, 0], device="cuda", dtype=torch.float64)).norm(dim=1).max()),
      float((dq[:, :4] - torch.tensor([1.0, 0, 0, 0], device="cuda", dtype=torch.float64)).norm(dim=1).median()), float(tr.max()), float(tr.median())))


def k3(T, Wt, skip, all_bounds=False):
    _lib.set_option("k3_skip", 3 if skip == "list" else ((2 if all_bounds else 1) if skip else 0))
    kernels.fuse_volume_dqb(T, Wt, live, sv.node_pos, dq, sv.node_w, 4, sf.ident_lw, tvox, res=(R, R, R), x_range=(0, R), workspace=sf.ws_dqb,
                            rebuild_candidates=False)
    _lib.set_option("k3_skip", None)


T0, W0 = sf.T.clone(), sf.Wt.clone()
Ta, Wa = T0.clone(), W0.clone()
Tb, Wb = T0.clone(), W0.clone()
k3(Ta, Wa, True, all_bounds=a.check_bound)
tabs = {kk: (v.clone() if isinstance(v, torch.Tensor) else v) for kk, v in kernels.dqb_skip_tables(sf.ws_dqb, (R, R, R), (R, R, R), N).items()}
k3(Tb, Wb, False)
torch.cuda.synchronize()
print("skip on == skip off: T %s, w %s; voxels changed by the call %d" % (bool(torch.equal(Ta, Tb)), bool(torch.equal(Wa, Wb)), int((Ta != T0).sum())))
b = tabs["bound"]
fin = torch.isfinite(b) & (b >= 0)
reach = tabs["reach"]
print("skip admitted %s; bricks %d: bound finite %.4f, min %.3g median %.3g max %.3g; reach 1: %.4f, 2: %.4f, none: %.4f" %
      (tabs["ok"], b.numel(), float(fin.float().mean()), float(b[fin].min()), float(b[fin].median()), float(b[fin].max()),
       float((reach == 1).float().mean()), float((reach == 2).float().mean()), float((reach == 255).float().mean())))
used = tabs["used"].int() & 0xffff
nu = (used < 0xfffe).sum(dim=1)
print("nodes a brick's voxels blend: mean %.2f, max %d; bricks with more than 16: %d" % (float(nu.float().mean()), int(nu.max()), int((used[:, 0] == 0xfffe).sum())))
pop = lambda t: int(sum(bin(int(v) & (2 ** 64 - 1)).count("1") for v in t.flatten().tolist()))
print("live cells set in U: %.4f; bricks taking the constant-live stream: %.4f; 16-voxel rows listed for the warp kernel: %d of %d = %.4f" %
      (pop(tabs["U"]) / (R // 4) ** 3, float(tabs["S"].float().mean()), tabs["n_listed"], tabs["n_runs"], tabs["n_listed"] / tabs["n_runs"]))

# what a uniform reach of 1 / 2 / 3 cells would leave for the warp kernel (the limit a tighter bound could approach)
Ub = ((cmin == tvox) & uni).view(R // 4, R // 4, R // 4)
for reach in (1, 2, 3):
    pad = torch.zeros((R // 4 + 2 * reach,) * 3, dtype=torch.float32, device="cuda")
    pad[reach:-reach, reach:-reach, reach:-reach] = Ub.float()
    er = -torch.nn.functional.max_pool3d(-pad[None, None], 2 * reach + 1, stride=1)[0, 0] > 0.5
    runs = er.view(R // 4, R // 4, R // 64, 16).all(dim=3)
    print("uniform reach %d: cells safe %.4f, runs safe %.4f (quarter runs: %.4f)" % (reach, float(er.float().mean()), float(runs.float().mean()),
          float(er.view(R // 4, R // 4, R // 16, 4).all(dim=3).float().mean())))

e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for skip in (True, False, "list", True, False, "list"):
    T, Wt = T0.clone(), W0.clone()
    k3(T, Wt, skip)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.reps):
        k3(T, Wt, skip)
    e1.record()
    torch.cuda.synchronize()
    print("K3 steady state, skip %-4s: %.1f us per call" % (skip if skip == "list" else ("on" if skip else "off"), e0.elapsed_time(e1) / a.reps * 1e3))

if a.check_bound:
    worst = 0.0
    viol = 0
    alld = []
    bb = b.view(R // 4, R // 4, R // 16)
    for x0 in range(0, R, 32):
        g = torch.stack(torch.meshgrid(torch.arange(x0, x0 + 32, device="cuda", dtype=torch.float64), torch.arange(R, device="cuda", dtype=torch.float64),
                                       torch.arange(R, device="cuda", dtype=torch.float64), indexing="ij"), dim=-1).reshape(-1, 3).contiguous()
        nbr, _ = sample_knn(g, sv.node_pos, sv.node_w, 4)
        wp, _ = warp_points(g, None, sf.ident_lw, nbr=nbr, node_dq=dq, node_pos=sv.node_pos, node_w=sv.node_w)
        d = (wp - g).norm(dim=1).view(8, 4, R // 4, 4, R // 16, 16).amax(dim=(1, 3, 5))
        lim = bb[x0 // 4:x0 // 4 + 8].double()
        viol += int((d > lim).sum())
        ratio = (d / lim)[torch.isfinite(lim)]
        worst = max(worst, float(ratio.max()))
        alld.append(d.flatten())
    print("bound check over all %d^3 voxels: bricks where a voxel moves further than the bound: %d; largest displacement / bound %.4f" % (R, viol, worst))
    alld = torch.cat(alld)
    qs = torch.quantile(alld, torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], device="cuda", dtype=torch.float64))
    print("largest displacement of a brick's voxels: median %.3g, 90 %% %.3g, 99 %% %.3g, 99.9 %% %.3g, max %.3g voxel" % tuple(float(v) for v in qs))
    print("bricks whose voxels all move less than 3 voxels (reach 1 would do): %.4f; less than 7: %.4f" % (float((alld < 3).float().mean()), float((alld < 7).float().mean())))
