#!/usr/bin/env python3
"""Does the composed frame loop follow a known motion?  Mean translation of the nodes facing camera 0 against the true
offset, for a few solver settings.  usage: python tools/track_probe.py --res 256 --nodes 512"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamicfusion_body_amd import scene
from dynamicfusion_body_amd.pipeline import SlabFrame
from dynamicfusion_body_amd.dq import qmul
ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--nodes", type=int, default=512)
ap.add_argument("--frames", type=int, default=16)
ap.add_argument("--views", type=int, default=3)
a = ap.parse_args()
R, N = a.res, a.nodes
H, W, fx, cx, cy = scene.CAMERAS["C2" if R <= 256 else "C5"]
K = scene.intrinsics(fx, cx, cy)
scale, center, tdist = scene.grid_params(R)
node_pos, node_w = scene.fibonacci_nodes(N, R)
lws = [scene.view_extrinsic(x) for x in ((0.0, 40.0, -40.0) if a.views == 3 else [45.0 * v for v in range(a.views)])]
front = node_pos[:, 2] < R / 2 - 0.25 * (scene.SPHERE_R / scale)
amp = np.array([0.8, -0.5, 0.4])
depth_seq = []
for t in range(a.frames):
    off = amp * np.sin(2 * np.pi * (t + 1) / 30.0)
    depth_seq.append((off, [torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0, sphere_offset=off * scale)).cuda() for lw in lws]))
for name, kw, pcg in (("default rw5 lm10 pcg10", {}, 10), ("pcg40", {}, 40), ("rw1 pcg10", {"rw": 1.0}, 10), ("rw1 pcg40", {"rw": 1.0}, 40),
                      ("rw0.2 lm1 pcg20", {"rw": 0.2, "lm_abs": 1.0}, 20)):
    sf = SlabFrame(K, scale, center, R, tdist / scale, node_pos, node_w, knn=4, pcg_iters=pcg, band=2.0, distributed=False)
    for lw in lws:
        sf.integrate(torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda(), lw)
    sf.refresh_samples()
    errs, mags, cnts = [], [], []
    for off, ds in depth_seq:
        cnts.append(sf.step(ds, lws, gn_iters=10, **kw))
        dq = sf.fs.solver.node_dq.cpu().numpy()
        rc = dq[:, :4] * np.array([1.0, -1, -1, -1])
        trans = 2.0 * np.stack([qmul(dq[i, 4:], rc[i]) for i in range(len(dq))])[:, 1:]
        errs.append(np.linalg.norm(trans[front].mean(axis=0) - off)); mags.append(np.linalg.norm(off))
    print("%-24s err/|off| %s   samples %d -> %d" % (name, " ".join("%.2f" % (e / m) for e, m in zip(errs, mags)), cnts[0], cnts[-1]))
