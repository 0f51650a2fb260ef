#!/usr/bin/env python3
"""Soak test of the persistent PCG / planned build: many composed frames and many solves; any barrier time-out
would show up as NaN in the warp field (the kernels never hang: bounded spins)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamicfusion_body_amd import scene
from dynamicfusion_body_amd.pipeline import SlabFrame
R, N = 256, 512
H, W, fx, cx, cy = scene.CAMERAS["C2"]
K = scene.intrinsics(fx, cx, cy)
scale, center, tdist = scene.grid_params(R)
node_pos, node_w = scene.fibonacci_nodes(N, R)
sf = SlabFrame(K, scale, center, R, tdist / scale, node_pos, node_w, knn=4, pcg_iters=10, band=2.0)
for ang in (0.0, 40.0, -40.0):
    lw = scene.view_extrinsic(ang)
    sf.integrate(torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda(), lw)
sf.refresh_samples()
lw_cam = scene.view_extrinsic(0.0)
lws = [scene.view_extrinsic(a) for a in (0.0, 40.0, -40.0)]
depths = [[torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0, sphere_offset=np.array([0.5, -0.3, 0.2]) * np.sin(2 * np.pi * f / 21) * scale,
                                               sphere_r=scene.SPHERE_R * (1.0 + 0.003 * np.cos(2 * np.pi * f / 21)))).cuda() for lw in lws] for f in range(21)]

t0 = time.perf_counter()
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
solves = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
for f in range(frames):
    sf.step(depths[f % 21], lws, gn_iters=10, huber=float(os.environ.get('HUBER', '0.5')), max_dist=float(os.environ.get('GATE', '2')),
            data_views=int(os.environ['VIEWS']) if os.environ.get('VIEWS') else None,      # VIEWS=1: the data term on the first view only
            relax=float(os.environ['RELAX']) if os.environ.get('RELAX') else None)
    c_, n_ = sf.fs.solver.cost()                      # (data + regulariser objective at the frame's last build, valid samples)
    hist = globals().setdefault("hist", [])
    hist.append((c_ / max(n_, 1), n_))
    if f % 50 == 49:
        h = np.array(hist[-50:])
        print("   last 50 frames: objective per valid sample %.4f (mean), valid samples %.0f (mean)" % (h[:, 0].mean(), h[:, 1].mean()), flush=True)
        dq = sf.fs.solver.node_dq
        assert torch.isfinite(dq).all(), "non-finite warp field at frame %d" % f
        d = dq.cpu().numpy()
        tr, rot = 2 * np.linalg.norm(d[:, 4:], axis=1), np.linalg.norm(d[:, 1:4], axis=1)
        # nodes that blend into at least one sample (the surface the cameras see) against the others (the unobserved back of the
        # object: no data row ever touches them, only the regulariser ties them to their neighbours)
        sv_ = sf.fs.solver
        sup = torch.bincount(sv_.snbr.view(-1).long(), minlength=N).cpu().numpy() > 0
        print("frame %d, %.1f s: max translation %.3f voxel over the %d nodes that blend into samples (median %.3f, 99th percentile %.3f), %.3f over the %d "
              "that do not; max rotation %.4f, samples %d" % (f + 1, time.perf_counter() - t0, tr[sup].max(), sup.sum(), np.median(tr[sup]),
                                                             np.percentile(tr[sup], 99), tr[~sup].max() if (~sup).any() else 0.0, (~sup).sum(),
                                                             rot.max(), sv_.S), flush=True)
        assert tr.max() < 3.0, "the warp field drifts: max translation %.2f voxel over all nodes at frame %d (scene moves +-0.6)" % (tr.max(), f + 1)
sv = sf.fs.solver
v0 = sv.vals.clone()
x_ref = None
for i in range(solves):
    sv.vals.copy_(v0)
    sv.solve_linear(1e-2, 1e-2)
    if i % 1000 == 999:
        x = sv.dx.clone()
        assert torch.isfinite(x).all()
        assert x_ref is None or torch.equal(x, x_ref), "solve not bit-reproducible"
        x_ref = x
        print("solve %d ok (bit-identical so far)" % (i + 1), flush=True)
print("soak ok")
