#!/usr/bin/env python3
"""Hop-by-hop wall-clock trace of the planned build's gather (block waves), BASELINE config 3's system.
Needs a library built with -DDFH_GATHER_TRACE (tools/build_variant.sh gtrace -DDFH_GATHER_TRACE; DFH_LIB_PATH=...)."""
import os, sys, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamicfusion_body_amd import scene, kernels
from dynamicfusion_body_amd.pipeline import FrameSolver
R, N = 256, 512
H, W, fx, cx, cy = scene.CAMERAS["C2"]
K = scene.intrinsics(fx, cx, cy)
Kinv = np.linalg.inv(K)
scale, center, tdist = scene.grid_params(R)
T = torch.full((R, R, R), tdist, dtype=torch.float32, device="cuda")
Wt = torch.zeros((R, R, R), dtype=torch.float32, device="cuda")
for a in (0.0, 40.0, -40.0):
    lw = scene.view_extrinsic(a)
    d = torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda()
    kernels.integrate_depth(T, Wt, d, K, Kinv, lw, scale, center, tdist)
fs = FrameSolver(K, scale, center, R / 2, knn=4, pcg_iters=10)
node_pos, node_w = scene.fibonacci_nodes(N, R)
ident = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0, 0]), (N, 1))
fs.set_graph(node_pos, ident, node_w)
fs.set_canonical(T, Wt, band=4.0, x0=0)
lw_cam = scene.view_extrinsic(0.0)
live = scene.render_depth(K, lw_cam, H, W, dtype=np.float32, sphere_offset=np.array([0.6, -0.4, 0.3]) * scale, sphere_r=scene.SPHERE_R * 1.02)
depth = torch.from_numpy(live).cuda()
sv = fs.solver
for _ in range(4):
    fs.gn_iteration(depth, lw_cam, rw=5.0, lm_abs=10.0, lm_rel=1e-2, max_dist=2.0, huber=0.5)
torch.cuda.synchronize()
print("samples", sv.S, "rows", sv.n_rows, "blocks", sv.B)
if hasattr(sv.lib, "dfh_debug_build_trace"):
    fn = sv.lib.dfh_debug_build_trace
    fn.restype = ctypes.c_int
    buf = (ctypes.c_ulonglong * (8192 * 8))()
    assert fn(buf) == 0
    nt = (sv.S + 127) // 128
    tr = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.int64)[:nt]
    tr = tr[tr[:, 5] > 0]                      # tiles with valid samples (the others leave early)
    t0 = tr[:, 0].min()
    st = (tr[:, :6] - t0) / 100.0
    print("build: %d tiles (%d with valid samples), span %.2f us" % (nt, len(tr), st[:, 5].max()))
    print("tile start: median %.2f max %.2f" % (np.median(st[:, 0]), st[:, 0].max()))
    d = np.diff(st, axis=1)
    for k, name in enumerate(["association (blend, warp, project)", "Jacobian rows -> LDS", "cost + run boundaries", "row counts + live flags", "Gram reduction + stores"]):
        print("  %-36s median %.2f  p90 %.2f  max %.2f us" % (name, np.median(d[:, k]), np.percentile(d[:, k], 90), d[:, k].max()))
    life = st[:, 5] - st[:, 0]
    nv = tr[:, 6] & 0xffff; nr = tr[:, 6] >> 16
    print("  tile life: median %.2f p90 %.2f max %.2f; valid/tile mean %.1f, runs/tile mean %.1f max %d" % (np.median(life), np.percentile(life, 90), life.max(), nv.mean(), nr.mean(), nr.max()))
    ts = np.linspace(0, st[:, 5].max(), 10)
    print("  tiles alive at t:", [(round(float(t), 1), int(((st[:, 0] <= t) & (st[:, 5] > t)).sum())) for t in ts])
if not hasattr(sv.lib, "dfh_debug_gather_trace"):
    sys.exit(0)
fn = sv.lib.dfh_debug_gather_trace
fn.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * (8192 * 8))()
assert fn(buf) == 0
tr = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.int64)[: sv.B]
t0 = tr[:, 0].min()
st = (tr[:, :5] - t0) / 100.0
print("kernel span (first start .. last end): %.2f us" % st[:, 4].max())
print("wave start: min %.2f median %.2f max %.2f" % (st[:, 0].min(), np.median(st[:, 0]), st[:, 0].max()))
d = np.diff(st, axis=1)
for k, name in enumerate(["hop 1 list bounds", "hop 2 entries", "hop 3 flags + compaction", "hop 4 values"]):
    print("%-26s median %.2f  p90 %.2f  max %.2f us" % (name, np.median(d[:, k]), np.percentile(d[:, k], 90), d[:, k].max()))
tot = st[:, 4] - st[:, 0]
print("wave life: median %.2f p90 %.2f max %.2f" % (np.median(tot), np.percentile(tot, 90), tot.max()))
L = tr[:, 5] & 0xffff; nl = (tr[:, 5] >> 16) & 0xffff; rl = (tr[:, 5] >> 32) & 0xffff; rnl = (tr[:, 5] >> 48) & 0xffff
print("list length: mean %.1f max %d; live: mean %.1f max %d; reg list mean %.1f live %.1f" % (L.mean(), L.max(), nl.mean(), nl.max(), rl.mean(), rnl.mean()))
order = np.argsort(-tot)[:8]
for b in order:
    print("  block %5d: start %.2f hops %s  list %d live %d" % (b, st[b, 0], np.round(d[b], 2), L[b], nl[b]))
# concurrency: number of waves alive over time
ts = np.linspace(0, st[:, 4].max(), 12)
print("waves alive at t:", [(round(float(t), 1), int(((st[:, 0] <= t) & (st[:, 4] > t)).sum())) for t in ts])
