#!/usr/bin/env python3
"""Persistent PCG alone: time per solve against the iteration count (slope = one iteration's hand-off chain, intercept =
prologue: block loads, 6x6 inverse, first SpMV).  System: BASELINE config 3's normal equations (512 nodes, ~5 900 blocks).
usage: python tools/kbench_pcg.py [nodes]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamicfusion_body_amd import scene, kernels
from dynamicfusion_body_amd.pipeline import FrameSolver
R = 256
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
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
fs.gn_iteration(depth, lw_cam, rw=5.0, lm_abs=10.0, lm_rel=1e-2, max_dist=2.0, huber=0.5)   # builds vals / rhs (and damps the diagonal once)
torch.cuda.synchronize()
print("nodes", sv.N, "blocks", sv.B)
reps = 40
res = []
for iters in (1, 2, 5, 10, 20, 40):
    sv.pcg_iters = iters
    nbytes = sv.lib.dfh_pcg_workspace_bytes(sv.N, iters)
    if sv.pcg_ws.numel() * 8 < nbytes:
        sv.pcg_ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device="cuda")
    for _ in range(5):
        sv.solve_linear(0.0, 0.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        sv.solve_linear(0.0, 0.0)
    e1.record()
    torch.cuda.synchronize()
    sv.check_status()
    us = e0.elapsed_time(e1) * 1e3 / reps
    res.append((iters, us))
    dies = int(sv.pcg_ws.view(torch.int32)[2 * (66 * sv.N + 3 * (iters + 1)) + 3])      # (option pcg_one_xcd: bit = XCC_ID of a die that took part)
    print("iters %3d: %.1f us per solve (memset + kernel), |x| %.6e%s" % (iters, us, float(sv.dx.norm()), "  dies 0x%x" % dies if dies else ""))
(i0, t0), (i1, t1) = res[2], res[-1]
print("slope %.2f us per iteration, intercept %.1f us" % ((t1 - t0) / (i1 - i0), t0 - i0 * (t1 - t0) / (i1 - i0)))
if hasattr(sv.lib, "dfh_debug_pcg_trace") or os.environ.get("DFH_LIB_PATH"):
    import ctypes
    try:
        fn = sv.lib.dfh_debug_pcg_trace
    except AttributeError:
        sys.exit(0)
    sv.pcg_iters = 10
    sv.solve_linear(0.0, 0.0)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (64 * 16 * 12))()
    fn.restype = ctypes.c_int
    assert fn(buf) == 0
    tr = np.frombuffer(buf, dtype=np.uint64).reshape(64, 16, 12).astype(np.int64)
    print("shader clock during the solve: %.0f MHz (s_memtime ticks per s_memrealtime tick x 100 MHz)" % (
        100.0 * (tr[0, 8, 9] - tr[0, 1, 9]) / (tr[0, 8, 0] - tr[0, 1, 0])))
    pro = tr[:, 15, :7]
    ok = pro[:, 6] > 0
    d = np.diff(pro[ok], axis=1) / 100.0
    print("prologue, one wave per workgroup, median / max us: block loads %.2f / %.2f | diagonal block to registers %.2f / %.2f | 6x6 inverse %.2f / %.2f | "
          "M^-1 r + publish %.2f / %.2f | first hand-off %.2f / %.2f | first SpMV %.2f / %.2f ; kernel start to loop %.2f" % (
              np.median(d[:, 0]), d[:, 0].max(), np.median(d[:, 1]), d[:, 1].max(), np.median(d[:, 2]), d[:, 2].max(), np.median(d[:, 3]), d[:, 3].max(),
              np.median(d[:, 4]), d[:, 4].max(), np.median(d[:, 5]), d[:, 5].max(), (pro[ok][:, 6].max() - pro[ok][:, 0].min()) / 100.0))
    it = 4
    print("iteration %d, one wave of every workgroup, us: start skew | to LDS barrier 1 | publish+request | wait | update | await | spmv+publish | iteration ; xcc se cu" % it)
    t00 = tr[:, it, 0].min()
    for b in range(64):
        t = tr[b, it]
        if t[0] == 0:
            continue
        hw = int(t[10]) & 0xffffffff
        xcc = (int(t[10]) >> 32) & 0xf
        print("  wg %2d: %+5.2f | %5.2f %5.2f %5.2f | %5.2f %5.2f %5.2f | %5.2f ; xcc %d se %d cu %2d simd %d" % (
            b, (t[0] - t00) / 100, (t[5] - t[0]) / 100, (t[6] - t[5]) / 100, (t[1] - t[6]) / 100, (t[2] - t[1]) / 100, (t[3] - t[2]) / 100,
            (t[4] - t[3]) / 100, (tr[b, it + 1, 0] - t[0]) / 100, xcc, (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3))
