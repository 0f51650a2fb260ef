#!/usr/bin/env python3
"""Where does a GN iteration's time go when several ranks share ONE GPU over gloo?  (41 GN-iters/s
at 4 ranks against 1 327 at 2.)  Times, per rank count, the parts of an iteration separately:

    A  all-reduce of the flat normal-equation buffer alone (gloo: device -> host -> ring -> device)
    B  associate + planned build alone (no collective)
    C  PCG, persistent single-launch kernel (every rank spinning in its own grid barrier)
    D  PCG, two launches per iteration (no spinning)
    E  the whole iteration with C        F  the whole iteration with D
    G  B followed by A (device work, then the host-staged collective)      H  B followed by C (no collective at all)
    I  one tiny kernel + a device synchronisation (what every gloo collective on device tensors pays first)

    python tools/diag_ranks.py --ranks 4          (spawns its own ranks; at most 6 processes may use the GPU)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=4)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--nodes", type=int, default=512)
    args = ap.parse_args()
    from dynamicfusion_body_amd import launch
    if args.ranks > 1 and not launch.under_launcher():
        sys.exit(launch.spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.ranks, timeout=500))
    import torch
    import torch.distributed as dist
    from dynamicfusion_body_amd import kernels, scene, _lib
    from dynamicfusion_body_amd import dist as D
    from dynamicfusion_body_amd.pipeline import FrameSolver
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo")
    lib = _lib.load()
    R, N, k = args.res, args.nodes, 4
    H, W, fx, cx, cy = scene.CAMERAS["C2"]
    K = scene.intrinsics(fx, cx, cy)
    Kinv = np.linalg.inv(K)
    scale, center, tdist = scene.grid_params(R)
    T = torch.full((R, R, R), tdist, dtype=torch.float32, device="cuda")
    Wt = torch.zeros_like(T)
    for a in (0.0, 40.0, -40.0):
        lw = scene.view_extrinsic(a)
        d = torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda()
        kernels.integrate_depth(T, Wt, d, K, Kinv, lw, scale, center, tdist)
    fs = FrameSolver(K, scale, center, R / 2, knn=k, pcg_iters=10)
    node_pos, node_w = scene.fibonacci_nodes(N, R)
    ident = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0, 0]), (N, 1))
    fs.set_graph(node_pos, ident, node_w)
    a, b = D.slab_range(R, rank, world)
    S = fs.set_canonical(T[a:b].contiguous(), Wt[a:b].contiguous(), band=4.0, x0=a)
    lw_cam = scene.view_extrinsic(0.0)
    depth = torch.from_numpy(scene.render_depth(K, lw_cam, H, W, dtype=np.float32, sphere_offset=np.array([0.6, -0.4, 0.3]) * scale,
                                                sphere_r=scene.SPHERE_R * 1.02)).cuda()
    sv = fs.solver

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def timeit(fn, reps=args.reps):
        fn(); sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps * 1e3
        sync()
        return D.max_over_ranks([dt])[0]

    def assoc_build_local():
        sv.associate_depth(depth, fs.K, fs.Kinv, lw_cam, scale, center, R / 2, fs.lw, 2.0)
        keep = sv.distributed
        sv.distributed = False
        try:
            sv.build(fs.lw, 5.0, 0.5)
        finally:
            sv.distributed = keep

    def pcg():
        sv.solve_linear(10.0, 1e-2)

    def full():
        fs.gn_iteration(depth, lw_cam, rw=5.0, lm_abs=10.0, lm_rel=1e-2, max_dist=2.0, huber=0.5)

    full(); sync()                                        # pattern, plans
    out = {"ranks": world, "samples_rank0": S, "system_MB": sv.system.numel() * 8 / 1e6}
    out["A_allreduce_ms"] = timeit(lambda: D.allreduce_system(sv.system))
    out["B_assoc_build_ms"] = timeit(assoc_build_local)
    out["G_build_then_allreduce_ms"] = timeit(lambda: (assoc_build_local(), D.allreduce_system(sv.system)))
    out["I_tiny_kernel_plus_sync_ms"] = timeit(lambda: (sv.apply(0.0), torch.cuda.synchronize()))
    lib.dfh_pcg_set_mode(0)
    out["H_build_then_pcg_persistent_no_collective_ms"] = timeit(lambda: (assoc_build_local(), pcg()))
    out["C_pcg_persistent_ms"] = timeit(pcg)
    lib.dfh_pcg_set_mode(2)
    out["D_pcg_multilaunch_ms"] = timeit(pcg)
    lib.dfh_pcg_set_mode(0)
    sv.node_dq.copy_(torch.from_numpy(ident).cuda())
    out["E_iteration_persistent_ms"] = timeit(full)
    lib.dfh_pcg_set_mode(2)
    sv.node_dq.copy_(torch.from_numpy(ident).cuda())
    out["F_iteration_multilaunch_ms"] = timeit(full)
    try:
        sv.check_status()
        out["pcg_timeouts"] = 0
    except _lib.DfhTimeout as e:
        out["pcg_timeouts"] = str(e)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
