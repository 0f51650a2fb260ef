"""The sharded solve's collective path on ONE GPU: a 1-rank RCCL ("nccl") group, the GN iteration of BASELINE config 3 with its
pack -> all-reduce -> unpack between build and solve (WarpSolver.force_collective), eager and captured into a hipGraph with the
collective inside, against the single-GPU one-call iteration.  What this can show without a multi-GPU node: that a live RCCL
communicator's all-reduce captures and replays with the solve's kernels around it, that the packed upper triangle round-trips to
the same bits, and what the two-call iteration + a collective launch cost on the host and on the device when the collective moves
nothing over a link (a lower bound of the model's `allreduce` and `eager` terms, dist.solve_mode).
Prints one JSON line.  python3 tools/rccl_capture_check.py [--res 256] [--nodes 512] [--solves 5]"""
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
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--nodes", type=int, default=512)
    ap.add_argument("--solves", type=int, default=5)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from dynamicfusion_body_amd import kernels, scene
    from dynamicfusion_body_amd.pipeline import FrameSolver
    R = args.res
    H, W, fx, cx, cy = scene.CAMERAS["C2" if R <= 256 else "C5"]
    K = scene.intrinsics(fx, cx, cy)
    Kinv = np.linalg.inv(K)
    scale, center, tdist = scene.grid_params(R)
    T = torch.full((R, R, R), tdist, dtype=torch.float32, device="cuda")
    Wt = torch.zeros((R, R, R), dtype=torch.float32, device="cuda")
    for a in (0.0, 40.0, -40.0):
        lw = scene.view_extrinsic(a)
        d = torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda()
        kernels.integrate_depth(T, Wt, d, K, Kinv, lw, scale, center, tdist)
    N, iters = args.nodes, 10
    node_pos, node_w = scene.fibonacci_nodes(N, R)
    ident = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0, 0]), (N, 1))
    ident_t = torch.from_numpy(ident).cuda()
    lw_cam = scene.view_extrinsic(0.0)
    live = scene.render_depth(K, lw_cam, H, W, dtype=np.float32, sphere_offset=np.array([0.6, -0.4, 0.3]) * scale,
                              sphere_r=scene.SPHERE_R * 1.02)
    depth = torch.from_numpy(live).cuda()

    def make(force):
        fs = FrameSolver(K, scale, center, R / 2, knn=4, pcg_iters=10, distributed=force)
        fs.solver.force_collective = force
        fs.set_graph(node_pos, ident, node_w)
        fs.set_canonical(T, Wt, band=4.0)
        return fs

    def timed(fs):
        sv = fs.solver

        def one_solve():
            sv.node_dq.copy_(ident_t)
            for _ in range(iters):
                fs.gn_iteration(depth, lw_cam, rw=5.0, lm_abs=10.0, lm_rel=1e-2, max_dist=2.0, huber=0.5)
        one_solve()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.solves):
            one_solve()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / (args.solves * iters) * 1e3
        dq_eager = sv.node_dq.clone()
        rec = {"eager_ms_per_iter": eager, "host_issue_ms_per_iter": t_issue / (args.solves * iters) * 1e3}
        try:
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                one_solve()
                torch.cuda.synchronize()
                time.sleep(0.3)      # the group's watchdog (100 ms poll) retires the eager collectives: nothing left to query during the capture
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    one_solve()
            torch.cuda.current_stream().wait_stream(side)
            g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.solves):
                g.replay()
            torch.cuda.synchronize()
            rec["graph_ms_per_iter"] = (time.perf_counter() - t0) / (args.solves * iters) * 1e3
            rec["graph_equals_eager"] = bool(torch.equal(sv.node_dq, dq_eager))
            rec["captured"] = True
        except Exception as e:
            rec["captured"] = False
            rec["capture_error"] = "%s: %s" % (type(e).__name__, str(e)[:200])
        rec["final_cost"] = sv.cost()[0]
        return rec, dq_eager

    single, dq_single = timed(make(False))
    fs2 = make(True)
    forced, dq_forced = timed(fs2)
    tri = fs2.solver
    out = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "res": R, "nodes": N,
           "single_gpu_one_call": single, "two_calls_with_all_reduce": forced,
           "collective_path_equals_single_gpu": bool(torch.equal(dq_single, dq_forced)),
           "max_abs_diff": float((dq_single - dq_forced).abs().max()),
           "packed_doubles": None if tri._tri is None else int(tri._tri[3].numel()), "system_doubles": int(tri.system.numel())}
    print(json.dumps(out))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
