#!/usr/bin/env python3
"""bench.py -- TSDF fusion throughput of the HIP hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (starts its own N ranks; also runs under torch.distributed.run)

A "step" is one depth view integrated into the resident 256^3 grid (BASELINE config 2: 256^3 grid,
640x480 synthetic depth, rigid TSDF integration).  At N>1 the SAME grid is cut into N axis-0 slabs, one per
rank, with no data-path collective (strong scaling, BASELINE config 4's partition; `--scaling weak` stacks N
cubes instead); `value` = grid voxels swept per step x steps / max-over-ranks time.  Inputs are resident in HBM
before the timed region.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     -- dominant kernel (integrate_depth_kernel): `achieved` = bytes the launch really loads and stores
                  (64 B per 4-voxel pack that has an updated voxel + the depth map; counted exactly per view,
                  cross-checked by the PMC `traffic`) over the kernel's mean launch duration measured with HIP
                  events on the launch stream, against the 8 TB/s HBM peak.  The SURVEY.md section 8(d) figure
                  (16 B/voxel over the whole grid + 4*H*W) is reported beside it as `algorithmic_*`: it credits
                  packs the kernel never touches, so it is an effective rate, not an HBM fraction.
  k1_512       -- the same kernel at 512^3 / 1280x720 (out of the Infinity Cache): the 0-degree view and a view
                  that updates every voxel.
  cpu_baseline -- the C restatement of the reference's CPU path (oracle/oracle_c.c, validated against the
                  reference's outputs) timed on this box's host cores over the same views; the reference itself
                  is a Python interpreter loop measured at 0.0835 Mvox/s.
  gn, frame    -- BASELINE config 3 (10 GN iterations at 256^3 / 512 nodes) and the composed per-frame loop.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
RES = 256
VIEW_ANGLES = (0.0, 30.0, -45.0, 60.0)      # all in front of the wall: every view updates voxels


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--res", type=int, default=RES, help="per-rank slab is res^3 (default = BASELINE config 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gn", action="store_true", help="skip the warp-solve (GN-iters/s) leg")
    ap.add_argument("--backend", default="auto", help="torch.distributed backend: nccl (= RCCL), gloo (only for rehearsing "
                                                      ">1 rank on a single GPU), auto = nccl when every rank has its own GPU")
    ap.add_argument("--scaling", default="strong", choices=("strong", "weak"),
                    help="strong (default): the SAME res^3 grid cut into --gpus axis-0 slabs (BASELINE configs 2/4); "
                         "weak: --gpus cubes stacked along axis 0")
    ap.add_argument("--launch", default="auto", choices=("auto", "eager", "graph"),
                    help="how the timed K1 steps are issued: eager ctypes calls, one hipGraph replay, auto = the faster")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the streaming micro-benchmark (a child process; profiler runs skip it)")
    ap.add_argument("--no-k1-512", action="store_true", help="skip the 512^3 K1 roofline leg (1 GPU only)")
    ap.add_argument("--job-timeout", type=int, default=1500, help="--gpus N started by this script: seconds after which all ranks are killed")
    ap.add_argument("--leg-timeout", type=int, default=300, help="N > 1: seconds the secondary legs (gn, frame) may take before "
                                                                   "rank 0 prints the line without them and every rank leaves (0 = off)")
    ap.add_argument("--no-frame", action="store_true", help="skip the end-to-end per-frame leg")
    ap.add_argument("--no-frame-512", action="store_true", help="skip the config-5-size frame leg (512^3, 8 views, 2 048 nodes; 1 GPU only)")
    ap.add_argument("--gn-nodes", type=int, default=512)
    ap.add_argument("--global-iters", type=int, default=None, help="frame legs: rigid-mode steps in front of the node iterations (default: SlabFrame's)")
    ap.add_argument("--gn-mode", default="auto", choices=("auto", "sharded", "replicated"), help="N > 1: how the warp solve runs")
    ap.add_argument("--gn-solves", type=int, default=5, help="timed solves of 10 GN iterations each")
    return ap.parse_args()


def gn_leg(args, torch, dist, scene, rank, world, barrier):
    """BASELINE config 3: 256^3 canonical volume, 512-node warp field, DQB warp + projective data
    association + 10 GN iterations per solve.  Strong scaling at N > 1: the canonical samples are
    sharded by axis-0 slab of the SAME grid and the normal equations are all-reduced each iteration."""
    import time as _t
    from dynamicfusion_body_amd import kernels
    from dynamicfusion_body_amd import dist as D
    from dynamicfusion_body_amd.pipeline import FrameSolver
    if os.environ.get("DFH_TEST_FAIL_GN_RANK") == str(rank):        # (test hook: the watchdog path of main())
        raise RuntimeError("injected failure on rank %d" % rank)
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
    N, k, iters = args.gn_nodes, 4, 10
    node_pos, node_w = scene.fibonacci_nodes(N, R)
    ident = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0, 0]), (N, 1))
    # Several GPUs: the solve runs sharded (every rank builds the normal equations of its slab's samples, one all-reduce per
    # iteration: BASELINE north star) or replicated (every rank solves the whole system on all samples, no per-iteration
    # collective), whichever dist.solve_mode's latency model predicts faster (--gn-mode forces); DESIGN.md section 6.
    band_voxels = int(((Wt > 0) & (T.abs() < tdist)).sum().item()) if world > 1 else 0       # ~ the sample count (band = 4 voxels = tdist)
    mode = "replicated" if world == 1 else (args.gn_mode if args.gn_mode != "auto" else D.solve_mode(max(band_voxels, 1), N, 12 * N, world))
    fs = FrameSolver(K, scale, center, R / 2, knn=k, pcg_iters=10, distributed=(world > 1 and mode == "sharded"))
    fs.set_graph(node_pos, ident, node_w)
    a, b = D.slab_range(R, rank, world)
    if mode == "sharded":
        # this rank's samples; normals are central differences across the slab faces too: one halo plane each side with
        # weight 0 (gradients only), so the union over ranks is the single-GPU sample set
        lo, hi = max(a - 1, 0), min(b + 1, R)
        Wp = Wt[lo:hi].clone()
        if lo < a:
            Wp[0] = 0
        if hi > b:
            Wp[-1] = 0
        S = fs.set_canonical(T[lo:hi].contiguous(), Wp, band=4.0, x0=lo)
    else:
        S = fs.set_canonical(T, Wt, band=4.0)            # (in the frame loop: the slabs' samples all-gathered once per frame)
    lw_cam = scene.view_extrinsic(0.0)
    live = scene.render_depth(K, lw_cam, H, W, dtype=np.float32, sphere_offset=np.array([0.6, -0.4, 0.3]) * scale,
                              sphere_r=scene.SPHERE_R * 1.02)
    depth = torch.from_numpy(live).cuda()
    sv = fs.solver
    ident_t = torch.from_numpy(ident).cuda()

    def one_solve():
        sv.node_dq.copy_(ident_t)
        for _ in range(iters):
            fs.gn_iteration(depth, lw_cam, rw=5.0, lm_abs=10.0, lm_rel=1e-2, max_dist=2.0, huber=0.5)

    one_solve()                                   # warm-up (also builds the block pattern)
    barrier()
    # eager: every kernel launched from Python through the C ABI
    t0 = _t.perf_counter()
    for _ in range(args.gn_solves):
        one_solve()
    t_issue = _t.perf_counter() - t0              # host time to enqueue (no sync yet)
    barrier()
    dt_eager = _t.perf_counter() - t0
    launch = "eager"
    dt = dt_eager
    graph_ms = None
    # one solve captured as a HIP graph and replayed.  Sharded solves hold an all-reduce per iteration: RCCL collectives can be
    # captured (tools/rccl_sanity.py: a live communicator inside a capture); gloo's cannot, a failed capture keeps the eager number
    if (world == 1 or mode == "replicated" or dist.get_backend() == "nccl") and not os.environ.get("DFH_NO_GRAPH"):
        try:
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                one_solve()
                quiesce_collectives(torch, dist, world)
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    one_solve()
            torch.cuda.current_stream().wait_stream(side)
            g.replay()
            barrier()
            t0 = _t.perf_counter()
            for _ in range(args.gn_solves):
                g.replay()
            barrier()
            dt_graph = _t.perf_counter() - t0
            graph_ms = dt_graph / (args.gn_solves * iters) * 1e3
            if dt_graph < dt_eager:
                dt, launch = dt_graph, "hipGraph replay of one 10-iteration solve"
        except Exception as e:                    # capture is an optimisation; eager numbers stay valid
            launch = "eager (graph capture failed: %s)" % str(e)[:80]
    dt = D.max_over_ranks([dt])[0]
    cost, cnt = sv.cost()
    tot = torch.tensor([float(S)], dtype=torch.float64, device="cuda" if (world == 1 or dist.get_backend() == "nccl") else "cpu")
    if world > 1 and mode == "sharded":
        dist.all_reduce(tot)
    A = int(tot.item())
    B = sv.B
    n_it = args.gn_solves * iters
    # HBM bytes one iteration has to move (fp64 layout of this build): associate reads pos/nbr/wts and
    # writes corr/valid; build reads pos/nrm/nbr/wts/corr/valid; system + PCG vectors per PCG iteration
    per_sample = (24 + 4 * k + 8 * k + 24 + 1) + (24 + 24 + 4 * k + 8 * k + 24 + 1)
    alg = per_sample * A / world + 4 * H * W + 8 * (36 * B + 6 * N) * (2 + sv.pcg_iters) + 64 * N
    oracle = {}
    try:
        # CPU value beside final_cost: the same 10-iteration loop with the same truncated PCG in numpy (oracle/gn_np.py), computed
        # by tests/test_gpu_configs.py::test_config3_benched_settings_vs_truncated_pcg_oracle on this problem and committed as data
        rec = json.load(open(os.path.join(ROOT, "tests", "golden", "config3_oracle_costs.json")))
        if R == 256 and N == 512 and rec["samples"] == A:
            oracle = {"oracle_final_cost": rec["final_cost_oracle"],
                      "oracle_final_cost_exact_linear_solve": rec["final_cost_oracle_exact_solve"],
                      "final_cost_rel_diff_vs_oracle": abs(cost - rec["final_cost_oracle"]) / rec["final_cost_oracle"],
                      "oracle_source": "tests/golden/config3_oracle_costs.json (numpy GN loop, same truncated Chronopoulos-Gear PCG)"}
    except Exception:
        pass
    return {**oracle, "gn_iters_per_s": n_it / dt, "ms_per_gn_iter": dt / n_it * 1e3, "gn_iters_per_solve": iters,
            "solves_timed": args.gn_solves, "active_samples": A, "nodes": N, "knn": k, "blocks_6x6": B,
            "pcg_iters": sv.pcg_iters, "scaling": "strong" if world > 1 else "n/a", "launch": launch,
            "solve_mode": mode if world > 1 else "single GPU", "pcg_path": pcg_path_name(N),
            "solve_mode_note": None if world == 1 else "sharded = slab samples + one all-reduce of the upper block triangle per iteration; "
                               "replicated = every rank solves the whole system (no per-iteration collective); chosen by "
                               "dynamicfusion_body_amd.dist.solve_mode, a latency model whose collective term is an estimate "
                               "(no multi-GPU node was available while this was written)",
            "eager_ms_per_gn_iter": dt_eager / n_it * 1e3, "graph_ms_per_gn_iter": graph_ms, "host_issue_ms_per_gn_iter": t_issue / n_it * 1e3,
            "final_cost": cost, "valid_samples_rank0": cnt,
            "hbm_bytes_per_iter_algorithmic": alg, "hbm_GBps_algorithmic": alg / (dt / n_it) / 1e9,
            "bound": "not HBM (SURVEY 8(d)): rows kernel = fp64 issue on the CUs + the matrix pipe (v_mfma_f64_16x16x4, 64 cycles each), "
                     "gather = latency of the longest lists, PCG = one cross-XCD hand-off per iteration; counters (256^3 / 512 nodes, "
                     "profiles/r2_gn_experiments.txt sections 2-4, 13): rows kernel VALU issue 33 % of resident wave cycles, waiting 53 %, "
                     "MFMA pipe busy 3.57 M cycles per launch, LDS bank conflicts 14 % of LDS cycles; gather walks the blocks with column >= row "
                     "only and stores their sums twice (round 3: 70 -> 50 MB per launch in the frame); PCG iteration 3.0 us of which ~1.8 us "
                     "hand-off; round 3's experiments on the rows kernel: profiles/r3_gn_experiments.txt",
            "workload": "%d^3 canonical volume, %d-node warp field, DQB warp + projective association + %d GN "
                        "iterations per solve (fp64), %s" % (R, N, iters, "samples sharded by axis-0 slab" if mode == "sharded" else
                                                           "all samples on every rank")}


def kernels_mod():
    from dynamicfusion_body_amd import kernels
    return kernels


def quiesce_collectives(torch, dist, world):
    """Before a stream capture in a process with a live RCCL group: drain the device and give the process group's watchdog thread
    (it polls every 100 ms) time to retire the collectives already issued, so that it has no event left to query while this
    thread captures.  thread_local capture mode is meant to allow such queries; one child of
    tests/test_gpu_dist_gloo.py::test_one_rank_rccl_sharded_iteration_captures_into_a_graph aborted (SIGABRT) in about twenty
    runs before this, none since."""
    torch.cuda.synchronize()
    if dist is not None and world >= 1 and dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        import time
        time.sleep(0.3)


def pcg_path_name(n_nodes):
    """Which PCG the solves of this process take (dfh_pcg_path): runs compare bit for bit only on the same path -- ranks that share
    one GPU (a rehearsal) take the two-launch kernels, one rank per GPU the persistent single-reduction kernel."""
    from dynamicfusion_body_amd import _lib
    code = _lib.load().dfh_pcg_path(int(n_nodes))
    return {1: "persistent single-reduction kernel", 2: "two launches per iteration (ranks share a GPU, or forced)"}.get(code, "error %d" % code)


def k23_of(torch, kernels, sf, tvox):
    """K2 (FusionDM.updateTSDF, core/fusion_dm.py:300-316) and K3 (Fusion.updateTSDF, core/fusion.py:153-198) on the state a frame
    loop has reached: the canonical slab, the live volume of its last frame, the solved warp field, K3's stored neighbourhoods
    (steady state, what every frame after the first runs).  HIP-event time per call; SURVEY 8(d)'s algorithmic bytes = 20 B per
    voxel (T and w read and written, each live sample counted once)."""
    from dynamicfusion_body_amd import _lib
    R = sf.R
    nv = float(sf.T.numel())
    sv = sf.fs.solver
    T, Wt = sf.T.clone(), sf.Wt.clone()
    alg = 20.0 * nv
    out = {}
    lw_rigid = np.array([0.9999995, 0.0005, -0.0007, 0.0004, 0.0, 0.05, -0.03, 0.02])        # a small rigid motion (unit to 1e-7: the reference's _lw)
    ms = time_launches(torch, lambda: kernels.fuse_volume_rigid(T, Wt, sf.live, lw_rigid, tvox, res=(R, R, R), x_range=(sf.a, sf.b)), 10)
    out["k2_rigid"] = {"kernel_ms": ms, "algorithmic_bytes": alg, "algorithmic_GBps": alg / (ms * 1e-3) / 1e9,
                       "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "mvox_per_s": nv / ms / 1e3}

    def k3():
        kernels.fuse_volume_dqb(T, Wt, sf.live, sv.node_pos, sv.node_dq, sv.node_w, sf.knn, sf.ident_lw, tvox, res=(R, R, R),
                                x_range=(sf.a, sf.b), workspace=sf.ws_dqb, rebuild_candidates=False)
    ms = time_launches(torch, k3, 10)
    tabs = kernels.dqb_skip_tables(sf.ws_dqb, (R, R, R), (R, R, R), int(sv.N), x_range=(sf.a, sf.b))
    skipped = float(tabs["S"].float().mean()) if tabs["ok"] else 0.0
    skip_on = _lib.get_option("k3_skip") > 0 or (_lib.get_option("k3_skip") < 0 and R ** 3 > (1 << 24))
    out["k3_dqb"] = {"kernel_ms": ms, "algorithmic_bytes": alg, "algorithmic_GBps": alg / (ms * 1e-3) / 1e9,
                     "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "mvox_per_s": nv / ms / 1e3, "nodes": int(sv.N),
                     "constant_live_skip": bool(skip_on and tabs["ok"]),
                     "bricks_skipping_the_warp": skipped if (skip_on and tabs["ok"]) else None,
                     "what": "steady state (stored neighbourhoods, float32 volumes, identity m_lw): every launch of the call -- "
                             "live-cell mask, reach, per-brick bound, constant-live stream, warp kernel, redo list"}
    _lib.set_option("k3_skip", 0 if skip_on else 1)                                          # the other setting, for the record
    try:
        ms2 = time_launches(torch, k3, 10)
        out["k3_dqb"]["kernel_ms_skip_%s" % ("off" if skip_on else "on")] = ms2
    finally:
        _lib.set_option("k3_skip", None)
    del T, Wt
    return out


def frame_leg(args, torch, dist, scene, rank, world, barrier, nframes=8, with_k23=False):
    """One non-rigid frame at config-3 scale, the loop of the reference's test.py:116-131 with this build's
    device path (pipeline.SlabFrame): live depth -> live TSDF slab (K1) -> all-gather of the live volume
    (N > 1) -> 10 GN iterations against the live depth (one all-reduce each) -> canonical slab <- live through
    the warp field (K3) -> surface samples for the next frame (+ marching cubes of the canonical volume on one
    GPU).  Strong scaling at N > 1: the SAME res^3 grid cut into axis-0 slabs."""
    import time as _t
    from dynamicfusion_body_amd import mesh
    from dynamicfusion_body_amd import dist as D
    from dynamicfusion_body_amd.pipeline import SlabFrame
    R = args.res
    H, W, fx, cx, cy = scene.CAMERAS["C2" if R <= 256 else "C5"]
    K = scene.intrinsics(fx, cx, cy)
    scale, center, tdist = scene.grid_params(R)
    N, iters = args.gn_nodes, 10
    node_pos, node_w = scene.fibonacci_nodes(N, R)
    # K1 stores min(tdist, sd) / scale, i.e. voxel units in [-4, 4]; this composed loop keeps every stage in those
    # units (fill value and the truncation of the TSDF->TSDF update = tdist / scale = 4 voxels), unlike the
    # reference's classes, which fill and truncate with the world-unit tdist
    sf = SlabFrame(K, scale, center, R, tdist / scale, node_pos, node_w, knn=4, pcg_iters=10, band=4.0)
    # config 3 (256^3): three views in front of the object; config 5 (--res 512): its 8-view orbit, 45 degrees apart
    angles = (0.0, 40.0, -40.0) if R <= 256 else tuple(45.0 * v for v in range(8))
    for a in angles:
        lw = scene.view_extrinsic(a)
        sf.integrate(torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda(), lw)
    sf.refresh_samples()
    lws = [scene.view_extrinsic(a) for a in angles]
    lw_cam = lws
    depths = []
    for f in range(nframes):                      # the sphere drifts and breathes a little every frame; three views per frame
        off = np.array([0.10, -0.07, 0.05]) * (f + 1) * scale
        depths.append([torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0, sphere_offset=off,
                                                           sphere_r=scene.SPHERE_R * (1.0 + 0.004 * (f + 1)))).cuda() for lw in lws])
    info = {"vertices": None, "faces": None}

    stages = {}

    mesh_stream = torch.cuda.Stream() if world == 1 else None

    def frame(f, timed=False):
        if mesh_stream is not None:
            torch.cuda.current_stream().wait_stream(mesh_stream)     # the previous frame's mesh has read the canonical volume
        pending = []

        def start_mesh():
            # the mesh needs the updated canonical volume only: its count pass is queued on a second stream the moment the
            # TSDF update is, beside the sample refresh (count -> emit -> node search -> sort) on the first; the emit
            # passes follow when the frame's own launches are queued
            with torch.cuda.stream(mesh_stream):
                mesh_stream.wait_event(sf.updated)
                pending.append(mesh.marching_cubes_begin(sf.T, 0.0))
        info["samples"] = sf.step(depths[f], lw_cam, gn_iters=iters, stage_ms=stages if timed else None,
                                  on_updated=None if (timed or mesh_stream is None) else start_mesh, global_iters=args.global_iters)
        if world == 1:
            t1 = _t.perf_counter()
            if timed:
                v, fc, n, val = mesh.marching_cubes(sf.T, 0.0)
            else:
                with torch.cuda.stream(mesh_stream):
                    v, fc, n, val = pending[0].finish()
            info["vertices"], info["faces"] = int(v.shape[0]), int(fc.shape[0])
            if timed:
                torch.cuda.synchronize()
                stages["mesh"] = stages.get("mesh", 0.0) + (_t.perf_counter() - t1) * 1e3

    nwarm = 2                                     # warm-up: frame 0 allocates, builds the block pattern and K3's stored
    for f in range(nwarm):                        # neighbourhoods; frame 1 is the first to take the steady-state paths
        frame(f)
    barrier()
    t0 = _t.perf_counter()
    for f in range(nwarm, nframes):
        frame(f)
    barrier()
    dt = D.max_over_ranks([(_t.perf_counter() - t0) / (nframes - nwarm)])[0]
    for f in range(nwarm, nframes):               # second, untimed-for-throughput pass with a sync after every stage
        frame(f, timed=True)
    cost, cnt = sf.fs.solver.cost()
    tot = torch.tensor([float(info["samples"])], dtype=torch.float64, device="cuda")
    if world > 1 and sf.solve_mode == "sharded":          # (replicated: every rank already holds the all-gathered sample set)
        dist.all_reduce(tot)
    k23 = k23_of(torch, kernels_mod(), sf, tdist / scale) if (with_k23 and world == 1) else None
    return {**({"k23": k23} if k23 is not None else {}),
            "ms_per_frame": dt * 1e3, "frames_per_s": 1.0 / dt, "frames_timed": nframes - nwarm, "scaling": "strong" if world > 1 else "n/a",
            "stage_ms_with_syncs": {kk: vv / (nframes - nwarm) for kk, vv in stages.items()},
            "gn_iters_per_frame": iters, "nodes": N, "samples": int(tot.item()), "mesh_vertices": info["vertices"],
            "mesh_faces": info["faces"], "final_cost": cost,
            "solve_mode": "single GPU" if world == 1 else sf.solve_mode,
            "pcg_path": pcg_path_name(N),
            "exchange": "none" if world == 1 else ("per frame: all-gather of the live volume (%.0f MB) + face-plane halo" % (R ** 3 * 4 / 1e6)) +
                        ("; per GN iteration: one all-reduce of the normal equations' upper block triangle" if sf.solve_mode == "sharded" else
                         " + all-gather of the slabs' samples (96 B each); no collective inside the GN iterations (every rank solves the whole system)"),
            "workload": "%d^3 grid in %d axis-0 slab(s), %d nodes: live TSDF (%d views of %dx%d, one sweep) + %d GN iterations + DQB "
                        "TSDF update + sample refresh%s, per frame" % (R, world, N, len(lws), W, H, iters,
                                                                     " + marching cubes (on a second stream beside the sample refresh)" if world == 1 else "")}


def pmc_traffic(kernel_substr, res, instance=None):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    of this same command (profiles/<tag>_summary.json, written by tools/summarize_profile.py:
    2*FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md §HBM).  The newest matching profile wins
    (files are named per round: r1x_..., r2x_...).  None when no matching profile."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json"))):
        try:
            d = json.load(open(f))
            line = json.loads(d["bench_line"])
            if line["config"]["grid"][1] != res or line.get("n_gpus", 1) != 1:
                continue
            for name, t in d["traffic"].items():
                if kernel_substr in name:
                    # the column sweep has several template instances (prefetch, non-temporal) in one profile -- the bench's
                    # 512^3 leg runs another one than the headline: `instance` (the template tail) picks the headline's
                    if instance is not None and instance not in name:
                        continue
                    best = (t["hbm_bytes_per_launch"], os.path.basename(f))
        except Exception:
            continue
    return best


def moved_bytes(torch, kernels, depth, K, Kinv, lw, scale, center, tdist, tsdf_res, res, x_range, H, W):
    """Bytes one single-view launch really loads and stores for this view, by the sweep the library takes for this slab
    (kernels.integrate_path):
      rows            T and w of a 16-byte pack (4 voxels along z) are read and written iff the view updates one of its voxels:
                      64 B per such pack;
      columns         every pack of the slab is read (32 B), updated packs are written (32 B);
      columns_culled  every pack of a brick that survives the classification is read (32 B: the mask array the sweep left in
                      its workspace says which), updated packs are written (32 B);
    plus the depth map once.  Counted exactly, outside any timed region: the view is integrated into a fresh volume pair and
    the packs with a non-zero weight are counted (torch plumbing; the PMC `traffic` figure cross-checks it).
    Returns (bytes, updated voxels, path)."""
    nx = x_range[1] - x_range[0]
    Tt = torch.full((nx, res[1], res[2]), float(tdist), dtype=torch.float32, device="cuda")
    Wt = torch.zeros_like(Tt)
    ws = kernels.integrate_workspace(1, H, W, res, x_range)
    path = kernels.integrate_path(Tt, depth, res=res, x_range=x_range)
    kernels.integrate_depth(Tt, Wt, depth, K, Kinv, lw, scale, center, tdist, 100.0, tsdf_res=tsdf_res, res=res, x_range=x_range, workspace=ws)
    upd = Wt > 0
    vox = int(upd.sum().item())
    n_packs = nx * res[1] * (res[2] // 4) if res[2] % 4 == 0 else nx * res[1] * res[2]
    if res[2] % 4 == 0:
        packs = int(upd.view(-1, 4).any(dim=1).sum().item())
    else:
        packs = vox                                   # scalar kernel: one voxel per "pack" (16 B each)
    per_pack = 32.0 if res[2] % 4 == 0 else 8.0       # T + w of one pack, one direction
    if path == "columns":
        loaded = n_packs
    elif path == "columns_culled":
        alive = int((kernels.brick_masks(ws, res, x_range) != 0).sum().item())
        loaded = min(n_packs, alive * 64)             # (bricks that stick out of a ragged grid have fewer packs: upper bound)
    else:
        loaded = packs
    del Tt, Wt, upd
    return per_pack * (loaded + packs) + 4.0 * H * W, vox, path


def time_launches(torch, fn, n, warm=2):
    """Mean duration (ms) of n back-to-back calls of fn on torch's current stream (HIP events on that stream)."""
    for _ in range(warm):
        fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def k1_512_leg(torch, kernels, scene):
    """K1 at 512^3 with 1280x720 depth (BASELINE configs 4/5): the working set (1.07 GB) is far beyond the 256 MiB
    Infinity Cache.  Two views: the bench's 0-degree view (updates about half the voxels) and a view that updates every
    voxel (camera moved back so that the frustum contains the whole grid, a wall far behind it: all free space), for which
    moved bytes = algorithmic bytes."""
    R = 512
    H, W, fx, cx, cy = scene.CAMERAS["C5"]
    K = scene.intrinsics(fx, cx, cy)
    Kinv = np.linalg.inv(K)
    scale, center, tdist = scene.grid_params(R)
    T = torch.full((R, R, R), tdist, dtype=torch.float32, device="cuda")
    Wt = torch.zeros_like(T)
    out = {}
    lw0 = scene.view_extrinsic(0.0)
    d0 = torch.from_numpy(scene.render_depth(K, lw0, H, W, dtype=np.float32)).cuda()
    lw_full = lw0.copy()
    lw_full[2, 3] += 1.5                                    # camera 1.5 m further back: the whole cube is in view
    d_full = torch.full((H, W), -8.0, dtype=torch.float32, device="cuda")
    for name, lw, d in (("view_0deg", lw0, d0), ("view_all_voxels", lw_full, d_full)):
        tb, vox, path = moved_bytes(torch, kernels, d, K, Kinv, lw, scale, center, tdist, R, (R, R, R), (0, R), H, W)
        ms = time_launches(torch, lambda: kernels.integrate_depth(T, Wt, d, K, Kinv, lw, scale, center, tdist, 100.0), 10)
        alg = 16.0 * R ** 3 + 4.0 * H * W
        out[name] = {"kernel_ms": ms, "sweep": path, "updated_fraction": vox / float(R ** 3), "moved_bytes": tb,
                     "moved_GBps": tb / (ms * 1e-3) / 1e9, "frac_moved": tb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "algorithmic_bytes": alg, "algorithmic_GBps": alg / (ms * 1e-3) / 1e9,
                     "frac_algorithmic": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "mvox_per_s": R ** 3 / ms / 1e3}
    out["workload"] = ("512^3 grid, 1280x720 depth, one rigid fuseDepths call per launch (depth pyramid + brick classification + "
                       "column sweep: kernel_ms is the whole call), 1 GPU")
    del T, Wt
    return out


def main():
    args = parse()
    from dynamicfusion_body_amd import launch
    if args.gpus > 1 and not launch.under_launcher():
        # `python bench.py --gpus N`: start the N ranks ourselves.  This parent never touches the GPU (no HIP call, no
        # torch.cuda query) and never re-execs; it relays rank 0's JSON line and the worst exit code.
        # (a finite limit: a rank stuck in a driver call that ignores SIGTERM must not make this command wait for ever)
        sys.exit(launch.spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus, json_only=True, timeout=args.job_timeout))
    import torch
    import torch.distributed as dist
    from dynamicfusion_body_amd import kernels, scene
    from dynamicfusion_body_amd import dist as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    n_dev = torch.cuda.device_count()
    backend = args.backend
    if backend == "auto":                                  # RCCL needs one GPU per rank; fewer GPUs = a rehearsal over gloo
        backend = "nccl" if n_dev >= world else "gloo"
    dev = local_rank % max(1, n_dev)                       # == local_rank on a real node
    torch.cuda.set_device(dev)
    distributed = world > 1
    if distributed:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)

    R = args.res
    cam = "C2" if R <= 256 else "C5"
    H, W, fx, cx, cy = scene.CAMERAS[cam]
    K = scene.intrinsics(fx, cx, cy)
    Kinv = np.linalg.inv(K)
    scale, center, tdist = scene.grid_params(R)
    tsdf_res = R
    center = center.copy()
    if args.scaling == "strong":
        # BASELINE configs 2 and 4: ONE res^3 grid cut into `world` axis-0 slabs, no data-path collective
        res = (R, R, R)
        x_range = D.slab_range(R, rank, world)
        vox_per_step = R * R * R
        workload = "%d^3 grid in %d axis-0 slab(s) (%d planes per GPU)" % (R, world, -(-R // world))
    else:
        # weak: `world` cubes stacked along axis 0, centred on the sphere; rank owns one cube.  Most of the outer cubes
        # lie outside every frustum and skip their loads -- this mode flatters and is not the default
        res = (R * world, R, R)
        x_range = (R * rank, R * (rank + 1))
        center[0] -= scale * (R * world / 2 - R / 2)
        vox_per_step = R * R * R * world
        workload = "%d^3 voxels per GPU (grid %dx%dx%d, axis-0 slabs)" % (R, res[0], res[1], res[2])
    nx = x_range[1] - x_range[0]

    lws = [scene.view_extrinsic(a) for a in VIEW_ANGLES]
    depths_np = [scene.render_depth(K, lw, H, W, dtype=np.float32) for lw in lws]
    depths = [torch.from_numpy(d).cuda() for d in depths_np]
    T = torch.full((nx, R, R), tdist, dtype=torch.float32, device="cuda")
    Wt = torch.zeros((nx, R, R), dtype=torch.float32, device="cuda")

    def step(i):
        v = i % len(lws)
        kernels.integrate_depth(T, Wt, depths[v], K, Kinv, lws[v], scale, center, tdist, 100.0,
                                tsdf_res=tsdf_res, res=res, x_range=x_range)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # The timed region is the same K launches either way; with `--launch graph` (default where it is faster: a slab of
    # 256^3 / 8 is a 6 us kernel, shorter than one Python call through ctypes) they are captured once, after the warm-up,
    # into a HIP graph on a side stream and the timed region replays it.  HIP events bracket the work on the stream it
    # runs on in both cases.
    for i in range(args.warmup):
        step(i)
    barrier()
    graph = None
    side = None
    if args.launch in ("graph", "auto") and not os.environ.get("DFH_NO_GRAPH"):
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            g = torch.cuda.CUDAGraph()
            quiesce_collectives(torch, dist, world)
            with torch.cuda.stream(side):
                # thread_local: RCCL's watchdog thread may query events while this thread captures; in the default global
                # mode that would invalidate the capture
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    for i in range(args.steps):
                        step(args.warmup + i)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = g
        except Exception as e:                                # capture is an optimisation; eager numbers stay valid
            graph = None
            sys.stderr.write("graph capture failed, timing eager launches: %s\n" % str(e)[:200])
            torch.cuda.synchronize()

    def timed(run):
        barrier()
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()                       # same stream the kernels run on (torch current stream)
        run()
        ev1.record()
        barrier()
        dt = time.perf_counter() - t0
        kern_ms = ev0.elapsed_time(ev1) / args.steps
        if distributed:
            tt = torch.tensor([dt, kern_ms], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt, kern_ms = float(tt[0]), float(tt[1])
        return dt, kern_ms

    def eager_run():
        for i in range(args.steps):
            step(args.warmup + i)
    dt_eager, kern_ms_eager = timed(eager_run)
    dt, kern_ms, launch_mode = dt_eager, kern_ms_eager, "eager (one ctypes call per step)"
    dt_graph = None
    if graph is not None:
        dt_graph, kern_ms_graph = timed(graph.replay)
        if args.launch == "graph" or dt_graph < dt_eager:
            dt, kern_ms, launch_mode = dt_graph, kern_ms_graph, "hipGraph replay of the %d steps" % args.steps

    value = vox_per_step * args.steps / dt / 1e6
    # ---- roofline of the dominant kernel (this rank's slab).  `achieved` counts the bytes the launch really moves
    # (see moved_bytes; cross-checked by the PMC `traffic`), not the 16 B/voxel of SURVEY section 8(d):
    # that figure credits packs no view updates, which the kernel never loads, and is kept as `algorithmic_*`.
    tb = [moved_bytes(torch, kernels, depths[v], K, Kinv, lws[v], scale, center, tdist, tsdf_res, res, x_range, H, W)
          for v in range(len(lws))]
    touched = sum(t[0] for t in tb) / len(tb)
    sweep_path = tb[0][2]
    upd_frac = sum(t[1] for t in tb) / len(tb) / float(max(1, nx) * R * R)
    alg_bytes = 16.0 * nx * R * R + 4.0 * H * W               # per launch (one rank's slab), SURVEY section 8(d)
    achieved = touched / (kern_ms * 1e-3) / 1e9
    alg_gbps = alg_bytes / (kern_ms * 1e-3) / 1e9
    k1_kernel = {"rows": "integrate_depth_kernel", "columns": "integrate_depth_column_kernel",
                 "columns_culled": "integrate_depth_column_kernel", "exact": "integrate_depth_exact_kernel"}[sweep_path]
    if sweep_path == "rows" and nx * R * R <= (1 << 23):
        k1_kernel = "integrate_depth_rows_early_kernel"
    out = {
        "metric": "Mvoxels/s TSDF fusion + GN-iters/s warp solve, 256³ grid, 1/2/4/8 GPU",
        "value": value,
        "unit": "Mvoxels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f64 geometry / f32 volume",
        "data": "synthetic",
        "config": {"workload": "%s, %dx%d synthetic depth, rigid TSDF integration (fuseDepths), %d views cycled"
                               % (workload, W, H, len(lws)),
                   "grid": list(res), "depth": [H, W], "views": len(lws), "partition": "slab%d" % world,
                   "backend": backend if distributed else "none", "launch": launch_mode},
        "launch": {"mode": launch_mode, "eager_ms_per_step": dt_eager / args.steps * 1e3,
                   "graph_ms_per_step": None if dt_graph is None else dt_graph / args.steps * 1e3},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": k1_kernel, "sweep": sweep_path, "kernel_ms": kern_ms,
                     "achieved_is": "bytes the launch loads and stores (see moved_bytes: by the sweep taken, here '%s') + the "
                                    "depth map, mean over the cycled views, / mean launch duration" % sweep_path,
                     "moved_bytes_per_launch": touched, "updated_voxel_fraction": upd_frac,
                     "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_GBps": alg_gbps,
                     "frac_algorithmic": alg_gbps / HBM_PEAK_GBS,
                     "limiter": "the life of a wave (launch, projection, dependent depth gathers, store drain), not bytes or VALU: a 256^3 "
                                "pair of volumes (134 MB) sits in the 256 MiB Infinity Cache; k1_512 below is the out-of-cache figure "
                                "(profiles/r3_k1_experiments.txt, DESIGN.md section 3)"},
    }

    # measured ceilings of THIS access pattern, hand-written (tools/ubench/rmw_stream.hip, built by __graft_entry__.build): a float4
    # copy and an in-place read-modify-write of two 512^3 float32 volumes (far beyond the 256 MiB Infinity Cache) as 1-KiB rows and
    # as 4 x 2 x 32 bricks, default and non-temporal cache policy.  The sweep's ceiling is the best RMW figure.  Run as a child
    # process on rank 0 only (its own 1 GiB of HBM).
    if rank == 0 and not args.no_ceiling:
        try:
            import subprocess
            ub = os.path.join(ROOT, "tools", "ubench", "rmw_stream")
            r = subprocess.run([ub, "512", "ceiling"], capture_output=True, text=True, timeout=120)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
            ceil = json.loads(line)
            ceil_gbs = max(ceil["rmw_rows_GBps"], ceil["rmw_rows_nt_GBps"], ceil["rmw_bricks_4x2x32_GBps"], ceil["rmw_bricks_4x2x32_nt_GBps"])
            out["roofline"]["copy_ceiling_GBps"] = ceil_gbs
            out["roofline"]["copy_ceiling_is"] = "best in-place read-modify-write of T and w at 512^3 (tools/ubench/rmw_stream.hip)"
            out["roofline"]["ceilings_GBps"] = ceil
            out["roofline"]["frac_of_copy_ceiling"] = achieved / ceil_gbs
        except Exception as e:                                                # a missing tool must not lose the line
            out["roofline"]["copy_ceiling_GBps"] = None
            out["roofline"]["copy_ceiling_error"] = "%s: %s" % (type(e).__name__, str(e)[:160])
    # secondary number: the same views fused in ONE sweep of the volume (dfh_integrate_depth_multi; bit-identical to the
    # consecutive sweeps timed above, compute-bound on the per-view projection -- never the headline `value`)
    try:
        Tm, Wm = torch.full_like(T, tdist), torch.zeros_like(Wt)
        wsv = kernels.integrate_workspace(len(lws), H, W, res, x_range)

        def sweep():
            kernels.integrate_depth_views(Tm, Wm, depths, K, Kinv, lws, scale, center, tdist, 100.0, tsdf_res=tsdf_res, res=res,
                                          x_range=x_range, workspace=wsv)
        ms_sweep = time_launches(torch, sweep, max(1, min(20, args.steps // len(lws))), warm=1)
        out["multi_view_sweep"] = {"views": len(lws), "ms_per_sweep": ms_sweep, "us_per_view": ms_sweep * 1e3 / len(lws),
                                   "mvox_per_s_per_gpu": nx * R * R * len(lws) / ms_sweep / 1e3,
                                   "hbm_GBps_algorithmic": (16.0 * nx * R * R + 4.0 * H * W * len(lws)) / (ms_sweep * 1e-3) / 1e9}
        del Tm, Wm
    except Exception as e:                                                   # a secondary figure must not lose the line
        out["multi_view_sweep"] = {"error": repr(e)}
    if world == 1:
        # (template tail of integrate_depth_column_kernel<DepthT, PINHOLE, BY, PREFETCH, NT>: slabs beyond the 256 MiB Infinity Cache
        # run with prefetch and non-temporal T / w, csrc/dfh_integrate.hip)
        big = nx * R * R * 8 > (256 << 20)
        inst = None if "column" not in k1_kernel else (", true, true>" if (big and sweep_path == "columns_culled") else
                                                        (", false, true>" if big else ", false, false>"))
        tr = pmc_traffic(k1_kernel, R, inst)
        if tr is not None:
            out["roofline"]["traffic"] = tr[0]
            out["roofline"]["traffic_source"] = "profiles/" + tr[1]

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # checker timed as the CPU baseline, never the product: the C restatement (OpenMP, all host
        # threads) and, for reference, the single-threaded numpy restatement
        from oracle import oracle_c as OC
        from oracle import oracle_np as O
        OC.build()
        nthr = OC.threads()
        To = np.zeros((R, R, R)) + tdist
        Wo = np.zeros((R, R, R))
        cycles = 5 if R <= 256 else 1
        # the GPU volume has seen warmup + steps (eager) [+ steps (graph)] sweeps; the mask after one cycle of the
        # views is the same as after any number of them
        t0 = time.perf_counter()
        for _ in range(cycles):
            for v in range(len(lws)):
                OC.fuse_depths(depths_np[v], lws[v], K, Kinv, To, Wo, tdist, tsdf_res=tsdf_res, scale=scale,
                               center=center, wmax=100.0, n_threads=nthr)
        cdt = time.perf_counter() - t0
        Tn = np.zeros((R, R, R)) + tdist
        Wn = np.zeros((R, R, R))
        t0 = time.perf_counter()
        O.fuse_depths(depths_np[0], lws[0], K, Kinv, Tn, Wn, tdist, tsdf_res=tsdf_res, scale=scale, center=center, wmax=100.0)
        ndt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": R ** 3 * cycles * len(lws) / cdt / 1e6, "unit": "Mvoxels/s", "cores": nthr,
                               "kind": "port",
                               "sample": "%d sweeps of %d^3 (the %d bench views x %d) with oracle/oracle_c.c, the C "
                                         "restatement of fuseDepths' CPU path (OpenMP, %d threads, fp64); numpy "
                                         "restatement on 1 core: %.1f Mvox/s; the reference's interpreter loop itself: "
                                         "0.0835 Mvox/s (BASELINE.md)" % (cycles * len(lws), R, len(lws), cycles, nthr,
                                                                          R ** 3 / ndt / 1e6),
                               # parity spot check: same update mask as the GPU volume after the same views
                               "mask_match": bool(np.array_equal(Wo > 0, (Wt > 0).cpu().numpy()))}
    del T, Wt
    k1_512_failed = False
    if world == 1 and not args.no_k1_512:
        try:
            out["k1_512"] = k1_512_leg(torch, kernels, scene)
        except Exception as e:
            out["k1_512"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
            k1_512_failed = True

    # The headline (K1) is measured; the secondary legs below run collectives.  If a rank fails inside one of them its peers would
    # wait in a collective for ever, so every rank runs a watchdog thread: it ends the rank when the legs' time limit passes OR when
    # another rank has posted "failed" in the process group's store (looked at twice a second), rank 0 printing the line with
    # what it has.  A run with a failed or timed-out leg still prints its line but EXITS NON-ZERO (3): a stalled collective must
    # not look like a clean run to whoever launched it.  (One rank: exceptions are caught, the line is printed, exit code 3.)
    import threading
    emit_lock = threading.Lock()
    EXIT_LEG_FAILED = 3
    store = None
    if distributed:
        try:
            store = dist.distributed_c10d._get_default_store()
        except Exception:
            store = None

    def emit_and_leave(bad=False, why=None):
        with emit_lock:                                   # (the watchdog thread and the main thread must not both print)
            if bad:
                for leg in ("gn", "frame"):
                    if leg not in out and not (args.no_gn or (leg == "frame" and args.no_frame)):
                        out[leg] = {"error": why or "did not finish within %d s (a rank failed or a collective stalled)" % args.leg_timeout}
            if rank == 0:
                print(json.dumps(out))
                sys.stdout.flush()
            if bad:
                sys.stderr.flush()
                os._exit(EXIT_LEG_FAILED)                 # (not sys.exit: the main thread may sit inside a collective)

    # (a watchdog THREAD, not SIGALRM: a rank stuck inside a collective is inside a C call, where Python runs no signal handler --
    # but torch releases the GIL there, so a thread does run)
    legs_done = threading.Event()

    def watch():
        deadline = time.time() + args.leg_timeout
        while not legs_done.wait(0.5):
            if time.time() > deadline:
                emit_and_leave(bad=True)
            try:
                if store is not None and store.check(["bench_leg_failed"]):
                    emit_and_leave(bad=True, why="another rank failed in this leg (see its stderr)")
            except Exception:
                pass

    if distributed and args.leg_timeout > 0:
        threading.Thread(target=watch, daemon=True).start()
    failed = k1_512_failed
    if not args.no_gn:
        try:
            out["gn"] = gn_leg(args, torch, dist, scene, rank, world, barrier)
        except Exception as e:                        # a failure here must not cost the headline line
            out["gn"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
            failed = True
        if not args.no_frame and not (failed and distributed):
            try:
                out["frame"] = frame_leg(args, torch, dist, scene, rank, world, barrier, with_k23=True)
                k23 = out["frame"].pop("k23", None)
                if k23 is not None:
                    out["k23"] = {"workload": "TSDF -> TSDF fusion of the frame loop's live volume into its canonical volume: K2 = "
                                              "FusionDM.updateTSDF (rigid), K3 = Fusion.updateTSDF (DQB warp field), float32 volumes, HIP events "
                                              "around 10 calls; bytes = SURVEY 8(d): 20 B per voxel; PMC traffic: profiles/ (DESIGN.md section 5)",
                                  "%d^3_%d_nodes" % (args.res, args.gn_nodes): k23}
            except Exception as e:                    # the composed leg must never cost the headline line
                out["frame"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
                failed = True
            if world == 1 and not args.no_frame_512 and args.res == 256 and not failed:
                # BASELINE config 5 on ONE GPU: 512^3, 8 views of 1280x720, 2 048 nodes (the 8-GPU run is the driver's)
                try:
                    import copy
                    a5 = copy.copy(args)
                    a5.res, a5.gn_nodes = 512, 2048
                    f5 = frame_leg(a5, torch, dist, scene, rank, world, barrier, nframes=5, with_k23=True)
                    k23 = f5.pop("k23", None)
                    if k23 is not None and "k23" in out:
                        out["k23"]["512^3_2048_nodes"] = k23
                    out["frame_512"] = f5
                except Exception as e:
                    out["frame_512"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
                    failed = True
    if distributed and failed:
        # this rank's collectives no longer match its peers': do not enter another one.  Tell the peers (their watchdogs look at
        # the store), give rank 0 a moment to print its line before a launcher that kills the job at the first non-zero exit
        # sees ours, and leave.
        try:
            if store is not None:
                store.set("bench_leg_failed", str(rank))
        except Exception:
            pass
        if rank != 0:
            time.sleep(3.0)
        emit_and_leave(bad=True, why="not run: an earlier leg failed on this rank")
    legs_done.set()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    emit_and_leave(bad=failed)


if __name__ == "__main__":
    main()
