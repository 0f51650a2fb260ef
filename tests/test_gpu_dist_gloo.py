"""GPU, 2 ranks sharing cuda:0 over gloo (RCCL refuses two ranks on one device; the collective
semantics are the same): canonical samples sharded by axis-0 slab, block pattern unioned, normal
equations all-reduced every GN iteration.  Must reproduce the non-distributed solver run over the
union of the same samples: identical pattern, system to 1e-12 relative, identical iterates."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        from dynamicfusion_body_amd import FusionDM, scene, kernels
        from dynamicfusion_body_amd import dist as D
        from dynamicfusion_body_amd.pipeline import FrameSolver, extract_surface_samples
        from dynamicfusion_body_amd.solve import WarpSolver
        torch.cuda.set_device(0)
        R, N, k = 64, 40, 4
        H, W, fx, cx, cy = scene.CAMERAS["C1"]
        K = scene.intrinsics(fx, cx, cy); Kinv = np.linalg.inv(K)
        scale, center, tdist = scene.grid_params(R)
        a, b = D.slab_range(R, rank, ws)
        # TSDF integration needs no exchange: every rank sweeps only its own slab
        T = torch.full((b - a, R, R), tdist, dtype=torch.float32, device="cuda"); Wt = torch.zeros_like(T)
        Tf = torch.full((R, R, R), tdist, dtype=torch.float32, device="cuda"); Wf = torch.zeros_like(Tf)
        for ang in (0.0, 40.0, -40.0):
            lw = scene.view_extrinsic(ang)
            d = torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda()
            kernels.integrate_depth(T, Wt, d, K, Kinv, lw, scale, center, tdist, tsdf_res=R, res=(R, R, R), x_range=(a, b))
            kernels.integrate_depth(Tf, Wf, d, K, Kinv, lw, scale, center, tdist)
        assert torch.equal(T, Tf[a:b]) and torch.equal(Wt, Wf[a:b])
        node_pos, node_w = scene.fibonacci_nodes(N, R)
        ident = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0, 0]), (N, 1))
        fs = FrameSolver(K, scale, center, R / 2, knn=k, pcg_iters=300)
        fs.set_graph(node_pos, ident, node_w)
        S = fs.set_canonical(T, Wt, band=2.0, x0=a)
        lw_cam = scene.view_extrinsic(0.0)
        live = scene.render_depth(K, lw_cam, H, W, dtype=np.float32, sphere_offset=np.array([0.6, -0.4, 0.3]) * scale,
                                  sphere_r=scene.SPHERE_R * 1.02)
        depth = torch.from_numpy(live).cuda()
        # reference: every rank also solves the union of all slabs' samples without collectives
        pos, nrm = extract_surface_samples(T, Wt, 2.0, x0=a)
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(ws)]
        dist.all_gather(sizes, torch.tensor([pos.shape[0]], dtype=torch.int64))
        m = int(max(int(s_) for s_ in sizes))
        buf = torch.zeros((m, 6), dtype=torch.float64)
        buf[:pos.shape[0], :3] = pos.cpu(); buf[:pos.shape[0], 3:] = nrm.cpu()
        parts = [torch.zeros_like(buf) for _ in range(ws)]
        dist.all_gather(parts, buf)
        allp = torch.cat([p_[:int(n_)] for p_, n_ in zip(parts, sizes)]).cuda()
        ref = FrameSolver(K, scale, center, R / 2, knn=k, pcg_iters=300)
        ref.solver = WarpSolver(knn=k, pcg_iters=300, distributed=False)
        ref.set_graph(node_pos, ident, node_w)
        ref.solver.set_samples(allp[:, :3].contiguous(), allp[:, 3:].contiguous())
        assert ref.solver.S == sum(int(s_) for s_ in sizes) and S == int(sizes[rank])
        for it in range(3):
            fs.gn_iteration(depth, lw_cam, rw=0.05, lm_abs=1e-2, lm_rel=1e-2, max_dist=4.0)
            ref.gn_iteration(depth, lw_cam, rw=0.05, lm_abs=1e-2, lm_rel=1e-2, max_dist=4.0)
            if it == 0:
                assert torch.equal(fs.solver.row_ptr, ref.solver.row_ptr) and torch.equal(fs.solver.col, ref.solver.col)
            c1, n1 = fs.solver.cost(); c2, n2 = ref.solver.cost()
            dv = float((fs.solver.vals - ref.solver.vals).abs().max() / ref.solver.vals.abs().max())
            dr = float((fs.solver.rhs - ref.solver.rhs).abs().max() / ref.solver.rhs.abs().max())
            dx = float((fs.solver.dx - ref.solver.dx).abs().max() / ref.solver.dx.abs().max())
            dq = float((fs.solver.node_dq - ref.solver.node_dq).abs().max())
            if it == 0:
                # same iterate on both sides: the system agrees to summation order
                assert n1 == n2 and n1 > 100, (n1, n2)
                assert abs(c1 - c2) <= 1e-11 * c2 and dv <= 1e-10 and dr <= 1e-10, (c1, c2, dv, dr)
            # 300 PCG iterations on 240 unknowns: converged, so the step is a well-defined function of the
            # system and the iterates stay together (a truncated PCG amplifies summation-order noise)
            assert dx <= 1e-6 and dq <= 1e-7, (it, dx, dq)
            assert abs(c1 - c2) <= 1e-7 * c2 and abs(n1 - n2) <= 1, (it, c1, c2, n1, n2)
        out[rank] = 1
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_match_single_process():
    ws = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Array("i", [0] * ws)
    procs = [ctx.Process(target=_worker, args=(r, ws, port, out)) for r in range(ws)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert all(p.exitcode == 0 for p in procs)
    assert list(out) == [1] * ws


def _frame_worker(rank, ws, port, out, sphere_r=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_WORLD_SIZE"] = str(ws)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        from dynamicfusion_body_amd import scene
        from dynamicfusion_body_amd import dist as D
        from dynamicfusion_body_amd.pipeline import SlabFrame
        torch.cuda.set_device(0)
        R, N = 63, 40                                  # 63 planes: uneven slabs (32 + 31), padded all-gather
        H, W, fx, cx, cy = scene.CAMERAS["C1"]
        K = scene.intrinsics(fx, cx, cy)
        scale, center, tdist = scene.grid_params(R)
        node_pos, node_w = scene.fibonacci_nodes(N, R)
        lw_cam = scene.view_extrinsic(0.0)
        frames = []
        sr = scene.SPHERE_R if sphere_r is None else sphere_r
        for f in range(2):
            off = np.array([0.3, -0.2, 0.15]) * (f + 1) * scale
            frames.append(torch.from_numpy(scene.render_depth(K, lw_cam, H, W, dtype=np.float32, sphere_offset=off,
                                                              sphere_r=sr * 1.01, wall_z=None if sphere_r is not None else scene.WALL_Z)).cuda())
        res = {}
        for mode in ("sharded", "replicated", "whole"):
            sf = SlabFrame(K, scale, center, R, tdist / scale, node_pos, node_w, knn=4, pcg_iters=300, band=2.0,
                           distributed=(mode != "whole"), solve_mode=mode if mode != "whole" else "auto")
            for ang in (0.0, 40.0, -40.0):
                lw = scene.view_extrinsic(ang)
                d = torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0, sphere_r=sr,
                                                        wall_z=None if sphere_r is not None else scene.WALL_Z)).cuda()
                sf.integrate(d, lw)
            n_s = sf.refresh_samples()
            if mode == "sharded":
                n_mine = n_s
            for d in frames:
                sf.step(d, lw_cam, gn_iters=3, lm_abs=1.0, global_stride=1)
            res[mode] = sf
        a, b = res["sharded"].a, res["sharded"].b
        assert (a, b) == D.slab_range(R, rank, ws) and (res["whole"].a, res["whole"].b) == (0, R)
        assert res["sharded"].solve_mode == "sharded" and res["replicated"].solve_mode == "replicated"
        Ts, Tw = res["sharded"].T, res["whole"].T[a:b]
        Ws, Ww = res["sharded"].Wt, res["whole"].Wt[a:b]
        dq_s, dq_w = res["sharded"].fs.solver.node_dq, res["whole"].fs.solver.node_dq
        # the replicated solve (slabs' samples all-gathered once per frame, no per-iteration collective) IS the whole-grid solve:
        # the same samples in the same order -> the same bits, on every rank
        rp = res["replicated"]
        assert rp.fs.solver.S == res["whole"].fs.solver.S
        rep_exact = bool(torch.equal(rp.fs.solver.node_dq, dq_w) and torch.equal(rp.T, Tw) and torch.equal(rp.Wt, Ww))
        out.put((rank, float((Ts - Tw).abs().max()), float((Ws - Ww).abs().max()), float(((Ws > 0) != (Ww > 0)).float().mean()),
                 float((dq_s - dq_w).abs().max()), float((Tw - tdist / scale).abs().max()),
                 None if rep_exact else "replicated solve differs from the whole-grid run: dq %.3g, T %.3g" %
                 (float((rp.fs.solver.node_dq - dq_w).abs().max()), float((rp.T - Tw).abs().max())), n_mine))
    except Exception as e:                                                  # pragma: no cover
        import traceback
        out.put((rank, 0, 0, 0, 0, 0, traceback.format_exc(), -1))
    finally:
        dist.destroy_process_group()


def test_slab_frame_two_ranks():
    """The per-frame loop with the canonical volume in two slabs (live slabs all-gathered, face planes exchanged
    as a halo for the sample normals, normal equations all-reduced) reproduces the whole-grid run: warp field to
    1e-6, canonical slab to 1e-4 voxel (the only difference left is the summation order of the all-reduce)."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_frame_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = [out.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, dT, dW, dmask, ddq, moved, err, n_mine in got:
        assert err is None, err
        assert ddq < 1e-6 and dT < 1e-4 and dW < 1e-4 and dmask < 1e-4
        assert moved > 1.0                                                   # the frames really changed the canonical volume


def test_slab_frame_with_ranks_that_own_no_surface():
    """Three slabs, a small sphere that lies entirely inside the middle one: ranks 0 and 2 have no samples at all (no rows, no
    plan, an all-zero contribution to the all-reduce) and must still take every collective the middle rank takes -- the 8-GPU job
    at 256^3 has such ranks (the sphere spans planes 48..208 of 256).  Same result as the whole-grid run."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_frame_worker, args=(r, 3, port, out, 0.15)) for r in range(3)]
    for p in procs:
        p.start()
    got = [out.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    counts = {}
    for rank, dT, dW, dmask, ddq, moved, err, n_mine in got:
        assert err is None, err
        counts[rank] = n_mine
        assert ddq < 1e-6 and dT < 1e-4 and dW < 1e-4 and dmask < 1e-4
    assert counts[0] == 0 and counts[2] == 0 and counts[1] > 500


def _frame_worker_full(rank, ws, port, out):
    """The bench's frame scene at its benched size (256^3, 512 nodes, three 640x480 views per frame, 10 GN x 10 PCG iterations,
    band 4): the kernels the bench runs -- 16-byte-pack sample extraction on a halo-padded slab, the multi-view FRESH column sweep
    on a slab, K3's LDS kernel + redo list with a slab offset and a level-2 workspace."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_WORLD_SIZE"] = str(ws)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        from dynamicfusion_body_amd import scene
        from dynamicfusion_body_amd import dist as D
        from dynamicfusion_body_amd.pipeline import SlabFrame
        torch.cuda.set_device(0)
        R, N = 256, 512
        H, W, fx, cx, cy = scene.CAMERAS["C2"]
        K = scene.intrinsics(fx, cx, cy)
        scale, center, tdist = scene.grid_params(R)
        node_pos, node_w = scene.fibonacci_nodes(N, R)
        angles = (0.0, 40.0, -40.0)
        lws = [scene.view_extrinsic(a) for a in angles]
        canon = [torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda() for lw in lws]
        frames = []
        for f in range(3):
            off = np.array([0.10, -0.07, 0.05]) * (f + 1) * scale
            frames.append([torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0, sphere_offset=off,
                                                               sphere_r=scene.SPHERE_R * (1.0 + 0.004 * (f + 1)))).cuda() for lw in lws])
        res = {}
        for mode in ("sharded", "replicated", "whole"):
            sf = SlabFrame(K, scale, center, R, tdist / scale, node_pos, node_w, knn=4, pcg_iters=10, band=4.0,
                           distributed=(mode != "whole"), solve_mode=mode if mode != "whole" else "auto")
            for d, lw in zip(canon, lws):
                sf.integrate(d, lw)
            sf.refresh_samples()
            counts = []
            for fr in frames:
                counts.append(sf.step(fr, lws, gn_iters=10, global_stride=1))    # (stride 1: the rigid-mode step over ALL samples is partition-independent)
            res[mode] = (sf, counts)
        (sh, n_sh), (rp, n_rp), (wh, n_wh) = res["sharded"], res["replicated"], res["whole"]
        a, b = sh.a, sh.b
        assert (a, b) == D.slab_range(R, rank, ws) and (rp.a, rp.b) == (a, b) and (wh.a, wh.b) == (0, R)
        assert sh.solve_mode == "sharded" and rp.solve_mode == "replicated"
        Tw, Ww, dq_w = wh.T[a:b], wh.Wt[a:b], wh.fs.solver.node_dq
        svr, svw = rp.fs.solver, wh.fs.solver
        rep = {"counts": n_rp == n_wh, "dq": bool(torch.equal(svr.node_dq, dq_w)), "T": bool(torch.equal(rp.T, Tw)),
               "W": bool(torch.equal(rp.Wt, Ww)), "samples": bool(svr.S == svw.S and torch.equal(svr.spos, svw.spos) and
                                                                  torch.equal(svr.snrm, svw.snrm) and torch.equal(svr.snbr, svw.snbr))}
        tot = torch.tensor([float(n_sh[-1])], dtype=torch.float64)
        dist.all_reduce(tot)
        shd = {"dq": float((sh.fs.solver.node_dq - dq_w).abs().max()), "T": float((sh.T - Tw).abs().max()),
               "T_off": int(((sh.T - Tw).abs() > 1e-3).sum()), "W": float((sh.Wt - Ww).abs().max()),
               "samples": abs(int(tot.item()) - n_wh[-1]) / n_wh[-1]}
        out.put((rank, rep, shd, n_wh[-1], float((Tw - tdist / scale).abs().max()), None))
    except Exception:                                                       # pragma: no cover
        import traceback
        out.put((rank, None, None, 0, 0.0, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_slab_frame_two_ranks_at_the_benched_size():
    """Round-3 verdict item 1: two ranks at 256^3 / 512 nodes (Z % 64 == 0: the vectorised extraction, the multi-view fresh column
    sweep and K3's LDS kernel on slabs -- none of which the 63^3 test above reaches).  The replicated solve must give the
    whole-grid run's BITS after three frames (canonical slab, weights, sample set, node DQs); the sharded solve differs by the
    summation order of its all-reduce only, amplified by the truncated PCG (warp field 1e-4, a handful of voxels, sample count 0.1 %).  Both runs of a worker use the
    same PCG launch shape (ranks that share a GPU take the two-launch PCG, in every solver of the process): the 0.7 % cost gap
    between round 3's 2-rank rehearsal and its 1-GPU bench line was the persistent single-reduction PCG against the two-launch
    one over 14 chaotic frames, not a slab stage (tools/slab_bisect.py: every stage bit-identical on slabs)."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_frame_worker_full, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = [out.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(120)
    for rank, rep, shd, n_whole, moved, err in got:
        assert err is None, err
        assert n_whole > 200000 and moved > 1.0
        assert all(rep.values()), (rank, rep)
        # (ten TRUNCATED PCG iterations per GN iteration amplify the all-reduce's summation order: 1e-16 grows to ~1e-6 in the warp
        # field over three frames, and a voxel whose update decision or float32 rounding sits that close to its threshold flips)
        assert shd["dq"] < 1e-4 and shd["T_off"] <= 64 and shd["T"] < 1.0 and shd["W"] < 1.0 and shd["samples"] < 1e-3, (rank, shd)


def _config4_worker(rank, ws, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_WORLD_SIZE"] = str(ws)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        from dynamicfusion_body_amd import scene, kernels
        from dynamicfusion_body_amd import dist as D
        from dynamicfusion_body_amd.pipeline import FrameSolver
        torch.cuda.set_device(0)
        R, N, k = 512, 2048, 4
        H, W, fx, cx, cy = scene.CAMERAS["C5"]
        K = scene.intrinsics(fx, cx, cy); Kinv = np.linalg.inv(K)
        scale, center, tdist = scene.grid_params(R)
        T = torch.full((R, R, R), tdist, dtype=torch.float32, device="cuda"); Wt = torch.zeros_like(T)
        for ang in (0.0, 40.0, -40.0):
            lw = scene.view_extrinsic(ang)
            d = torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda()
            kernels.integrate_depth(T, Wt, d, K, Kinv, lw, scale, center, tdist)
        node_pos, node_w = scene.fibonacci_nodes(N, R)
        ident = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0, 0]), (N, 1))
        rng = np.random.default_rng(7)
        from oracle import gn_np as G
        dq0 = torch.from_numpy(G.apply_twists(ident, rng.normal(scale=[2e-3] * 3 + [0.15] * 3, size=(N, 6)))).cuda()
        lw_cam = scene.view_extrinsic(0.0)
        depth = torch.from_numpy(scene.render_depth(K, lw_cam, H, W, dtype=np.float32, sphere_offset=np.array([0.6, -0.4, 0.3]) * scale,
                                                    sphere_r=scene.SPHERE_R * 1.005)).cuda()
        a, b = D.slab_range(R, rank, ws)
        lo, hi = max(a - 1, 0), min(b + 1, R)
        Wp = Wt[lo:hi].clone()
        if lo < a:
            Wp[0] = 0
        if hi > b:
            Wp[-1] = 0
        systems = {}
        for mode in ("sharded", "whole"):
            fs = FrameSolver(K, scale, center, R / 2, knn=k, pcg_iters=10, distributed=(mode == "sharded"))
            fs.set_graph(node_pos, ident, node_w)
            if mode == "sharded":
                S = fs.set_canonical(T[lo:hi].contiguous(), Wp, band=4.0, x0=lo)
            else:
                S = fs.set_canonical(T, Wt, band=4.0)
            sv = fs.solver
            sv.node_dq.copy_(dq0)
            sv.associate_depth(depth, fs.K, fs.Kinv, lw_cam, scale, center, R / 2, fs.lw, 2.0)
            sv.build(fs.lw, 5.0, 0.5)                         # sharded: all-reduce of the flat system inside
            c, n = sv.cost()
            systems[mode] = (S, sv._pattern_keys.clone(), sv.vals.clone(), sv.rhs.clone(), c, n)
        tot = torch.tensor([float(systems["sharded"][0])], dtype=torch.float64)
        dist.all_reduce(tot)
        Ss, ks, vs, rs, cs, ns = systems["sharded"]
        Sw, kw, vw, rw_, cw, nw = systems["whole"]
        same_pattern = bool(ks.numel() == kw.numel() and torch.equal(ks, kw))
        dv = float((vs - vw).abs().max() / vw.abs().max()) if same_pattern else -1.0
        dr = float((rs - rw_).abs().max() / rw_.abs().max())
        out.put((rank, int(tot.item()), Sw, same_pattern, dv, dr, abs(cs - cw) / cw, ns, nw, None))
    except Exception:                                                       # pragma: no cover
        import traceback
        out.put((rank, 0, 0, False, 0, 0, 0, 0, 0, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_config4_two_rank_split():
    """BASELINE config 4 (512^3, 2 048 nodes) cut into two axis-0 slabs: each rank builds the Huber-weighted normal
    equations of its own slab's samples, ONE all-reduce of the flat buffer sums them, and the result is the whole-grid
    system: same sample union, same block pattern, blocks / J^T r / cost to summation order (1e-11 relative)."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_config4_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = [out.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(120)
    for rank, s_union, s_whole, same_pattern, dv, dr, dc, n_valid, n_valid_whole, err in got:
        assert err is None, err
        assert s_union == s_whole and s_whole > 500000
        assert n_valid == n_valid_whole and n_valid > 100000                # the all-reduced count is the whole grid's
        assert same_pattern
        assert 0 <= dv <= 1e-11 and dr <= 1e-11 and dc <= 1e-12


def test_bench_with_a_failed_leg_exits_non_zero_and_still_prints_its_line():
    """bench.py --gpus 2 with an injected failure of rank 1 inside the warp-solve leg (DFH_TEST_FAIL_GN_RANK): the headline line is
    still printed by rank 0 (with an `error` in the failed leg), every rank leaves within seconds -- the failed rank posts its
    state in the process group's store, the others' watchdogs look there twice a second -- and the command's exit code is NOT 0:
    a stalled collective must not look like a clean run to whoever launched it (round-2 ADVICE)."""
    import json
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DFH_TEST_FAIL_GN_RANK="1")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline",
                        "--leg-timeout", "120", "--res", "64", "--gn-nodes", "48"], env=env, capture_output=True, text=True, timeout=280)
    took = time.time() - t0
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode != 0, (r.returncode, r.stderr[-400:])
    assert len(lines) == 1, (r.stdout[-400:], r.stderr[-400:])
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["n_gpus"] == 2
    assert "error" in d["gn"]
    assert took < 110, took                       # (well inside the legs' own time limit: the ranks told each other)


def test_one_rank_rccl_sharded_iteration_captures_into_a_graph():
    """The sharded solve's collective path over RCCL, as far as one GPU can take it (round-2 verdict item 4b): a child process
    with a 1-rank "nccl" group runs config 3's GN iterations as build -> pack upper triangle -> all_reduce -> unpack -> solve
    (WarpSolver.force_collective), eager and captured into a hipGraph with the collective inside.  Asserted: the capture
    succeeds, the replay gives the eager bits, and both give the bits of the single-GPU one-call iteration."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_capture_check.py"), "--res", "128", "--nodes", "128", "--solves", "2"],
                       env=env, capture_output=True, text=True, timeout=280)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-300:], r.stderr[-600:])
    d = json.loads(lines[0])
    assert d["backend"] == "nccl" and d["world_size"] == 1
    forced, single = d["two_calls_with_all_reduce"], d["single_gpu_one_call"]
    assert forced["captured"] and forced["graph_equals_eager"], forced
    assert single["captured"] and single["graph_equals_eager"], single
    assert d["collective_path_equals_single_gpu"] and d["max_abs_diff"] == 0.0
    assert d["packed_doubles"] < 0.62 * d["system_doubles"]               # the upper block triangle is what travels
