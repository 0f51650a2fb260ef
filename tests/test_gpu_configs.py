"""GPU: the BASELINE.json configurations at FULL size, with the settings the bench and the frame loop ship.

  config 3  256^3 canonical volume, 512-node warp field, 10 GN iterations with the benched settings (10 truncated PCG
            iterations, Huber IRLS delta 0.5, rw 5, lm_abs 10, lm_rel 1e-2, 2-voxel association gate): cost at every
            build against oracle/gn_np.gn_loop_truncated -- the same loop on the CPU with the SAME truncated
            Chronopoulos-Gear PCG -- to 1e-4 relative (the north-star residual bar); the gap to the exactly solved
            oracle loop is measured and recorded.  The oracle's costs are written to gpurun_out/ and committed as
            tests/golden/config3_oracle_costs.json, which bench.py quotes beside its own `gn.final_cost`.
  config 4  512^3, 2 048 nodes: one Huber-weighted build and 10 PCG iterations against the block-sparse oracle
            assembly (the 2-rank split of it: tests/test_gpu_dist_gloo.py::test_config4_two_rank_split).
  config 5  512^3, 8 views of 1280x720 (45 degrees apart): the one-sweep multi-view kernel against eight consecutive
            sweeps (bit-identical), and a 30-frame sequence of the composed frame loop.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import gn_np as G
from dynamicfusion_body_amd import kernels, scene
from dynamicfusion_body_amd.pipeline import FrameSolver, SlabFrame

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IDENT = np.array([1.0, 0, 0, 0, 0, 0, 0, 0])


def canonical(R, cam, angles=(0.0, 40.0, -40.0)):
    H, W, fx, cx, cy = scene.CAMERAS[cam]
    K = scene.intrinsics(fx, cx, cy)
    Kinv = np.linalg.inv(K)
    scale, center, tdist = scene.grid_params(R)
    T = torch.full((R, R, R), tdist, dtype=torch.float32, device="cuda")
    Wt = torch.zeros_like(T)
    for a in angles:
        lw = scene.view_extrinsic(a)
        d = torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda()
        kernels.integrate_depth(T, Wt, d, K, Kinv, lw, scale, center, tdist)
    return K, Kinv, (H, W), scale, center, tdist, T, Wt


def bench_problem(R, N, cam, pcg_iters=10, inflate=1.02):
    """The problem of bench.py's gn leg: canonical volume from three views, Fibonacci nodes, band-4 samples, the live
    frame = the sphere displaced by (0.6, -0.4, 0.3) voxels and inflated by 2 %."""
    K, Kinv, (H, W), scale, center, tdist, T, Wt = canonical(R, cam)
    fs = FrameSolver(K, scale, center, R / 2, knn=4, pcg_iters=pcg_iters, distributed=False)
    node_pos, node_w = scene.fibonacci_nodes(N, R)
    ident = np.tile(IDENT, (N, 1))
    fs.set_graph(node_pos, ident, node_w)
    S = fs.set_canonical(T, Wt, band=4.0)
    lw_cam = scene.view_extrinsic(0.0)
    live = scene.render_depth(K, lw_cam, H, W, dtype=np.float32, sphere_offset=np.array([0.6, -0.4, 0.3]) * scale,
                              sphere_r=scene.SPHERE_R * inflate)
    return fs, S, node_pos, node_w, ident, lw_cam, live, (K, Kinv, scale, center)


def host_arrays(sv):
    return (sv.spos.cpu().numpy(), sv.snrm.cpu().numpy(), sv.snbr.cpu().numpy().astype(np.int64),
            sv.node_nbr.cpu().numpy().astype(np.int64))


def make_associate(K, Kinv, lw_cam, live, scale, center, half, max_dist):
    def associate(warped):
        co, vo = G.associate_depth(warped, K, Kinv, lw_cam, live, scale, center, half)
        d = co - warped
        vo = vo & ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]) <= max_dist * max_dist)
        return co, vo
    return associate


def test_config3_benched_settings_vs_truncated_pcg_oracle():
    R, N, iters = 256, 512, 10
    rw, lm_abs, lm_rel, max_dist, huber, pcg_iters = 5.0, 10.0, 1e-2, 2.0, 0.5, 10       # bench.py gn_leg / SlabFrame.step defaults
    fs, S, node_pos, node_w, ident, lw_cam, live, (K, Kinv, scale, center) = bench_problem(R, N, "C2", pcg_iters)
    assert S > 150000
    sv = fs.solver
    depth = torch.from_numpy(live).cuda()
    gpu_costs, gpu_counts = [], []
    for _ in range(iters):
        fs.gn_iteration(depth, lw_cam, rw=rw, lm_abs=lm_abs, lm_rel=lm_rel, max_dist=max_dist, huber=huber)
        c, n = sv.cost()
        gpu_costs.append(c); gpu_counts.append(n)
    dq_gpu = sv.node_dq.cpu().numpy()
    pos, nrm, nbr, node_nbr = host_arrays(sv)
    assoc = make_associate(K, Kinv, lw_cam, live, scale, center, R / 2, max_dist)
    or_costs, or_counts, dq_or = G.gn_loop_truncated(ident, pos, nrm, nbr, node_nbr, node_pos, node_w, IDENT, assoc, iters, rw,
                                                    lm_abs, lm_rel, huber, pcg_iters)
    ex_costs, ex_counts, dq_ex = G.gn_loop_truncated(ident, pos, nrm, nbr, node_nbr, node_pos, node_w, IDENT, assoc, iters, rw,
                                                    lm_abs, lm_rel, huber, pcg_iters, exact=True)
    rel = np.abs(np.array(gpu_costs) - np.array(or_costs)) / np.array(or_costs)
    gap = np.abs(np.array(or_costs) - np.array(ex_costs)) / np.array(ex_costs)
    rec = {"workload": "256^3 canonical volume, 512 Fibonacci nodes, band-4 samples, 10 GN iterations: pcg_iters 10, huber 0.5, "
                       "rw 5, lm_abs 10, lm_rel 1e-2, max_dist 2 (bench.py gn leg)", "samples": int(S),
           "gpu_costs": gpu_costs, "oracle_truncated_pcg_costs": or_costs, "oracle_exact_solve_costs": ex_costs,
           "gpu_valid": gpu_counts, "oracle_valid": or_counts, "max_rel_gpu_vs_truncated_oracle": float(rel.max()),
           "max_rel_truncated_vs_exact_oracle": float(gap.max()), "max_abs_dq_gpu_vs_oracle": float(np.abs(dq_gpu - dq_or).max()),
           "final_cost_oracle": or_costs[-1], "final_cost_oracle_exact_solve": ex_costs[-1]}
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(rec, open(os.path.join(ROOT, "gpurun_out", "config3_oracle_costs.json"), "w"), indent=1)
    except OSError:
        pass
    print(json.dumps(rec))
    # the north-star bar: residual (cost) to 1e-4 relative at every iteration, same truncated linear solve on both sides
    assert rel.max() <= 1e-4, rec
    assert max(abs(a - b) for a, b in zip(gpu_counts, or_counts)) <= 3            # a sample exactly on the gate may flip
    assert np.abs(dq_gpu - dq_or).max() <= 1e-5
    # the gate admits more samples as the field converges (42 k -> 74 k), so the cost PER VALID SAMPLE is what must fall
    assert gpu_costs[-1] / gpu_counts[-1] < 0.5 * gpu_costs[0] / gpu_counts[0]
    # the committed record the bench quotes must be this run's (regenerate it when settings change)
    gold = os.path.join(ROOT, "tests", "golden", "config3_oracle_costs.json")
    if os.path.exists(gold):
        g = json.load(open(gold))
        assert abs(g["final_cost_oracle"] - or_costs[-1]) <= 1e-6 * or_costs[-1]


def test_config4_build_and_pcg_vs_sparse_oracle():
    R, N = 512, 2048
    rw, lm_abs, lm_rel, max_dist, huber = 5.0, 10.0, 1e-2, 2.0, 0.5
    # (a 2 % inflation is 3.2 voxels at 512^3, beyond the 2-voxel gate: 0.5 % here)
    fs, S, node_pos, node_w, ident, lw_cam, live, (K, Kinv, scale, center) = bench_problem(R, N, "C5", 10, inflate=1.005)
    assert S > 500000
    sv = fs.solver
    depth = torch.from_numpy(live).cuda()
    # move off the identity so that Jacobians, regulariser and Huber weights are all exercised
    rng = np.random.default_rng(7)
    dq0 = G.apply_twists(ident, rng.normal(scale=[2e-3] * 3 + [0.15] * 3, size=(N, 6)))
    sv.node_dq.copy_(torch.from_numpy(dq0).cuda())
    sv.associate_depth(depth, fs.K, fs.Kinv, lw_cam, scale, center, R / 2, fs.lw, max_dist)
    sv.build(fs.lw, rw, huber)
    cost, cnt = sv.cost()
    keys_g = sv._pattern_keys.cpu().numpy()
    vals_g = sv.vals.cpu().numpy().reshape(-1, 6, 6)
    rhs_g = sv.rhs.cpu().numpy().reshape(N, 6)
    pos, nrm, nbr, node_nbr = host_arrays(sv)
    assoc = make_associate(K, Kinv, lw_cam, live, scale, center, R / 2, max_dist)
    from oracle import oracle_np as O
    warped = O.warp(pos, dq0[nbr], node_pos[nbr], node_w[nbr], m_lw=IDENT)
    co, vo = assoc(warped)
    assert np.array_equal(sv.valid.cpu().numpy().astype(bool), vo) and cnt == int(vo.sum()) and cnt > 100000
    sel = np.flatnonzero(vo)
    r, J = G.data_residual_jacobian(dq0, pos[sel], nrm[sel], co[sel], nbr[sel], node_pos, node_w, IDENT)
    sc, obj = G.huber_scale(r, huber)
    assert 0.02 < (sc < 1).mean() < 0.98                                          # the Huber weights are in play
    rho, nb, Ji, Jj = G.reg_residual_jacobian(dq0, np.arange(N), node_nbr, node_pos, node_w, rw)
    keys_o, blocks_o, Jtr_o, _ = G.assemble_blocks(N, r * sc, J * sc[:, None, None], nbr[sel], rho, nb, Ji, Jj)
    cost_o = obj + 0.5 * float(np.sum(rho * rho))
    assert abs(cost - cost_o) <= 1e-10 * cost_o
    # the device pattern contains every oracle block (it may hold more: all-zero head-room blocks)
    idx = np.searchsorted(keys_g, keys_o)
    assert np.array_equal(keys_g[idx], keys_o)
    dense = np.zeros_like(vals_g)
    dense[idx] = blocks_o
    scale_v = np.abs(blocks_o).max()
    assert np.abs(vals_g - dense).max() <= 1e-10 * scale_v
    assert np.abs(rhs_g - Jtr_o).max() <= 1e-10 * np.abs(Jtr_o).max()
    # ten iterations of the single-reduction PCG, persistent kernel (N = 2048 = its largest grid) vs the numpy recurrence
    sv.solve_linear(lm_abs, lm_rel)
    sv.check_status()
    x = sv.dx.cpu().numpy()
    xo = G.pcg_cg1(N, keys_o, blocks_o, Jtr_o, 10, lm_abs, lm_rel)
    assert np.isfinite(x).all() and np.abs(x - xo).max() <= 1e-8 * np.abs(xo).max()


def config5_views(K, H, W, n=8, **kw):
    lws = [scene.view_extrinsic(45.0 * v) for v in range(n)]
    return lws, [scene.render_depth(K, lw, H, W, dtype=np.float32, **kw) for lw in lws]


def test_config5_multiview_sweep_equals_eight_sweeps():
    R = 512
    H, W, fx, cx, cy = scene.CAMERAS["C5"]
    K = scene.intrinsics(fx, cx, cy)
    Kinv = np.linalg.inv(K)
    scale, center, tdist = scene.grid_params(R)
    lws, dms = config5_views(K, H, W)
    depths = [torch.from_numpy(d).cuda() for d in dms]
    Ta = torch.full((R, R, R), tdist, dtype=torch.float32, device="cuda"); Wa = torch.zeros_like(Ta)
    Tb, Wb = Ta.clone(), Wa.clone()
    for rep in range(2):                                            # second pass: non-trivial T, w going in
        for d, lw in zip(depths, lws):
            kernels.integrate_depth(Ta, Wa, d, K, Kinv, lw, scale, center, tdist)
        kernels.integrate_depth_views(Tb, Wb, depths, K, Kinv, lws, scale, center, tdist)
        assert torch.equal(Ta, Tb) and torch.equal(Wa, Wb)
    upd = float((Wa > 0).float().mean())
    assert 0.5 < upd < 0.99
    # the same in four slabs of uneven size
    Tc = torch.full((R, R, R), tdist, dtype=torch.float32, device="cuda"); Wc = torch.zeros_like(Tc)
    for rep in range(2):
        for a, b in ((0, 100), (100, 101), (101, 384), (384, 512)):
            Ts, Ws = Tc[a:b], Wc[a:b]                               # contiguous views of whole planes
            kernels.integrate_depth_views(Ts, Ws, depths, K, Kinv, lws, scale, center, tdist, tsdf_res=R, res=(R, R, R), x_range=(a, b))
    assert torch.equal(Tc, Ta) and torch.equal(Wc, Wa)


def test_config5_thirty_frame_sequence():
    """512^3 grid, 2 048 nodes, 8 views of 1280x720 per frame, 30 frames of a deformation ~ sin(2 pi t / 30): the composed
    loop (live TSDF from all 8 views in one sweep -> 10 GN iterations -> DQB TSDF update -> sample refresh) follows the
    motion, stays finite and bounded, keeps its band."""
    R, N = 512, 2048
    H, W, fx, cx, cy = scene.CAMERAS["C5"]
    K = scene.intrinsics(fx, cx, cy)
    scale, center, tdist = scene.grid_params(R)
    node_pos, node_w = scene.fibonacci_nodes(N, R)
    lws = [scene.view_extrinsic(45.0 * v) for v in range(8)]
    sf = SlabFrame(K, scale, center, R, tdist / scale, node_pos, node_w, knn=4, pcg_iters=10, band=2.0, distributed=False)
    for lw in lws:
        sf.integrate(torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda(), lw)
    n0 = sf.refresh_samples()
    assert n0 > 400000
    amp = np.array([0.8, -0.5, 0.4])                                 # voxels
    front = node_pos[:, 2] < R / 2 - 0.25 * (scene.SPHERE_R / scale)  # nodes facing camera 0
    counts, tmax, err_front = [], [], []
    for t in range(30):
        off_vox = amp * np.sin(2 * np.pi * (t + 1) / 30.0)
        ds = [torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0, sphere_offset=off_vox * scale)).cuda()
              for lw in lws]
        counts.append(sf.step(ds, lws, gn_iters=10))
        dq = sf.fs.solver.node_dq.cpu().numpy()
        assert np.isfinite(dq).all()
        trans = 2.0 * G.qmul(dq[:, 4:], G.qconj(dq[:, :4]))[:, 1:]   # translation of every node's DQ (voxels)
        tmax.append(float(np.linalg.norm(trans, axis=1).max()))
        err_front.append(float(np.linalg.norm(trans[front].mean(axis=0) - off_vox)))
    print("config5 sequence: samples", counts, "max node translation", [round(t, 3) for t in tmax], "front-node tracking error",
          [round(e, 3) for e in err_front])
    # What follows the motion here is mostly the canonical VOLUME, not the warp field: Fusion.updateTSDF weighs the live sample
    # with wi = mean distance to the knn nodes (tens of voxels, core/fusion.py:180-190) against a canonical weight that starts
    # at the view count, so the first update already moves the canonical surface ~90 % of the way to the live one and the node
    # DQs only carry the rest (tools/track_probe.py: the front nodes' mean translation is 5-10 % of the true offset at any
    # regulariser / damping / PCG setting).  Reproduced reference semantics; asserted here: nothing runs away, the band stays
    # bounded (repeated re-averaging with shifted live surfaces thickens the |T| < 2 shell from 4 towards 6 voxels).
    assert all(np.isfinite(err_front))
    assert max(tmax) < 2.5
    assert np.linalg.norm(dq[:, 1:4], axis=1).max() < 0.02
    assert max(counts) < 1.7 * min(counts)
    cost, cnt = sf.fs.solver.cost()
    assert np.isfinite(cost) and cnt > 50000
    # the data term sees ALL eight views (round 2: the first only): with one view the samples camera 0 cannot see -- the far
    # side of the sphere, two thirds of the surface -- had no data row at all
    sv = sf.fs.solver
    half = R / 2
    sv.associate_depth(ds[0], sf.K, sf.Kinv, lws[0], sf.scale, sf.center, half, sf.fs.lw, 2.0)
    n_one = int(sv.valid.sum())
    sv.associate_depth(ds, sf.K, sf.Kinv, lws, sf.scale, sf.center, half, sf.fs.lw, 2.0)
    n_all = int(sv.valid.sum())
    print("config5: valid samples against view 0 only %d, against all 8 views %d (of %d)" % (n_one, n_all, sv.S))
    assert n_all > 2 * n_one and cnt >= 0.9 * n_all


def test_multi_view_association_and_gn_loop_vs_oracle():
    """Data term over several live views (dfh_gn_associate_views / dfh_gn_iteration_views) at R = 64, three views 50 degrees
    apart: the chosen correspondences (closest valid view, ties to the lower index) equal oracle/gn_np.associate_depth_views',
    one view through the same entry points equals dfh_gn_associate bit for bit, and six GN iterations with the benched
    settings follow the CPU loop (same truncated PCG) to 1e-4 relative in cost at every iteration -- with MORE valid samples
    than any single view gives."""
    R, N, iters = 64, 96, 6
    rw, lm_abs, lm_rel, max_dist, huber, pcg_iters = 5.0, 10.0, 1e-2, 2.0, 0.5, 10
    K, Kinv, (H, W), scale, center, tdist, T, Wt = canonical(R, "C1", angles=(0.0, 50.0, -50.0, 130.0, -130.0))
    fs = FrameSolver(K, scale, center, R / 2, knn=4, pcg_iters=pcg_iters, distributed=False)
    node_pos, node_w = scene.fibonacci_nodes(N, R)
    ident = np.tile(IDENT, (N, 1))
    fs.set_graph(node_pos, ident, node_w)
    S = fs.set_canonical(T, Wt, band=3.0)
    assert S > 3000
    off = np.array([0.5, -0.35, 0.25]) * scale
    lws = [scene.view_extrinsic(a) for a in (0.0, 50.0, -50.0)]
    lives = [scene.render_depth(K, lw, H, W, dtype=np.float32, sphere_offset=off, sphere_r=scene.SPHERE_R * 1.02) for lw in lws]
    depths = [torch.from_numpy(d).cuda() for d in lives]
    sv = fs.solver
    pos, nrm, nbr, node_nbr = host_arrays(sv)
    from oracle import oracle_np as O
    # association alone, off the identity
    rng = np.random.default_rng(11)
    dq0 = G.apply_twists(ident, rng.normal(scale=[2e-3] * 3 + [0.15] * 3, size=(N, 6)))
    sv.node_dq.copy_(torch.from_numpy(dq0).cuda())
    warped = O.warp(pos, dq0[nbr], node_pos[nbr], node_w[nbr], m_lw=IDENT)
    sv.associate_depth(depths, K, Kinv, lws, scale, center, R / 2, fs.lw, max_dist)
    co, vo, view = G.associate_depth_views(warped, K, Kinv, lws, lives, scale, center, R / 2, max_dist)
    assert np.array_equal(sv.valid.cpu().numpy().astype(bool), vo)
    assert np.abs(sv.corr.cpu().numpy() - co).max() <= 1e-9
    per_view = []
    for d, lw in zip(depths, lws):
        sv.associate_depth(d, K, Kinv, lw, scale, center, R / 2, fs.lw, max_dist)
        per_view.append((sv.corr.clone(), sv.valid.clone()))
        sv.associate_depth([d], K, Kinv, [lw], scale, center, R / 2, fs.lw, max_dist)      # one view as a list: same entry point as before
        assert torch.equal(sv.corr, per_view[-1][0]) and torch.equal(sv.valid, per_view[-1][1])
    assert int(vo.sum()) > max(int(v.sum()) for _, v in per_view) and len(set(view[vo])) == 3
    # the GN loop on all three views against the CPU loop
    sv.node_dq.copy_(torch.from_numpy(ident).cuda())

    def assoc(w):
        c, v, _ = G.associate_depth_views(w, K, Kinv, lws, lives, scale, center, R / 2, max_dist)
        return c, v
    gpu_costs, gpu_counts = [], []
    for _ in range(iters):
        fs.gn_iteration(depths, lws, rw=rw, lm_abs=lm_abs, lm_rel=lm_rel, max_dist=max_dist, huber=huber)
        c, n = sv.cost()
        gpu_costs.append(c); gpu_counts.append(n)
    or_costs, or_counts, dq_or = G.gn_loop_truncated(ident, pos, nrm, nbr, node_nbr, node_pos, node_w, IDENT, assoc, iters, rw,
                                                    lm_abs, lm_rel, huber, pcg_iters)
    rel = np.abs(np.array(gpu_costs) - np.array(or_costs)) / np.array(or_costs)
    assert rel.max() <= 1e-4, (gpu_costs, or_costs)
    assert max(abs(a - b) for a, b in zip(gpu_counts, or_counts)) <= 3
    assert np.abs(sv.node_dq.cpu().numpy() - dq_or).max() <= 1e-5
    assert gpu_costs[-1] / gpu_counts[-1] < gpu_costs[0] / gpu_counts[0]


def test_view_culling_per_tile_changes_nothing():
    """Config 5's data term: eight orbit views, 45 degrees apart.  The fused build drops, per 128-sample tile, the views none
    of the tile's samples can be valid in (csrc/dfh_solve.hip: tile_view_mask -- the tile's box projects outside the image, or
    onto pixels whose valid depths lie further than the gate from the box's depth range) before projecting a sample into them.
    Asserted: corr / valid, the normal equations and the cost are bit for bit those of the build that tries every view
    (option gn_no_view_cull) and corr / valid those of the stand-alone association kernel; five of the eight views win somewhere (the three behind the back wall see the wall, not the sphere)."""
    from dynamicfusion_body_amd import _lib
    R, N = 128, 256
    angles = tuple(45.0 * v for v in range(8))
    K, Kinv, (H, W), scale, center, tdist, T, Wt = canonical(R, "C2", angles=angles)
    fs = FrameSolver(K, scale, center, R / 2, knn=4, pcg_iters=10, distributed=False)
    node_pos, node_w = scene.fibonacci_nodes(N, R)
    ident = np.tile(IDENT, (N, 1))
    fs.set_graph(node_pos, ident, node_w)
    S = fs.set_canonical(T, Wt, band=4.0)
    assert S > 50000
    off = np.array([0.5, -0.35, 0.25]) * scale
    lws = [scene.view_extrinsic(a) for a in angles]
    lives = [scene.render_depth(K, lw, H, W, dtype=np.float32, sphere_offset=off, sphere_r=scene.SPHERE_R * 1.01) for lw in lws]
    depths = [torch.from_numpy(d).cuda() for d in lives]
    sv = fs.solver
    rng = np.random.default_rng(23)
    dq0 = G.apply_twists(ident, rng.normal(scale=[2e-3] * 3 + [0.15] * 3, size=(N, 6)))
    sv.node_dq.copy_(torch.from_numpy(dq0).cuda())
    max_dist = 2.0
    sv.associate_depth(depths, K, Kinv, lws, scale, center, R / 2, fs.lw, max_dist)            # every view, sample by sample
    ref_c, ref_v = sv.corr.clone(), sv.valid.clone()
    assert int(ref_v.sum()) > 20000
    outs = []
    for opt in (None, 1):
        _lib.set_option("gn_no_view_cull", opt)
        sv.corr.zero_(); sv.valid.zero_()
        sv.build_associated(depths, K, Kinv, lws, scale, center, R / 2, fs.lw, 5.0, max_dist, 0.5)
        c, n = sv.cost()
        assert torch.equal(sv.corr, ref_c) and torch.equal(sv.valid, ref_v), opt
        outs.append((sv.vals.clone(), sv.rhs.clone(), c, n))
    _lib.set_option("gn_no_view_cull", None)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and outs[0][2:] == outs[1][2:]
    # every one of the eight views is the closest valid one for some samples (the culling never removes a winner)
    pos, nrm, nbr, node_nbr = host_arrays(sv)
    from oracle import oracle_np as O
    sel = rng.choice(len(pos), size=4000, replace=False)
    warped = O.warp(pos[sel], dq0[nbr[sel]], node_pos[nbr[sel]], node_w[nbr[sel]], m_lw=IDENT)
    co, vo, view = G.associate_depth_views(warped, K, Kinv, lws, lives, scale, center, R / 2, max_dist)
    # (the views behind the back wall see the wall, not the sphere: they win nowhere -- and are dropped for every tile)
    assert np.array_equal(ref_v.cpu().numpy()[sel].astype(bool), vo) and len(set(view[vo])) >= 5


def _recovery_metrics(dqs, pos, nrm, nbr, valid, node_pos, node_w, truth):
    """How much of a known rigid translation `truth` (voxels) a warp field carries: the warped samples' displacement along
    their normals -- what a point-to-plane data term can see -- as a share of truth . n (samples with a valid correspondence and
    |truth . n| >= 0.2 voxel), and the whole displacement vector's projection on the truth."""
    from oracle import oracle_np as O
    disp = O.warp(pos, dqs[nbr], node_pos[nbr], node_w[nbr], m_lw=IDENT) - pos
    tn = nrm @ truth
    sel = valid & (np.abs(tn) >= 0.2)
    along = float(np.mean((disp[sel] * nrm[sel]).sum(axis=1) / tn[sel]))
    vec = float((disp[valid].mean(axis=0) @ truth) / (truth @ truth))
    left = float(np.sqrt(np.mean(((disp[sel] - truth) * nrm[sel]).sum(axis=1) ** 2)))
    return {"normal_share": along, "vector_share": vec, "rms_point_to_plane_left_voxel": left, "samples": int(sel.sum())}


def test_solve_recovers_a_known_translation():
    """Round-3 verdict item 6: how well does the shipped solve solve?  A static canonical sphere at 128^3; the live frame is the
    same sphere translated by (0.6, -0.4, 0.3) voxel, seen from three views; from the identity, the shipped frame loop's solve: two
    rigid-mode steps (dfh_gn_global_step) and ten GN iterations (10 truncated PCG iterations, Huber 0.5, rw 5, lm_abs 10, lm_rel 1e-2,
    2-voxel gate).  Reported and asserted: the share
    of the true displacement the warped SAMPLES carry along their normals (a point-to-plane term sees nothing else), for the GPU
    loop, for the numpy loop with the same truncated solve (must agree: 1e-6) and -- the achievable bound for this objective --
    for the numpy loop with the exact sparse solve run to convergence (40 iterations).  Written to gpurun_out/ and committed as
    tests/golden/solve_recovery.json; DESIGN.md section 6 quotes it."""
    R, N, iters = 128, 256, 10
    rw, lm_abs, lm_rel, max_dist, huber, pcg_iters = 5.0, 10.0, 1e-2, 2.0, 0.5, 10
    angles = (0.0, 40.0, -40.0)
    K, Kinv, (H, W), scale, center, tdist, T, Wt = canonical(R, "C2", angles=angles)
    fs = FrameSolver(K, scale, center, R / 2, knn=4, pcg_iters=pcg_iters, distributed=False)
    node_pos, node_w = scene.fibonacci_nodes(N, R)
    ident = np.tile(IDENT, (N, 1))
    fs.set_graph(node_pos, ident, node_w)
    S = fs.set_canonical(T, Wt, band=2.0)
    assert S > 10000
    truth = np.array([0.6, -0.4, 0.3])
    lws = [scene.view_extrinsic(a) for a in angles]
    lives = [scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0, sphere_offset=truth * scale) for lw in lws]
    depths = [torch.from_numpy(d).cuda() for d in lives]
    sv = fs.solver
    ident_t = torch.from_numpy(ident).cuda()
    # round 3's loop (node iterations only), then the shipped frame loop's (SlabFrame.step: two rigid-mode steps first)
    for _ in range(iters):
        fs.gn_iteration(depths, lws, rw=rw, lm_abs=lm_abs, lm_rel=lm_rel, max_dist=max_dist, huber=huber)
    sv.cost()
    valid_r3 = sv.valid.cpu().numpy().astype(bool)
    dq_r3 = sv.node_dq.cpu().numpy()
    sv.node_dq.copy_(ident_t)
    fs.global_iteration(depths, lws, rw=rw, max_dist=max_dist, huber=huber, lm_rel=0.1, n_iters=2)
    xi_gpu = sv.global_xi.cpu().numpy()
    gpu_costs = []
    for _ in range(iters):
        fs.gn_iteration(depths, lws, rw=rw, lm_abs=lm_abs, lm_rel=lm_rel, max_dist=max_dist, huber=huber)
        gpu_costs.append(sv.cost())
    pos, nrm, nbr, node_nbr = host_arrays(sv)
    valid_gpu = sv.valid.cpu().numpy().astype(bool)
    dq_gpu = sv.node_dq.cpu().numpy()

    def assoc(w):
        c, v, _ = G.associate_depth_views(w, K, Kinv, lws, lives, scale, center, R / 2, max_dist)
        return c, v
    _, _, dq_tr = G.gn_loop_truncated(ident, pos, nrm, nbr, node_nbr, node_pos, node_w, IDENT, assoc, iters, rw, lm_abs, lm_rel, huber, pcg_iters,
                                      global_iters=2, global_lm=0.1)
    ex_costs, ex_counts, dq_ex = G.gn_loop_truncated(ident, pos, nrm, nbr, node_nbr, node_pos, node_w, IDENT, assoc, 40, rw, lm_abs, lm_rel, huber,
                                                     pcg_iters, exact=True)
    from oracle import oracle_np as O
    _, valid_ex = assoc(O.warp(pos, dq_ex[nbr], node_pos[nbr], node_w[nbr], m_lw=IDENT))
    m_gpu = _recovery_metrics(dq_gpu, pos, nrm, nbr, valid_gpu, node_pos, node_w, truth)
    m_tr = _recovery_metrics(dq_tr, pos, nrm, nbr, valid_gpu, node_pos, node_w, truth)
    m_ex = _recovery_metrics(dq_ex, pos, nrm, nbr, valid_ex, node_pos, node_w, truth)
    m_0 = _recovery_metrics(ident, pos, nrm, nbr, valid_gpu, node_pos, node_w, truth)
    m_r3 = _recovery_metrics(dq_r3, pos, nrm, nbr, valid_r3, node_pos, node_w, truth)
    rec = {"workload": "128^3 static sphere, 256 Fibonacci nodes, band-2 samples, live = the sphere translated by (0.6, -0.4, 0.3) voxel, "
                       "3 views of 640x480; from the identity: 2 rigid-mode steps (lm 0.1) + 10 GN iterations: pcg_iters 10, huber 0.5, rw 5, "
                       "lm_abs 10, lm_rel 1e-2, max_dist 2 (SlabFrame.step's defaults)",
           "truth_voxel": truth.tolist(), "samples": int(S), "identity": m_0, "gpu_10_iterations_without_the_rigid_mode_steps": m_r3,
           "gpu_10_iterations": m_gpu, "numpy_same_truncated_solve": m_tr,
           "numpy_exact_solve_40_iterations": m_ex, "gpu_objective_first_last": [gpu_costs[0][0], gpu_costs[-1][0]],
           "gpu_valid_first_last": [gpu_costs[0][1], gpu_costs[-1][1]], "exact_objective_last": ex_costs[-1], "exact_valid_last": ex_counts[-1],
           "exact_objective_change_last_iteration": abs(ex_costs[-1] - ex_costs[-2]) / ex_costs[-1]}
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(rec, open(os.path.join(ROOT, "gpurun_out", "solve_recovery.json"), "w"), indent=1)
    except OSError:
        pass
    print(json.dumps(rec))
    assert abs(m_gpu["normal_share"] - m_tr["normal_share"]) <= 1e-6 and np.abs(dq_gpu - dq_tr).max() <= 1e-6, (m_gpu, m_tr, xi_gpu)
    assert rec["exact_objective_change_last_iteration"] < 1e-2                       # the exact loop has (nearly) converged
    assert abs(m_0["normal_share"]) < 1e-6
    # the shipped loop carries at least what the converged exact solve of the node iterations alone carries; round 3's (no
    # rigid-mode steps) carried 28 %
    assert m_gpu["normal_share"] >= m_ex["normal_share"] and m_gpu["normal_share"] > 0.8, rec
    assert m_r3["normal_share"] < 0.5 * m_gpu["normal_share"], rec
    assert m_gpu["rms_point_to_plane_left_voxel"] < 0.6 * m_0["rms_point_to_plane_left_voxel"], rec
    gold = os.path.join(ROOT, "tests", "golden", "solve_recovery.json")
    if os.path.exists(gold):
        g = json.load(open(gold))
        assert abs(g["gpu_10_iterations"]["normal_share"] - m_gpu["normal_share"]) <= 1e-3


def test_rigid_mode_step_variants_and_oracle():
    """dfh_gn_global_sampled_views (the frame loop's rigid-mode step: one twist for all nodes from the data rows) against its numpy
    restatement (oracle/gn_np.global_step_sampled: all tiles, and every 3rd tile), and against the step from the BUILT normal
    equations (dfh_gn_global_step) with the regulariser switched off -- the same system summed another way."""
    R, N = 96, 128
    K, Kinv, (H, W), scale, center, tdist, T, Wt = canonical(R, "C2")
    fs = FrameSolver(K, scale, center, R / 2, knn=4, pcg_iters=10, distributed=False)
    node_pos, node_w = scene.fibonacci_nodes(N, R)
    ident = np.tile(IDENT, (N, 1))
    fs.set_graph(node_pos, ident, node_w)
    S = fs.set_canonical(T, Wt, band=2.0)
    assert S > 5000
    lws = [scene.view_extrinsic(a) for a in (0.0, 40.0, -40.0)]
    lives = [scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0, sphere_offset=np.array([0.5, -0.3, 0.2]) * scale) for lw in lws]
    depths = [torch.from_numpy(d).cuda() for d in lives]
    sv = fs.solver
    rng = np.random.default_rng(4)
    dq0 = G.apply_twists(ident, rng.normal(scale=[1e-3] * 3 + [0.1] * 3, size=(N, 6)))
    pos, nrm, nbr, node_nbr = host_arrays(sv)

    def assoc(w):
        c, v, _ = G.associate_depth_views(w, K, Kinv, lws, lives, scale, center, R / 2, 2.0)
        return c, v
    for stride in (1, 3):
        sv.node_dq.copy_(torch.from_numpy(dq0).cuda())
        fs.global_iteration(depths, lws, max_dist=2.0, huber=0.5, lm_rel=0.1, n_iters=1, stride=stride)
        xi = sv.global_xi.cpu().numpy()
        dq_o, xi_o, n_o = G.global_step_sampled(dq0, pos, nrm, nbr, node_pos, node_w, IDENT, assoc, 0.5, 0.1, stride=stride)
        assert int(xi[7]) == n_o and n_o > 1000
        assert np.abs(xi[:6] - xi_o).max() <= 1e-9 * max(1.0, np.abs(xi_o).max()), (stride, xi[:6], xi_o)
        assert np.abs(sv.node_dq.cpu().numpy() - dq_o).max() <= 1e-9
        if stride == 1:
            xi_all = xi[:6].copy()
    assert np.linalg.norm(xi_all[3:]) > 0.05                                      # it does move: the live sphere is 0.6 voxel away
    # the built system without the regulariser holds the same sums
    sv.node_dq.copy_(torch.from_numpy(dq0).cuda())
    fs.global_iteration(depths, lws, rw=0.0, max_dist=2.0, huber=0.5, lm_rel=0.1, n_iters=1, built=True)
    xi_b = sv.global_xi.cpu().numpy()[:6]
    assert np.abs(xi_b - xi_all).max() <= 1e-9 * max(1.0, np.abs(xi_all).max()), (xi_b, xi_all)
