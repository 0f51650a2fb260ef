"""GPU: the per-frame pipeline end to end on a small config-3-shaped problem (canonical sphere
fused from depth, Fibonacci node graph, live frame = displaced + inflated sphere): projective
association + GN iterations reduce the point-to-plane cost and move the warped surface onto the
live surface; one iteration's system matches the fp64 oracle."""
import numpy as np
import pytest
import torch

from oracle import gn_np as G
from oracle import oracle_np as O
from dynamicfusion_body_amd import FusionDM, scene
from dynamicfusion_body_amd.pipeline import FrameSolver, extract_surface_samples

pytestmark = pytest.mark.gpu


def build_canonical(R, cam):
    H, W, fx, cx, cy = scene.CAMERAS[cam]
    K = scene.intrinsics(fx, cx, cy)
    scale, center, tdist = scene.grid_params(R)
    f = FusionDM(tdist, K, tsdf_res=R)
    T, Wt = f._new_volume_pair()
    for a in (0.0, 40.0, -40.0):
        lw = scene.view_extrinsic(a)
        dm = scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)
        f.fuseDepths(torch.from_numpy(dm).cuda(), lw, T, Wt, scale=scale, center=center)
    return K, (H, W), scale, center, tdist, T, Wt


def test_sample_extraction_lies_on_the_sphere():
    R = 64
    K, hw, scale, center, tdist, T, Wt = build_canonical(R, "C1")
    pos, nrm = extract_surface_samples(T, Wt, band=1.0)
    assert pos.shape[0] > 1000
    p = pos.cpu().numpy(); n = nrm.cpu().numpy()
    rad = np.linalg.norm(p - R / 2, axis=1) * scale
    assert np.abs(rad - scene.SPHERE_R).max() < 1.5 * scale           # within 1.5 voxels of the true surface
    assert np.median(np.abs(rad - scene.SPHERE_R)) < 0.3 * scale
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0)
    # sd = depth - voxel depth is positive in front of the surface: the gradient is the outward normal
    outward = (p - R / 2) / np.linalg.norm(p - R / 2, axis=1, keepdims=True)
    assert np.mean(np.sum(n * outward, axis=1)) > 0.9


def test_frame_solve_converges_and_matches_oracle():
    R, N, k = 64, 48, 4
    K, (H, W), scale, center, tdist, T, Wt = build_canonical(R, "C1")
    fs = FrameSolver(K, scale, center, R / 2, knn=k, pcg_iters=60)
    node_pos, node_w = scene.fibonacci_nodes(N, R)
    ident = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0, 0]), (N, 1))
    fs.set_graph(node_pos, ident, node_w)
    S = fs.set_canonical(T, Wt, band=1.0)
    assert S > 1000
    lw_cam = scene.view_extrinsic(0.0)
    shift_vox = np.array([0.6, -0.4, 0.3])
    live = scene.render_depth(K, lw_cam, H, W, dtype=np.float32, invalid_frac=0.01, seed=5,
                              sphere_offset=shift_vox * scale, sphere_r=scene.SPHERE_R * 1.02)
    depth = torch.from_numpy(live).cuda()
    sv = fs.solver
    # --- one iteration against the oracle on the same (sorted) samples
    sv.associate_depth(depth, fs.K, fs.Kinv, lw_cam, scale, center, R / 2, fs.lw, 4.0)
    sv.build(fs.lw, 0.05)
    A, b = sv.dense_normal_equations()
    cost, cnt = sv.cost()
    pos, nrm = sv.spos.cpu().numpy(), sv.snrm.cpu().numpy()
    nbr = sv.snbr.cpu().numpy().astype(np.int64)
    warped = O.warp(pos, ident[nbr], node_pos[nbr], node_w[nbr], m_lw=fs.lw)
    co, vo = G.associate_depth(warped, fs.K, fs.Kinv, lw_cam, live, scale, center, R / 2)
    vo &= np.linalg.norm(co - warped, axis=1) <= 4.0
    assert np.array_equal(sv.valid.cpu().numpy().astype(bool), vo) and cnt == int(vo.sum())
    assert cnt > 0.3 * S
    r, J = G.data_residual_jacobian(ident, pos, nrm, co, nbr, node_pos, node_w, fs.lw)
    node_nbr = sv.node_nbr.cpu().numpy().astype(np.int64)
    rho, nb, Ji, Jj = G.reg_residual_jacobian(ident, np.arange(N), node_nbr, node_pos, node_w, 0.05)
    Ao, bo, c_or = G.assemble_dense(N, r, J, nbr, rho, nb, Ji, Jj, valid=vo)
    assert abs(cost - c_or) <= 1e-9 * c_or
    assert np.abs(A - Ao).max() <= 1e-9 * np.abs(Ao).max() and np.abs(b - bo).max() <= 1e-9 * np.abs(bo).max()
    # --- six more iterations, GPU (PCG to convergence) vs the oracle loop (dense exact solve), same
    #     damping: total cost per iteration to 1e-4 relative -- the north-star residual bar.  The
    #     data RMS floors at ~0.3 voxel here: nearest-pixel association at 320x240 over a 64^3 grid.
    rw, lm = 0.05, 1.0
    fs.solver.pcg_iters = 400
    fs.solver._pattern = None
    fs.solver.node_dq.copy_(torch.from_numpy(ident).cuda())
    gpu_costs = fs.solve(depth, lw_cam, rw=rw, iters=6, lm_abs=lm, lm_rel=lm, max_dist=4.0)
    dqs = ident.copy()
    or_costs, or_rms = [], []
    for it in range(6):
        warped = O.warp(pos, dqs[nbr], node_pos[nbr], node_w[nbr], m_lw=fs.lw)
        co, vo = G.associate_depth(warped, fs.K, fs.Kinv, lw_cam, live, scale, center, R / 2)
        vo &= np.linalg.norm(co - warped, axis=1) <= 4.0
        r, J = G.data_residual_jacobian(dqs, pos, nrm, co, nbr, node_pos, node_w, fs.lw)
        rho, nb, Ji, Jj = G.reg_residual_jacobian(dqs, np.arange(N), node_nbr, node_pos, node_w, rw)
        Ao, bo, c_or = G.assemble_dense(N, r, J, nbr, rho, nb, Ji, Jj, valid=vo)
        or_costs.append(c_or)
        or_rms.append(np.sqrt(np.sum(np.where(vo, r, 0) ** 2) / vo.sum()))
        Ad = Ao + lm * np.eye(6 * N) + lm * np.diag(np.diag(Ao))
        dqs = G.apply_twists(dqs, np.linalg.solve(Ad, -bo).reshape(N, 6))
    assert np.allclose([c for c, n in gpu_costs], or_costs, rtol=1e-4)
    assert or_rms[-1] < 0.8 * or_rms[0]
    dq = sv.node_dq.cpu().numpy()
    assert np.allclose(np.sum(dq[:, :4] ** 2, axis=1), 1.0, atol=1e-10)
    assert np.abs(dq - dqs).max() < 1e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_device_extraction_matches_torch_definition(dtype):
    """csrc/dfh_extract.hip (count / scan / emit) against the same sample definition on torch ops:
    identical count and voxel order, positions and normals to fp64 rounding; ragged shapes, slab
    offset, capacity truncation, empty result."""
    from dynamicfusion_body_amd.pipeline import extract_surface_samples_torch
    rng = np.random.default_rng(12)
    for shape, x0 in (((40, 33, 37), 0), ((7, 64, 50), 13), ((9, 21, 44), 5), ((1, 40, 48), 3), ((64, 64, 64), 0)):   # (z rows of 44, 48, 64: the 16-byte-pack kernels; a one-plane slab)
        g = np.stack(np.meshgrid(*[np.arange(s, dtype=np.float64) for s in shape], indexing="ij"), axis=-1)
        T = np.linalg.norm(g - (np.array(shape) / 2.0 + 0.3), axis=-1) - max(min(shape), 12) / 3.0 + 0.05 * rng.normal(size=shape)
        W = (rng.random(shape) < 0.8).astype(np.float64)
        Td = torch.from_numpy(T).to("cuda", dtype=dtype).contiguous()
        Wd = torch.from_numpy(W).to("cuda", dtype=dtype).contiguous()
        pos, nrm = extract_surface_samples(Td, Wd, 1.5, x0=x0)
        pt, nt = extract_surface_samples_torch(Td, Wd, 1.5, x0=x0)
        assert pos.shape == pt.shape and pos.shape[0] > 50
        assert float((pos - pt).abs().max()) <= 1e-12 and float((nrm - nt).abs().max()) <= 1e-12
        for cap in (pos.shape[0] // 3, pos.shape[0] - 1, 1, 7):           # an even subsample over the whole surface, not a prefix
            n_all = pos.shape[0]
            sel = (torch.arange(cap, device="cuda", dtype=torch.int64) * n_all + cap - 1) // cap
            p2, n2 = extract_surface_samples(Td, Wd, 1.5, x0=x0, max_samples=cap)
            assert p2.shape[0] == cap and torch.equal(p2, pos[sel]) and torch.equal(n2, nrm[sel])
        assert int(sel[-1]) > n_all // 2
        p3, _ = extract_surface_samples(Td, Wd, 1.5, x0=x0, max_samples=n_all + 5)
        assert torch.equal(p3, pos)
        pe, ne = extract_surface_samples(Td, torch.zeros_like(Wd), 1.5)
        assert pe.shape == (0, 3) and ne.shape == (0, 3)
    with pytest.raises(ValueError):
        extract_surface_samples(Td, Wd[:2], 1.5)


@pytest.mark.parametrize("shape", [(258, 256, 260), (128, 128, 256), (129, 128, 256), (131, 130, 252)])
def test_device_extraction_more_than_one_scan_chunk(shape):
    """The scan of the block counts (1 024 voxels per block) works in chunks of 4 096 counts, one workgroup each, and a second
    launch adds the chunk bases (csrc/dfh_extract.hip): 16 770 blocks = five chunks, the last one partly filled; exactly 4 096 blocks
    (the single-workgroup case, full); 4 128 (a second chunk of 32 blocks); 4 191 blocks whose last one is cut by the volume's end."""
    from dynamicfusion_body_amd.pipeline import extract_surface_samples_torch
    ax = [torch.arange(s, device="cuda", dtype=torch.float64) for s in shape]
    c = [0.505 * s for s in shape]
    d = torch.sqrt((ax[0][:, None, None] - c[0]) ** 2 + (ax[1][None, :, None] - c[1]) ** 2 + (ax[2][None, None, :] - c[2]) ** 2)
    Td = (d - 0.35 * min(shape)).to(torch.float32).contiguous()
    Wd = ((ax[0][:, None, None] + ax[1][None, :, None] * 3 + ax[2][None, None, :] * 7) % 5 != 0).to(torch.float32).contiguous()
    pos, nrm = extract_surface_samples(Td, Wd, 1.5)
    pt, nt = extract_surface_samples_torch(Td, Wd, 1.5)
    assert pos.shape == pt.shape and pos.shape[0] > 20000
    assert float((pos - pt).abs().max()) <= 1e-12 and float((nrm - nt).abs().max()) <= 1e-12
    assert float(pos[-1, 0]) > 0.75 * shape[0]                           # samples behind the first chunk(s) are there
    cap = pos.shape[0] // 3                                               # (the even subsample reads the total behind the last chunk)
    sel = (torch.arange(cap, device="cuda", dtype=torch.int64) * pos.shape[0] + cap - 1) // cap
    p2, _ = extract_surface_samples(Td, Wd, 1.5, max_samples=cap)
    assert torch.equal(p2, pos[sel])


def test_composed_frame_loop_tracks_without_drift():
    """pipeline.SlabFrame over 60 frames of a +-0.6 voxel oscillation (three-view live volumes, default damping /
    gate): the warp field follows the motion and does not drift, the band-sample count stays put.  (With a 4-voxel
    association gate and weak damping the same loop drifts by tens of voxels within 40 frames.)"""
    from dynamicfusion_body_amd.pipeline import SlabFrame
    R, N = 128, 256
    H, W, fx, cx, cy = scene.CAMERAS["C2"]
    K = scene.intrinsics(fx, cx, cy)
    scale, center, tdist = scene.grid_params(R)
    node_pos, node_w = scene.fibonacci_nodes(N, R)
    lws = [scene.view_extrinsic(a) for a in (0.0, 40.0, -40.0)]
    sf = SlabFrame(K, scale, center, R, tdist / scale, node_pos, node_w, knn=4, pcg_iters=10, band=2.0)
    for lw in lws:
        sf.integrate(torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda(), lw)
    sf.refresh_samples()
    amp = np.array([0.5, -0.3, 0.2])
    counts, tmax = [], []
    for f in range(60):
        off = amp * np.sin(0.3 * f) * scale
        ds = [torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0, sphere_offset=off)).cuda() for lw in lws]
        counts.append(sf.step(ds, lws, gn_iters=10))
        dq = sf.fs.solver.node_dq.cpu().numpy()
        assert np.isfinite(dq).all()
        tmax.append(2 * np.linalg.norm(dq[:, 4:], axis=1).max())
    assert max(tmax) < 1.5 and max(tmax[20:]) > 0.3                       # follows the 0.62-voxel amplitude, no run-away
    assert np.linalg.norm(dq[:, 1:4], axis=1).max() < 0.02
    assert max(counts[20:]) < 1.15 * min(counts[20:])                     # steady band


def test_mesh_started_inside_the_frame_equals_the_mesh_after_it():
    """SlabFrame.step(on_updated=...) calls back once the TSDF update is queued: a mesh whose count pass is queued there on a
    second stream (mesh.marching_cubes_begin) and finished after the frame equals, element for element, the mesh extracted from
    the canonical volume afterwards -- the sample refresh that runs beside it only reads the volume."""
    from dynamicfusion_body_amd import mesh
    from dynamicfusion_body_amd.pipeline import SlabFrame
    R, N = 96, 160
    H, W, fx, cx, cy = scene.CAMERAS["C2"]
    K = scene.intrinsics(fx, cx, cy)
    scale, center, tdist = scene.grid_params(R)
    node_pos, node_w = scene.fibonacci_nodes(N, R)
    lws = [scene.view_extrinsic(a) for a in (0.0, 40.0, -40.0)]
    sf = SlabFrame(K, scale, center, R, tdist / scale, node_pos, node_w, knn=4, pcg_iters=10, band=2.0, distributed=False)
    for lw in lws:
        sf.integrate(torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda(), lw)
    sf.refresh_samples()
    side = torch.cuda.Stream()
    for f in range(3):
        off = np.array([0.4, -0.25, 0.15]) * np.sin(0.5 * (f + 1)) * scale
        ds = [torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0, sphere_offset=off)).cuda() for lw in lws]
        pending, calls = [], []

        def start():
            calls.append(1)
            with torch.cuda.stream(side):
                side.wait_event(sf.updated)
                pending.append(mesh.marching_cubes_begin(sf.T, 0.0))
        torch.cuda.current_stream().wait_stream(side)             # the previous frame's mesh has read the canonical volume
        sf.step(ds, lws, gn_iters=4, on_updated=start)
        assert len(calls) == 1
        with torch.cuda.stream(side):
            got = pending[0].finish()
        side.synchronize()
        torch.cuda.synchronize()
        ref = mesh.marching_cubes(sf.T, 0.0)
        assert got[0].shape[0] > 1000
        for a, b in zip(got, ref):
            assert torch.equal(a, b)


def test_frame_loop_variants_give_the_same_bits(monkeypatch):
    """SlabFrame.step overlaps the live-volume sweep (side stream) with the plan build and reads the plan's counts back through
    pinned memory behind an event; the plan's lists are built by counting + per-list sorts.  The sequential loop
    (DFH_NO_SIDE_STREAM), the radix-sort plan (DFH_PLAN_RADIX), the stage-timed loop (a synchronisation after every stage) and the
    loop that reads its counts through device scalars instead of host-visible words (HostScalar off) and the loop whose GN
    iterations are two calls (build, solve) instead of dfh_gn_iteration (DFH_GN_NO_FUSED_ITER), and the loop whose gather walks the
    lists of ALL blocks instead of those with column >= row (option gn_gather_full) must give the same warp
    field and the same canonical volume, bit for bit: a race between the streams, or a count read too early, would show here."""
    from dynamicfusion_body_amd.pipeline import SlabFrame
    from dynamicfusion_body_amd.device import HostScalar
    R, N = 96, 160
    H, W, fx, cx, cy = scene.CAMERAS["C2"]
    K = scene.intrinsics(fx, cx, cy)
    scale, center, tdist = scene.grid_params(R)
    node_pos, node_w = scene.fibonacci_nodes(N, R)
    lws = [scene.view_extrinsic(a) for a in (0.0, 40.0, -40.0)]
    first = [torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda() for lw in lws]
    frames = []
    for f in range(5):
        off = np.array([0.4, -0.25, 0.15]) * np.sin(0.5 * (f + 1)) * scale
        frames.append([torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0, sphere_offset=off)).cuda() for lw in lws])

    def run(env, timed, data_views=None):
        monkeypatch.setattr(HostScalar, "enabled", "NO_HOST_SCALARS" not in env)
        from dynamicfusion_body_amd import _lib
        # every switch lives in the library's option table (dfh_set_option): the C side's and the Python layer's (py_*)
        _lib.set_option("plan_radix", 1 if "DFH_PLAN_RADIX" in env else None)
        _lib.set_option("gn_gather_full", 1 if "GN_GATHER_FULL" in env else None)    # every block's list walked (default: column >= row, sums stored twice)
        _lib.set_option("py_no_side_stream", 1 if "DFH_NO_SIDE_STREAM" in env else None)
        _lib.set_option("py_gn_no_fused_iter", 1 if "DFH_GN_NO_FUSED_ITER" in env else None)
        _lib.set_option("py_gn_iter_per_call", 1 if "DFH_GN_ITER_PER_CALL" in env else None)
        sf = SlabFrame(K, scale, center, R, tdist / scale, node_pos, node_w, knn=4, pcg_iters=10, band=2.0, distributed=False)
        for d, lw in zip(first, lws):
            sf.integrate(d, lw)
        sf.refresh_samples()
        counts = [sf.step(ds, lws, gn_iters=6, stage_ms={} if timed else None, data_views=data_views) for ds in frames]
        torch.cuda.synchronize()
        return counts, sf.fs.solver.node_dq.clone(), sf.T.clone(), sf.Wt.clone()
    ref = run((), False)
    assert HostScalar.enabled
    for env, timed in ((("DFH_NO_SIDE_STREAM",), False), (("DFH_PLAN_RADIX",), False), (("NO_HOST_SCALARS",), False),
                       (("DFH_GN_NO_FUSED_ITER",), False), (("GN_GATHER_FULL",), False), ((), True), ((), False)):
        got = run(env, timed)
        assert got[0] == ref[0], (env, timed)
        for a, b in zip(got[1:], ref[1:]):
            assert torch.equal(a, b), (env, timed)
    # the data term on the first view only (round 2's): the frame's iterations in one call through a one-view table against one
    # dfh_gn_iteration call per iteration (DFH_GN_ITER_PER_CALL) -- the same bits; and the three-view data term is a different solve
    ref1 = run((), False, data_views=1)
    got1 = run(("DFH_GN_ITER_PER_CALL",), False, data_views=1)
    assert got1[0] == ref1[0]
    for a, b in zip(got1[1:], ref1[1:]):
        assert torch.equal(a, b)
    assert not torch.equal(ref1[1], ref[1])


def test_association_inside_the_build_is_bit_identical(monkeypatch):
    """dfh_gn_build_planned_assoc (association folded into the data-row kernel) against dfh_gn_associate followed by
    dfh_gn_build_planned: same correspondences, same validity, same normal equations and cost, bit for bit -- at identity and
    after moving the field, with and without Huber weights and a gate, and the whole 5-iteration loop."""
    R, N, k = 64, 48, 4
    K, (H, W), scale, center, tdist, T, Wt = build_canonical(R, "C1")
    node_pos, node_w = scene.fibonacci_nodes(N, R)
    ident = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0, 0]), (N, 1))
    lw_cam = scene.view_extrinsic(0.0)
    live = scene.render_depth(K, lw_cam, H, W, dtype=np.float32, invalid_frac=0.01, seed=5,
                              sphere_offset=np.array([0.6, -0.4, 0.3]) * scale, sphere_r=scene.SPHERE_R * 1.02)
    depth = torch.from_numpy(live).cuda()
    rng = np.random.default_rng(2)
    dq1 = G.apply_twists(ident, rng.normal(scale=[3e-3] * 3 + [0.2] * 3, size=(N, 6)))
    out = {}
    for mode in ("fused", "separate"):
        from dynamicfusion_body_amd import _lib
        _lib.set_option("py_gn_no_fused_assoc", 1 if mode == "separate" else None)
        fs = FrameSolver(K, scale, center, R / 2, knn=k, pcg_iters=10, distributed=False)
        fs.set_graph(node_pos, ident, node_w)
        fs.set_canonical(T, Wt, band=2.0)
        sv = fs.solver
        snaps = []
        for dq, max_dist, huber in ((ident, 0.0, 0.0), (dq1, 2.0, 0.5), (dq1, 4.0, 0.0)):
            sv.node_dq.copy_(torch.from_numpy(dq).cuda())
            sv.build_associated(depth, fs.K, fs.Kinv, lw_cam, scale, center, R / 2, fs.lw, 0.7, max_dist, huber)
            snaps.append((sv.corr.clone(), sv.valid.clone(), sv.vals.clone(), sv.rhs.clone(), sv.cost_count.clone()))
        sv.node_dq.copy_(torch.from_numpy(ident).cuda())
        costs = fs.solve(depth, lw_cam, rw=5.0, iters=5, lm_abs=10.0, lm_rel=1e-2, max_dist=2.0, huber=0.5)
        out[mode] = (snaps, costs, sv.node_dq.clone())
    for a, b in zip(out["fused"][0], out["separate"][0]):
        assert int(a[1].sum()) > 500
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert out["fused"][1] == out["separate"][1] and torch.equal(out["fused"][2], out["separate"][2])


def test_soak_300_frames():
    """The guarantee of DESIGN.md section 6, pinned (round-3 verdict item 7): 300 frames of the composed loop at the benched size
    (256^3, 512 nodes, three views, the shipped defaults of SlabFrame.step) on a sphere that oscillates by +-0.6 voxel and
    breathes by 0.3 %.  The warp field stays finite and bounded over ALL nodes (the ones that blend into samples and the ones on
    the unobserved back), the sample count stays where it was at frame 50 (round 3's |T| < band shell thickened: 129 k -> 261 k),
    and repeated solves of one system are bit-identical."""
    from dynamicfusion_body_amd.pipeline import SlabFrame
    R, N = 256, 512
    H, W, fx, cx, cy = scene.CAMERAS["C2"]
    K = scene.intrinsics(fx, cx, cy)
    scale, center, tdist = scene.grid_params(R)
    node_pos, node_w = scene.fibonacci_nodes(N, R)
    sf = SlabFrame(K, scale, center, R, tdist / scale, node_pos, node_w, knn=4, pcg_iters=10, band=2.0, distributed=False)
    lws = [scene.view_extrinsic(a) for a in (0.0, 40.0, -40.0)]
    for lw in lws:
        sf.integrate(torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda(), lw)
    sf.refresh_samples()
    period = 21
    depths = [[torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0,
                                                   sphere_offset=np.array([0.5, -0.3, 0.2]) * np.sin(2 * np.pi * f / period) * scale,
                                                   sphere_r=scene.SPHERE_R * (1.0 + 0.003 * np.cos(2 * np.pi * f / period)))).cuda() for lw in lws]
              for f in range(period)]
    n50 = None
    worst = 0.0
    for f in range(300):
        n = sf.step(depths[f % period], lws, gn_iters=10)
        if f % 50 == 49:
            dq = sf.fs.solver.node_dq
            assert bool(torch.isfinite(dq).all()), f
            tr = float(2.0 * dq[:, 4:].norm(dim=1).max())                      # |t| of every node's motion about the grid origin
            worst = max(worst, tr)
            assert tr < 1.5, "the warp field drifts: max translation %.2f voxel at frame %d (the scene moves +-0.6)" % (tr, f + 1)
            if n50 is None:
                n50 = n
            assert abs(n - n50) <= 0.1 * n50, (n, n50, f)
    assert worst > 0.05                                                          # (the field did carry motion)
    sv = sf.fs.solver
    v0 = sv.vals.clone()
    xs = []
    for _ in range(5):
        sv.vals.copy_(v0)
        sv.solve_linear(1e-2, 1e-2)
        xs.append(sv.dx.clone())
    assert all(torch.equal(x, xs[0]) for x in xs[1:]) and bool(torch.isfinite(xs[0]).all())
