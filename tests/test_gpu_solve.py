"""GPU parity of the warp-field solve (csrc/dfh_solve.hip through the C ABI).

  * residual evaluators vs the REFERENCE's outputs (golden g5): <= 1e-12 (fp64, same operations;
    exp() of the blend weights is the only last-ulp difference);
  * normal equations / PCG / twist update vs the fp64 numpy GN oracle (oracle/gn_np.py), whose
    Jacobians are pinned against finite differences of the reference's residual
    (tests/test_gn_oracle.py): J^T J, J^T r relative 1e-10; cost relative 1e-12; LM iterates and
    costs relative 1e-6 (the "warp-solve residual to 1e-4" bar of BASELINE north_star).
"""
import numpy as np
import pytest
import torch

from oracle import gn_np as G
from oracle import oracle_np as O
from dynamicfusion_body_amd import _lib, scene, solve

pytestmark = pytest.mark.gpu


def load(golden):
    g = golden("g5_residuals")
    return (g, g["verts"], g["norms"], g["corr"], g["nbr"], g["vidx"], g["node_pos"], g["node_dq"], g["node_w"], g["lw"],
            float(g["rw"]))


def test_residuals_match_reference(golden):
    g, verts, norms, corr, nbr, vidx, npos, ndq, nw, lw, rw = load(golden)
    V = len(verts)
    fd = solve.residual_data(ndq, verts, norms, corr, nbr, npos, nw, lw).cpu().numpy()
    assert np.abs(fd - g["computef_out"][:V]).max() <= 1e-12
    fr = solve.residual_reg(ndq, nbr[vidx], npos, nw, rw).cpu().numpy()
    assert len(fr) == 3 * nbr.shape[1] * len(npos)
    assert np.abs(fr - g["computef_out"][V:]).max() <= 1e-12
    f = np.concatenate([fd, fr])
    assert abs(0.5 * f @ f - float(g["cost"])) <= 1e-9
    fl = solve.residual_data(ndq, verts, norms, corr, nbr, npos, nw, g["lw2"]).cpu().numpy()        # Fusion.computef_lw
    assert np.abs(fl - g["computef_lw_out"]).max() <= 1e-12
    keep = g["rigid_keep"]
    fr2 = solve.residual_rigid(g["rigid_x"], verts[keep], norms[keep], corr[keep]).cpu().numpy()    # FusionDM.computef_lw
    assert np.abs(fr2 - g["rigid_out"]).max() <= 1e-12


def test_residual_errors():
    z3 = np.zeros((4, 3))
    with pytest.raises(ValueError):
        solve.residual_rigid(np.zeros(8), z3, z3, np.zeros((3, 3)))
    with pytest.raises(ValueError):
        solve.residual_data(np.zeros((2, 8)), z3, z3, np.zeros((5, 3)), np.zeros((4, 2), dtype=int), np.zeros((2, 3)), np.ones(2), np.zeros(8))
    with pytest.raises(ValueError):
        solve.residual_data(np.zeros((2, 8)), z3, z3, z3, np.full((4, 2), 7), np.zeros((2, 3)), np.ones(2), np.zeros(8))
    assert solve.residual_rigid(np.zeros(8), np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 3))).numel() == 0


def test_sample_knn_vs_oracle():
    rng = np.random.default_rng(1)
    for N, k, S in ((300, 4, 1000), (9, 8, 257), (40, 1, 64)):
        npos = rng.uniform(0, 50, size=(N, 3)); nw = rng.uniform(2, 6, size=N)
        pts = rng.uniform(-5, 55, size=(S, 3))
        nbr, wts = solve.sample_knn(pts, npos, nw, k)
        loc = O.knn_bruteforce(pts, npos, k)
        assert np.array_equal(nbr.cpu().numpy(), loc)
        w = G.blend_weights(pts, npos[loc], nw[loc])
        assert np.abs(wts.cpu().numpy() - w).max() <= 1e-15 * 4
    # spatially coherent samples (consecutive band voxels along z, as the extraction emits them): the bounding-box
    # pruning is active; more nodes than the candidate capacity; duplicated nodes = exact distance ties (lower index wins)
    for N, k in ((1500, 4), (700, 8), (600, 3)):
        npos = rng.uniform(0, 120, size=(N, 3)); nw = rng.uniform(2, 6, size=N)
        npos[N // 2:N // 2 + 20] = npos[:20]
        z = np.arange(900, dtype=np.float64)
        pts = np.stack([20.0 + (z // 300) + 0.3 * np.sin(z), 33.0 + 0.2 * np.cos(z), (z % 300) * 0.4], axis=1)
        pts = np.concatenate([pts, rng.uniform(0, 120, size=(300, 3))])      # and one incoherent tail (no pruning there)
        nbr, wts = solve.sample_knn(pts, npos, nw, k)
        loc = O.knn_bruteforce(pts, npos, k)
        assert np.array_equal(nbr.cpu().numpy(), loc)
        assert np.abs(wts.cpu().numpy() - G.blend_weights(pts, npos[loc], nw[loc])).max() <= 1e-15 * 4


@pytest.mark.parametrize("res,slab,N,k", [((32, 24, 40), (0, 32), 80, 4), ((20, 20, 36), (6, 15), 300, 4), ((16, 16, 16), (0, 16), 9, 8),
                                          ((24, 24, 48), (0, 24), 40, 1)])
def test_sample_knn_through_brick_lists(res, slab, N, k):
    """dfh_sample_knn_bricks (candidate lists of the K3 workspace) == dfh_sample_knn, bit for bit: off-lattice points
    (up to sqrt(3)/2 from a voxel centre), points outside the slab / grid (full scan), duplicate nodes (ties by node
    index), a node cluster that overflows a brick's list."""
    from dynamicfusion_body_amd import kernels
    rng = np.random.default_rng(N + k)
    hi = np.array(res) - 1.0
    node_pos = rng.uniform(0, hi, size=(N, 3))
    if N >= 300:
        node_pos[:280] = np.array(res) / 2.0 + rng.normal(size=(280, 3)) * 1.2          # > 256 candidates around one brick
    node_pos[N // 2] = node_pos[N // 3]                                                    # exact tie
    node_w = rng.uniform(2.0, 5.0, size=N)
    pts = [rng.uniform(-0.49, hi + 0.49, size=(6000, 3)),                                 # anywhere in the lattice's cells
           np.rint(rng.uniform(0, hi, size=(500, 3))) + rng.choice([-0.5, 0.5], size=(500, 3)),   # cell corners: rounding edge
           rng.uniform(-6, hi + 6, size=(500, 3)),                                         # partly outside the grid
           node_pos[:50] + 0.0]                                                            # on nodes: zero distance
    pts = np.concatenate(pts)
    ws = kernels.dqb_workspace(res, slab, knn=k, n_nodes=N)
    kernels.dqb_build_candidates(ws, res, node_pos, k, slab)
    nbr0, w0 = solve.sample_knn(pts, node_pos, node_w, k)
    nbr1, w1 = solve.sample_knn(pts, node_pos, node_w, k, bricks=(res, slab, ws))
    assert torch.equal(nbr0, nbr1) and torch.equal(w0, w1)
    inside = np.all((np.rint(pts) >= [slab[0], 0, 0]) & (np.rint(pts) <= [slab[1] - 1, hi[1], hi[2]]), axis=1)
    assert inside.any() and (~inside).any()
    # the lists really are used: they are short compared with N on most bricks
    cnt = ws.view(torch.int32)[:(ws.numel())].cpu().numpy()
    nb = (-(-(slab[1] - slab[0]) // 4)) * (-(-res[1] // 4)) * (-(-res[2] // 16))
    counts = cnt[:nb * 257].reshape(nb, 257)[:, 0]
    assert counts.max() <= 256
    if N >= 300:
        assert (counts < 0).any()                                                          # the overflow path was exercised
    else:
        assert (counts >= 0).all() and (counts < N).any() or N <= 2 * k


def make_solver(npos, ndq, nw, nbr, vidx, verts, norms, corr, knn, pcg_iters=400, sort=True, valid=None):
    sv = solve.WarpSolver(knn=knn, pcg_iters=pcg_iters)
    sv.set_graph(npos, ndq, nw, node_nbr=nbr[vidx])
    sv.set_samples(verts, norms, nbr=nbr, sort=sort)
    sv.set_correspondences(corr, valid)
    return sv


@pytest.mark.parametrize("sort", [True, False])
def test_normal_equations_vs_oracle(golden, sort):
    g, verts, norms, corr, nbr, vidx, npos, ndq, nw, lw, rw = load(golden)
    N = len(npos)
    rng = np.random.default_rng(4)
    valid = rng.random(len(verts)) < 0.9
    sv = make_solver(npos, ndq, nw, nbr, vidx, verts, norms, corr, nbr.shape[1], sort=sort, valid=valid)
    sv.build(lw, rw)
    A, b = sv.dense_normal_equations()
    cost, cnt = sv.cost()
    r, J = G.data_residual_jacobian(ndq, verts, norms, corr, nbr, npos, nw, lw)
    rho, nb, Ji, Jj = G.reg_residual_jacobian(ndq, vidx, nbr, npos, nw, rw)
    Ao, bo, co = G.assemble_dense(N, r, J, nbr, rho, nb, Ji, Jj, valid=valid)
    assert cnt == int(valid.sum())
    assert abs(cost - co) <= 1e-12 * co
    assert np.abs(A - Ao).max() <= 1e-10 * np.abs(Ao).max()
    assert np.abs(b - bo).max() <= 1e-10 * np.abs(bo).max()
    assert np.abs(A - A.T).max() <= 1e-9 * np.abs(A).max()
    # data term only (rw = 0 skips the regulariser)
    sv.build(lw, 0.0)
    A0, b0 = sv.dense_normal_equations()
    Ao0, bo0, _ = G.assemble_dense(N, r, J, nbr, rho * 0, nb, Ji * 0, Jj * 0, valid=valid)
    assert np.abs(A0 - Ao0).max() <= 1e-10 * np.abs(Ao0).max()


def test_pcg_and_twist_vs_oracle(golden):
    g, verts, norms, corr, nbr, vidx, npos, ndq, nw, lw, rw = load(golden)
    N = len(npos)
    sv = make_solver(npos, ndq, nw, nbr, vidx, verts, norms, corr, nbr.shape[1], pcg_iters=600)
    sv.build(lw, rw)
    A, b = sv.dense_normal_equations()
    lm_abs, lm_rel = 0.5, 1e-3
    Ad = A + lm_abs * np.eye(6 * N) + lm_rel * np.diag(np.diag(A))
    xo = np.linalg.solve(Ad, -b)
    sv.solve_linear(lm_abs, lm_rel)
    x = sv.dx.cpu().numpy()
    assert np.abs(x - xo).max() <= 1e-8 * np.abs(xo).max()
    sv.apply()
    dq_new = sv.node_dq.cpu().numpy()
    assert np.abs(dq_new - G.apply_twists(ndq, xo.reshape(N, 6))).max() <= 1e-8
    # few iterations: still a descent direction with decreasing residual norm
    sv2 = make_solver(npos, ndq, nw, nbr, vidx, verts, norms, corr, nbr.shape[1], pcg_iters=5)
    sv2.build(lw, rw)
    sv2.solve_linear(lm_abs, lm_rel)
    x5 = sv2.dx.cpu().numpy()
    assert x5 @ b < 0 and np.linalg.norm(Ad @ x5 + b) < np.linalg.norm(b)


def test_pcg_paths_agree_and_are_deterministic(golden, monkeypatch):
    """The persistent single-launch PCG and the two-launches-per-iteration PCG are the same algorithm
    (same operation order per row; only the order of the dot-product sums differs), and neither has
    floating-point atomics (the two-launch path adds per-workgroup partials in a fixed order): two runs of
    either are bit-identical."""
    g, verts, norms, corr, nbr, vidx, npos, ndq, nw, lw, rw = load(golden)
    runs = []
    for mode in ("persistent", "persistent", "multilaunch"):
        if mode == "multilaunch":
            _lib.set_option("pcg_multilaunch", 1)
        else:
            _lib.set_option("pcg_multilaunch", None)
        sv = make_solver(npos, ndq, nw, nbr, vidx, verts, norms, corr, nbr.shape[1], pcg_iters=12)
        sv.build(lw, rw)
        sv.solve_linear(0.5, 1e-3)
        runs.append((sv.dx.cpu().numpy().copy(), sv.vals.cpu().numpy().copy()))
    _lib.set_option("pcg_multilaunch", None)
    # the build uses fp64 atomics, so compare each solve against its own matrix only through the solution scale
    assert np.isfinite(runs[0][0]).all()
    assert np.abs(runs[0][0] - runs[2][0]).max() <= 1e-9 * np.abs(runs[2][0]).max()
    assert np.abs(runs[0][1] - runs[2][1]).max() <= 1e-12 * np.abs(runs[2][1]).max()     # same damping written into the matrix
    # determinism of the solve itself: same system twice through the persistent path
    sv = make_solver(npos, ndq, nw, nbr, vidx, verts, norms, corr, nbr.shape[1], pcg_iters=12)
    sv.build(lw, rw)
    v0 = sv.vals.clone()
    sv.solve_linear(0.5, 1e-3)
    x1 = sv.dx.clone()
    sv.vals.copy_(v0)
    sv.solve_linear(0.5, 1e-3)
    assert torch.equal(x1, sv.dx)
    # ... and through the two-launch path, five times (its dot products were atomic sums once: last bits differed run to run)
    _lib.set_option("pcg_multilaunch", 1)
    xs = []
    for _ in range(5):
        sv.vals.copy_(v0)
        sv.solve_linear(0.5, 1e-3)
        xs.append(sv.dx.clone())
    _lib.set_option("pcg_multilaunch", None)
    assert all(torch.equal(xs[0], x) for x in xs[1:])
    assert float((xs[0] - x1).abs().max()) <= 1e-9 * float(x1.abs().max())


@pytest.mark.parametrize("multilaunch", [False, True])
def test_pcg_rows_wider_than_the_register_cache(multilaunch, monkeypatch):
    """dfh_pcg_solve through the C ABI on a synthetic block-sparse SPD system whose rows have 6 .. 48 blocks: the
    persistent kernel keeps 30 blocks of a row in registers (three per lane slot) and walks the rest from memory,
    waiting for every neighbour value on its own -- the golden systems have no such row.  Checked against a dense solve."""
    import ctypes
    from dynamicfusion_body_amd import _lib
    from dynamicfusion_body_amd.device import current_stream_ptr
    lib = _lib.load()
    if multilaunch:
        _lib.set_option("pcg_multilaunch", 1)
    else:
        _lib.set_option("pcg_multilaunch", None)
    rng = np.random.default_rng(11)
    N = 48
    adj = np.eye(N, dtype=bool)
    adj[:4, :] = True                                            # four hubs: rows of 48 blocks
    for a in range(N):
        adj[a, (a + 1) % N] = adj[a, (a - 1) % N] = True         # a ring
    adj[10, 5:30] = True                                         # 25-ish blocks: two full cache rounds and a part of the third
    adj[20, 32:40] = True                                        # 15-ish blocks
    adj = adj | adj.T
    A = np.zeros((6 * N, 6 * N))
    for a in range(N):
        for b in range(a + 1, N):
            if adj[a, b]:
                blk = rng.standard_normal((6, 6))
                A[6 * a:6 * a + 6, 6 * b:6 * b + 6] = blk
                A[6 * b:6 * b + 6, 6 * a:6 * a + 6] = blk.T
    for a in range(N):
        d = rng.standard_normal((6, 6))
        A[6 * a:6 * a + 6, 6 * a:6 * a + 6] = d @ d.T
    A += np.diag(np.abs(A).sum(axis=1) * 0.6)                    # SPD, condition number of a few tens
    rhs = rng.standard_normal(6 * N)
    row_ptr, col, vals = [0], [], []
    for a in range(N):
        for b in np.flatnonzero(adj[a]):
            col.append(int(b))
            vals.append(A[6 * a:6 * a + 6, 6 * b:6 * b + 6].copy())
        row_ptr.append(len(col))
    widths = np.diff(row_ptr)
    assert widths.max() == 48 and widths.min() <= 8 and ((widths > 10) & (widths <= 20)).any() and ((widths > 20) & (widths <= 30)).any()
    dev = "cuda"
    rp = torch.tensor(row_ptr, dtype=torch.int32, device=dev)
    cl = torch.tensor(col, dtype=torch.int32, device=dev)
    vl = torch.from_numpy(np.stack(vals)).to(dev).contiguous()
    rh = torch.from_numpy(rhs).to(dev)
    iters = 120
    nbytes = lib.dfh_pcg_workspace_bytes(N, iters)
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=dev)
    xs = []
    for _ in range(2):
        x = torch.full((6 * N,), float("nan"), dtype=torch.float64, device=dev)
        _lib.check(lib.dfh_pcg_solve(rp.data_ptr(), cl.data_ptr(), vl.data_ptr(), rh.data_ptr(), N, iters, 0.0, 0.0, x.data_ptr(),
                                     ws.data_ptr(), ws.numel() * 8, current_stream_ptr()), "dfh_pcg_solve")
        torch.cuda.synchronize()
        xs.append(x.cpu().numpy())
    xo = np.linalg.solve(A, -rhs)
    assert np.abs(xs[0] - xo).max() <= 1e-9 * np.abs(xo).max()
    if not multilaunch:
        assert np.array_equal(xs[0], xs[1])                      # the persistent path has no atomics: same bits
    aborted = ctypes.c_long(-1)
    _lib.check(lib.dfh_pcg_status(current_stream_ptr(), ctypes.byref(aborted)), "dfh_pcg_status")
    assert aborted.value == 0


def test_persistent_pcg_timeout_is_reported(golden, monkeypatch):
    """A grid barrier of the persistent PCG that does not complete within its spin bound makes every workgroup leave
    (x = NaN, node_dq untouched) -- and the host must hear about it: the next synchronising call raises DfhTimeout,
    the process then takes the multi-launch path, and the next solve is sound.  The time-out is forced with a spin
    bound of 0 polls (option pcg_spin_limit), which no 2-workgroup barrier meets."""
    from dynamicfusion_body_amd import _lib
    g, verts, norms, corr, nbr, vidx, npos, ndq, nw, lw, rw = load(golden)
    lib = _lib.load()
    lib.dfh_pcg_set_mode(0)
    try:
        sv = make_solver(npos, ndq, nw, nbr, vidx, verts, norms, corr, nbr.shape[1], pcg_iters=12)
        assert sv.N > 8                                                  # more than one workgroup: a real grid barrier
        sv.build(lw, rw)
        sv.check_status()                                                # nothing pending
        v0 = sv.vals.clone()
        dq0 = sv.node_dq.clone()
        _lib.set_option("pcg_spin_limit", 0)
        sv.step(lw, rw, 0.5, 1e-3)                                       # build + PCG + twist update in one go
        _lib.set_option("pcg_spin_limit", None)
        with pytest.raises(_lib.DfhTimeout):
            sv.cost()
        assert torch.isnan(sv.dx).all() and torch.equal(sv.node_dq, dq0)  # no update was applied
        sv.check_status()                                                # the counter was cleared by the raise
        sv.check_status(completed_only=True)
        # the same through the host-side flag (what SlabFrame.step asks after its own synchronisation): silent while the
        # timed-out solve is still in flight or none has happened, raising once it has completed
        lib.dfh_pcg_set_mode(0)
        sv.vals.copy_(v0)
        _lib.set_option("pcg_spin_limit", 0)
        sv.step(lw, rw, 0.5, 1e-3)
        _lib.set_option("pcg_spin_limit", None)
        torch.cuda.synchronize()
        with pytest.raises(_lib.DfhTimeout):
            sv.check_status(completed_only=True)
        sv.check_status(completed_only=True)                             # cleared
        sv.check_status()
        # the process has fallen back to the multi-launch PCG: same system, sound answer
        sv.vals.copy_(v0)
        sv.build(lw, rw)
        sv.solve_linear(0.5, 1e-3)
        x_fallback = sv.dx.clone()
        assert torch.isfinite(x_fallback).all()
        lib.dfh_pcg_set_mode(0)
        sv.build(lw, rw)
        sv.solve_linear(0.5, 1e-3)
        sv.check_status()
        assert (sv.dx - x_fallback).abs().max() <= 1e-9 * x_fallback.abs().max()
    finally:
        lib.dfh_pcg_set_mode(0)


def test_planned_gather_walks_lists_of_every_length():
    """Few nodes and many samples: the block lists of the planned build run from a handful of rows to thousands (the
    golden scene's longest has a few dozen).  Lists of up to 12 live rows are added by their own wave, longer ones by the
    four waves of the workgroup together, those beyond 256 entries by the chunked walk -- all against the dense oracle."""
    rng = np.random.default_rng(21)
    N, k, S = 40, 4, 24000
    npos = rng.uniform(0, 40, size=(N, 3))
    nw = rng.uniform(8, 14, size=N)
    tw = rng.standard_normal((N, 6)) * np.array([0.02, 0.02, 0.02, 0.3, 0.3, 0.3])
    ndq = G.apply_twists(np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0, 0]), (N, 1)), tw)
    verts = np.concatenate([npos, rng.uniform(0, 40, size=(S - N, 3))])          # the first N samples are the nodes: nbr[:N] = node graph
    norms = rng.standard_normal((S, 3)); norms /= np.linalg.norm(norms, axis=1, keepdims=True)
    corr = verts + rng.standard_normal((S, 3)) * 0.3
    nbr = O.knn_bruteforce(verts, npos, k)
    vidx = np.arange(N)
    lw = np.array([1.0, 0, 0, 0, 0, 0, 0, 0])
    valid = rng.random(S) < 0.6
    sv = make_solver(npos, ndq, nw, nbr, vidx, verts, norms, corr, k, valid=valid)
    sv.build(lw, 0.0)
    lens = np.diff(sv.blk_ptr.cpu().numpy())
    assert lens.max() > 256 and ((lens > 12) & (lens <= 256)).any() and (lens <= 12).any()
    A, b = sv.dense_normal_equations()
    cost, cnt = sv.cost()
    r, J = G.data_residual_jacobian(ndq, verts, norms, corr, nbr, npos, nw, lw)
    rho, nb, Ji, Jj = G.reg_residual_jacobian(ndq, vidx, nbr, npos, nw, 1.0)
    Ao, bo, co = G.assemble_dense(N, r, J, nbr, rho * 0, nb, Ji * 0, Jj * 0, valid=valid)
    assert cnt == int(valid.sum())
    assert abs(cost - co) <= 1e-12 * co
    assert np.abs(A - Ao).max() <= 1e-11 * np.abs(Ao).max()
    assert np.abs(b - bo).max() <= 1e-11 * np.abs(bo).max()
    s1 = sv.system.clone()
    sv.build(lw, 0.0)
    assert torch.equal(s1, sv.system)                                             # no atomics anywhere: same bits


def test_planned_build_is_deterministic_and_matches_the_atomic_build(golden, monkeypatch):
    """dfh_gn_build_planned (per-run partial rows + gather, regulariser included) has no floating-point
    atomics: two builds give the same bits; dfh_gn_build (atomics) gives the same system to rounding."""
    g, verts, norms, corr, nbr, vidx, npos, ndq, nw, lw, rw = load(golden)
    rng = np.random.default_rng(3)
    valid = (rng.random(len(verts)) < 0.7).astype(np.uint8)
    _lib.set_option("py_gn_atomic", None)
    sv = make_solver(npos, ndq, nw, nbr, vidx, verts, norms, corr, nbr.shape[1], pcg_iters=5, valid=valid)
    sv.build(lw, rw)
    s1 = sv.system.clone()
    sv.build(lw, rw)
    assert torch.equal(s1, sv.system)
    # the regulariser's rows / lists ride along in the data rows' launches; in launches of their own: the same bits
    for switch in ("gn_reg_own_gather", "gn_reg_own_launch"):
        _lib.set_option(switch, 1)
        sv.build(lw, rw)
        assert torch.equal(s1, sv.system), switch
        _lib.set_option(switch, None)
    sv.build(lw, 0.0)                                   # without the regulariser rows
    s0 = sv.system.clone()
    _lib.set_option("py_gn_atomic", 1)
    sv.build(lw, rw)
    sa = sv.system.clone()
    sv.build(lw, 0.0)
    sa0 = sv.system.clone()
    _lib.set_option("py_gn_atomic", None)
    for p_, a_ in ((s1, sa), (s0, sa0)):
        assert float((p_ - a_).abs().max()) <= 1e-12 * float(a_.abs().max())
        assert p_[-1] == a_[-1] and p_[-1] == float(valid.sum())          # valid-sample count
    assert float((s1 - s0).abs().max()) > 0                               # the regulariser really contributes


def test_device_plan_builder_equals_the_torch_plan(monkeypatch):
    """dfh_gn_sort_samples / dfh_gn_plan_count / dfh_gn_plan_build (run scan, block look-up, two stable radix sorts) against the
    same bookkeeping assembled from torch ops: sample order, tuple keys, run ids, and all four CSR arrays, element for
    element -- ragged sample counts, several tiles, a pattern that covers the pairs and one that does not; lists short enough
    for the per-list sort's register and LDS networks and lists beyond them (five nodes, 800 000 samples)."""
    rng = np.random.default_rng(11)
    for N, k, S in ((40, 4, 1000), (300, 4, 70000), (9, 2, 257), (64, 3, 256), (5, 4, 800000)):   # the last: lists of thousands of entries
        npos = rng.uniform(0, 60, size=(N, 3)); nw = rng.uniform(3, 6, size=N)
        ndq = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0, 0]), (N, 1))
        pts = rng.uniform(0, 60, size=(S, 3)); nrm = rng.normal(size=(S, 3))
        node_nbr, _ = solve.sample_knn(npos, npos, nw, k)
        plans = {}
        for mode in ("device", "torch"):
            _lib.set_option("py_plan_torch", 1 if mode == "torch" else None)
            sv = solve.WarpSolver(knn=k, pcg_iters=5, distributed=False)
            sv.set_graph(npos, ndq, nw, node_nbr=node_nbr)
            sv.set_samples(pts, nrm)
            sv.set_correspondences(pts + 0.1)
            sv.build(np.array([1.0, 0, 0, 0, 0, 0, 0, 0]), 0.5)
            plans[mode] = sv
        a, b = plans["device"], plans["torch"]
        assert torch.equal(a._tuple_key, b._tuple_key) and torch.equal(a._order_index(), b._order_index())
        assert torch.equal(a.spos, b.spos) and torch.equal(a.snbr, b.snbr) and torch.equal(a.swts, b.swts) and torch.equal(a.snrm, b.snrm)
        assert a.n_rows == b.n_rows and torch.equal(a.run_id, b.run_id) and torch.equal(a._row_first, b._row_first)
        for name in ("blk_ptr", "blk_ent", "node_ptr", "node_ent", "rblk_ptr", "rblk_ent", "rnode_ptr", "rnode_ent"):
            assert torch.equal(getattr(a, name), getattr(b, name)), name
        assert torch.equal(a.vals, b.vals) and torch.equal(a.rhs, b.rhs)            # same plan, same kernels: same bits
        # a second sample set against the kept pattern: covered or not, both builders must agree on the verdict and the result
        pts2 = rng.uniform(0, 60, size=(S // 2 + 3, 3))
        for sv in (a, b):
            _lib.set_option("py_plan_torch", 1 if sv is b else None)
            sv.set_samples(pts2, rng.normal(size=pts2.shape) * 0 + 1.0)
            sv.set_correspondences(pts2 + 0.1)
            sv.build(np.array([1.0, 0, 0, 0, 0, 0, 0, 0]), 0.5)
        assert a.B == b.B and torch.equal(a.col, b.col) and torch.equal(a.blk_ent, b.blk_ent) and torch.equal(a.vals, b.vals)
    _lib.set_option("py_plan_torch", None)


def test_block_pattern_kept_across_sample_sets(golden):
    """New samples on an unchanged graph: the block pattern is kept when it covers the new node pairs (only the data
    plan is rebuilt) and grows otherwise; either way the system and the step equal a fresh solver's."""
    g, verts, norms, corr, nbr, vidx, npos, ndq, nw, lw, rw = load(golden)
    S = len(verts)
    rng = np.random.default_rng(11)
    sv = make_solver(npos, ndq, nw, nbr, vidx, verts, norms, corr, nbr.shape[1], pcg_iters=8)
    sv.build(lw, rw)
    keys_full = sv._pattern_keys.clone()
    first = np.arange(S) < S // 3

    def fresh_solver(sel):
        f = solve.WarpSolver(knn=nbr.shape[1], pcg_iters=8)
        f.set_graph(npos, ndq, nw, node_nbr=nbr[vidx])
        f.set_samples(verts[sel], norms[sel], nbr=nbr[sel])
        f.set_correspondences(corr[sel])
        return f
    for sel, grows in ((rng.random(S) < 0.5, False), (first, False)):
        sv.set_samples(verts[sel], norms[sel], nbr=nbr[sel])
        sv.set_correspondences(corr[sel])
        sv.build(lw, rw)
        assert torch.equal(sv._pattern_keys, keys_full)                       # covered: same pattern object contents
        fresh = fresh_solver(sel)
        fresh.build(lw, rw)
        A, b = sv.dense_normal_equations()
        Af, bf = fresh.dense_normal_equations()
        assert np.array_equal(A, Af) and np.array_equal(b, bf)
        assert sv.cost() == fresh.cost()
        sv.solve_linear(1e-3, 0.0); fresh.solve_linear(1e-3, 0.0)
        dxs, dxf = sv.dx.cpu().numpy(), fresh.dx.cpu().numpy()
        assert np.abs(dxs - dxf).max() <= 1e-12 * np.abs(dxf).max()           # extra all-zero blocks only reorder nothing
    # start from the small set, then hand over the full one: the pattern has to grow
    small = fresh_solver(first)
    small.build(lw, rw)
    b_small = small.B
    small.set_samples(verts, norms, nbr=nbr)
    small.set_correspondences(corr)
    small.build(lw, rw)
    assert small.B >= b_small
    A, b = small.dense_normal_equations()
    sv.set_samples(verts, norms, nbr=nbr); sv.set_correspondences(corr); sv.build(lw, rw)
    Af, bf = sv.dense_normal_equations()
    assert np.array_equal(A, Af) and np.array_equal(b, bf)


def test_huber_weights_vs_oracle(golden):
    """huber > 0: every data row and its residual scaled by sqrt(min(1, huber / |r|)) -- the IRLS form of the loss the
    reference's solver uses (least_squares(loss='huber'), core/fusion.py:389) -- against the oracle's J, r with the
    same weights; huber = 0 is the plain system."""
    g, verts, norms, corr, nbr, vidx, npos, ndq, nw, lw, rw = load(golden)
    N = len(npos)
    sv = make_solver(npos, ndq, nw, nbr, vidx, verts, norms, corr, nbr.shape[1])
    r, J = G.data_residual_jacobian(ndq, verts, norms, corr, nbr, npos, nw, lw)
    rho, nb, Ji, Jj = G.reg_residual_jacobian(ndq, vidx, nbr, npos, nw, rw)
    delta = float(np.percentile(np.abs(r), 60))                       # 40 % of the rows are down-weighted
    sc = np.sqrt(np.minimum(1.0, delta / np.maximum(np.abs(r), 1e-300)))
    assert 0.2 < (sc < 1).mean() < 0.6
    sv.build(lw, rw, huber=delta)
    A, b = sv.dense_normal_equations()
    cost, cnt = sv.cost()
    Jw = J * sc.reshape((-1,) + (1,) * (J.ndim - 1))
    Ao, bo, co = G.assemble_dense(N, r * sc, Jw, nbr, rho, nb, Ji, Jj)
    a_ = np.abs(r)
    huber_obj = float(np.where(a_ <= delta, 0.5 * r * r, delta * (a_ - 0.5 * delta)).sum() + 0.5 * float((rho * rho).sum()))
    assert abs(cost - huber_obj) <= 1e-12 * huber_obj and cnt == len(verts)           # the Huber objective, not the IRLS-weighted one
    assert np.abs(A - Ao).max() <= 1e-10 * np.abs(Ao).max() and np.abs(b - bo).max() <= 1e-10 * np.abs(bo).max()
    sv.build(lw, rw)
    A0, b0 = sv.dense_normal_equations()
    Ap, bp, _ = G.assemble_dense(N, r, J, nbr, rho, nb, Ji, Jj)
    assert np.abs(A0 - Ap).max() <= 1e-10 * np.abs(Ap).max() and np.abs(A0 - A).max() > 1e-6 * np.abs(A0).max()


def test_lm_loop_vs_oracle(golden):
    """Noise-free target from a known field, identity start: GPU LM costs follow the oracle's GN
    with the same damping schedule to 1e-6 relative; cost falls by > 100x."""
    g, verts, norms, corr, nbr, vidx, npos, ndq_true, nw, lw, rw = load(golden)
    N = len(npos)
    ndq_true = ndq_true / np.sqrt(np.sum(ndq_true[:, :4] ** 2, axis=1, keepdims=True))
    target, _ = O.warp(verts, ndq_true[nbr], npos[nbr], nw[nbr], normal=norms, m_lw=lw)
    ident = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0, 0]), (N, 1))
    rw = 1e-3
    sv = make_solver(npos, ident, nw, nbr, vidx, verts, norms, target, nbr.shape[1], pcg_iters=800)
    # 8 iterations: the damping falls 3x per iteration and from the 9th on the steps are so weakly damped that last-bit
    # differences of the linear solve grow ~1000x per iteration (observed 3e-9 / 3e-6 / 1e-5 at iterations 8 / 9 / 10)
    costs = sv.solve_lm(lw, rw, iters=8, lm_abs=1.0, lm_rel=0.0, adaptive=False)
    dqs = ident.copy()
    ocosts = []
    for it in range(8):
        dqs, c, dx = G.gn_step(dqs, verts, norms, target, nbr, vidx, npos, nw, lw, rw, lm=1.0 / 3.0 ** it)
        ocosts.append(c)
    assert np.allclose(costs[:8], ocosts, rtol=1e-6)
    assert costs[-1] < 1e-2 * costs[0]
    assert np.abs(sv.node_dq.cpu().numpy() - dqs).max() <= 1e-5
    # the reference's own cost function evaluated at the GPU's solution agrees with the reported cost
    f = O.computef(sv.node_dq.cpu().numpy().reshape(-1), verts, norms, target, nbr, vidx, npos, nw, lw, rw)
    assert abs(0.5 * f @ f - costs[-1]) <= 1e-9 * max(1.0, costs[-1])
    # adaptive LM never increases the cost, even from a bad damping
    sv2 = make_solver(npos, ident, nw, nbr, vidx, verts, norms, target, nbr.shape[1], pcg_iters=200)
    c2 = sv2.solve_lm(lw, 0.1, iters=8, lm_abs=1e-9, adaptive=True)
    assert all(b <= a * (1 + 1e-12) for a, b in zip(c2, c2[1:])) and c2[-1] < c2[0]


def test_rigid_gn_vs_oracle(golden):
    g = golden("g5_residuals")
    verts, norms = g["verts"], g["norms"]
    x_true = G.twist_exp_dq(np.array([0.05, -0.03, 0.08, 0.4, -0.2, 0.3]))
    corr = O.dqb_warp(x_true, verts)
    x, costs = solve.solve_rigid_gn(np.array([1.0, 0, 0, 0, 0, 0, 0, 0]), verts, norms, corr, iters=6)
    xo = np.array([1.0, 0, 0, 0, 0, 0, 0, 0]); oc = []
    for it in range(6):
        xo, c, dx = G.gn_step_rigid(xo, verts, norms, corr)
        oc.append(c)
    assert np.allclose(costs, oc, rtol=1e-9, atol=1e-14)
    assert np.abs(x - xo).max() <= 1e-10
    assert np.abs(O.DQTSE3(x) - O.DQTSE3(x_true)).max() < 1e-3
    # masked rows
    val = torch.from_numpy((np.arange(len(verts)) % 3 != 0))
    x2, c2 = solve.solve_rigid_gn(np.array([1.0, 0, 0, 0, 0, 0, 0, 0]), verts, norms, corr, valid=val, iters=1)
    keep = val.numpy()
    _, co, _ = G.gn_step_rigid(np.array([1.0, 0, 0, 0, 0, 0, 0, 0]), verts[keep], norms[keep], corr[keep])
    assert abs(c2[0] - co) <= 1e-12 * co
    # many workgroups: their sums are added in a fixed order (no atomics), so a solve repeats bit for bit
    rng = np.random.default_rng(31)
    V = rng.uniform(-1, 1, size=(300000, 3)); Nn = rng.normal(size=V.shape); Nn /= np.linalg.norm(Nn, axis=1, keepdims=True)
    C = O.dqb_warp(x_true, V) + 1e-3 * rng.normal(size=V.shape)
    runs = [solve.solve_rigid_gn(np.array([1.0, 0, 0, 0, 0, 0, 0, 0]), V, Nn, C, iters=4) for _ in range(3)]
    for xr, cr in runs[1:]:
        assert np.array_equal(xr, runs[0][0]) and list(cr) == list(runs[0][1])
    xo = np.array([1.0, 0, 0, 0, 0, 0, 0, 0])
    for it in range(4):
        xo, c, dx = G.gn_step_rigid(xo, V, Nn, C)
    assert np.abs(runs[0][0] - xo).max() <= 1e-9


def test_projective_association_vs_oracle():
    R = 64
    H, W, fx, cx, cy = scene.CAMERAS["C1"]
    K = scene.intrinsics(fx, cx, cy); Kinv = np.linalg.inv(K)
    scale, center, tdist = scene.grid_params(R)
    lw_cam = scene.view_extrinsic(25.0)
    dm = scene.render_depth(K, lw_cam, H, W, invalid_frac=0.05, seed=3)
    rng = np.random.default_rng(7)
    N, k, S = 40, 4, 3000
    npos, nw = scene.fibonacci_nodes(N, R)
    ndq = np.array([G.twist_exp_dq(rng.normal(size=6) * np.array([.02, .02, .02, .5, .5, .5])) for _ in range(N)])
    lw = G.twist_exp_dq(np.array([0.01, -0.02, 0.015, 0.3, -0.2, 0.1]))
    d = rng.normal(size=(S, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts = R / 2 + d * (scene.SPHERE_R / scale) * rng.uniform(0.8, 1.3, size=(S, 1))
    sv = solve.WarpSolver(knn=k)
    sv.set_graph(npos, ndq, nw)
    sv.set_samples(pts, d, sort=False)
    sv.associate_depth(torch.from_numpy(dm).cuda(), K, Kinv, lw_cam, scale, center, R / 2, lw, max_dist=0.0)
    loc = sv.snbr.cpu().numpy().astype(np.int64)
    warped = O.warp(pts, ndq[loc], npos[loc], nw[loc], m_lw=lw)
    co, vo = G.associate_depth(warped, K, Kinv, lw_cam, dm, scale, center, R / 2)
    assert np.array_equal(sv.valid.cpu().numpy().astype(bool), vo)
    assert vo.any() and (~vo).any()
    assert np.abs(sv.corr.cpu().numpy() - co).max() <= 1e-9
    # distance gate
    sv.associate_depth(torch.from_numpy(dm).cuda(), K, Kinv, lw_cam, scale, center, R / 2, lw, max_dist=3.0)
    gate = vo & (np.linalg.norm(co - warped, axis=1) <= 3.0)
    assert np.array_equal(sv.valid.cpu().numpy().astype(bool), gate)


def test_class_surface_matches_reference(golden):
    """Fusion.computef / computef_lw / warp / dq_blend / solve and FusionDM.computef_lw / solve with
    the reference's call shapes; values against the reference's golden outputs."""
    from dynamicfusion_body_amd import Fusion, FusionDM
    g, verts, norms, corr, nbr, vidx, npos, ndq, nw, lw, rw = load(golden)
    fu = Fusion(np.zeros((4, 4, 4)), 1.0, knn=nbr.shape[1])
    fu._nodes = [(int(vidx[i]), npos[i], ndq[i], float(nw[i])) for i in range(len(npos))]
    fu._vertices, fu._normals, fu._correspondences = verts, norms, corr
    fu._neighbor_look_up = [row for row in nbr]
    fu._lw = lw
    f = fu.computef(ndq.flatten(), 0.2, 0.001, rw)
    assert f.shape == g["computef_out"].shape and np.abs(f - g["computef_out"]).max() <= 1e-12
    assert np.abs(fu.computef_lw(g["lw2"], 0.2, 1) - g["computef_lw_out"]).max() <= 1e-12
    with pytest.raises(ValueError):
        fu.solve(correspondences=corr[:10])
    fu._correspondences = corr
    c0 = 0.5 * float(f @ f)
    fu.solve(precompute_lw=False, regularization_weight=rw, iterations=5)
    x = np.concatenate([n[2] for n in fu._nodes])
    f1 = fu.computef(x, 0.2, 0.001, rw)
    assert 0.5 * float(f1 @ f1) < 0.5 * c0
    # single-point helpers against the reference's outputs (golden g4)
    g4 = golden("g4_dqb")
    fu2 = Fusion(np.zeros((4, 4, 4)), 1.0, knn=int(g4["knn"]))
    fu2._nodes = [(0, g4["node_pos"][i], g4["node_dq"][i], float(g4["node_w"][i])) for i in range(len(g4["node_pos"]))]
    for i in range(6):
        p, n = g4["warp_P"][i], g4["warp_N"][i]
        loc = g4["warp_loc"][i]
        dqs = [fu2._nodes[j][2] for j in loc]
        assert np.abs(fu2.dq_blend(p, dqs, loc) - g4["blend_out"][i]).max() <= 1e-14
        a, b = fu2.warp(p, dqs, loc, normal=n, m_lw=g4["lw"])
        assert np.abs(a - g4["warp_pos_out"][i]).max() <= 1e-12 and np.abs(b - g4["warp_nrm_out"][i]).max() <= 1e-12
        a2, b2 = fu2.warp(p, normal=n, m_lw=g4["lw"])             # kd-tree form: knn+1 query, last dropped
        assert np.abs(a2 - a).max() <= 1e-12
    # FusionDM
    fd = FusionDM(1.0, np.eye(3), tsdf_res=4)
    keep = g["rigid_keep"]
    fd._vertices, fd._normals = verts, norms
    fd._corridx = list(keep); fd._correspondences = [corr[i] for i in keep]
    assert np.abs(fd.computef_lw(g["rigid_x"]) - g["rigid_out"]).max() <= 1e-12
    x_true = G.twist_exp_dq(np.array([0.03, 0.01, -0.04, 0.2, 0.1, -0.3]))
    fd._correspondences = list(O.dqb_warp(x_true, verts[keep]))
    fd._lw = np.array([1, 0, 0, 0, 0, 0, 0, 0], dtype=np.float32)
    fd.solve(None)
    assert np.abs(O.DQTSE3(fd._lw) - O.DQTSE3(x_true)).max() < 1e-3
    assert fd.last_costs[-1][-1] < 1e-8
    # FusionDM's second copy of the non-rigid methods (core/fusion_dm.py:369-560: the same statements as Fusion's)
    fd2 = FusionDM(1.0, np.eye(3), tsdf_res=4, knn=nbr.shape[1], write_warpfield=False)
    fd2._nodes = [(int(vidx[i]), npos[i], ndq[i], float(nw[i])) for i in range(len(npos))]
    fd2._vertices, fd2._normals, fd2._correspondences = verts, norms, corr
    fd2._neighbor_look_up = [row for row in nbr]
    fd2._lw = lw
    f2 = fd2.computef(ndq.flatten(), 0.2, 0.001, rw)
    assert np.abs(f2 - g["computef_out"]).max() <= 1e-12
    S = fd2.computeSparsity(len(f2), 8 * len(npos))
    assert S.shape == (len(f2), 8 * len(npos)) and S[0, 8 * int(nbr[0][0])] == 1
    for i in range(3):
        p_, n_ = g4["warp_P"][i], g4["warp_N"][i]
        fd3 = FusionDM(1.0, np.eye(3), tsdf_res=4, knn=int(g4["knn"]))
        fd3._nodes = fu2._nodes
        loc = g4["warp_loc"][i]
        dqs = [fu2._nodes[j][2] for j in loc]
        assert np.abs(fd3.dq_blend(p_, dqs, loc) - g4["blend_out"][i]).max() <= 1e-14
        a, b = fd3.warp(p_, dqs, loc, normal=n_, m_lw=g4["lw"])
        assert np.abs(a - g4["warp_pos_out"][i]).max() <= 1e-12 and np.abs(b - g4["warp_nrm_out"][i]).max() <= 1e-12
    g8 = golden("g8_graph_io")
    fd4 = FusionDM(1.0, np.eye(3), tsdf_res=4, knn=int(g8["knn"]), write_warpfield=False)
    fd4._vertices, fd4._radius = g8["verts"], float(g8["radius"])
    fd4.construct_graph()
    assert np.array_equal(np.array([n[1] for n in fd4._nodes]), g8["cg_pos"]) and np.array_equal(np.asarray(fd4._neighbor_look_up), g8["cg_lookup"])


def test_setup_correspondences_matches_reference(golden):
    """FusionDM.setupCorrespondences / the batch warp / the selection kernel against the reference's
    outputs (golden g7: marching cubes patched to return given live vertices)."""
    from dynamicfusion_body_amd import FusionDM
    g = golden("g7_correspondences")
    k = int(g["knn"])
    fd = FusionDM(1.0, np.eye(3), tsdf_res=4, knn=k)
    fd._vertices, fd._normals, fd._lw = g["verts"], g["norms"], g["lw"]
    for tol in (1.0, 0.35):
        fd.setupCorrespondences(None, tolerance=tol, live_vertices=g["lverts"])
        assert np.array_equal(np.array(fd._corridx), g["dm_corridx_%g" % tol])
        assert np.array_equal(np.array(fd._correspondences), g["dm_corr_%g" % tol])
    # non-rigid: warp through the node graph, then the same selection (Fusion.setupCorrespondences 'clpts')
    vp, wn = solve.warp_points(g["verts"], g["norms"], g["lw"], nbr=g["nbr"], node_dq=g["node_dq"], node_pos=g["node_pos"], node_w=g["node_w"])
    vo, no = O.warp(g["verts"], g["node_dq"][g["nbr"]], g["node_pos"][g["nbr"]], g["node_w"][g["nbr"]], normal=g["norms"], m_lw=g["lw"])
    assert np.abs(vp.cpu().numpy() - vo).max() <= 1e-12 and np.abs(wn.cpu().numpy() - no).max() <= 1e-12
    corr, cost, keep = solve.closest_correspondences(vp, wn, g["lverts"], k, 0.2)
    assert np.array_equal(corr.cpu().numpy(), g["nr_corr"])
    bo, co, ko = O.closest_correspondences(vo, no, g["lverts"], k, 0.2)
    assert np.abs(cost.cpu().numpy() - co).max() <= 1e-12 and np.array_equal(keep.cpu().numpy().astype(bool), ko)
    with pytest.raises(ValueError):
        solve.closest_correspondences(vp, wn, g["lverts"][:2], k, 0.2)
    # the class method: Fusion.setupCorrespondences, closest-points branch (core/fusion.py:243-314)
    from dynamicfusion_body_amd import Fusion
    def fresh():
        fu = Fusion(np.zeros((4, 4, 4)), 1.0, knn=k, write_warpfield=False)
        fu._vertices, fu._normals, fu._lw = g["verts"].copy(), g["norms"].copy(), g["lw"]
        fu._neighbor_look_up = g["nbr"].copy()
        fu._nodes = [(0, g["node_pos"][i], g["node_dq"][i], float(g["node_w"][i])) for i in range(len(g["node_pos"]))]
        fu._radius = 1.5
        return fu
    fu = fresh()
    fu.setupCorrespondences(None, method='clpts', prune_result=False, tolerance=0.2, live_vertices=g["lverts"])
    assert np.array_equal(np.asarray(fu._correspondences), g["nr_corr"]) and len(fu._vertices) == len(g["verts"])
    fu = fresh()
    fu.setupCorrespondences(None, method='clpts', prune_result=True, tolerance=0.2, live_vertices=g["lverts"])
    assert 0 < ko.sum() < len(ko)
    assert np.array_equal(fu._vertices, g["verts"][ko]) and np.array_equal(fu._normals, g["norms"][ko])
    assert np.array_equal(fu._neighbor_look_up, g["nbr"][ko]) and np.array_equal(fu._correspondences, g["nr_corr"][ko])
    assert fu._faces is None
    for nd in fu._nodes:                                             # nodes re-anchored to their nearest kept vertex
        assert nd[0] == int(np.argmin(np.linalg.norm(g["verts"][ko] - nd[1], axis=1))) and nd[3] == 3.0


def test_fusion_frame_loop_with_mesh_correspondences():
    """The reference's non-rigid frame loop on the Fusion class, every step on the device path: initial mesh and
    graph, live mesh + closest-point correspondences, three rounds of [associate -> minimise] (`solve(method=
    'clpts')` re-associating against the stored live volume), DQB TSDF update, graph update."""
    from dynamicfusion_body_amd import Fusion
    R = 40
    X, Y, Z = np.meshgrid(*(np.arange(R),) * 3, indexing="ij")
    sd = lambda c, r: np.clip(np.sqrt((X - c[0]) ** 2 + (Y - c[1]) ** 2 + (Z - c[2]) ** 2) - r, -3.0, 3.0).astype(np.float32)
    fu = Fusion(sd((19.6, 20.2, 19.9), 11.5), 3.0, subsample_rate=3.0, knn=4, marching_cubes_step_size=1, write_warpfield=False)
    fu._lw = np.array([1.0, 0, 0, 0, 0, 0, 0, 0])
    fu.initialize_canonical()
    nv0, nn0 = len(fu._vertices), len(fu._nodes)
    live = sd((20.3, 19.8, 20.4), 11.7)                               # moved and slightly inflated
    fu.setupCorrespondences(live, method='clpts', prune_result=True, tolerance=2.0)
    assert 0.5 * nv0 < len(fu._vertices) <= nv0 and len(fu._correspondences) == len(fu._vertices)
    f0 = fu.computef(np.concatenate([n[2] for n in fu._nodes]), 0.2, 0.001, 1)
    c0 = 0.5 * float(f0 @ f0)
    fu.solve(method='clpts', precompute_lw=True, regularization_weight=1, iterations=8)
    f1 = fu.computef(np.concatenate([n[2] for n in fu._nodes]), 0.2, 0.001, 1)
    c1 = 0.5 * float(f1 @ f1)
    assert c1 < 0.2 * c0, (c0, c1)
    assert 1 <= len(fu.last_costs) <= 3
    T0 = fu._tsdf.copy()
    fu.updateTSDF(live)
    assert np.abs(fu._tsdf - T0).max() > 0.05
    fu.update_graph()
    assert len(fu._nodes) >= nn0 and fu._faces is not None and len(fu._correspondences) == 0


def test_icp_compute_live_tsdf_recovers_rigid_motion():
    """compute_live_tsdf(useICP=True) (reference core/fusion_dm.py:149-164): view 0 becomes the
    canonical volume, the second (displaced scene) view is aligned by three rounds of
    [setupCorrespondences -> GN on `_lw`] and fused with updateTSDF."""
    from dynamicfusion_body_amd import FusionDM
    R = 64
    H, W, fx, cx, cy = scene.CAMERAS["C1"]
    K = scene.intrinsics(fx, cx, cy)
    f = FusionDM(0.6, K, tsdf_res=R)
    avg = np.array([-0.03, -0.43, -5.6], dtype='float32'); c = avg.astype(np.float64)
    lw = scene.view_extrinsic(0.0, centre=c)
    scale = 10 * 1.3 / R
    shift = np.array([1.2, -0.8, 0.6]) * scale                    # metres: about one voxel
    d0 = scene.render_depth(K, lw, H, W, invalid_frac=0.0, sphere_c=c, sphere_r=2.5, wall_z=-1.0)
    d1 = scene.render_depth(K, lw, H, W, invalid_frac=0.0, sphere_c=c + shift, sphere_r=2.5, wall_z=-1.0 + shift[2])
    T, Wt = f.compute_live_tsdf([d0, d1], [lw, lw], useICP=True)
    assert len(f.last_costs) == 3 and f.last_costs[-1][-1] < f.last_costs[0][0]
    M = O.DQTSE3(f._lw)
    # `_lw` maps canonical index space onto the live frame.  Point-to-plane on a sphere observes the
    # motion along the viewing direction well (z) and tangential sliding only weakly (x, y): the
    # right sign and part of the magnitude
    t_true = shift / scale
    assert np.abs(M[:3, :3] - np.eye(3)).max() < 0.05
    assert abs(M[2, 3] - t_true[2]) < 0.15
    assert M[0, 3] * t_true[0] > 0 and M[1, 3] * t_true[1] > 0 and np.abs(M[:2, 3]).max() < 1.5 * np.abs(t_true[:2]).max()
    assert (Wt > 1).any()                                          # the second view was fused in
