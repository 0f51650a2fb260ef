"""Generate tests/golden/*.npz by IMPORTING AND RUNNING THE REFERENCE (build container only).

    cd /root/repo && python -B tests/golden/make_golden.py

The reference tree cannot travel to the GPU box, so its outputs on seeded inputs are
committed here as data (numeric arrays only).  Inputs are stored next to the expected
outputs so the tests never need the reference.  Set names follow SURVEY.md §8(c):
  g1_primitives  util.py primitives (+ the reference's doctest constants)
  g2_fuse_depths FusionDM.fuseDepths   (A1)  R=20, 48x64 depth, rotated lw, wmax=3
  g3_rigid       FusionDM.updateTSDF   (A3)  R=20, non-unit DQ, 1 and 4 repeats
  g4_dqb         Fusion.updateTSDF/warp/dq_blend (A4/A5) R=14, N=24, k=4
  g5_residuals   FusionDM.computef_lw (A9), Fusion.computef/computef_lw (A10) + FD
                 Jacobian columns w.r.t. left twists exp(eps e_k) (x) dq_j
  g6_config1     BASELINE config 1 (R=64, 320x240): statistics + 4096 sampled voxels
"""
import contextlib
import io
import os
import sys

import numpy as np
from scipy.spatial import KDTree

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import ref_import  # noqa: E402
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("scene", os.path.join(ROOT, "dynamicfusion_body_amd", "scene.py"))
scene = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(scene)

util, FusionDM, Fusion = ref_import.load()


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def rand_unit_dq(rng, trans=1.0):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    if q[0] < 0:
        q = -q
    t = rng.normal(size=3) * trans
    qe = 0.5 * util.quaternion_multiply([0, t[0], t[1], t[2]], q)
    return np.append(q, qe)


def small_dq(rng, rot=0.05, trans=0.5, scale_jitter=0.0):
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    ang = rng.normal() * rot
    q = np.append(np.cos(ang / 2), np.sin(ang / 2) * ax)
    t = rng.normal(size=3) * trans
    qe = 0.5 * util.quaternion_multiply([0, t[0], t[1], t[2]], q)
    dq = np.append(q, qe)
    return dq * (1.0 + scale_jitter * rng.normal())


# ---------------------------------------------------------------------------- G1
def g1():
    rng = np.random.default_rng(101)
    n = 64
    out = {}
    q1 = rng.normal(size=(n, 4)); q0 = rng.normal(size=(n, 4))
    out["qmul_q1"], out["qmul_q0"] = q1, q0
    out["qmul_out"] = np.array([util.quaternion_multiply(a, b) for a, b in zip(q1, q0)])
    out["qmul_doctest"] = util.quaternion_multiply([4, 1, -2, 3], [8, -5, 6, 7])
    d1 = rng.normal(size=(n, 8)); d2 = rng.normal(size=(n, 8))
    out["dqmul_a"], out["dqmul_b"] = d1, d2
    out["dqmul_out"] = np.array([util.dual_quaternion_multiply(a, b) for a, b in zip(d1, d2)])
    out["dqconj_out"] = np.array([util.dual_quaternion_conjugate(a) for a in d1])
    out["dqconj_probe"] = util.dual_quaternion_conjugate(np.arange(1.0, 9.0))
    # dqb_warp: unit, non-unit (f64) and float32-typed DQs
    dq_unit = np.array([rand_unit_dq(rng) for _ in range(n)])
    dq_non = dq_unit * rng.uniform(0.7, 1.4, size=(n, 1)) + 0.05 * rng.normal(size=(n, 8))
    dq_f32 = dq_non.astype(np.float32)
    pos = rng.uniform(-5, 60, size=(n, 3))
    out["warp_dq_unit"], out["warp_dq_non"], out["warp_dq_f32"], out["warp_pos"] = dq_unit, dq_non, dq_f32, pos
    out["warp_unit_out"] = np.array([util.dqb_warp(d, p) for d, p in zip(dq_unit, pos)])
    out["warp_non_out"] = np.array([util.dqb_warp(d, p) for d, p in zip(dq_non, pos)])
    out["warp_f32_out"] = np.array([util.dqb_warp(d, p) for d, p in zip(dq_f32, pos)])
    nrm = rng.normal(size=(n, 3))
    out["warpn_n"] = nrm
    out["warpn_non_out"] = np.array([util.dqb_warp_normal(d, p) for d, p in zip(dq_non, nrm)])
    out["warpn_f32_out"] = np.array([util.dqb_warp_normal(d, p) for d, p in zip(dq_f32, nrm)])
    # SE3 <-> DQ
    out["dqtse3_out"] = np.array([util.DQTSE3(d) for d in dq_unit])
    out["se3tdq_roundtrip"] = np.array([util.SE3TDQ(util.DQTSE3(d)) for d in dq_unit])
    out["qmat_doctest"] = util.quaternion_matrix([0.99810947, 0.06146124, 0, 0])
    # projection
    K = scene.intrinsics(60.3, 31.7, 23.6)
    pp = rng.normal(size=(n, 3)) * [1, 1, 0.5] + [0, 0, 2]
    pp[:4, 2] = 0.0
    uv = [util.project_to_pixel(K, p) for p in pp]
    out["proj_K"], out["proj_pos"] = K, pp
    out["proj_ok"] = np.array([a[0] is not None for a in uv])
    out["proj_u"] = np.array([np.nan if a[0] is None else a[0] for a in uv])
    out["proj_v"] = np.array([np.nan if a[1] is None else a[1] for a in uv])
    # trilinear sampler: interior, integer coords, faces == R-1, invalid, swapped-fraction probe
    vol = rng.normal(size=(9, 10, 11))
    ipos = [rng.uniform(0, [8, 9, 10]) for _ in range(40)]
    ipos += [np.array([3.0, 4.0, 5.0]), np.array([0.0, 0.0, 0.0]), np.array([8.0, 9.0, 10.0]),
             np.array([8.0, 2.5, 3.25]), np.array([1.5, 9.0, 0.0]), np.array([2.0, 3.5, 10.0])]
    ipos += [np.array([-0.01, 1, 1]), np.array([1, -1e-9, 1]), np.array([1, 1, -3.0]),
             np.array([8.0000001, 1, 1]), np.array([1, 9.5, 1]), np.array([1, 1, 10.001])]
    ipos = np.array(ipos)
    iv = [util.interpolate_tsdf(p, vol) for p in ipos]
    out["interp_vol"], out["interp_pos"] = vol, ipos
    out["interp_valid"] = np.array([x is not None for x in iv])
    out["interp_out"] = np.array([np.nan if x is None else x for x in iv])
    lin = np.fromfunction(lambda x, y, z: 9 * x + 3 * y + z, (3, 3, 3))
    out["interp_probe"] = np.array(util.interpolate_tsdf(np.array([0.5, 0.25, 1.75]), lin))
    # robust losses
    xs = rng.normal(size=n) * 2
    out["loss_x"] = xs
    out["huber_out"] = np.array([util.huber_loss(x, 0.7) for x in xs])
    out["tukey_out"] = np.array([util.tukey_biweight_loss(x, 1.3) for x in xs])
    np.savez_compressed(os.path.join(HERE, "g1_primitives.npz"), **out)
    print("g1 ok")


# ---------------------------------------------------------------------------- G2
def make_dm_fixture(K, lw, H, W, seed, invalid=0.05):
    return scene.render_depth(K, lw, H, W, invalid_frac=invalid, seed=seed)


def g2():
    R = 20
    H, W = 48, 64
    K = scene.intrinsics(61.37, 31.71, 23.63)      # chosen so no voxel sits on a .5 pixel tie
    scale, center, tdist = scene.grid_params(R)
    center32 = center.astype(np.float32)          # compute_live_tsdf passes a float32 centre (fusion_dm.py:106)
    out = dict(K=K, scale=scale, center=center32, tdist=tdist, wmax=3.0, R=R)
    f = FusionDM(tdist, K, tsdf_res=R)
    tsdf = np.zeros((R, R, R)) + tdist
    tsdfw = np.zeros((R, R, R))
    angles = [0.0, 20.0, -35.0, 50.0, 5.0]
    lws, dms = [], []
    for i, a in enumerate(angles):
        lw = scene.view_extrinsic(a)
        dm = make_dm_fixture(K, lw, H, W, seed=7 + i)
        lws.append(lw); dms.append(dm)
        tsdf, tsdfw = quiet(f.fuseDepths, dm, lw, tsdf, tsdfw, scale=scale, center=center32, wmax=3.0)
        if i == 0:
            out["T_after1"], out["W_after1"] = tsdf.copy(), tsdfw.copy()
    out["lws"], out["dms"] = np.array(lws), np.array(dms)
    out["T_after5"], out["W_after5"] = tsdf, tsdfw
    np.savez_compressed(os.path.join(HERE, "g2_fuse_depths.npz"), **out)
    print("g2 ok: updated", (out["W_after1"] > 0).mean(), (tsdfw > 0).mean(), "sat", (tsdfw == 3).mean())


# ---------------------------------------------------------------------------- G3
def sphere_volume(R, centre, radius, tdist):
    g = np.stack(np.meshgrid(*[np.arange(R, dtype=np.float64)] * 3, indexing="ij"), axis=-1)
    return np.clip(np.linalg.norm(g - centre, axis=-1) - radius, -tdist * 1.5, tdist * 1.5)


def g3():
    rng = np.random.default_rng(303)
    R = 20
    tdist = 2.0
    K = np.eye(3)
    f = FusionDM(tdist, K, tsdf_res=R)
    f._tsdf = sphere_volume(R, np.array([9.5, 10.2, 9.1]), 6.0, tdist)
    f._tsdfw = (rng.random((R, R, R)) < 0.5).astype(np.float64) * rng.integers(1, 4, size=(R, R, R))
    lw = small_dq(rng, rot=0.12, trans=0.8) * 1.03      # non-unit on purpose (fusion_dm.py:282)
    f._lw = lw
    out = dict(T0=f._tsdf.copy(), W0=f._tsdfw.copy(), lw=lw, tdist=tdist, wmax=5.0)
    lives = []
    for r in range(4):
        live = sphere_volume(R, np.array([10.3, 9.7, 9.9]) + 0.2 * r, 6.2, tdist) + 0.01 * rng.normal(size=(R, R, R))
        lives.append(live)
        quiet(f.updateTSDF, live, wmax=5.0)
        if r == 0:
            out["T_after1"], out["W_after1"] = f._tsdf.copy(), f._tsdfw.copy()
    out["lives"] = np.array(lives)
    out["T_after4"], out["W_after4"] = f._tsdf.copy(), f._tsdfw.copy()
    np.savez_compressed(os.path.join(HERE, "g3_rigid.npz"), **out)
    print("g3 ok: updated", (out["W_after1"] != out["W0"]).mean())


# ---------------------------------------------------------------------------- G4
def bare_fusion(tdist, knn, node_pos, node_dq, node_w, lw, vert_idx=None):
    fu = Fusion.__new__(Fusion)         # ctor is broken at HEAD (core/fusion.py:51)
    fu._tdist = tdist
    fu._knn = knn
    fu._verbose = False
    fu._lw = lw
    fu._curr_tsdf = None
    n = len(node_pos)
    vert_idx = np.zeros(n, dtype=int) if vert_idx is None else vert_idx
    fu._nodes = [(int(vert_idx[i]), node_pos[i], node_dq[i], float(node_w[i])) for i in range(n)]
    fu._kdtree = KDTree(node_pos)
    return fu


def g4():
    rng = np.random.default_rng(404)
    R, N, k = 14, 24, 4
    tdist = 2.0
    node_pos = rng.uniform(1.5, R - 2.5, size=(N, 3))
    node_dq = np.array([small_dq(rng, rot=0.08, trans=0.4, scale_jitter=0.02) for _ in range(N)])
    node_w = rng.uniform(2.0, 4.0, size=N)
    lw = small_dq(rng, rot=0.05, trans=0.3) * 0.98
    fu = bare_fusion(tdist, k, node_pos, node_dq, node_w, lw)
    fu._tsdf = sphere_volume(R, np.array([6.6, 7.1, 6.4]), 4.0, tdist)
    fu._tsdfw = (rng.random((R, R, R)) < 0.6).astype(np.float64) * rng.uniform(0.5, 3.0, size=(R, R, R))
    out = dict(T0=fu._tsdf.copy(), W0=fu._tsdfw.copy(), node_pos=node_pos, node_dq=node_dq, node_w=node_w,
               lw=lw, tdist=tdist, wmax=9.0, knn=k)
    lives = []
    for r in range(3):
        live = sphere_volume(R, np.array([6.9, 6.8, 6.7]) + 0.15 * r, 4.2, tdist) + 0.01 * rng.normal(size=(R, R, R))
        lives.append(live)
        quiet(fu.updateTSDF, live, wmax=9.0)
        if r == 0:
            out["T_after1"], out["W_after1"] = fu._tsdf.copy(), fu._tsdfw.copy()
    out["lives"] = np.array(lives)
    out["T_after3"], out["W_after3"] = fu._tsdf.copy(), fu._tsdfw.copy()
    # warp / dq_blend on free points (kd-tree form and explicit form), with normals
    P = rng.uniform(0, R - 1, size=(48, 3))
    Nn = rng.normal(size=(48, 3)); Nn /= np.linalg.norm(Nn, axis=1, keepdims=True)
    wp, wn, bl, locs = [], [], [], []
    for p, nn in zip(P, Nn):
        d, idx = fu._kdtree.query(p, k=k + 1)
        loc = idx[:-1]
        dqs = [fu._nodes[i][2] for i in loc]
        a, b = fu.warp(p, dqs, loc, normal=nn, m_lw=lw)
        wp.append(a); wn.append(b); locs.append(loc)
        bl.append(fu.dq_blend(p, dqs, loc))
    out.update(warp_P=P, warp_N=Nn, warp_loc=np.array(locs), warp_pos_out=np.array(wp),
               warp_nrm_out=np.array(wn), blend_out=np.array(bl))
    # dmax form and the zero-blend guard
    out["blend_dmax_out"] = np.array([fu.dq_blend(p, [fu._nodes[i][2] for i in l], l, dmax=3.5) for p, l in zip(P[:8], locs[:8])])
    np.savez_compressed(os.path.join(HERE, "g4_dqb.npz"), **out)
    print("g4 ok: updated", (out["W_after1"] != out["W0"]).mean())


# ---------------------------------------------------------------------------- G5
def twist_exp_dq(xi):
    """unit dual quaternion exp of a twist xi = (omega, v): rotation exp(omega), translation v
    (first-order in v is all the FD probe needs; we use the exact rotation and t = v)."""
    om, v = xi[:3], xi[3:]
    th = np.linalg.norm(om)
    if th < 1e-300:
        q = np.array([1.0, 0, 0, 0])
    else:
        q = np.append(np.cos(th / 2), np.sin(th / 2) * om / th)
    qe = 0.5 * util.quaternion_multiply([0, v[0], v[1], v[2]], q)
    return np.append(q, qe)


def g5():
    rng = np.random.default_rng(505)
    V, N, k = 200, 20, 4
    # vertices on a sphere of radius 8 around (16,16,16), outward normals
    d = rng.normal(size=(V, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    verts = 16 + 8 * d + 0.05 * rng.normal(size=(V, 3))
    norms = d + 0.05 * rng.normal(size=(V, 3)); norms /= np.linalg.norm(norms, axis=1, keepdims=True)
    vidx = rng.choice(V, size=N, replace=False)
    node_pos = verts[vidx].copy()
    node_dq = np.array([small_dq(rng, rot=0.06, trans=0.3, scale_jitter=0.01) for _ in range(N)])
    node_w = rng.uniform(5.0, 7.0, size=N)
    lw = small_dq(rng, rot=0.04, trans=0.2)
    corr = verts + 0.3 * rng.normal(size=(V, 3))
    fu = bare_fusion(1.0, k, node_pos, node_dq, node_w, lw, vert_idx=vidx)
    fu._vertices, fu._normals, fu._correspondences = verts, norms, corr
    fu._neighbor_look_up = [fu._kdtree.query(v, k=k)[1] for v in verts]      # core/fusion.py:121-123
    x = node_dq.flatten()
    rw = 0.7
    f0 = fu.computef(x, 0.2, 0.001, rw)
    out = dict(verts=verts, norms=norms, corr=corr, vidx=vidx, node_pos=node_pos, node_dq=node_dq,
               node_w=node_w, lw=lw, rw=rw, nbr=np.array(fu._neighbor_look_up), knn=k,
               computef_out=f0, cost=0.5 * np.inner(f0, f0))
    lw2 = small_dq(rng, rot=0.1, trans=0.5) * 1.02
    out["lw2"] = lw2
    out["computef_lw_out"] = fu.computef_lw(lw2, 0.2, 1)
    # FD Jacobian columns of computef w.r.t. left twists on a few nodes.  The reference rounds the
    # intermediate warped point to float32 (core/util.py:69), so its residual is a staircase with
    # ~2e-6 steps: use LARGE steps h, 2h and Richardson extrapolation (4 D(h) - D(2h))/3, which
    # leaves O(h^4) truncation and ~1e-6/h staircase noise (~1e-4 for h = 1e-2).
    eps = 1e-2
    probe_nodes = np.array([0, 3, 7, 12, 19])

    def central(fun, base, j, c, h):
        xi = np.zeros(6); xi[c] = h
        xp = base.copy(); xm = base.copy()
        xp[j] = util.dual_quaternion_multiply(twist_exp_dq(xi), base[j])
        xm[j] = util.dual_quaternion_multiply(twist_exp_dq(-xi), base[j])
        return (fun(xp) - fun(xm)) / (2 * h)

    fun = lambda dq: fu.computef(dq.flatten(), 0.2, 0.001, rw)
    cols = np.zeros((len(probe_nodes), 6, len(f0)))
    for a, j in enumerate(probe_nodes):
        for c in range(6):
            cols[a, c] = (4 * central(fun, node_dq, j, c, eps) - central(fun, node_dq, j, c, 2 * eps)) / 3
    out["fd_nodes"], out["fd_cols"], out["fd_eps"] = probe_nodes, cols, eps
    # rigid residual (FusionDM.computef_lw), non-unit x
    fd = FusionDM(1.0, np.eye(3), tsdf_res=4)
    keep = np.sort(rng.choice(V, size=150, replace=False))
    fd._vertices, fd._normals = verts, norms
    fd._corridx = list(keep)
    fd._correspondences = [corr[i] for i in keep]
    out["rigid_keep"] = keep
    out["rigid_x"] = lw2
    out["rigid_out"] = fd.computef_lw(lw2)
    cols6 = np.zeros((6, len(keep)))

    def central6(c, h):
        xi = np.zeros(6); xi[c] = h
        return (fd.computef_lw(util.dual_quaternion_multiply(twist_exp_dq(xi), lw2)) -
                fd.computef_lw(util.dual_quaternion_multiply(twist_exp_dq(-xi), lw2))) / (2 * h)
    for c in range(6):
        cols6[c] = (4 * central6(c, eps) - central6(c, 2 * eps)) / 3
    out["rigid_fd_cols"] = cols6
    np.savez_compressed(os.path.join(HERE, "g5_residuals.npz"), **out)
    print("g5 ok: len", len(f0), "cost", out["cost"])


# ---------------------------------------------------------------------------- G6
def g6():
    R = 64
    H, W, fx, cx, cy = scene.CAMERAS["C1"]
    K = scene.intrinsics(fx, cx, cy)
    scale, center, tdist = scene.grid_params(R)
    lw = scene.view_extrinsic(0.0)
    dm = scene.render_depth(K, lw, H, W)
    f = FusionDM(tdist, K, tsdf_res=R)
    tsdf = np.zeros((R, R, R)) + tdist
    tsdfw = np.zeros((R, R, R))
    import time
    t0 = time.perf_counter()
    tsdf, tsdfw = quiet(f.fuseDepths, dm, lw, tsdf, tsdfw, scale=scale, center=center)
    dt = time.perf_counter() - t0
    rng = np.random.default_rng(606)
    samp = rng.choice(R ** 3, size=4096, replace=False)
    out = dict(R=R, scale=scale, center=center, tdist=tdist, updated=int((tsdfw > 0).sum()),
               sumW=tsdfw.sum(), sumT=tsdf.sum(), minT=tsdf.min(), maxT=tsdf.max(),
               sample_idx=samp, sample_T=tsdf.reshape(-1)[samp], sample_W=tsdfw.reshape(-1)[samp],
               # packed update mask: bit-exact visibility/update parity for the whole 64^3 volume
               mask_packed=np.packbits((tsdfw > 0).reshape(-1)),
               ref_seconds=dt)
    np.savez_compressed(os.path.join(HERE, "g6_config1.npz"), **out)
    print("g6 ok: updated", out["updated"], "of", R ** 3, "ref time", dt)


# ---------------------------------------------------------------------------- G7
def g7():
    """setupCorrespondences of both classes with marching cubes (skimage, absent) monkey-patched to
    return a given live vertex set: everything after that call is the reference's own code."""
    rng = np.random.default_rng(707)
    V, L, k = 300, 420, 4
    d = rng.normal(size=(V, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    verts = 16 + 8 * d
    norms = d + 0.05 * rng.normal(size=(V, 3)); norms /= np.linalg.norm(norms, axis=1, keepdims=True)
    dl = rng.normal(size=(L, 3)); dl /= np.linalg.norm(dl, axis=1, keepdims=True)
    lverts = 16.3 + 8.2 * dl + 0.4 * rng.normal(size=(L, 3))
    lw = small_dq(rng, rot=0.05, trans=0.4) * 1.01
    out = dict(verts=verts, norms=norms, lverts=lverts, lw=lw, knn=k)
    fd = FusionDM(1.0, np.eye(3), tsdf_res=4, knn=k)
    fd._vertices, fd._normals, fd._lw = verts, norms, lw
    fd.marching_cubes = lambda tsdf=None, step_size=1: (lverts, None, None, None)
    for tol in (1.0, 0.35):
        quiet(fd.setupCorrespondences, np.zeros((2, 2, 2)), tolerance=tol)
        out["dm_corridx_%g" % tol] = np.array(fd._corridx)
        out["dm_corr_%g" % tol] = np.array(fd._correspondences)
    # non-rigid: Fusion.setupCorrespondences(method='clpts', prune_result=False)
    N = 24
    vidx = rng.choice(V, size=N, replace=False)
    node_pos = verts[vidx].copy()
    node_dq = np.array([small_dq(rng, rot=0.06, trans=0.3, scale_jitter=0.01) for _ in range(N)])
    node_w = rng.uniform(5.0, 7.0, size=N)
    fu = bare_fusion(1.0, k, node_pos, node_dq, node_w, lw, vert_idx=vidx)
    fu._vertices, fu._normals = verts, norms
    fu._neighbor_look_up = [fu._kdtree.query(v, k=k)[1] for v in verts]
    fu._sess = None
    fu.marching_cubes = lambda tsdf=None, step_size=0: (lverts, None, None, None)
    quiet(fu.setupCorrespondences, np.zeros((2, 2, 2)), method='clpts', prune_result=False)
    out.update(node_pos=node_pos, node_dq=node_dq, node_w=node_w, vidx=vidx, nbr=np.array(fu._neighbor_look_up),
               nr_corr=np.array(fu._correspondences))
    np.savez_compressed(os.path.join(HERE, "g7_correspondences.npz"), **out)
    print("g7 ok: kept", len(out["dm_corridx_1"]), len(out["dm_corridx_0.35"]), "of", V)


# ---------------------------------------------------------------------------- G8
def g8():
    """Graph maintenance (uniform_sample, construct_graph, update_graph with marching cubes patched
    out) and the file readers (load_sdf on a file written by this package's writer, read_proj_matrix)."""
    import tempfile
    from core.sdf import load_sdf
    rng = np.random.default_rng(808)
    V, k = 260, 4
    d = rng.normal(size=(V, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    verts = 16 + 8 * d
    out = dict(verts=verts, knn=k)
    us_v, us_i = util.uniform_sample(verts, 3.1)
    out["us_v"], out["us_i"], out["radius"] = us_v, us_i, 3.1
    fu = Fusion.__new__(Fusion)
    fu._vertices, fu._radius, fu._knn, fu._verbose, fu._nodes = verts, 3.1, k, False, []
    fu.construct_graph()
    out["cg_idx"] = np.array([n[0] for n in fu._nodes])
    out["cg_pos"] = np.array([n[1] for n in fu._nodes])
    out["cg_dq"] = np.array([n[2] for n in fu._nodes])
    out["cg_w"] = np.array([n[3] for n in fu._nodes])
    out["cg_lookup"] = np.array(fu._neighbor_look_up)
    # update_graph: new surface = old vertices + an extra patch no node supports
    d2 = rng.normal(size=(60, 3)); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    verts2 = np.concatenate([verts[:200] + 0.05 * rng.normal(size=(200, 3)), np.array([34.0, 16, 16]) + 5 * d2])
    for i in range(len(fu._nodes)):
        nd = fu._nodes[i]
        fu._nodes[i] = (nd[0], nd[1], small_dq(rng, 0.05, 0.3), nd[3])
    out["ug_dq_in"] = np.array([n[2] for n in fu._nodes])

    def fake_mc(tsdf=None, step_size=0):
        fu._vertices = verts2
    fu.marching_cubes = fake_mc
    fu._write_warpfield = False
    quiet(fu.update_graph)
    out["verts2"] = verts2
    out["ug_idx"] = np.array([n[0] for n in fu._nodes])
    out["ug_pos"] = np.array([n[1] for n in fu._nodes])
    out["ug_dq"] = np.array([np.asarray(n[2], dtype=np.float64) for n in fu._nodes])
    out["ug_w"] = np.array([n[3] for n in fu._nodes])
    out["ug_lookup"] = np.array(fu._neighbor_look_up)
    # file formats
    _spec2 = importlib.util.spec_from_file_location("dfio", os.path.join(ROOT, "dynamicfusion_body_amd", "io.py"))
    dfio = importlib.util.module_from_spec(_spec2); _spec2.loader.exec_module(dfio)
    vol = rng.normal(size=(5, 6, 8)).astype(np.float32)
    cp = rng.normal(size=(5, 6, 8, 3)).astype(np.float32)
    bmin, bmax = np.array([4.9, 0.0, -1.5]), np.array([59.1, 64.0, 64.25])
    with tempfile.TemporaryDirectory() as td:
        fn = os.path.join(td, "t.dist")
        dfio.write_sdf(fn, bmin, bmax, vol, cp)
        a, b, v, c = quiet(load_sdf, fn, read_closest_points=True)
        out["sdf_bytes"] = np.frombuffer(open(fn, "rb").read(), dtype=np.uint8)
        out["sdf_bmin"], out["sdf_bmax"], out["sdf_vol"], out["sdf_cp"] = a, b, np.ascontiguousarray(v), np.ascontiguousarray(c)
        pn = os.path.join(td, "proj0.txt")
        txt = "2000.5 0 800 12.25\n0 -2000.25 600 -3e-1\n0 0 1 5.5\n"
        open(pn, "w").write(txt)
        out["proj_txt"] = np.frombuffer(txt.encode(), dtype=np.uint8)
        out["proj_out"] = util.read_proj_matrix(pn)
    np.savez_compressed(os.path.join(HERE, "g8_graph_io.npz"), **out)
    print("g8 ok: nodes", len(out["cg_idx"]), "->", len(out["ug_idx"]))


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8"]
    for w in which:
        globals()[w]()
