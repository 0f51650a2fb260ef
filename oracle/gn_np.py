"""fp64 numpy Gauss-Newton for the warp-field solve -- TEST INFRASTRUCTURE ONLY.

The reference has no Gauss-Newton: it hands its residual functions (FusionDM.computef_lw,
core/fusion_dm.py:285-297; Fusion.computef, core/fusion.py:459-491) to
scipy.optimize.least_squares with finite-difference Jacobians.  What is pinned against the
reference is therefore (1) the residual vectors (oracle_np.computef*, golden g5) and (2) the
analytic Jacobians below, which must match central finite differences of the REFERENCE's
own computef / computef_lw with respect to left twists  dq <- exp(xi) (x) dq  (golden g5:
fd_cols, rigid_fd_cols).  The normal-equation assembly, the linear solve and the update are
this build's own algorithm; this file is its CPU statement, used to check the HIP kernels
(tests/test_gn_oracle.py, tests/test_gpu_solve.py).

Parametrisation: xi = (omega, v) in R^6 per node, exp(xi) = unit dual quaternion with
rotation exp(omega) and translation v (to first order [1, omega/2 | 0, v/2]).
"""
import numpy as np

from . import oracle_np as O


# ---------------------------------------------------------------- quaternion helpers (w first)
def qmul(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw], axis=-1)


def qconj(a):
    return np.asarray(a, dtype=np.float64) * np.array([1.0, -1.0, -1.0, -1.0])


def pure(v):
    v = np.asarray(v, dtype=np.float64)
    return np.concatenate([np.zeros(v.shape[:-1] + (1,)), v], axis=-1)


def dq_mul(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.concatenate([qmul(a[..., :4], b[..., :4]), qmul(a[..., :4], b[..., 4:]) + qmul(a[..., 4:], b[..., :4])], axis=-1)


def twist_exp_dq(xi):
    """exp of a twist: rotation exp(omega) (exact), translation v."""
    xi = np.asarray(xi, dtype=np.float64)
    om, v = xi[..., :3], xi[..., 3:]
    th = np.sqrt(np.sum(om * om, axis=-1))
    half = 0.5 * th
    # sin(th/2)/th with the series near 0
    small = th < 1e-8
    s = np.where(small, 0.5 - th * th / 48.0, np.sin(half) / np.where(small, 1.0, th))
    q = np.concatenate([np.cos(half)[..., None], s[..., None] * om], axis=-1)
    qe = 0.5 * qmul(pure(v), q)
    return np.concatenate([q, qe], axis=-1)


def apply_twists(dqs, xi):
    """dq_a <- exp(xi_a) (x) dq_a for every node (keeps the 8-float layout)."""
    return dq_mul(twist_exp_dq(xi), dqs)


def warp_closed(q, p):
    """dqb_warp without the float32 rounding: vec(r P r*) + 2 vec(d r*)."""
    r, d = q[..., :4], q[..., 4:]
    return (qmul(qmul(r, pure(p)), qconj(r)) + 2.0 * qmul(d, qconj(r)))[..., 1:]


def f32(x):
    return np.asarray(x).astype(np.float32).astype(np.float64)


# ---------------------------------------------------------------- rigid (global lw) term, A9
def rigid_residual_jacobian(x, vertices, normals, corr):
    """r_i = dqb_warp_normal(x, n_i) . (dqb_warp(x, v_i) - c_i) and d r_i / d xi for the left
    twist on x:  J = [ c x m | s*m ],  m = warped normal, s = |r_x|^2."""
    x = np.asarray(x, dtype=np.float64)
    m = O.dqb_warp_normal(x, normals)
    y = O.dqb_warp(x, vertices)
    c = np.asarray(corr, dtype=np.float64)
    r = np.sum(m * (y - c), axis=-1)
    s = np.sum(x[:4] * x[:4])
    J = np.concatenate([np.cross(c, m), s * m], axis=-1)
    return r, J


# ---------------------------------------------------------------- non-rigid data term, A10
def blend_weights(vertices, node_pos_k, node_w_k):
    """Gaussian DQB weights of core/fusion.py:537 (pos rounded to float32 by the caller's
    convention is NOT applied here: dq_blend uses the raw position)."""
    d = np.asarray(vertices, dtype=np.float64)[..., None, :] - node_pos_k
    dist = np.sqrt(d[..., 0] ** 2 + d[..., 1] ** 2 + d[..., 2] ** 2)
    return np.exp(-1.0 * (dist / (2 * node_w_k)) ** 2)


def data_residual_jacobian(dqs, vertices, normals, corr, nbr, node_pos, node_w, lw, round_f32=True):
    """Data rows of Fusion.computef and their Jacobian w.r.t. the left twists of the k nodes of
    each vertex.  Returns r (V,), J (V,k,6).  Derivation: r = n'.(x'-c), x' = A x1 + t,
    n' = A n1 (A = |r_lw|^2 R(lw)), x1 = W(b^,p), n1 = Wn(b^,n0), b^ = b/|b|_8, b = sum w_a q_a.
    With u = A^T n', h = A^T (x'-c):  grad_{b^} = [ -2 U r P - 2 U d - 2 H r N | 2 U r ],
    grad_b = (grad_{b^} - (grad_{b^}.b^) b^)/|b|,  J_a[omega] = w_a/2 (vec(g_r r_a*) + vec(g_d d_a*)),
    J_a[v] = w_a/2 vec(g_d r_a*).  round_f32=False drops the reference's float32 roundings
    (util.py:69), giving the smooth function whose derivative the Jacobian is."""
    dqs = np.asarray(dqs, dtype=np.float64)
    lw = np.asarray(lw, dtype=np.float64)
    V = len(vertices)
    qk = dqs[nbr]                                          # (V,k,8)
    w = blend_weights(vertices, node_pos[nbr], node_w[nbr])            # (V,k)
    b = np.sum(w[..., None] * qk, axis=1)                  # (V,8)
    nb = np.sqrt(np.sum(b * b, axis=-1))
    bh = b / nb[:, None]
    rnd = f32 if round_f32 else (lambda a: np.asarray(a, dtype=np.float64))
    p = rnd(vertices); n0 = rnd(normals)
    x1 = warp_closed(bh, p)
    n1 = qmul(qmul(bh[:, :4], pure(n0)), qconj(bh[:, :4]))[:, 1:]
    x1r, n1r = rnd(x1), rnd(n1)
    xp = warp_closed(lw, x1r)
    rl = lw[:4]
    npr = qmul(qmul(rl, pure(n1r)), qconj(rl))[..., 1:]
    c = np.asarray(corr, dtype=np.float64)
    r = np.sum(npr * (xp - c), axis=-1)
    # A^T y = vec(rl* Y rl)
    def AT(yv):
        return qmul(qmul(qconj(rl), pure(yv)), rl)[..., 1:]
    U = pure(AT(npr)); H = pure(AT(xp - c))
    rr, dd = bh[:, :4], bh[:, 4:]
    g_r = -2.0 * qmul(qmul(U, rr), pure(p)) - 2.0 * qmul(U, dd) - 2.0 * qmul(qmul(H, rr), pure(n0))
    g_d = 2.0 * qmul(U, rr)
    g = np.concatenate([g_r, g_d], axis=-1)
    g = (g - np.sum(g * bh, axis=-1, keepdims=True) * bh) / nb[:, None]
    gr, gd = g[:, None, :4], g[:, None, 4:]
    ra, da = qk[..., :4], qk[..., 4:]
    Jw = 0.5 * (qmul(gr, qconj(ra)) + qmul(gd, qconj(da)))[..., 1:]
    Jv = 0.5 * qmul(gd, qconj(ra))[..., 1:]
    J = w[..., None] * np.concatenate([Jw, Jv], axis=-1)   # (V,k,6)
    return r, J


def reg_residual_jacobian(dqs, node_vidx, nbr, node_pos, node_w, rw):
    """Regularisation rows (core/fusion.py:475-484) rho_ij = c_ij (W(q_i,v_j) - W(q_j,v_j)),
    c_ij = rw*max(w_i,w_j), and their Jacobians w.r.t. xi_i and xi_j:
      d W(exp(xi) q, p)/d xi = [ -[y]x | s I ],  y = W(q,p), s = |r_q|^2.
    Returns rho (N,k,3), nb (N,k) neighbour node index, Ji (N,k,3,6), Jj (N,k,3,6)."""
    dqs = np.asarray(dqs, dtype=np.float64)
    nb = nbr[node_vidx]
    vj = node_pos[nb]
    yi = O.dqb_warp(dqs[:, None, :], vj)
    yj = O.dqb_warp(dqs[nb], vj)
    cij = rw * np.maximum(node_w[:, None], node_w[nb])
    rho = cij[..., None] * (yi - yj)

    def skew(y):
        z = np.zeros(y.shape[:-1])
        return np.stack([np.stack([z, -y[..., 2], y[..., 1]], -1),
                         np.stack([y[..., 2], z, -y[..., 0]], -1),
                         np.stack([-y[..., 1], y[..., 0], z], -1)], -2)
    s_all = np.sum(dqs[:, :4] ** 2, axis=-1)
    si = np.broadcast_to(s_all[:, None], nb.shape)
    sj = s_all[nb]
    I3 = np.eye(3)
    Ji = cij[..., None, None] * np.concatenate([-skew(yi), si[..., None, None] * I3], axis=-1)
    Jj = -cij[..., None, None] * np.concatenate([-skew(yj), sj[..., None, None] * I3], axis=-1)
    same = (nb == np.arange(len(dqs))[:, None])
    Ji = np.where(same[..., None, None], 0.0, Ji)
    Jj = np.where(same[..., None, None], 0.0, Jj)
    return rho, nb, Ji, Jj


def assemble_dense(N, r_data, J_data, nbr, rho, nb, Ji, Jj, valid=None, lm=0.0):
    """Dense normal equations  (J^T J + lm I) dx = -J^T r  over all rows; returns JtJ (6N,6N),
    Jtr (6N,), cost = 0.5*|r|^2."""
    k = nbr.shape[1]
    JtJ = np.zeros((6 * N, 6 * N)); Jtr = np.zeros(6 * N)
    V = len(r_data)
    if valid is None:
        valid = np.ones(V, dtype=bool)
    rd = np.where(valid, r_data, 0.0)
    Jd = np.where(valid[:, None, None], J_data, 0.0)
    for a in range(k):
        ia = nbr[:, a]
        np.add.at(Jtr.reshape(N, 6), ia, Jd[:, a, :] * rd[:, None])
        for b_ in range(k):
            ib = nbr[:, b_]
            blk = Jd[:, a, :, None] * Jd[:, b_, None, :]
            np.add.at(JtJ.reshape(N, 6, N, 6).transpose(0, 2, 1, 3), (ia, ib), blk)
    cost = 0.5 * float(np.sum(rd * rd))
    Nn, kk = nb.shape
    ii = np.repeat(np.arange(Nn)[:, None], kk, axis=1)
    JtJ4 = JtJ.reshape(N, 6, N, 6).transpose(0, 2, 1, 3)
    np.add.at(JtJ4, (ii, ii), np.einsum('nkci,nkcj->nkij', Ji, Ji))
    np.add.at(JtJ4, (nb, nb), np.einsum('nkci,nkcj->nkij', Jj, Jj))
    np.add.at(JtJ4, (ii, nb), np.einsum('nkci,nkcj->nkij', Ji, Jj))
    np.add.at(JtJ4, (nb, ii), np.einsum('nkci,nkcj->nkij', Jj, Ji))
    np.add.at(Jtr.reshape(N, 6), ii, np.einsum('nkci,nkc->nki', Ji, rho))
    np.add.at(Jtr.reshape(N, 6), nb, np.einsum('nkci,nkc->nki', Jj, rho))
    cost += 0.5 * float(np.sum(rho * rho))
    JtJ += lm * np.eye(6 * N)
    return JtJ, Jtr, cost


def gn_step(dqs, vertices, normals, corr, nbr, node_vidx, node_pos, node_w, lw, rw, valid=None, lm=1e-6):
    """One Gauss-Newton step on the reference's residual (computef) -> (new dqs, cost before, dx)."""
    N = len(dqs)
    r, J = data_residual_jacobian(dqs, vertices, normals, corr, nbr, node_pos, node_w, lw)
    rho, nb, Ji, Jj = reg_residual_jacobian(dqs, node_vidx, nbr, node_pos, node_w, rw)
    JtJ, Jtr, cost = assemble_dense(N, r, J, nbr, rho, nb, Ji, Jj, valid=valid, lm=lm)
    dx = np.linalg.solve(JtJ, -Jtr)
    return apply_twists(dqs, dx.reshape(N, 6)), cost, dx


def gn_step_rigid(x, vertices, normals, corr, lm=0.0):
    r, J = rigid_residual_jacobian(x, vertices, normals, corr)
    dx = np.linalg.solve(J.T @ J + lm * np.eye(6), -(J.T @ r))
    return dq_mul(twist_exp_dq(dx), x), 0.5 * float(r @ r), dx


# ---------------------------------------------------------------- projective data association
def associate_depth(points_idx, K, Kinv, lw_cam, dm, scale, center, half):
    """Projective association of warped points (voxel-index space) against a depth map, with the
    reference's primitives: index -> world (fusion_dm.py:191), world -> camera (:193), pixel
    (util.py:317-320), nearest pixel round-half-even and z = -dm (:196), back-projection
    K^-1 (z [u,v,1]) (:198-200), then back to index space through the inverse extrinsic.
    Returns (corr_idx (S,3), valid (S,))."""
    P = np.asarray(points_idx, dtype=np.float64)
    H, W = dm.shape
    world = scale * (P - half) + center
    R, t = lw_cam[:, :3], lw_cam[:, 3]
    cam = world @ R.T + t
    u, v, ok = O.project_to_pixel(K, cam)
    vis = ok & (u >= 0) & (u < W - 1) & (v >= 0) & (v < H - 1)
    ui = np.where(vis, np.rint(u), 0).astype(np.int64)
    vi = np.where(vis, np.rint(v), 0).astype(np.int64)
    z = -1.0 * dm[vi, ui].astype(np.float64)
    valid = vis & (z > 0)
    uc = z[:, None] * np.stack([u, v, np.ones_like(u)], axis=-1)
    ccam = uc @ np.asarray(Kinv, dtype=np.float64).T
    cworld = (ccam - t) @ np.linalg.inv(R).T
    cidx = (cworld - center) / scale + half
    return np.where(valid[:, None], cidx, 0.0), valid


def associate_depth_views(points_idx, K, Kinv, lw_cams, dms, scale, center, half, max_dist=0.0):
    """Association against several live views (the device's dfh_gn_associate_views; no reference counterpart): every view is
    tried with associate_depth, a point keeps the correspondence of the view in which it lies closest to the observed surface
    -- smallest |c - x'| among the views where it is valid and (max_dist > 0) within the gate; ties go to the lower view
    index.  Returns (corr_idx (S,3), valid (S,), chosen view (S,), -1 where invalid)."""
    P = np.asarray(points_idx, dtype=np.float64)
    best = np.full(len(P), np.inf)
    corr = np.zeros_like(P)
    view = np.full(len(P), -1, dtype=np.int64)
    for v, (lw_cam, dm) in enumerate(zip(lw_cams, dms)):
        c, ok = associate_depth(P, K, Kinv, np.asarray(lw_cam, dtype=np.float64), dm, scale, center, half)
        d = c - P
        d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
        if max_dist > 0:
            ok = ok & (d2 <= max_dist * max_dist)
        take = ok & (d2 < best)
        best = np.where(take, d2, best)
        corr = np.where(take[:, None], c, corr)
        view = np.where(take, v, view)
    return corr, view >= 0, view


# ---------------------------------------------------------------- block-sparse assembly + truncated PCG
# (what the HIP build ships: csrc/dfh_solve.hip gn_build_* + pcg_cg1_kernel).  Same algorithm as assemble_dense
# + a dense solve, but sized for BASELINE configs 3/4 (512 / 2 048 nodes, 1e5..1e6 samples) and with the SAME
# truncated linear solve the device runs, so that the benched settings (10 PCG iterations) have a CPU value beside them.
def _reduce_by_key(keys, vals):
    """Sum vals (n, ...) over equal keys -> (sorted unique keys, sums)."""
    order = np.argsort(keys, kind="stable")
    ks = keys[order]
    start = np.flatnonzero(np.concatenate([[True], ks[1:] != ks[:-1]]))
    return ks[start], np.add.reduceat(vals[order], start, axis=0)


def assemble_blocks(N, r_data, J_data, nbr, rho, nb, Ji, Jj, valid=None):
    """Block-sparse normal equations of the same rows as assemble_dense: returns
    (keys (B,) sorted unique i*N+j, blocks (B,6,6), Jtr (N,6), cost).  Only valid data rows are touched."""
    k = nbr.shape[1]
    if valid is not None:
        sel = np.flatnonzero(valid)
        r_data, J_data, nbr = r_data[sel], J_data[sel], nbr[sel]
    key_parts, blk_parts = [], []
    Jtr = np.zeros((N, 6))
    for a in range(k):
        np.add.at(Jtr, nbr[:, a], J_data[:, a, :] * r_data[:, None])
        for b_ in range(k):
            kk, bb = _reduce_by_key(nbr[:, a].astype(np.int64) * N + nbr[:, b_], J_data[:, a, :, None] * J_data[:, b_, None, :])
            key_parts.append(kk); blk_parts.append(bb)
    cost = 0.5 * float(np.sum(r_data * r_data))
    Nn, kn = nb.shape
    ii = np.repeat(np.arange(Nn)[:, None], kn, axis=1).reshape(-1)
    jj = nb.reshape(-1)
    Ji_, Jj_ = Ji.reshape(-1, 3, 6), Jj.reshape(-1, 3, 6)
    for (ra, rb, A_, B_) in ((ii, ii, Ji_, Ji_), (jj, jj, Jj_, Jj_), (ii, jj, Ji_, Jj_), (jj, ii, Jj_, Ji_)):
        kk, bb = _reduce_by_key(ra.astype(np.int64) * N + rb, np.einsum('nci,ncj->nij', A_, B_))
        key_parts.append(kk); blk_parts.append(bb)
    np.add.at(Jtr, ii, np.einsum('nci,nc->ni', Ji_, rho.reshape(-1, 3)))
    np.add.at(Jtr, jj, np.einsum('nci,nc->ni', Jj_, rho.reshape(-1, 3)))
    cost += 0.5 * float(np.sum(rho * rho))
    keys, blocks = _reduce_by_key(np.concatenate(key_parts), np.concatenate(blk_parts))
    return keys, blocks, Jtr, cost


def blocks_to_bsr(N, keys, blocks):
    import scipy.sparse as sp
    rows = keys // N
    indptr = np.searchsorted(rows, np.arange(N + 1))
    return sp.bsr_matrix((blocks, (keys % N).astype(np.int64), indptr), shape=(6 * N, 6 * N))


def damp_blocks(N, keys, blocks, lm_abs, lm_rel):
    """The device's damping: d <- d + lm_abs + lm_rel * d on the scalar diagonal (written into the matrix)."""
    out = blocks.copy()
    diag = np.flatnonzero(keys // N == keys % N)
    idx = np.arange(6)
    out[diag[:, None], idx, idx] = out[diag[:, None], idx, idx] + lm_abs + lm_rel * out[diag[:, None], idx, idx]
    return out


def pcg_cg1(N, keys, blocks, Jtr, iters, lm_abs=0.0, lm_rel=0.0):
    """x after `iters` iterations of the single-reduction preconditioned CG of Chronopoulos & Gear on
    (A + lm_abs I + lm_rel diag A) x = -J^T r with the block-Jacobi preconditioner -- the recurrence of
    pcg_cg1_kernel (csrc/dfh_solve.hip): u = M^-1 r, w = A u, gamma = r.u, delta = w.u, beta = gamma/gamma_old,
    alpha = gamma / (delta - beta gamma / alpha_old), p = u + beta p, s = w + beta s, t = v + beta t (v = M^-1 w),
    x += alpha p, r -= alpha s, u -= alpha t."""
    Bd = damp_blocks(N, keys, blocks, lm_abs, lm_rel)
    A = blocks_to_bsr(N, keys, Bd)
    diag = np.full(N, -1, dtype=np.int64)
    dsel = np.flatnonzero(keys // N == keys % N)
    diag[keys[dsel] // N] = dsel
    D = np.where((diag >= 0)[:, None, None], Bd[np.maximum(diag, 0)], np.eye(6)[None])
    Minv = np.linalg.inv(D)
    M = lambda v: np.einsum('nij,nj->ni', Minv, v.reshape(N, 6)).reshape(-1)
    x = np.zeros(6 * N)
    r = -Jtr.reshape(-1).copy()
    u = M(r)
    w = A @ u
    v = M(w)
    gamma, delta = float(r @ u), float(w @ u)
    p = np.zeros_like(x); s = np.zeros_like(x); t = np.zeros_like(x)
    gamma_prev = alpha_prev = 0.0
    for it in range(iters):
        beta = gamma / gamma_prev if gamma_prev != 0.0 else 0.0
        denom = delta - (beta * gamma) / alpha_prev if alpha_prev != 0.0 else delta
        alpha = gamma / denom if denom != 0.0 else 0.0
        p = u + beta * p; s = w + beta * s; t = v + beta * t
        x = x + alpha * p; r = r - alpha * s; u = u - alpha * t
        if it == iters - 1:
            break
        w = A @ u
        v = M(w)
        gamma_prev, alpha_prev = gamma, alpha
        gamma, delta = float(r @ u), float(w @ u)
    return x


def global_step(dqs, blocks, Jtr, lm_rel):
    """The rigid mode of the normal equations solved on its own (dfh_gn_global_step, csrc/dfh_solve.hip): all nodes share ONE
    twist xi -- (sum of all 6x6 blocks, symmetrised, + lm_rel diag) xi = -(sum of all J^T r) -- applied to every node.
    Returns (new dqs, xi)."""
    A = blocks.sum(axis=0)
    A = 0.5 * (A + A.T)
    A = A + lm_rel * np.diag(np.diag(A))
    xi = -np.linalg.solve(A, Jtr.sum(axis=0))
    return apply_twists(dqs, np.tile(xi, (len(dqs), 1))), xi


def global_step_sampled(dqs, pos, nrm, nbr, node_pos, node_w, lw, associate, huber, lm_rel, stride=1, tile=128):
    """The rigid-mode step straight from the samples (dfh_gn_global_sampled_views): the data rows of every `stride`-th `tile`-sample
    tile, a sample's Jacobian for the shared twist = the sum of its k node blocks, Huber weights as in the builds, no regulariser;
    (A_g + lm_rel diag A_g) xi = -g_g, xi applied to every node.  Returns (new dqs, xi, valid count)."""
    warped = O.warp(pos, dqs[nbr], node_pos[nbr], node_w[nbr], m_lw=lw)
    corr, valid = associate(warped)
    sel = np.flatnonzero(valid & ((np.arange(len(pos)) // tile) % stride == 0))
    r, J = data_residual_jacobian(dqs, pos[sel], nrm[sel], corr[sel], nbr[sel], node_pos, node_w, lw)
    if huber > 0.0:
        sc, _ = huber_scale(r, huber)
        r, J = r * sc, J * sc[:, None, None]
    Jg = J.sum(axis=1)
    A = Jg.T @ Jg
    A = A + lm_rel * np.diag(np.diag(A))
    xi = -np.linalg.solve(A, Jg.T @ r) if len(sel) >= 6 else np.zeros(6)
    return apply_twists(dqs, np.tile(xi, (len(dqs), 1))), xi, len(sel)


def huber_scale(r, delta):
    """(sqrt of the IRLS weight per row, Huber objective sum rho(r)) -- dfh_gn_build_planned's huber_delta."""
    a = np.abs(r)
    sc = np.sqrt(np.minimum(1.0, delta / np.maximum(a, 1e-300)))
    obj = float(np.where(a <= delta, 0.5 * r * r, delta * (a - 0.5 * delta)).sum())
    return sc, obj


def gn_loop_truncated(dqs, pos, nrm, nbr, node_nbr, node_pos, node_w, lw, associate, iters, rw, lm_abs, lm_rel, huber, pcg_iters,
                      exact=False, global_iters=0, global_lm=0.1, global_sampled=True):
    """The shipped GN loop (pipeline.FrameSolver.gn_iteration x iters) on the CPU: per iteration associate ->
    Huber-weighted normal equations -> `pcg_iters` iterations of pcg_cg1 (exact=True: sparse direct solve) -> twist
    update.  associate(warped_points) -> (corr, valid).  global_iters: that many rigid-mode steps first, as
    pipeline.SlabFrame.step takes them (global_sampled: from the data rows, global_step_sampled; else from the built normal
    equations, global_step).  Returns (costs at every build of the node
    iterations, valid counts, final dqs)."""
    N = len(dqs)
    dqs = np.asarray(dqs, dtype=np.float64).copy()
    costs, counts = [], []
    if global_iters and global_sampled:
        for _ in range(global_iters):
            dqs, _, _ = global_step_sampled(dqs, pos, nrm, nbr, node_pos, node_w, lw, associate, huber, global_lm)
        global_iters = 0
    for it_ in range(global_iters + iters):
        warped = O.warp(pos, dqs[nbr], node_pos[nbr], node_w[nbr], m_lw=lw)
        corr, valid = associate(warped)
        sel = np.flatnonzero(valid)
        r, J = data_residual_jacobian(dqs, pos[sel], nrm[sel], corr[sel], nbr[sel], node_pos, node_w, lw)
        obj = 0.5 * float(r @ r)
        if huber > 0.0:
            sc, obj = huber_scale(r, huber)
            r, J = r * sc, J * sc[:, None, None]
        rho, nb, Ji, Jj = reg_residual_jacobian(dqs, np.arange(N), node_nbr, node_pos, node_w, rw)
        keys, blocks, Jtr, _ = assemble_blocks(N, r, J, nbr[sel], rho, nb, Ji, Jj)
        if it_ < global_iters:
            dqs, _ = global_step(dqs, blocks, Jtr, global_lm)
            continue
        costs.append(obj + 0.5 * float(np.sum(rho * rho)))
        counts.append(int(len(sel)))
        if exact:
            import scipy.sparse.linalg as spla
            A = blocks_to_bsr(N, keys, damp_blocks(N, keys, blocks, lm_abs, lm_rel)).tocsc()
            dx = spla.spsolve(A, -Jtr.reshape(-1))
        else:
            dx = pcg_cg1(N, keys, blocks, Jtr, pcg_iters, lm_abs, lm_rel)
        dqs = apply_twists(dqs, dx.reshape(N, 6))
    return costs, counts, dqs
