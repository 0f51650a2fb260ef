"""numpy fp64 restatement of the reference's per-frame hot path -- TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only tests/, __graft_entry__.smoke() and
bench.py's `cpu_baseline` leg import it.  Every function cites the reference lines it
restates (paths relative to the reference repo root).  Parity is PINNED: the functions
below are checked in tests/test_oracle_golden.py against vectors produced by importing and
running the reference itself in the build container (tests/golden/make_golden.py, vectors
committed under tests/golden/*.npz) plus the reference's own doctest constants
(core/util.py:258-260, :146-154).

The reference loops over voxels / vertices in the Python interpreter; here the same
arithmetic is vectorised over the leading axes, keeping the reference's operation order
inside each expression so masks and weights are reproduced exactly and values to ~1e-14.
"""
import numpy as np

# --------------------------------------------------------------------------------------
# A7  dual-quaternion algebra                                   core/util.py:68-76,255-304
# --------------------------------------------------------------------------------------

def quaternion_multiply(q1, q0):
    """core/util.py:255-269 -- w-first layout; products are formed in the operands' own
    dtype (f32*f32 stays f32) and only then cast to f64, exactly like the reference's
    scalar unpacking followed by np.array(..., dtype=np.float64)."""
    q1 = np.asarray(q1)
    q0 = np.asarray(q0)
    w0, x0, y0, z0 = q0[..., 0], q0[..., 1], q0[..., 2], q0[..., 3]
    w1, x1, y1, z1 = q1[..., 0], q1[..., 1], q1[..., 2], q1[..., 3]
    return np.stack([
        -x1 * x0 - y1 * y0 - z1 * z0 + w1 * w0,
        x1 * w0 + y1 * z0 - z1 * y0 + w1 * x0,
        -x1 * z0 + y1 * w0 + z1 * x0 + w1 * y0,
        x1 * y0 - y1 * x0 + z1 * w0 + w1 * z0], axis=-1).astype(np.float64)


def dual_quaternion_multiply(q1, q2):
    """core/util.py:275-282"""
    q1 = np.asarray(q1)
    q2 = np.asarray(q2)
    qr1, qd1 = q1[..., :4], q1[..., 4:]
    qr2, qd2 = q2[..., :4], q2[..., 4:]
    qr = quaternion_multiply(qr1, qr2)
    qd = quaternion_multiply(qr1, qd2) + quaternion_multiply(qd1, qr2)
    return np.concatenate([qr, qd], axis=-1)


def dual_quaternion_conjugate(dq):
    """core/util.py:299-304 -- (w,-x,-y,-z,-d0,+d1,+d2,+d3)"""
    dq = np.array(dq, dtype=np.float64, copy=True)
    dq[..., 1:5] = -dq[..., 1:5]
    return dq


def dqb_warp(dq, pos):
    """core/util.py:68-72.  `pos` is rounded to float32 first (vq dtype, :69)."""
    dq = np.asarray(dq)
    pos = np.asarray(pos)
    shp = np.broadcast_shapes(dq.shape[:-1], pos.shape[:-1])
    vq = np.zeros(shp + (8,), dtype=np.float32)
    vq[..., 0] = 1
    vq[..., 5:] = pos.astype(np.float32)
    dqv = dual_quaternion_multiply(np.broadcast_to(dq, shp + (8,)), vq)
    out = dual_quaternion_multiply(dqv, dual_quaternion_conjugate(np.broadcast_to(dq, shp + (8,))))
    return out[..., 5:]


def dqb_warp_normal(dq, n):
    """core/util.py:74-76 -- dual part zeroed, result is NOT renormalised."""
    dq = np.asarray(dq)
    rq = np.concatenate([dq[..., :4].astype(np.float64), np.zeros(dq.shape[:-1] + (4,))], axis=-1)
    return dqb_warp(rq, n)


def quaternion_matrix(q):
    """core/util.py:143-167 (Gohlke)."""
    q = np.array(q, dtype=np.float64, copy=True)
    n = np.dot(q, q)
    if n < np.finfo(float).eps * 4.0:
        return np.identity(4)
    q *= np.sqrt(2.0 / n)
    q = np.outer(q, q)
    return np.array([
        [1.0 - q[2, 2] - q[3, 3], q[1, 2] - q[3, 0], q[1, 3] + q[2, 0], 0.0],
        [q[1, 2] + q[3, 0], 1.0 - q[1, 1] - q[3, 3], q[2, 3] - q[1, 0], 0.0],
        [q[1, 3] - q[2, 0], q[2, 3] + q[1, 0], 1.0 - q[1, 1] - q[2, 2], 0.0],
        [0.0, 0.0, 0.0, 1.0]])


def DQTSE3(dq):
    """core/util.py:86-89 -- dual quaternion -> 4x4 rigid matrix."""
    dq = np.asarray(dq, dtype=np.float64)
    M = np.identity(4)
    M[:3, :3] = quaternion_matrix(dq[:4])[:3, :3]
    conj = dq[:4] * np.array([1.0, -1.0, -1.0, -1.0])
    t = quaternion_multiply(2 * dq[4:], conj)
    M[:3, 3] = t[1:]
    return M


def SE3TDQ_from_Rt(R, t):
    """core/util.py:79-84 restated for a proper rotation R (unit quaternion from R via
    the closed form; the reference goes through an eigen-decomposition, :232-252, which
    agrees up to rounding and sign; sign is fixed to w >= 0 like :250-251)."""
    R = np.asarray(R, dtype=np.float64)
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    q = q / np.linalg.norm(q)
    if q[0] < 0:
        q = -q
    qe = 0.5 * quaternion_multiply(np.array([0.0, t[0], t[1], t[2]]), q)
    return np.concatenate([q, qe])


# --------------------------------------------------------------------------------------
# robust losses                                                        core/util.py:50-60
# --------------------------------------------------------------------------------------

def huber_loss(x, c):
    x = np.asarray(x, dtype=np.float64)
    return np.where(np.abs(x) <= c, 0.5 * (x ** 2), c * (np.abs(x) - 0.5 * c))


def tukey_biweight_loss(x, c):
    x = np.asarray(x, dtype=np.float64)
    return np.where(np.abs(x) > c, 0.0, x * (1 - (x / c) ** 2) ** 2)


# --------------------------------------------------------------------------------------
# A8  pinhole projection                                             core/util.py:312-320
# --------------------------------------------------------------------------------------

def project_to_pixel(K, pos):
    """core/util.py:317-320 (lw=None branch).  Returns (u, v, ok); ok False <=> p2 == 0
    (the reference returns (None, None))."""
    K = np.asarray(K, dtype=np.float64)
    pos = np.asarray(pos, dtype=np.float64)
    p0 = K[0, 0] * pos[..., 0] + K[0, 1] * pos[..., 1] + K[0, 2] * pos[..., 2]
    p1 = K[1, 0] * pos[..., 0] + K[1, 1] * pos[..., 1] + K[1, 2] * pos[..., 2]
    p2 = K[2, 0] * pos[..., 0] + K[2, 1] * pos[..., 1] + K[2, 2] * pos[..., 2]
    ok = p2 != 0
    p2s = np.where(ok, p2, 1.0)
    return p0 / p2s, p1 / p2s, ok


# --------------------------------------------------------------------------------------
# A6  trilinear sampler with the reference's swapped y/z fractions   core/util.py:102-137
# --------------------------------------------------------------------------------------

def interpolate_tsdf(pos, tsdf):
    """Returns (value, valid).  valid False <=> the reference returns None (:107-108).
    x1/y1/z1 are ceil() (:113-115) so integer coordinates sample one voxel twice; the
    y-fraction blends the z1 samples and the z-fraction the y1 samples (:121-137)."""
    pos = np.asarray(pos, dtype=np.float64)
    rx, ry, rz = tsdf.shape
    px, py, pz = pos[..., 0], pos[..., 1], pos[..., 2]
    valid = ~((np.minimum(np.minimum(px, py), pz) < 0) | (px > rx - 1) | (py > ry - 1) | (pz > rz - 1))
    # NaN compares False everywhere in the reference test -> it would index with NaN and
    # raise; treat as invalid here.
    valid &= np.isfinite(px) & np.isfinite(py) & np.isfinite(pz)
    sx = np.where(valid, px, 0.0)
    sy = np.where(valid, py, 0.0)
    sz = np.where(valid, pz, 0.0)
    x0 = np.floor(sx).astype(np.int64); x1 = np.ceil(sx).astype(np.int64)
    y0 = np.floor(sy).astype(np.int64); y1 = np.ceil(sy).astype(np.int64)
    z0 = np.floor(sz).astype(np.int64); z1 = np.ceil(sz).astype(np.int64)
    xd = sx - x0; yd = sy - y0; zd = sz - z0
    c000 = tsdf[x0, y0, z0]; c100 = tsdf[x1, y0, z0]
    c001 = tsdf[x0, y1, z0]; c101 = tsdf[x1, y1, z0]
    c010 = tsdf[x0, y0, z1]; c110 = tsdf[x1, y0, z1]
    c011 = tsdf[x0, y1, z1]; c111 = tsdf[x1, y1, z1]
    c00 = c000 * (1 - xd) + c100 * xd
    c01 = c001 * (1 - xd) + c101 * xd
    c10 = c010 * (1 - xd) + c110 * xd
    c11 = c011 * (1 - xd) + c111 * xd
    c0 = c00 * (1 - yd) + c10 * yd
    c1 = c01 * (1 - yd) + c11 * yd
    return c0 * (1 - zd) + c1 * zd, valid


# --------------------------------------------------------------------------------------
# A1  depth map -> TSDF integration (CPU semantics)             core/fusion_dm.py:180-217
# --------------------------------------------------------------------------------------

def _voxel_index_grid(shape, x0, x1):
    """float32 multi_index of np.nditer in C order (fusion_dm.py:186-188), as f64."""
    X, Y, Z = shape
    ix = np.arange(x0, x1, dtype=np.float32).astype(np.float64)[:, None, None]
    iy = np.arange(Y, dtype=np.float32).astype(np.float64)[None, :, None]
    iz = np.arange(Z, dtype=np.float32).astype(np.float64)[None, None, :]
    return ix, iy, iz


def fuse_depths(dm, lw, K, Kinv, tsdf, tsdf_w, tdist, tsdf_res=None, scale=1.0,
                center=np.zeros(3), wmax=100.0, x_range=None, chunk=8, return_mask=False,
                margin_out=None):
    """FusionDM.fuseDepths, core/fusion_dm.py:180-217.  Mutates tsdf / tsdf_w in place
    (any float dtype; arithmetic is fp64) and returns them.  `tsdf_res` is the ctor's
    `tsdf_res` (sdf_center = tsdf_res/2 on all three axes, :183).  `x_range=(a,b)`
    restricts the sweep to array-axis-0 planes [a,b) (slab partition; indices stay global).

    `margin_out` (a 1-element list) receives the smallest distance of any voxel to one of
    the reference's decision boundaries (.5 pixel tie of round(), frustum edge, z > 0,
    sd > -tdist).  At an exact tie the reference's own answer depends on the summation
    order inside its platform BLAS (np.matmul, fusion_dm.py:193 / util.py:317), so parity
    is only defined -- and only claimed -- for margin > 0; fixtures assert that."""
    dm = np.asarray(dm)
    H, W = dm.shape                                   # (dmx, dmy) = dm.shape, :181
    lw = np.asarray(lw, dtype=np.float64)
    K = np.asarray(K, dtype=np.float64)
    Kinv = np.asarray(Kinv, dtype=np.float64)
    center = np.asarray(center).astype(np.float64)
    X, Y, Z = tsdf.shape
    if tsdf_res is None:
        tsdf_res = X
    c = tsdf_res / 2                                  # :183
    a, b = (0, X) if x_range is None else x_range
    mask_out = np.zeros(tsdf.shape, dtype=bool) if return_mask else None
    for s in range(a, b, chunk):
        e = min(b, s + chunk)
        ix, iy, iz = _voxel_index_grid(tsdf.shape, s, e)
        px = scale * (ix - c) + center[0]             # :191
        py = scale * (iy - c) + center[1]
        pz = scale * (iz - c) + center[2]
        # lpos = lw @ [pos,1]   (:193)
        l0 = lw[0, 0] * px + lw[0, 1] * py + lw[0, 2] * pz + lw[0, 3]
        l1 = lw[1, 0] * px + lw[1, 1] * py + lw[1, 2] * pz + lw[1, 3]
        l2 = lw[2, 0] * px + lw[2, 1] * py + lw[2, 2] * pz + lw[2, 3]
        # project_to_pixel(K, lpos)   (:194 -> util.py:317-320)
        p0 = K[0, 0] * l0 + K[0, 1] * l1 + K[0, 2] * l2
        p1 = K[1, 0] * l0 + K[1, 1] * l1 + K[1, 2] * l2
        p2 = K[2, 0] * l0 + K[2, 1] * l1 + K[2, 2] * l2
        ok = p2 != 0
        p2s = np.where(ok, p2, 1.0)
        u = p0 / p2s
        v = p1 / p2s
        vis = ok & (u >= 0) & (u < W - 1) & (v >= 0) & (v < H - 1)      # :195
        ui = np.where(vis, np.rint(u), 0).astype(np.int64)              # round-half-even, :196
        vi = np.where(vis, np.rint(v), 0).astype(np.int64)
        z = -1 * dm[vi, ui].astype(np.float64)                          # :196
        val = vis & (z > 0)                                             # :197
        # cpos = Kinv @ (z*[u,v,1]); tsdf_l = cpos[2] - lpos[2]   (:198-201)
        cz = Kinv[2, 0] * (z * u) + Kinv[2, 1] * (z * v) + Kinv[2, 2] * (z * 1.0)
        sd = cz - l2
        upd = val & (sd > -1 * tdist)                                   # :203
        if margin_out is not None:
            big = np.inf
            m = np.minimum(np.abs(u - np.floor(u) - 0.5), np.abs(v - np.floor(v) - 0.5))
            m = np.where(vis, m, big)
            edge = np.minimum(np.minimum(np.abs(u), np.abs(u - (W - 1))), np.minimum(np.abs(v), np.abs(v - (H - 1))))
            m = np.minimum(m, np.where(ok, edge, big))
            m = np.minimum(m, np.where(val, np.abs(sd + tdist), big))
            cur = float(np.min(m)) if m.size else big
            margin_out[0] = cur if margin_out[0] is None else min(margin_out[0], cur)
        T = tsdf[s:e].astype(np.float64)
        Wt = tsdf_w[s:e].astype(np.float64)
        newT = (scale * T * Wt + np.minimum(tdist, sd) * 1) / (scale * (1 + Wt))   # :209
        newW = np.minimum(1 + Wt, wmax)                                            # :210
        tsdf[s:e] = np.where(upd, newT, T)
        tsdf_w[s:e] = np.where(upd, newW, Wt)
        if return_mask:
            mask_out[s:e] = upd
    if return_mask:
        return tsdf, tsdf_w, mask_out
    return tsdf, tsdf_w


# --------------------------------------------------------------------------------------
# A3  rigid TSDF -> TSDF fusion                                 core/fusion_dm.py:300-316
# --------------------------------------------------------------------------------------

def _avg_update(T, Wt, s, wi, tdist, wmax):
    newT = (T * Wt + np.minimum(tdist, s) * wi) / (wi + Wt)
    newW = np.minimum(wi + Wt, wmax)
    return newT, newW


def update_tsdf_rigid(tsdf, tsdf_w, curr_tsdf, lw_dq, tdist, wmax=100.0, x_range=None,
                      chunk=8, return_mask=False):
    """FusionDM.updateTSDF, core/fusion_dm.py:300-316.  `lw_dq` is the 8-vector `_lw`
    (voxel-index space, possibly non-unit).  In place on tsdf / tsdf_w."""
    X, Y, Z = tsdf.shape
    a, b = (0, X) if x_range is None else x_range
    mask_out = np.zeros(tsdf.shape, dtype=bool) if return_mask else None
    for s in range(a, b, chunk):
        e = min(b, s + chunk)
        ix, iy, iz = _voxel_index_grid(tsdf.shape, s, e)
        pos = np.stack(np.broadcast_arrays(ix, iy, iz), axis=-1)
        q = dqb_warp(lw_dq, pos)                                        # :306
        sv, valid = interpolate_tsdf(q, curr_tsdf)                      # :307
        upd = valid & (sv > -1 * tdist)                                 # :308
        T = tsdf[s:e].astype(np.float64)
        Wt = tsdf_w[s:e].astype(np.float64)
        newT, newW = _avg_update(T, Wt, sv, 1, tdist, wmax)             # :309-312
        tsdf[s:e] = np.where(upd, newT, T)
        tsdf_w[s:e] = np.where(upd, newW, Wt)
        if return_mask:
            mask_out[s:e] = upd
    return (tsdf, tsdf_w, mask_out) if return_mask else (tsdf, tsdf_w)


# --------------------------------------------------------------------------------------
# A4/A5  non-rigid (DQB) warp and TSDF -> TSDF fusion     core/fusion.py:153-198,502-551
# --------------------------------------------------------------------------------------

def knn_bruteforce(pos, node_pos, k):
    """k nearest nodes by Euclidean distance, ascending -- what
    KDTree.query(pos, k=knn+1)[1][:-1] returns (core/fusion.py:175-176)."""
    pos = np.asarray(pos, dtype=np.float64)
    node_pos = np.asarray(node_pos, dtype=np.float64)
    d = pos[..., None, :] - node_pos
    d2 = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2]
    idx = np.argsort(d2, axis=-1, kind="stable")[..., :k]
    return idx


def _norm3(d):
    return np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2])


def dq_blend(pos, dqs, node_v, node_w, dmax=None):
    """Fusion.dq_blend, core/fusion.py:527-551.  pos (...,3); dqs (...,k,8); node_v
    (...,k,3); node_w (...,k).  Gaussian weights with sigma = 2*dg_w (:537) or dmax
    (:540); normalisation by the full 8-norm (:551); identity DQ if the blend is exactly
    zero (:544-549)."""
    pos = np.asarray(pos)
    dqs = np.asarray(dqs, dtype=np.float64)
    k = dqs.shape[-2]
    dqb = np.zeros(np.broadcast_shapes(pos.shape[:-1], dqs.shape[:-2]) + (8,))
    for j in range(k):
        dist = _norm3(pos.astype(np.float64) - node_v[..., j, :])
        if dmax is None:
            w = np.exp(-1.0 * (dist / (2 * node_w[..., j])) ** 2)
        else:
            w = np.exp(-1.0 * (dist / dmax) ** 2)
        dqb = dqb + w[..., None] * dqs[..., j, :]
    n = np.sqrt(np.sum(dqb * dqb, axis=-1))
    zero = n == 0
    out = dqb / np.where(zero, 1.0, n)[..., None]
    if np.any(zero):
        ident = np.zeros(8); ident[0] = 1
        out = np.where(zero[..., None], ident, out)
    return out


def warp(pos, dqs, node_v, node_w, normal=None, dmax=None, m_lw=None):
    """Fusion.warp, core/fusion.py:502-520 (explicit dqs/locations form)."""
    se3 = dq_blend(pos, dqs, node_v, node_w, dmax)                      # :508
    pw = dqb_warp(se3, pos)                                             # :510
    if m_lw is not None:
        pw = dqb_warp(m_lw, pw)                                         # :512 (re-rounds to f32)
    if normal is None:
        return pw
    nw = dqb_warp_normal(se3, normal)                                   # :515
    if m_lw is not None:
        nw = dqb_warp_normal(m_lw, nw)                                  # :517
    return pw, nw


def update_tsdf_dqb(tsdf, tsdf_w, curr_tsdf, node_pos, node_dq, node_w, knn, lw_dq, tdist,
                    wmax=100.0, x_range=None, chunk=2, return_mask=False):
    """Fusion.updateTSDF, core/fusion.py:153-198.  node_pos (N,3), node_dq (N,8),
    node_w (N,) = nodes' 4th tuple entry (2*radius, :116).  In place."""
    node_pos = np.asarray(node_pos, dtype=np.float64)
    node_dq = np.asarray(node_dq, dtype=np.float64)
    node_w = np.asarray(node_w, dtype=np.float64)
    X, Y, Z = tsdf.shape
    a, b = (0, X) if x_range is None else x_range
    mask_out = np.zeros(tsdf.shape, dtype=bool) if return_mask else None
    for s in range(a, b, chunk):
        e = min(b, s + chunk)
        ix, iy, iz = _voxel_index_grid(tsdf.shape, s, e)
        pos = np.stack(np.broadcast_arrays(ix, iy, iz), axis=-1)        # f32-exact indices
        loc = knn_bruteforce(pos, node_pos, knn)                        # :175-176
        q = warp(pos, node_dq[loc], node_pos[loc], node_w[loc], m_lw=lw_dq)   # :178
        sv, valid = interpolate_tsdf(q, curr_tsdf)
        upd = valid & (sv > -1 * tdist)                                 # :179
        wi = np.zeros(pos.shape[:-1])
        for j in range(knn):                                            # :182-183
            wi = wi + _norm3(node_pos[loc[..., j]] - pos) / knn
        T = tsdf[s:e].astype(np.float64)
        Wt0 = tsdf_w[s:e].astype(np.float64)
        Wt = np.where(Wt0 == 0, wi, Wt0)                                # :186-187
        newT, newW = _avg_update(T, Wt, sv, wi, tdist, wmax)            # :189-190
        tsdf[s:e] = np.where(upd, newT, T)
        tsdf_w[s:e] = np.where(upd, newW, Wt0)
        if return_mask:
            mask_out[s:e] = upd
    return (tsdf, tsdf_w, mask_out) if return_mask else (tsdf, tsdf_w)


# --------------------------------------------------------------------------------------
# A9  rigid point-to-plane residual                             core/fusion_dm.py:285-297
# --------------------------------------------------------------------------------------

def computef_lw_rigid(x, vertices, normals, correspondences):
    """FusionDM.computef_lw.  vertices/normals are already restricted to `_corridx`
    (row i pairs with correspondences[i])."""
    wn = dqb_warp_normal(x, normals)
    vp = dqb_warp(x, vertices)
    d = vp - np.asarray(correspondences, dtype=np.float64)
    return wn[..., 0] * d[..., 0] + wn[..., 1] * d[..., 1] + wn[..., 2] * d[..., 2]


# --------------------------------------------------------------------------------------
# A10  non-rigid residual                                       core/fusion.py:444-491
# --------------------------------------------------------------------------------------

def computef_data(dqs, vertices, normals, correspondences, nbr, node_pos, node_w, lw_dq):
    """Data rows of Fusion.computef (core/fusion.py:466-473) / computef_lw (:448-454).
    dqs (N,8); nbr (V,k) = _neighbor_look_up."""
    dqs = np.asarray(dqs, dtype=np.float64)
    pw, nw = warp(vertices, dqs[nbr], node_pos[nbr], node_w[nbr], normal=normals, m_lw=lw_dq)
    d = pw - np.asarray(correspondences, dtype=np.float64)
    return nw[..., 0] * d[..., 0] + nw[..., 1] * d[..., 1] + nw[..., 2] * d[..., 2]


def computef_reg(dqs, node_vidx, nbr, node_pos, node_w, rw):
    """Regularisation rows of Fusion.computef (core/fusion.py:475-484): node-major,
    neighbour-major, xyz.  node_vidx (N,) = nodes' 1st tuple entry (vertex index)."""
    dqs = np.asarray(dqs, dtype=np.float64)
    nb = nbr[node_vidx]                                   # (N,k) node indices
    vj = node_pos[nb]                                     # dgj_v, :478
    diff = dqb_warp(dqs[:, None, :], vj) - dqb_warp(dqs[nb], vj)        # :480
    wmx = np.maximum(node_w[:, None], node_w[nb])                       # :482
    return ((rw * wmx)[..., None] * diff).reshape(-1)


def computef(x, vertices, normals, correspondences, nbr, node_vidx, node_pos, node_w, lw_dq, rw):
    """Fusion.computef(x, tdw, trw, rw), core/fusion.py:459-491 (tdw/trw are unused
    there)."""
    dqs = np.asarray(x, dtype=np.float64).reshape(-1, 8)
    fd = computef_data(dqs, vertices, normals, correspondences, nbr, node_pos, node_w, lw_dq)
    fr = computef_reg(dqs, node_vidx, nbr, node_pos, node_w, rw)
    return np.concatenate([fd, fr])


# --------------------------------------------------------------------------------------
# correspondence search     core/fusion_dm.py:219-244, core/fusion.py:255-276
# --------------------------------------------------------------------------------------

def closest_correspondences(warped_pos, warped_nrm, live_verts, k, tolerance):
    """The selection loop shared by FusionDM.setupCorrespondences (fusion_dm.py:229-244) and
    Fusion.setupCorrespondences('clpts') (fusion.py:258-276): for every warped canonical vertex
    vp with warped normal wn, the k nearest live vertices (KDTree.query order: nearest first);
    best = the first one whose cost |wn.(vp - p)| is smallest AND < 1 (best_cost starts at 1,
    best_pt at the nearest neighbour, :233-241); kept iff best_cost <= tolerance (:242).
    Returns (best_pt (V,3), best_cost (V,), keep (V,) bool)."""
    vp = np.asarray(warped_pos, dtype=np.float64)
    wn = np.asarray(warped_nrm, dtype=np.float64)
    lv = np.asarray(live_verts, dtype=np.float64)
    nidx = knn_bruteforce(vp, lv, k)                      # (V,k)
    P = lv[nidx]
    d = vp[:, None, :] - P
    cost = np.abs(wn[:, None, 0] * d[..., 0] + wn[:, None, 1] * d[..., 1] + wn[:, None, 2] * d[..., 2])
    best_cost = np.ones(len(vp))
    best = P[:, 0, :].copy()
    for j in range(k):
        better = cost[:, j] < best_cost
        best_cost = np.where(better, cost[:, j], best_cost)
        best = np.where(better[:, None], P[:, j, :], best)
    return best, best_cost, best_cost <= tolerance


# ------------------------------------------------------------------------------------------------ A2 (optional)
def fuse_depths_ocl(dm, lw, K, Kinv, IND, tsdf, tsdf_w, tdist, wmax=100.0):
    """The reference's OpenCL kernel `fuse_depth` and its host code (core/fusion_dm.py:600-703) in numpy float32, operation by
    operation in the kernel text's order.  PARITY UNPINNED: no OpenCL device or pyopencl here to run the reference's kernel, and
    OpenCL C may contract a*b+c; this is the uncontracted IEEE-float32 reading of the text.  Returns new float32 arrays."""
    f32 = np.float32
    H, W = dm.shape
    T = np.asarray(tsdf).astype(f32).copy(); Wt = np.asarray(tsdf_w).astype(f32).copy()              # :690-691
    proj = np.matmul(K, np.matmul(lw, IND)).astype(f32)                                               # :695
    k2 = np.asarray(Kinv).astype(f32)[2]
    depth = np.asarray(dm).astype(f32).reshape(-1)
    TD, WM = f32(float("%f" % tdist)), f32(float("%f" % wmax))                                        # :682-687
    X, Y, Z = T.shape
    x, y, z = [a.astype(f32) for a in np.meshgrid(np.arange(X), np.arange(Y), np.arange(Z), indexing="ij")]
    with np.errstate(all="ignore"):
        u = ((proj[0, 0] * x + proj[0, 1] * y) + proj[0, 2] * z) + proj[0, 3]                         # :640-642
        v = ((proj[1, 0] * x + proj[1, 1] * y) + proj[1, 2] * z) + proj[1, 3]
        w = ((proj[2, 0] * x + proj[2, 1] * y) + proj[2, 2] * z) + proj[2, 3]
        px = u / w; py = v / w                                                                        # :645-646
        vis = (px >= 0) & (py >= 0) & (px < f32(W - 1)) & (py < f32(H - 1))                           # :647 (NaN: skipped, see the kernel)
        pxs, pys = np.where(vis, px, f32(0)), np.where(vis, py, f32(0))
        ix = np.floor(pxs).astype(np.int64); iy = np.floor(pys).astype(np.int64)                      # :607-608
        wx = pxs - ix.astype(f32); wy = pys - iy.astype(f32)
        lu = iy * W + ix; lb = (iy + 1) * W + ix
        one = f32(1)
        up = depth[lu] * (one - wx) + depth[lu + 1] * wx                                              # :617-619
        bot = depth[lb] * (one - wx) + depth[lb + 1] * wx
        pz = -(up * (one - wy) + bot * wy)                                                            # :649
        near = pz <= TD
        pxz, pyz = pxs * pz, pys * pz                                                                 # :655-656
        dz = -((k2[0] * (pxz - u) + k2[1] * (pyz - v)) + k2[2] * (pz - w))                            # :657-658
        dz = np.where(near, -TD, dz).astype(f32)                                                      # :652-653
        upd = vis & (dz < TD)                                                                         # :667
        nw = np.minimum(one + Wt, WM)                                                                 # :670
        newT = ((nw - one) * T + one * np.maximum(-TD, dz)) / nw                                      # :671
    T = np.where(upd, newT, T).astype(f32)
    Wt = np.where(upd, nw, Wt).astype(f32)
    return T, Wt
