#!/usr/bin/env python3
"""Builds tests/golden/g9_mesh.npz from the reference's own marching-cubes output
(/root/reference/meshes/original.obj, written by FusionDM.write_canonical_mesh -> skimage
marching_cubes_lewiner; test.py:112).  Runs in the build container only; the .npz travels.

Content:
  verts, normals (fp32), faces (int32, as stored: 0-based)   -- the reference's data file
  sdf  (65^3 fp32)  -- a signed distance field of THAT mesh (positive outside), truncated at +-3
         voxels, computed here (exact point-triangle distances near the surface, sign from the
         mesh's vertex normals, far voxels signed by flood fill).  Marching cubes of `sdf` at level 0
         must give the reference's mesh back (same crossed lattice edges, nearby positions, same
         normal / winding conventions): the round trip the mesh tests check.
"""
import os
import numpy as np
from scipy.spatial import cKDTree
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/meshes/original.obj"
TRUNC = 3.0


def read_obj(path):
    V, N, F = [], [], []
    for line in open(path):
        t = line.split()
        if not t:
            continue
        if t[0] == "v":
            V.append([float(x) for x in t[1:4]])
        elif t[0] == "vn":
            N.append([float(x) for x in t[1:4]])
        elif t[0] == "f":
            F.append([int(x.split("/")[0]) for x in t[1:4]])
    return np.array(V), np.array(N), np.array(F)


def closest_on_triangles(P, A, B, C):
    """Closest points of P (n,3) on triangles (A,B,C) (n,3 each); returns points and barycentric weights.
    Ericson, Real-Time Collision Detection 5.1.5, vectorised."""
    ab, ac, ap = B - A, C - A, P - A
    d1, d2 = (ab * ap).sum(1), (ac * ap).sum(1)
    bp = P - B
    d3, d4 = (ab * bp).sum(1), (ac * bp).sum(1)
    cp = P - C
    d5, d6 = (ab * cp).sum(1), (ac * cp).sum(1)
    vc = d1 * d4 - d3 * d2
    vb = d5 * d2 - d1 * d6
    va = d3 * d6 - d5 * d4
    n = len(P)
    w = np.zeros((n, 3))
    done = np.zeros(n, dtype=bool)

    def put(mask, wa, wb, wc):
        m = mask & ~done
        w[m, 0], w[m, 1], w[m, 2] = wa[m], wb[m], wc[m]
        done[m] = True

    one, zero = np.ones(n), np.zeros(n)
    put((d1 <= 0) & (d2 <= 0), one, zero, zero)
    put((d3 >= 0) & (d4 <= d3), zero, one, zero)
    with np.errstate(divide="ignore", invalid="ignore"):
        v = d1 / (d1 - d3)
        put((vc <= 0) & (d1 >= 0) & (d3 <= 0), 1 - v, v, zero)
        put((d6 >= 0) & (d5 <= d6), zero, zero, one)
        wv = d2 / (d2 - d6)
        put((vb <= 0) & (d2 >= 0) & (d6 <= 0), 1 - wv, zero, wv)
        u = (d4 - d3) / ((d4 - d3) + (d5 - d6))
        put((va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0), zero, 1 - u, u)
        den = 1.0 / (va + vb + vc)
        put(np.ones(n, dtype=bool), 1 - vb * den - vc * den, vb * den, vc * den)
    Q = w[:, :1] * A + w[:, 1:2] * B + w[:, 2:3] * C
    return Q, w


def main():
    V, N, F = read_obj(SRC)
    assert F.min() == 0                                   # stored 0-based
    R = 65
    grid = np.stack(np.meshgrid(np.arange(R), np.arange(R), np.arange(R), indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
    tree_v = cKDTree(V)
    dv, _ = tree_v.query(grid)
    near = np.nonzero(dv < TRUNC + 1.5)[0]
    P = grid[near]
    cen = V[F].mean(1)
    tree_f = cKDTree(cen)
    K = 24
    _, cand = tree_f.query(P, k=K)
    best = np.full(len(P), np.inf)
    sgn = np.zeros(len(P))
    for j in range(K):
        f = F[cand[:, j]]
        Q, w = closest_on_triangles(P, V[f[:, 0]], V[f[:, 1]], V[f[:, 2]])
        d = np.linalg.norm(P - Q, axis=1)
        nout = -(w[:, :1] * N[f[:, 0]] + w[:, 1:2] * N[f[:, 1]] + w[:, 2:3] * N[f[:, 2]])     # stored normals point inward
        s = np.sign(((P - Q) * nout).sum(1))
        better = d < best
        best[better] = d[better]
        sgn[better] = s[better]
    sgn[sgn == 0] = 1.0
    sdf = np.zeros(R ** 3)
    known = np.zeros(R ** 3, dtype=bool)
    ok = best <= TRUNC
    sdf[near[ok]] = (best * sgn)[ok]
    known[near[ok]] = True
    sdf = sdf.reshape(R, R, R)
    known = known.reshape(R, R, R)
    # far voxels: sign of the band voxels their connected component touches
    lab, nlab = ndimage.label(~known)
    band_sign = np.where(known, np.sign(sdf), 0.0)
    for l in range(1, nlab + 1):
        comp = lab == l
        ring = ndimage.binary_dilation(comp) & known
        s = band_sign[ring]
        assert len(s) > 0 and (np.all(s > 0) or np.all(s < 0)), (l, len(s), (s > 0).sum(), (s < 0).sum())
        sdf[comp] = TRUNC * (1.0 if s[0] > 0 else -1.0)
    out = os.path.join(HERE, "g9_mesh.npz")
    np.savez_compressed(out, verts=V.astype(np.float32), normals=N.astype(np.float32), faces=F.astype(np.int32),
                        sdf=sdf.astype(np.float32), trunc=np.float64(TRUNC))
    print("wrote", out, os.path.getsize(out), "bytes; band voxels", int(known.sum()), "components", nlab)


if __name__ == "__main__":
    main()
