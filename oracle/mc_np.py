"""TEST INFRASTRUCTURE ONLY: numpy statement of the marching-cubes mesh extraction that
csrc/dfh_mesh.hip implements, used by tests/ as the checker.

What it stands for in the reference: `measure.marching_cubes_lewiner(tsdf, level, step_size,
allow_degenerate=False)` as called at core/fusion_dm.py:319-331,342 and core/fusion.py:554-568
(skimage 0.13.1: a third-party dependency, not vendored, not installed here).  Its published
algorithm: Lewiner et al. 2003 marching cubes -- vertices on the lattice edges at the linearly
interpolated crossing of the level, per-configuration triangle tables.  skimage's tables are not
available, so the table is derived (tools/gen_mc_table.py).  Pinned by the reference's own output
mesh meshes/original.obj (tests/golden/g9_mesh.npz; tests/test_mesh_oracle.py):
  * vertices on lattice edges, array-index coordinates; unit normals pointing DOWN the gradient; faces
    wound with their right-hand normal UP the gradient; zero-area faces dropped;
  * faces ordered cube by cube in C order, vertices numbered by first use in the right-to-left rows;
  * the triangulation of the 88 sign configurations that occur in that mesh (adopted from it), and
    the cut of an ambiguous cube face (above-level corners stay connected, as in its two instances);
  * run on a signed distance field of that mesh, the mesh's face array comes back bit for bit.
"Parity unpinned" (nothing in the reference covers it): the triangulation of the other 168
configurations, skimage's value-dependent resolution of ambiguous faces / cube interiors (a fixed
rule here), its vertex normals away from smooth fields (here: lattice central differences,
interpolated along the edge), `values`, and `level=None` = (min+max)/2 (skimage's documented default).

Output order, order="lattice" (what dfh_mc_emit produces):
  vertices: by owning lattice point in C order of the (sub-sampled) volume, then by edge axis 0,1,2;
  faces:    by cube in C order, then in table order.
order="reference" (dfh_mc_reorder on top): faces unchanged, vertices renumbered by first use when the
face rows are read right-to-left, unused vertices dropped -- the order of the reference's file
(skimage numbers vertices as its faces create them and flips the rows afterwards).  With the 88
configurations of that file triangulated as in the file, its face array is reproduced exactly.
"""
import numpy as np

from .mc_table import TABLE


def _edge_info(e):
    a, r = divmod(e, 4)
    o1, o2 = r & 1, r >> 1
    others = [x for x in range(3) if x != a]
    off = [0, 0, 0]
    off[others[0]] = o1
    off[others[1]] = o2
    return a, tuple(off)


def _gradient(V):
    """Central differences, one-sided on the faces of the volume (fp64)."""
    G = np.zeros(V.shape + (3,), dtype=np.float64)
    for a in range(3):
        n = V.shape[a]
        Vm = np.moveaxis(V, a, 0)
        g = np.zeros_like(Vm)
        if n > 1:
            g[1:-1] = (Vm[2:] - Vm[:-2]) * 0.5
            g[0] = Vm[1] - Vm[0]
            g[-1] = Vm[-1] - Vm[-2]
        G[..., a] = np.moveaxis(g, 0, a)
    return G


def reorder_first_use(verts, faces, normals, values):
    """Vertices renumbered by first use in the face rows read right-to-left; unused ones dropped."""
    flat = faces[:, ::-1].reshape(-1)
    used, first = np.unique(flat, return_index=True)
    order = used[np.argsort(first, kind="stable")]           # old ids in new order
    newid = np.full(len(verts), -1, dtype=np.int64)
    newid[order] = np.arange(len(order))
    return verts[order], newid[faces].astype(np.int32), normals[order], values[order]


def marching_cubes(vol, level=None, step_size=1, allow_degenerate=False, order="reference"):
    """-> verts (V,3) f32, faces (F,3) i32, normals (V,3) f32, values (V,) f32."""
    out = _marching_cubes_lattice(vol, level, step_size, allow_degenerate)
    return reorder_first_use(*out) if order == "reference" else out


def _marching_cubes_lattice(vol, level=None, step_size=1, allow_degenerate=False):
    vol = np.asarray(vol)
    if level is None:
        level = 0.5 * (float(vol.min()) + float(vol.max()))
    level = float(level)
    s = int(step_size)
    V = np.ascontiguousarray(vol[::s, ::s, ::s]).astype(np.float64)
    NX, NY, NZ = V.shape
    above = V > level
    # --- vertices -----------------------------------------------------------------------------
    cross = np.zeros((NX, NY, NZ, 3), dtype=bool)
    cross[:-1, :, :, 0] = above[:-1] != above[1:]
    cross[:, :-1, :, 1] = above[:, :-1] != above[:, 1:]
    cross[:, :, :-1, 2] = above[:, :, :-1] != above[:, :, 1:]
    flat = cross.reshape(-1)
    base = np.cumsum(flat) - flat                       # exclusive, order = (point, axis)
    vid = base.reshape(NX, NY, NZ, 3)
    idx = np.argwhere(cross)                            # rows (x, y, z, a) in the same order
    nv = len(idx)
    G = _gradient(V)
    p0 = idx[:, :3]
    a = idx[:, 3]
    p1 = p0.copy()
    p1[np.arange(nv), a] += 1
    f0 = V[p0[:, 0], p0[:, 1], p0[:, 2]]
    f1 = V[p1[:, 0], p1[:, 1], p1[:, 2]]
    t = (level - f0) / (f1 - f0)
    pos = p0.astype(np.float64)
    pos[np.arange(nv), a] = pos[np.arange(nv), a] + t
    verts = (pos * float(s)).astype(np.float32)
    g0 = G[p0[:, 0], p0[:, 1], p0[:, 2]]
    g1 = G[p1[:, 0], p1[:, 1], p1[:, 2]]
    g = g0 + t[:, None] * (g1 - g0)
    n2 = (g[:, 0] * g[:, 0] + g[:, 1] * g[:, 1]) + g[:, 2] * g[:, 2]
    nrm = np.sqrt(n2)
    inv = np.where(nrm > 0.0, -1.0 / np.where(nrm > 0.0, nrm, 1.0), 0.0)
    normals = (g * inv[:, None]).astype(np.float32)
    values = np.maximum(f0, f1).astype(np.float32)
    # --- faces --------------------------------------------------------------------------------
    if min(NX, NY, NZ) < 2:
        return verts, np.zeros((0, 3), np.int32), normals, values
    case = np.zeros((NX - 1, NY - 1, NZ - 1), dtype=np.int32)
    eq = np.zeros_like(case)                             # corners sitting exactly on the level
    for c in range(8):
        ox, oy, oz = c & 1, (c >> 1) & 1, (c >> 2) & 1
        sub = (slice(ox, NX - 1 + ox), slice(oy, NY - 1 + oy), slice(oz, NZ - 1 + oz))
        case |= above[sub].astype(np.int32) << c
        eq |= (V[sub] == level).astype(np.int32) << c
    cells = np.argwhere((case != 0) & (case != 255))
    faces = []
    for (x, y, z) in cells:
        cs = int(case[x, y, z])
        e_q = int(eq[x, y, z])
        row = TABLE[cs]
        for ti in range(int(row[0])):
            tri = []
            collapsed = []
            for e in row[1 + 3 * ti:4 + 3 * ti]:
                ax, off = _edge_info(int(e))
                tri.append(int(vid[x + off[0], y + off[1], z + off[2], ax]))
                c0 = off[0] | (off[1] << 1) | (off[2] << 2)
                c1 = c0 | (1 << ax)
                # the vertex sits on a corner iff that corner's value equals the level (t = 0 or 1)
                collapsed.append(c0 if (e_q >> c0) & 1 else (c1 if (e_q >> c1) & 1 else -1 - len(collapsed)))
            if not allow_degenerate and len(set(collapsed)) < 3:
                continue
            faces.append(tri)
    faces = np.array(faces, dtype=np.int32).reshape(-1, 3)
    return verts, faces, normals, values


def mesh_report(verts, faces):
    """Topology facts used by the tests: directed-edge consistency, boundary edges, Euler number."""
    E = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
    und = np.sort(E, axis=1)
    u, inv, cnt = np.unique(und, axis=0, return_inverse=True, return_counts=True)
    inv = inv.reshape(-1)
    sign = np.where(E[:, 0] < E[:, 1], 1, -1)
    bal = np.bincount(inv, weights=sign, minlength=len(u))
    used = np.unique(faces)
    return {"edges": len(u), "boundary_edges": int((cnt == 1).sum()), "nonmanifold_edges": int((cnt > 2).sum()),
            "misoriented_edges": int(((cnt == 2) & (bal != 0)).sum()),
            "euler": int(len(used) - len(u) + len(faces)), "used_vertices": int(len(used))}
