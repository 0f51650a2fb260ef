"""Marching cubes, CPU side: the numpy oracle (oracle/mc_np.py) against the reference's own
marching-cubes output (meshes/original.obj -> tests/golden/g9_mesh.npz, made by
tests/golden/make_golden_mesh.py) and against topology properties every cube configuration must
satisfy; the generated triangle table against its generator."""
import os
import sys

import numpy as np
import pytest

from oracle import mc_np
from oracle.mc_table import TABLE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sphere(shape, c, r):
    X, Y, Z = np.meshgrid(*(np.arange(s) for s in shape), indexing="ij")
    return np.sqrt((X - c[0]) ** 2 + (Y - c[1]) ** 2 + (Z - c[2]) ** 2) - r


def test_reference_mesh_round_trip(golden):
    """MC of the signed distance field of the reference's mesh gives the reference's mesh back: the face
    array bit for bit (same cubes, same triangles, same rotation, same vertex numbering), vertices on
    the same lattice edges (positions differ only because the field is re-derived from the mesh),
    normals and winding with the same conventions."""
    g = golden("g9_mesh")
    V, N, F, sdf = g["verts"], g["normals"], g["faces"], g["sdf"]
    v, f, n, val = mc_np.marching_cubes(sdf, 0.0)
    assert np.array_equal(f, F)
    assert v.shape == V.shape
    # same lattice edge: the two integer coordinates agree exactly, the third lies in the same unit interval
    frac = np.abs(V - np.round(V)) > 1e-6
    assert np.all(frac.sum(1) == 1)                                    # the reference's vertices sit on lattice edges
    assert np.array_equal(np.where(frac, 0, np.round(V)), np.where(frac, 0, np.round(v)))
    assert np.array_equal(np.floor(V[frac]), np.floor(v[frac]))
    d = np.linalg.norm(v.astype(np.float64) - V, axis=1)
    assert np.median(d) < 0.01 and np.percentile(d, 99) < 0.06
    dots = (n * N).sum(1)
    assert np.median(dots) > 0.9999 and np.percentile(dots, 1) > 0.97 and dots.min() > 0.9
    for (vv, ff, nn) in ((V, F, N), (v, f, n)):                        # conventions, checked on both
        assert np.allclose(np.linalg.norm(nn, axis=1), 1.0, atol=1e-5)
        gn = np.cross(vv[ff[:, 1]] - vv[ff[:, 0]], vv[ff[:, 2]] - vv[ff[:, 0]])
        assert np.all((gn * nn[ff].mean(1)).sum(1) < 0)                # right-hand face normal opposes the vertex normals
    assert mc_np.mesh_report(v, f) == mc_np.mesh_report(V, F)


def test_reference_vertex_order_rule(golden):
    """The rule dfh_mc_reorder implements, verified on the reference's file itself: vertex ids
    increase with first use when face rows are read right-to-left, faces are grouped by cube in C order."""
    g = golden("g9_mesh")
    V, F = g["verts"].astype(np.float64), g["faces"]
    _, first = np.unique(F[:, ::-1].reshape(-1), return_index=True)
    assert np.array_equal(np.argsort(first, kind="stable"), np.arange(len(V)))
    cell = np.floor(V[F].mean(1)).astype(np.int64)
    lin = (cell[:, 0] * 100 + cell[:, 1]) * 100 + cell[:, 2]
    assert np.all(np.diff(lin) >= 0)


@pytest.mark.parametrize("shape,seed", [((14, 13, 12), 0), ((24, 9, 17), 1)])
def test_all_configurations_watertight(shape, seed):
    """White noise hits all 256 configurations, ambiguous faces included: every interior edge has
    exactly two triangles with opposite directions, open edges only on the volume border."""
    vol = np.random.default_rng(seed).normal(size=shape)
    v, f, n, val = mc_np.marching_cubes(vol, 0.0, order="lattice")
    rep = mc_np.mesh_report(v, f)
    assert rep["nonmanifold_edges"] == 0 and rep["misoriented_edges"] == 0
    E = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
    u, c = np.unique(E, axis=0, return_counts=True)
    b = u[c == 1]
    mx = np.array(shape) - 1
    on = lambda p: np.any((p == 0) | (p == mx), axis=1)
    assert np.all(on(v[b[:, 0]]) & on(v[b[:, 1]]))
    # every configuration occurred
    ab = vol > 0
    case = np.zeros(tuple(s - 1 for s in shape), dtype=int)
    for c_ in range(8):
        ox, oy, oz = c_ & 1, (c_ >> 1) & 1, (c_ >> 2) & 1
        case |= ab[ox:shape[0] - 1 + ox, oy:shape[1] - 1 + oy, oz:shape[2] - 1 + oz].astype(int) << c_
    assert len(np.unique(case)) == (256 if min(shape) >= 12 else len(np.unique(case)))


def test_sphere_geometry():
    c, r = np.array([19.3, 17.1, 21.7]), 12.4
    sd = sphere((40, 36, 44), c, r)
    v, f, n, val = mc_np.marching_cubes(sd.astype(np.float32), 0.0)
    rep = mc_np.mesh_report(v, f)
    assert rep["euler"] == 2 and rep["boundary_edges"] == 0 and rep["nonmanifold_edges"] == 0 and rep["misoriented_edges"] == 0
    assert np.abs(np.linalg.norm(v - c, axis=1) - r).max() < 0.02
    radial = (v - c) / np.linalg.norm(v - c, axis=1)[:, None]
    assert ((n * radial).sum(1)).max() < -0.999                          # normals point down the gradient (inward)
    gn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    assert np.all((gn * (v[f].mean(1) - c)).sum(1) > 0)                  # faces wound outward
    assert np.all(val >= 0.0)                                            # max of the two samples of a crossed edge


def test_level_default_step_and_orders():
    sd = sphere((33, 30, 35), (16.2, 14.9, 17.3), 9.7)
    v0, f0, _, _ = mc_np.marching_cubes(sd, None)
    lvl = 0.5 * (sd.min() + sd.max())
    v1, f1, _, _ = mc_np.marching_cubes(sd, lvl)
    assert np.array_equal(v0, v1) and np.array_equal(f0, f1)             # level=None is (min+max)/2
    v2, f2, n2, _ = mc_np.marching_cubes(sd, 0.0, step_size=3)
    assert np.all(np.abs(v2 / 3 - np.round(v2 / 3)).min(axis=1) < 1e-6)  # vertices on the coarse lattice's edges
    assert mc_np.mesh_report(v2, f2)["euler"] == 2
    vl, fl, nl, _ = mc_np.marching_cubes(sd, 0.0, order="lattice")
    vr, fr, nr, _ = mc_np.marching_cubes(sd, 0.0, order="reference")
    assert len(vl) == len(vr) and np.array_equal(vl[fl], vr[fr])         # same triangles, renumbered vertices
    _, first = np.unique(fr[:, ::-1].reshape(-1), return_index=True)
    assert np.array_equal(np.argsort(first, kind="stable"), np.arange(len(vr)))


def test_degenerate_faces_dropped():
    """Samples exactly on the level put vertices on lattice points; triangles collapsing there are
    dropped (allow_degenerate=False), and with them the vertices nothing uses any more."""
    vol = np.round(np.random.default_rng(5).normal(size=(12, 11, 13)) * 2) / 2
    v, f, n, _ = mc_np.marching_cubes(vol, 0.0)
    va, fa, _, _ = mc_np.marching_cubes(vol, 0.0, allow_degenerate=True, order="lattice")
    area = lambda vv, ff: np.linalg.norm(np.cross(vv[ff[:, 1]] - vv[ff[:, 0]], vv[ff[:, 2]] - vv[ff[:, 0]]), axis=1)
    nz = int((area(va, fa) == 0).sum())
    assert nz > 50 and len(f) == len(fa) - nz and np.all(area(v, f) > 0)
    assert len(np.unique(f)) == len(v)


def test_table_matches_generator():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_mc_table
    tab, max_t = gen_mc_table.build()
    assert np.array_equal(tab, TABLE)
    hdr = open(os.path.join(ROOT, "dynamicfusion_body_amd", "csrc", "dfh_mc_table.h")).read()
    body = hdr[hdr.index("{", hdr.index("kMcTable")) + 1:hdr.rindex("};")]
    vals = np.array([int(x) for x in body.replace("\n", " ").split(",") if x.strip()], dtype=np.int8)
    assert np.array_equal(vals.reshape(256, -1), TABLE)
    assert int(TABLE[:, 0].sum()) == 820 and max_t == 5
