#!/usr/bin/env python3
"""Generates the marching-cubes triangle table used by csrc/dfh_mesh.hip and oracle/mc_np.py.

The table is DERIVED, not transcribed: for each of the 256 sign configurations the isosurface
polygons are found by (1) cutting each of the six cube faces with one segment per pair of crossed
face edges, (2) chaining the segments into closed loops, (3) orienting every loop so that its
right-hand normal points from the below-level corners to the above-level corners (the orientation of
the faces in the reference's own marching-cubes output, meshes/original.obj: geometric face normal
opposite to the stored vertex normal, which points down the gradient), and (4) triangulating each
polygon with diagonals that never lie in a cube face (see triangulate()).
A face whose four corners alternate (ambiguous face) is cut so that the two below-level corners are
separated (the above-level ones stay connected across the face -- what the reference's mesh shows in
the two ambiguous faces it contains); the rule only looks at the face's own corner signs, so the two
cubes sharing the face cut it identically and the mesh is watertight.  (skimage's Lewiner tables
resolve these faces, and some cube interiors, from the data values instead; that is not reproduced.)
Where the reference's mesh shows how skimage triangulates a configuration (tools/mc_observed.json,
88 of the 256), that triangulation is adopted.

Conventions:
  corner c in 0..7 has offset ((c>>0)&1, (c>>1)&1, (c>>2)&1) along axes (0, 1, 2);
  case index bit c is set when corner c is ABOVE the level (value > level);
  edge e = 4*a + o1 + 2*o2 runs along axis a from the corner with offsets (o1, o2) on the two other
  axes (in increasing axis order) and offset 0 on axis a;
  row `case` of the table = [n_triangles, e00, e01, e02, e10, ...] padded with -1.

usage: tools/gen_mc_table.py   (rewrites csrc/dfh_mc_table.h and oracle/mc_table.py)
"""
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def corner_off(c):
    return np.array([(c >> 0) & 1, (c >> 1) & 1, (c >> 2) & 1])


def edge_info(e):
    a, r = divmod(e, 4)
    o1, o2 = r & 1, r >> 1
    others = [x for x in range(3) if x != a]
    off0 = np.zeros(3, dtype=int)
    off0[others[0]] = o1
    off0[others[1]] = o2
    off1 = off0.copy()
    off1[a] = 1
    return a, off0, off1


def corner_id(off):
    return int(off[0]) | (int(off[1]) << 1) | (int(off[2]) << 2)


EDGE_BY_CORNERS = {}
for _e in range(12):
    _a, _o0, _o1 = edge_info(_e)
    EDGE_BY_CORNERS[frozenset((corner_id(_o0), corner_id(_o1)))] = _e


def face_cycles():
    """Six faces, each as its four corners in cyclic order."""
    out = []
    for n in range(3):
        u, v = [x for x in range(3) if x != n]
        for s in (0, 1):
            cyc = []
            for (du, dv) in ((0, 0), (1, 0), (1, 1), (0, 1)):
                off = np.zeros(3, dtype=int)
                off[n] = s
                off[u] = du
                off[v] = dv
                cyc.append(corner_id(off))
            out.append(cyc)
    return out


FACES = face_cycles()


def case_loops(case):
    above = [(case >> c) & 1 for c in range(8)]
    seg = {}                                        # edge -> list of neighbouring edges

    def link(e0, e1):
        seg.setdefault(e0, []).append(e1)
        seg.setdefault(e1, []).append(e0)

    for cyc in FACES:
        crossed = []                                # position i: edge between cyc[i] and cyc[i+1]
        for i in range(4):
            c0, c1 = cyc[i], cyc[(i + 1) % 4]
            if above[c0] != above[c1]:
                crossed.append(i)
        if len(crossed) == 2:
            e = [EDGE_BY_CORNERS[frozenset((cyc[i], cyc[(i + 1) % 4]))] for i in crossed]
            link(e[0], e[1])
        elif len(crossed) == 4:
            # corners alternate: cut off each BELOW corner with the two face edges that meet in it
            for i in range(4):
                if not above[cyc[i]]:
                    e_prev = EDGE_BY_CORNERS[frozenset((cyc[(i - 1) % 4], cyc[i]))]
                    e_next = EDGE_BY_CORNERS[frozenset((cyc[i], cyc[(i + 1) % 4]))]
                    link(e_prev, e_next)
        else:
            assert len(crossed) == 0
    for e, nb in seg.items():
        assert len(nb) == 2, (case, e, nb)
    loops = []
    todo = set(seg)
    while todo:
        start = min(todo)
        loop = [start]
        todo.discard(start)
        prev, cur = None, start
        while True:
            a, b = seg[cur]
            nxt = a if a != prev else b
            if a == b:                              # two segments between the same pair cannot happen
                raise AssertionError((case, cur))
            if nxt == start:
                break
            loop.append(nxt)
            todo.discard(nxt)
            prev, cur = cur, nxt
        loops.append(loop)
    # orientation
    out = []
    for loop in loops:
        mid = []
        g = np.zeros(3)
        for e in loop:
            a, o0, o1 = edge_info(e)
            mid.append((o0 + o1) / 2.0)
            up, dn = (o1, o0) if above[corner_id(o1)] else (o0, o1)
            g += up - dn
        mid = np.array(mid)
        n = np.zeros(3)
        for i in range(len(mid)):                  # Newell
            p, q = mid[i], mid[(i + 1) % len(mid)]
            n += np.cross(p, q)
        d = float(np.dot(n, g))
        assert abs(d) > 1e-9, (case, loop, n, g)
        if d < 0:
            loop = [loop[0]] + loop[:0:-1]
        k = loop.index(min(loop))                   # canonical start: smallest edge id
        out.append(loop[k:] + loop[:k])
    out.sort(key=lambda l: l[0])
    return out


def edge_faces(e):
    """The two cube faces (axis, side) an edge lies on."""
    a, o0, _ = edge_info(e)
    return {(n, int(o0[n])) for n in range(3) if n != a}


def on_common_face(e0, e1):
    return bool(edge_faces(e0) & edge_faces(e1))


def triangulate(loop):
    """All triangulations of the polygon are tried; the one with the fewest diagonals lying in a cube
    face wins (a diagonal in a face could coincide with the neighbouring cube's diagonal in the same,
    ambiguous, face: an edge with four triangles).  Ties: lexicographically smallest triangle list."""
    n = len(loop)
    best = {}

    def solve(i, j):                                # polygon loop[i..j], i<j, chord (i,j) already paid for
        if j - i < 2:
            return (0, [])
        key = (i, j)
        if key in best:
            return best[key]
        res = None
        for k in range(i + 1, j):
            c = 0
            if k - i >= 2:
                c += on_common_face(loop[i], loop[k])
            if j - k >= 2:
                c += on_common_face(loop[k], loop[j])
            cl, tl = solve(i, k)
            cr, tr = solve(k, j)
            cand = (c + cl + cr, sorted([(loop[i], loop[k], loop[j])] + tl + tr))
            if res is None or cand < res:
                res = cand
        best[key] = res
        return res

    cost, tris = solve(0, n - 1)
    return cost, tris


def directed_boundary(tris):
    """Directed edges used by exactly one triangle (an interior diagonal appears in both directions)."""
    d = {}
    for t in tris:
        for i in range(3):
            a, b = t[i], t[(i + 1) % 3]
            d[(a, b)] = d.get((a, b), 0) + 1
    return {e for e, c in d.items() if c == 1 and (e[1], e[0]) not in d}, d


def observed_rows():
    """tools/mc_observed.json (tools/learn_mc_triangulation.py): how the reference's own marching-cubes
    output (meshes/original.obj) triangulates each sign configuration that occurs in it."""
    path = os.path.join(ROOT, "tools", "mc_observed.json")
    if not os.path.exists(path):
        return {}
    import json
    return {int(k): [tuple(t) for t in v["tris"]] for k, v in json.load(open(path)).items()}


def build():
    rows = []
    worst = 0
    obs = observed_rows()
    adopted = rejected = 0
    for case in range(256):
        tris = []
        segs = set()
        for loop in case_loops(case):
            cost, t = triangulate(loop)
            worst = max(worst, cost)
            segs |= {(loop[i], loop[(i + 1) % len(loop)]) for i in range(len(loop))}
            # keep the loop's orientation: rotate every triangle so that it starts at its smallest edge id
            for tri in t:
                k = tri.index(min(tri))
                tris.append(tuple(tri[k:] + tri[:k]))
        tris.sort()
        if case in obs:
            o = list(obs[case])                     # order inside the cube and rotation as in the reference's file
            bnd, used = directed_boundary(o)
            diag = [e for e in used if (e[1], e[0]) in used]
            ok = bnd == segs and len(o) == len(tris) and all(c == 1 for c in used.values()) \
                and not any(on_common_face(a, b) for a, b in diag)
            if ok:
                adopted += 1
                tris = o
            else:
                rejected += 1
        rows.append(tris)
    print("diagonals lying in a cube face (worst case over all polygons):", worst)
    print("triangulations taken from the reference's mesh: %d cases (%d observed ones not compatible)" % (adopted, rejected))
    max_t = max(len(t) for t in rows)
    tab = -np.ones((256, 1 + 3 * max_t), dtype=np.int8)
    for c, tris in enumerate(rows):
        tab[c, 0] = len(tris)
        for i, t in enumerate(tris):
            tab[c, 1 + 3 * i:4 + 3 * i] = t
    return tab, max_t


def main():
    tab, max_t = build()
    hdr = os.path.join(ROOT, "dynamicfusion_body_amd", "csrc", "dfh_mc_table.h")
    with open(hdr, "w") as f:
        f.write("// GENERATED by tools/gen_mc_table.py -- do not edit.  Conventions: see that script.\n")
        f.write("#pragma once\nnamespace dfh {\n")
        f.write("constexpr int kMcMaxTris = %d;\nconstexpr int kMcRow = %d;\n" % (max_t, tab.shape[1]))
        f.write("__device__ __constant__ signed char kMcTable[256 * %d] = {\n" % tab.shape[1])
        for c in range(256):
            f.write("    " + ", ".join("%d" % v for v in tab[c]) + ",\n")
        f.write("};\n}  // namespace dfh\n")
    py = os.path.join(ROOT, "oracle", "mc_table.py")
    with open(py, "w") as f:
        f.write('"""GENERATED by tools/gen_mc_table.py -- do not edit (test infrastructure: the oracle\'s copy of the\n'
                'triangle table; row = [n_triangles, edge ids ...], conventions in the generator)."""\n')
        f.write("import numpy as np\n\nMAX_TRIS = %d\nTABLE = np.array([\n" % max_t)
        for c in range(256):
            f.write("    [" + ", ".join("%d" % v for v in tab[c]) + "],\n")
        f.write("], dtype=np.int8)\n")
    print("max triangles per cube:", max_t, " total triangles over 256 cases:", int(tab[:, 0].sum()))


if __name__ == "__main__":
    main()
