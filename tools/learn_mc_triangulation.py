#!/usr/bin/env python3
"""Reads the reference's marching-cubes output (tests/golden/g9_mesh.npz: mesh + the signed distance
field derived from it) and records, for every sign configuration that occurs in it, how skimage
split the cube's polygon(s) into triangles, in the stored order and rotation -> tools/mc_observed.json
{case: [[e0,e1,e2], ...]}.
tools/gen_mc_table.py adopts an observed triangulation when it covers exactly the polygons its own
rules produce and uses no diagonal lying in a cube face (edge / corner numbering: gen_mc_table.py)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = np.load(os.path.join(ROOT, "tests", "golden", "g9_mesh.npz"))
V, F, sdf = g["verts"].astype(np.float64), g["faces"], g["sdf"]
above = sdf > 0.0
cen = V[F].mean(1)
cell = np.floor(cen).astype(int)
# vertex -> (lattice point, axis)
fr = np.abs(V - np.round(V)) > 1e-6
ax = np.argmax(fr, axis=1)
base = np.round(V).astype(int)
base[np.arange(len(V)), ax] = np.floor(V[np.arange(len(V)), ax]).astype(int)

def edge_id(a, off):
    others = [x for x in range(3) if x != a]
    return 4 * a + off[others[0]] + 2 * off[others[1]]

obs = {}
bad = 0
order = np.lexsort((cell[:, 2], cell[:, 1], cell[:, 0]))
i = 0
while i < len(order):
    j = i
    c = cell[order[i]]
    while j < len(order) and np.all(cell[order[j]] == c):
        j += 1
    case = 0
    for k in range(8):
        o = np.array([k & 1, (k >> 1) & 1, (k >> 2) & 1])
        p = c + o
        case |= int(above[p[0], p[1], p[2]]) << k
    tris = []
    okc = True
    for fi in order[i:j]:
        t = []
        for v in F[fi]:
            off = base[v] - c
            if off.min() < 0 or off.max() > 1 or off[ax[v]] != 0:
                okc = False
                break
            t.append(int(edge_id(int(ax[v]), off)))
        if not okc:
            break
        tris.append(tuple(t))                 # as stored: order inside the cube and rotation kept
    if okc:
        tris = tuple(tris)
        obs.setdefault(case, {})
        obs[case][tris] = obs[case].get(tris, 0) + 1
    else:
        bad += 1
    i = j
out = {}
for case, d in sorted(obs.items()):
    best = max(d.items(), key=lambda kv: kv[1])
    out[str(case)] = {"tris": [list(t) for t in best[0]], "count": best[1], "variants": len(d)}
json.dump(out, open(os.path.join(ROOT, "tools", "mc_observed.json"), "w"), indent=0, sort_keys=True)
print("cases observed:", len(out), " cubes skipped:", bad, " cases with >1 variant:", sum(1 for v in out.values() if v["variants"] > 1))
