"""Deformation-graph maintenance, host-side like in the reference:
radius subsampling (core/util.py:27-47), graph construction (core/fusion.py:101-123) and the
per-frame graph update (core/fusion.py:201-239).  Runs once per frame on O(vertices) data."""
import numpy as np
from scipy.spatial import KDTree

NEW_NODE_DQ = np.array([1, 0.00, 0.00, 0.00, 0.00, 0.01, 0.01, 0.00], dtype=np.float32)     # core/fusion.py:115


def uniform_sample(arr, radius):
    """Greedy radius subsampling, reference core/util.py:27-47: repeatedly take the first remaining
    candidate and drop every candidate closer than `radius` to it (itself included).
    Returns (samples, their indices in `arr`)."""
    candidates = np.array(arr, dtype=np.float64).copy()
    if candidates.size == 0:
        return np.array([]), np.array([])
    locations = np.arange(len(candidates))
    result, result_idx = [], []
    while candidates.size > 0:
        sample = candidates[0]
        result.append(sample)
        result_idx.append(locations[0])
        d = candidates - sample
        keep = ~(np.sqrt(np.sum(d * d, axis=1)) < radius)
        candidates, locations = candidates[keep], locations[keep]
    return np.array(result), np.array(result_idx)


def construct_graph(vertices, radius, knn):
    """Reference core/fusion.py:101-123.  Returns (nodes, kdtree, neighbor_look_up): nodes is the
    list of 4-tuples (vertex index, position, DQ, weight = 2*radius)."""
    nodes_v, nodes_idx = uniform_sample(vertices, radius)
    nodes = [(nodes_idx[i], nodes_v[i], NEW_NODE_DQ.copy(), 2 * radius) for i in range(len(nodes_v))]
    kdtree = KDTree(nodes_v)
    lookup = [kdtree.query(v, k=knn)[1] for v in vertices]
    return nodes, kdtree, lookup


def update_graph(nodes, kdtree, vertices, radius, knn, dq_blend):
    """Reference core/fusion.py:203-233 after the marching-cubes refresh: re-anchor every node on
    its nearest vertex, find the vertices no node supports (min over their knn nodes of
    |node - v| / w >= 1), subsample them into new nodes whose DQ is the blend of the OLD graph at
    that point (`dq_blend(pos)`, :222), rebuild the KD-tree and the vertex -> node table.
    Returns (nodes, kdtree, neighbor_look_up, number of inserted nodes)."""
    vert_kdtree = KDTree(vertices)
    nodes = list(nodes)
    for i in range(len(nodes)):
        pos, se3 = nodes[i][1], nodes[i][2]
        _, vidx = vert_kdtree.query(pos)
        nodes[i] = (vidx, pos, se3, 2 * radius)
    unsupported = []
    for vert in vertices:
        _, kdidx = kdtree.query(vert, k=knn)
        if min([np.linalg.norm(nodes[idx][1] - vert) / nodes[idx][3] for idx in np.atleast_1d(kdidx)]) >= 1:
            unsupported.append(vert)
    new_v, new_idx = uniform_sample(unsupported, radius)
    for i in range(len(new_v)):
        nodes.append((new_idx[i], new_v[i], dq_blend(new_v[i]), 2 * radius))
    kdtree = KDTree(np.array([n[1] for n in nodes]))
    lookup = [kdtree.query(v, k=knn)[1] for v in vertices]
    return nodes, kdtree, lookup, len(new_v)
