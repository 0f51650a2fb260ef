"""TEST INFRASTRUCTURE -- CPU restatement of the reference's deformation-graph maintenance, statement for statement:
construct_graph (core/fusion.py:101-123) and update_graph (core/fusion.py:203-233), numpy + scipy.spatial.KDTree as in the
reference.  Pinned by golden g8 (tests/test_graph_io.py: the reference's own outputs).  Only tests may import this module;
the product path (dynamicfusion_body_amd/graph.py) runs these steps on the device and builds no KD-tree."""
import numpy as np
from scipy.spatial import KDTree

from dynamicfusion_body_amd.graph import NEW_NODE_DQ, uniform_sample   # (the greedy radius subsampling is host code in the product too)


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
