"""Deformation-graph maintenance: radius subsampling (core/util.py:27-47), graph
construction (core/fusion.py:101-123) and the per-frame graph update (core/fusion.py:201-239).

Device path (`construct_graph_device`, `update_graph_device`; what Fusion and pipeline.SlabFrame use): every
O(vertices x nodes) step is a HIP kernel behind the C ABI -- anchor vertices (dfh_nearest_points), the vertex -> node
table (dfh_sample_knn), the unsupported-vertex test (dfh_graph_unsupported), the new nodes' DQs (dfh_dq_blend_points).
Only the greedy radius subsampling stays on the host: it is sequential by definition and runs on the unsupported set only.
The reference's own loops (numpy + a KD-tree, statement for statement) live in oracle/graph_np.py as the checker of this
path; nothing here builds a KD-tree: the `_kdtree` attribute the reference's callers touch is a `NodeIndex`, whose
query() runs on the device."""
import numpy as np

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


# ------------------------------------------------------------------------------------------------- device path
def _dev64(a):
    import torch
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64)))
    return t.to(device="cuda", dtype=torch.float64).contiguous()


def nearest_points(query, cloud):
    """Index of the nearest cloud point of every query point (CUDA int32 tensor): KDTree(cloud).query(q)[1]."""
    import torch
    from . import _lib
    from .device import current_stream_ptr, require_gpu
    require_gpu()
    lib = _lib.load()
    Q, C = _dev64(query), _dev64(cloud)
    if Q.dim() != 2 or Q.shape[1] != 3 or C.dim() != 2 or C.shape[1] != 3 or C.shape[0] < 1:
        raise ValueError("nearest_points needs (n,3) queries and a non-empty (m,3) cloud")
    idx = torch.empty(Q.shape[0], dtype=torch.int32, device="cuda")
    _lib.check(lib.dfh_nearest_points(Q.data_ptr(), Q.shape[0], C.data_ptr(), C.shape[0], idx.data_ptr(), 0, current_stream_ptr()),
               "dfh_nearest_points")
    return idx


def unsupported_vertices(vertices, nbr, node_pos, node_w):
    """uint8 CUDA flags: 1 where no node of the vertex's `nbr` row supports it (min |node - v| / w >= 1)."""
    import torch
    from . import _lib
    from .device import current_stream_ptr, require_gpu
    require_gpu()
    lib = _lib.load()
    V, P, Wn = _dev64(vertices), _dev64(node_pos), _dev64(node_w)
    nb = nbr.to(device="cuda", dtype=torch.int32).contiguous()
    if nb.dim() != 2 or nb.shape[0] != V.shape[0]:
        raise ValueError("neighbour table must be (n_vertices, knn)")
    flag = torch.empty(V.shape[0], dtype=torch.uint8, device="cuda")
    _lib.check(lib.dfh_graph_unsupported(V.data_ptr(), V.shape[0], nb.data_ptr(), nb.shape[1], P.data_ptr(), Wn.data_ptr(), P.shape[0],
                                         flag.data_ptr(), current_stream_ptr()), "dfh_graph_unsupported")
    return flag


def dq_blend_points(points, nbr, node_dq, node_pos, node_w):
    """Fusion.dq_blend for a batch of points over the nodes `nbr[p]` -> (P,8) CUDA fp64."""
    import torch
    from . import _lib
    from .device import current_stream_ptr, require_gpu
    require_gpu()
    lib = _lib.load()
    X, Q, P, Wn = _dev64(points), _dev64(node_dq), _dev64(node_pos), _dev64(node_w)
    nb = nbr.to(device="cuda", dtype=torch.int32).contiguous()
    out = torch.empty((X.shape[0], 8), dtype=torch.float64, device="cuda")
    _lib.check(lib.dfh_dq_blend_points(X.data_ptr(), X.shape[0], nb.data_ptr(), nb.shape[1], Q.data_ptr(), P.data_ptr(), Wn.data_ptr(),
                                       P.shape[0], out.data_ptr(), current_stream_ptr()), "dfh_dq_blend_points")
    return out


class NodeIndex:
    """What the reference keeps in `_kdtree` (a scipy KDTree over the node positions, core/fusion.py:120,232) as far as its
    callers use it: `.data` and `.query(x, k)` -> (distances, indices), nearest first, ties to the lower index -- answered
    by dfh_sample_knn on the device (a brute-force scan: the graph has a few thousand nodes at most)."""

    def __init__(self, points):
        self.data = np.ascontiguousarray(np.asarray(points, dtype=np.float64).reshape(-1, 3))
        self.n = len(self.data)

    def query(self, x, k=1):
        from .solve import sample_knn
        x = np.asarray(x, dtype=np.float64)
        single = x.ndim == 1
        X = np.ascontiguousarray(x.reshape(-1, 3))
        kk = min(int(k), self.n)
        nbr, _ = sample_knn(X, self.data, np.ones(self.n), kk)
        idx = nbr.cpu().numpy().astype(np.int64)
        d = np.linalg.norm(self.data[idx] - X[:, None, :], axis=2)
        if kk < int(k):                                   # scipy pads missing neighbours with (inf, n)
            pad = int(k) - kk
            idx = np.concatenate([idx, np.full((len(X), pad), self.n, dtype=np.int64)], axis=1)
            d = np.concatenate([d, np.full((len(X), pad), np.inf)], axis=1)
        if int(k) == 1:
            d, idx = d[:, 0], idx[:, 0]
        return (d[0], idx[0]) if single else (d, idx)


def construct_graph_device(vertices, radius, knn):
    """construct_graph with the vertex -> node table computed on the device.  Returns (node_vidx (N,) int64 numpy,
    node_pos (N,3) numpy, node_dq (N,8) float32 numpy, node_w (N,) numpy, lookup (V,knn) CUDA int32)."""
    from .solve import sample_knn
    nodes_v, nodes_idx = uniform_sample(vertices, radius)
    N = len(nodes_v)
    node_w = np.full(N, 2.0 * radius)
    lookup, _ = sample_knn(vertices, nodes_v, node_w, min(knn, N))
    return np.asarray(nodes_idx, dtype=np.int64), np.asarray(nodes_v, dtype=np.float64), np.tile(NEW_NODE_DQ, (N, 1)), node_w, lookup


def update_graph_device(node_pos, node_dq, node_w, vertices, radius, knn, gather_unsupported=None):
    """update_graph (reference core/fusion.py:203-233) on device arrays.
    node_pos (N,3), node_dq (N,8), node_w (N,): the OLD graph (numpy or CUDA); vertices (V,3): the refreshed surface.
    Returns (node_vidx (N+n,) CUDA int32, node_pos, node_dq, node_w (CUDA fp64, old nodes first), lookup (V,knn) CUDA
    int32 against the new graph, n_new).  gather_unsupported: optional callable mapping this rank's unsupported
    vertices (numpy (u,3)) to the concatenation over all ranks (so that every rank inserts the same nodes)."""
    import torch
    from .solve import sample_knn
    V = _dev64(vertices)
    P, Q, Wn = _dev64(node_pos), _dev64(node_dq), _dev64(node_w)
    N = P.shape[0]
    k = min(int(knn), N)
    # :209-212 re-anchoring also resets every old node's weight to 2 * radius BEFORE the support test and the blend
    Wn = torch.full((N,), 2.0 * float(radius), dtype=torch.float64, device="cuda")
    # :215-219 unsupported vertices: the OLD graph's knn nodes of every vertex
    nbr_old, _ = sample_knn(V, P, Wn, k)
    flag = unsupported_vertices(V, nbr_old, P, Wn)
    uns_idx = torch.nonzero(flag).reshape(-1)
    uns = V[uns_idx].cpu().numpy()
    if gather_unsupported is not None:
        uns = gather_unsupported(uns)
    # :221 greedy radius subsampling of the unsupported set (sequential by definition: host)
    new_v, new_i = uniform_sample(uns, radius)
    n_new = len(new_v)
    Wnew = torch.full((N + n_new,), 2.0 * float(radius), dtype=torch.float64, device="cuda")       # :212, :225: every node's weight
    if n_new:
        Xn = _dev64(new_v)
        nb_new, _ = sample_knn(Xn, P, Wn, k)                               # :222 dq_blend(pos) queries the OLD tree
        Qn = dq_blend_points(Xn, nb_new, Q, P, Wn)
        P2, Q2 = torch.cat([P, Xn]), torch.cat([Q, Qn])
    else:
        P2, Q2 = P, Q
    # :209-212 anchors of the old nodes = their nearest vertex; new nodes carry their index in the unsupported list (:223)
    if V.shape[0] > 0:
        vidx_old = nearest_points(P, V)
    else:
        vidx_old = torch.zeros(N, dtype=torch.int32, device="cuda")
    vidx = torch.cat([vidx_old, torch.from_numpy(np.asarray(new_i, dtype=np.int32)).cuda()]) if n_new else vidx_old
    lookup, _ = sample_knn(V, P2, Wnew, min(int(knn), N + n_new))          # :229-233
    return vidx, P2, Q2, Wnew, lookup, n_new
