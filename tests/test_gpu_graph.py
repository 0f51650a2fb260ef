"""GPU: deformation-graph maintenance on the device path against the reference's outputs
(golden g8: construct_graph / update_graph run in the reference with marching cubes patched out) and inside the frame loop."""
import numpy as np
import pytest
import torch

from dynamicfusion_body_amd import Fusion, graph, scene
from dynamicfusion_body_amd.pipeline import SlabFrame

pytestmark = pytest.mark.gpu


def test_nearest_points_and_support_test_vs_numpy():
    rng = np.random.default_rng(3)
    cloud = rng.uniform(0, 40, size=(5000, 3))
    cloud[77] = cloud[4000]                                             # an exact tie: the lower index wins
    q = np.concatenate([rng.uniform(-5, 45, size=(300, 3)), cloud[[4000, 12]]])
    idx = graph.nearest_points(q, cloud).cpu().numpy()
    d2 = ((q[:, None, :] - cloud[None, :, :]) ** 2).sum(axis=2)
    assert np.array_equal(idx, d2.argmin(axis=1))
    assert idx[-2] == 77
    npos = rng.uniform(0, 40, size=(60, 3)); nw = rng.uniform(1.0, 6.0, size=60)
    nbr = torch.from_numpy(np.argsort(((cloud[:, None, :] - npos[None]) ** 2).sum(axis=2), axis=1)[:, :4].astype(np.int32)).cuda()
    flag = graph.unsupported_vertices(cloud, nbr, npos, nw).cpu().numpy().astype(bool)
    nb = nbr.cpu().numpy()
    want = (np.linalg.norm(npos[nb] - cloud[:, None, :], axis=2) / nw[nb]).min(axis=1) >= 1
    assert np.array_equal(flag, want) and 0 < want.sum() < len(want)


def test_construct_and_update_graph_device_match_reference(golden):
    g = golden("g8_graph_io")
    k = int(g["knn"])
    fu = Fusion(np.zeros((4, 4, 4)), 1.0, knn=k, write_warpfield=False)
    fu._vertices, fu._radius = g["verts"], float(g["radius"])
    fu.construct_graph()                                                # device vertex -> node table
    assert np.array_equal(np.array([n[0] for n in fu._nodes]), g["cg_idx"])
    assert np.array_equal(np.array([n[1] for n in fu._nodes]), g["cg_pos"])
    assert np.array_equal(np.array([n[2] for n in fu._nodes]), g["cg_dq"]) and fu._nodes[0][2].dtype == np.float32
    assert np.array_equal(np.array([n[3] for n in fu._nodes]), g["cg_w"])
    assert np.array_equal(np.asarray(fu._neighbor_look_up), g["cg_lookup"])
    fu._nodes = [(n[0], n[1], g["ug_dq_in"][i], n[3]) for i, n in enumerate(fu._nodes)]
    fu._vertices = g["verts2"]
    n_new = fu.update_graph(refresh_surface=False)                      # device path
    assert n_new == len(g["ug_idx"]) - len(g["cg_idx"]) and n_new > 0
    assert np.array_equal(np.array([n[0] for n in fu._nodes]), g["ug_idx"])
    assert np.array_equal(np.array([n[1] for n in fu._nodes]), g["ug_pos"])
    assert np.abs(np.array([np.asarray(n[2], dtype=np.float64) for n in fu._nodes]) - g["ug_dq"]).max() <= 1e-12   # device exp vs libm
    assert np.array_equal(np.array([n[3] for n in fu._nodes]), g["ug_w"])
    assert np.array_equal(np.asarray(fu._neighbor_look_up), g["ug_lookup"])
    assert fu._curr_tsdf is None and fu._correspondences == []


def test_frame_loop_inserts_nodes_where_the_graph_has_none():
    """A graph that covers only the x < R/2 half of the observed surface: the first frame with update_graph=True inserts nodes on the
    rest of the surface (every band sample supported afterwards, a second update inserts nothing), K3's stored
    neighbourhoods and the solver's block pattern follow the new graph, and the loop keeps tracking."""
    R, N = 128, 256
    H, W, fx, cx, cy = scene.CAMERAS["C2"]
    K = scene.intrinsics(fx, cx, cy)
    scale, center, tdist = scene.grid_params(R)
    node_pos, node_w = scene.fibonacci_nodes(N, R)
    front = node_pos[:, 0] < R / 2                                      # (the cameras see both x halves of the front)
    lws = [scene.view_extrinsic(a) for a in (0.0, 40.0, -40.0)]
    sf = SlabFrame(K, scale, center, R, tdist / scale, node_pos[front], node_w[front], knn=4, pcg_iters=10, band=2.0, distributed=False)
    for lw in lws:
        sf.integrate(torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda(), lw)
    sf.refresh_samples()
    sv = sf.fs.solver
    n0 = sv.N
    flag0 = graph.unsupported_vertices(sv.spos, sv.snbr, sv.node_pos, sv.node_w)
    assert int(flag0.sum()) > 1000                                      # the other half of the surface has no node within reach
    amp = np.array([0.5, -0.3, 0.2])
    counts = []
    for f in range(6):
        off = amp * np.sin(0.3 * (f + 1)) * scale
        ds = [torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0, sphere_offset=off)).cuda() for lw in lws]
        counts.append(sf.step(ds, lws, gn_iters=10, update_graph=True))
        if f == 0:
            n1 = sv.N
            assert n1 > n0 + 10
            from dynamicfusion_body_amd.solve import sample_knn
            nbr_plain, _ = sample_knn(sv.spos, sv.node_pos, sv.node_w, 4)            # exhaustive search, no brick lists
            bad = (nbr_plain != sv.snbr).any(dim=1)
            assert int(bad.sum()) == 0, ("brick-list node search differs from the exhaustive one after the graph grew",
                                         int(bad.sum()), sv.spos[bad][:5].cpu().numpy(), nbr_plain[bad][:5].cpu().numpy(), sv.snbr[bad][:5].cpu().numpy())
            flag1 = graph.unsupported_vertices(sv.spos, sv.snbr, sv.node_pos, sv.node_w)
            assert int(flag1.sum()) == 0                                # every sample now has a node within its weight radius
            new_dq = sv.node_dq[n0:].cpu().numpy()
            assert np.allclose(np.sqrt((new_dq ** 2).sum(axis=1)), 1.0, atol=1e-12)     # dq_blend normalises by the 8-norm
    assert sv.N - n1 <= 3                                               # the graph settles (a moving band may still expose a point)
    dq = sv.node_dq.cpu().numpy()
    assert np.isfinite(dq).all() and 2 * np.linalg.norm(dq[:, 4:], axis=1).max() < 2.0
    assert sv.node_nbr.shape[0] == sv.N and int(sv.snbr.max()) < sv.N and int(sv.snbr.max()) >= n0    # samples use the new nodes
    cost, cnt = sv.cost()
    assert np.isfinite(cost) and cnt > 1000
    # the stored-neighbourhood K3 path after the rebuild equals a search-every-call run on the same volumes
    from dynamicfusion_body_amd import kernels
    T1, W1 = sf.T.clone(), sf.Wt.clone()
    T2, W2 = sf.T.clone(), sf.Wt.clone()
    live = sf.live.clone()
    if sf._first:                                # the last frame still inserted a node: its workspace is new, nothing stored in it yet
        Ts, Ws = sf.T.clone(), sf.Wt.clone()
        kernels.fuse_volume_dqb(Ts, Ws, live, sv.node_pos, sv.node_dq, sv.node_w, 4, sf.ident_lw, sf.tvox, workspace=sf.ws_dqb, rebuild_candidates=True)
    kernels.fuse_volume_dqb(T1, W1, live, sv.node_pos, sv.node_dq, sv.node_w, 4, sf.ident_lw, sf.tvox, workspace=sf.ws_dqb, rebuild_candidates=False)
    kernels.fuse_volume_dqb(T2, W2, live, sv.node_pos, sv.node_dq, sv.node_w, 4, sf.ident_lw, sf.tvox)
    assert torch.equal(T1, T2) and torch.equal(W1, W2)
