"""GPU parity: marching cubes (dfh_mc_count / dfh_mc_emit / dfh_mc_reorder, csrc/dfh_mesh.hip) against
the numpy oracle -- face indices bit-exact, vertex positions bit-exact (fp64 interpolation, fp32
store, same operation order), normals to 2 fp32 ulp (fp64 sqrt) -- against the reference's own
marching-cubes output (tests/golden/g9_mesh.npz), and through the FusionDM / Fusion methods that call
it in the reference."""
import os

import numpy as np
import pytest
import torch

from oracle import mc_np
from dynamicfusion_body_amd import Fusion, FusionDM, mesh, scene

pytestmark = pytest.mark.gpu


def sphere(shape, c, r):
    X, Y, Z = np.meshgrid(*(np.arange(s) for s in shape), indexing="ij")
    return np.sqrt((X - c[0]) ** 2 + (Y - c[1]) ** 2 + (Z - c[2]) ** 2) - r


def run(vol, level, step=1, order="reference", dtype=torch.float32):
    t = torch.from_numpy(np.ascontiguousarray(vol)).to(device="cuda", dtype=dtype)
    return mesh.marching_cubes(t, level, step, as_numpy=True, order=order)


def check_equal(got, want):
    v, f, n, val = got
    V, F, N, VAL = want
    assert f.dtype == np.int32 and v.dtype == np.float32 and n.dtype == np.float32
    assert np.array_equal(f, F)
    assert np.array_equal(v, V)
    assert np.array_equal(val, VAL)
    assert np.abs(n - N).max() <= 2.5e-7                                  # 2 ulp of fp32 at |n| <= 1


@pytest.mark.parametrize("order", ["lattice", "reference"])
@pytest.mark.parametrize("shape,seed,dtype", [((14, 13, 12), 0, torch.float32), ((9, 31, 20), 3, torch.float64),
                                              ((2, 2, 2), 4, torch.float32), ((5, 2, 70), 6, torch.float64)])
def test_noise_all_configurations(shape, seed, dtype, order):
    """White noise: all 256 cube configurations (at the larger sizes), ragged shapes, both dtypes."""
    vol = np.random.default_rng(seed).normal(size=shape)
    vol = vol.astype(np.float32) if dtype == torch.float32 else vol
    check_equal(run(vol, 0.1, order=order, dtype=dtype), mc_np.marching_cubes(vol, 0.1, order=order))


def test_reference_mesh_round_trip(golden):
    """The signed distance field of the reference's mesh -> the reference's face array, bit for bit."""
    g = golden("g9_mesh")
    v, f, n, val = run(g["sdf"], 0.0)
    assert np.array_equal(f, g["faces"])
    assert v.shape == g["verts"].shape
    d = np.linalg.norm(v.astype(np.float64) - g["verts"], axis=1)
    assert np.median(d) < 0.01 and np.percentile(d, 99) < 0.06
    assert np.median((n * g["normals"]).sum(1)) > 0.9999
    check_equal((v, f, n, val), mc_np.marching_cubes(g["sdf"], 0.0))


@pytest.mark.parametrize("step", [1, 2, 3])
def test_sphere_step_and_default_level(step):
    sd = sphere((41, 38, 45), (19.3, 17.1, 21.7), 12.4).astype(np.float32)
    check_equal(run(sd, None, step), mc_np.marching_cubes(sd, None, step))
    v, f, n, _ = run(sd, 0.0, step)
    rep = mc_np.mesh_report(v, f)
    assert rep["euler"] == 2 and rep["boundary_edges"] == 0 and rep["nonmanifold_edges"] == 0 and rep["misoriented_edges"] == 0


def test_emit_over_all_tiles_is_the_same():
    vol = np.random.default_rng(11).normal(size=(9, 14, 300)).astype(np.float32)       # two z segments per row
    t = torch.from_numpy(vol).cuda()
    a = mesh.marching_cubes(t, 0.0, as_numpy=True)
    b = mesh.marching_cubes(t, 0.0, as_numpy=True, visit_all_tiles=True)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    check_equal(a, mc_np.marching_cubes(vol, 0.0))
    lvl = 0.1234567                                                                    # not an fp32 number: native-type thresholds
    check_equal(run(vol, lvl), mc_np.marching_cubes(vol, lvl))
    lvl = float(vol[3, 4, 5])                                                          # exactly a sample: equality path
    check_equal(run(vol, lvl), mc_np.marching_cubes(vol, lvl))


def test_degenerate_faces_and_unused_vertices():
    vol = (np.round(np.random.default_rng(5).normal(size=(12, 11, 13)) * 2) / 2).astype(np.float32)
    got = run(vol, 0.0)
    check_equal(got, mc_np.marching_cubes(vol, 0.0))
    v, f = got[0], got[1]
    assert np.all(np.linalg.norm(np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]]), axis=1) > 0)
    assert len(np.unique(f)) == len(v)
    check_equal(run(vol, 0.0, order="lattice"), mc_np.marching_cubes(vol, 0.0, order="lattice"))


def test_no_surface_and_errors():
    vol = np.ones((8, 8, 8), dtype=np.float32)
    v, f, n, val = run(vol, 0.0)
    assert v.shape == (0, 3) and f.shape == (0, 3) and n.shape == (0, 3)
    with pytest.raises(ValueError):
        mesh.marching_cubes(torch.ones((1, 8, 8), device="cuda"), 0.0)
    with pytest.raises(ValueError):
        mesh.marching_cubes(torch.ones((8, 8, 8), device="cuda"), 0.0, step_size=0)
    with pytest.raises(ValueError):
        mesh.marching_cubes(torch.ones((8, 8, 8)), 0.0)                     # CPU tensor
    with pytest.raises(ValueError):
        mesh.marching_cubes(torch.ones((8, 8, 8), device="cuda"), float("nan"))


def test_256_cubed_properties():
    """Config-2/3 size: closed sphere, consistent orientation; size-independent facts only."""
    R = 256
    ax = torch.arange(R, device="cuda", dtype=torch.float32)
    sd = torch.sqrt((ax[:, None, None] - 127.3) ** 2 + (ax[None, :, None] - 129.1) ** 2 + (ax[None, None, :] - 126.6) ** 2) - 90.2
    v, f, n, val = mesh.marching_cubes(sd, 0.0, as_numpy=True)
    rep = mc_np.mesh_report(v, f)
    assert rep["euler"] == 2 and rep["boundary_edges"] == 0 and rep["nonmanifold_edges"] == 0 and rep["misoriented_edges"] == 0
    c = np.array([127.3, 129.1, 126.6])
    assert np.abs(np.linalg.norm(v - c, axis=1) - 90.2).max() < 5e-3
    assert ((n * (v - c)).sum(1) / np.linalg.norm(v - c, axis=1)).max() < -0.9999
    assert len(f) == 2 * len(v) - 4


def fused_sphere(R):
    H, W, fx, cx, cy = scene.CAMERAS["C1"]
    K = scene.intrinsics(fx, cx, cy)
    scale, center, tdist = scene.grid_params(R)
    fu = FusionDM(tdist, K, tsdf_res=R, marching_cubes_step_size=2)
    fu._ensure_volumes()
    for a in (0.0, 40.0, -40.0):
        lw = scene.view_extrinsic(a)
        dm = scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)
        fu.fuseDepths(torch.from_numpy(dm).cuda(), lw, fu._T, fu._Wt, scale=scale, center=center)
    return fu, scale


def test_fusion_dm_methods(tmp_path):
    """FusionDM.marching_cubes (no level: skimage's default (min+max)/2) and write_canonical_mesh
    (level 0, world coordinates through _IND, `f a//a` rows)."""
    R = 48
    fu, scale = fused_sphere(R)
    T = fu._tsdf
    fu.marching_cubes()
    want = mc_np.marching_cubes(T.astype(np.float32), None, 1)
    check_equal((fu._vertices, fu._faces, fu._normals, want[3]), want)
    got = fu.marching_cubes(T, step_size=0)                                 # < 1 -> _marching_cubes_step_size = 2
    check_equal(got, mc_np.marching_cubes(T.astype(np.float32), None, 2))
    fu._IND = np.array([[0.5, 0, 0, 1.0], [0, 0.5, 0, -2.0], [0, 0, 0.5, 3.0], [0, 0, 0, 1.0]])
    fu.write_canonical_mesh(str(tmp_path), "canon.obj")
    V, F, N = mesh.read_obj(os.path.join(str(tmp_path), "canon.obj"))
    w = mc_np.marching_cubes(T.astype(np.float32), 0.0, 1)
    assert np.array_equal(F, w[1].astype(np.int64) + 1)
    assert np.abs(V - (w[0].astype(np.float64) * 0.5 + np.array([1.0, -2.0, 3.0]))).max() < 1e-6      # %f rows
    assert np.abs(N - w[2].astype(np.float64) * 0.5).max() < 1e-6
    lines = open(os.path.join(str(tmp_path), "canon.obj")).read().splitlines()
    assert lines[0].startswith("v ") and lines[len(V)].startswith("vn ") and "//" in lines[-1]
    # write_warp_field (core/fusion_dm.py:334-336): pickle of `_nodes`, <name>__<itercounter>.p
    import pickle
    fu._nodes = [(3, np.array([1.0, 2, 3]), np.arange(8.0), 6.2)]
    fu._itercounter = 4
    fn = fu.write_warp_field(str(tmp_path), "wf")
    assert os.path.basename(fn) == "wf__4.p"
    back = pickle.load(open(fn, "rb"))
    assert back[0][0] == 3 and np.array_equal(back[0][2], np.arange(8.0))


def test_compute_live_tsdf_output_mesh_writes_the_reference_files(tmp_path):
    """compute_live_tsdf(outputMesh=True) (core/fusion_dm.py:174-176): `tsdf_temp.npy` and `test.obj`."""
    R = 32
    H, W_, fx, cx, cy = scene.CAMERAS["C1"]
    K = scene.intrinsics(fx, cx, cy)
    f = FusionDM(0.6, K, tsdf_res=R)
    avg = np.array([-0.03, -0.43, -5.6], dtype='float32')
    c = avg.astype(np.float64)
    lws = [scene.view_extrinsic(a, centre=c) for a in (0.0, 30.0)]
    dms = [scene.render_depth(K, lw, H, W_, invalid_frac=0.0, sphere_c=c, sphere_r=2.0, wall_z=None) for lw in lws]
    T, Wt = f.compute_live_tsdf(dms, lws, outputMesh=True, mesh_path=str(tmp_path))
    saved = np.load(os.path.join(str(tmp_path), "tsdf_temp.npy"))
    assert saved.shape == (R, R, R) and np.array_equal(saved.astype(np.float64), np.asarray(T, dtype=np.float64))
    V, F, N = mesh.read_obj(os.path.join(str(tmp_path), "test.obj"))
    assert len(V) > 100 and len(F) > 100


def test_fusion_initial_graph():
    """Fusion.initialize_canonical = the reference constructor's tail (core/fusion.py:86-96): marching
    cubes with the configured step, radius from the mean face edge length, deformation graph."""
    R = 40
    sd = np.clip(sphere((R, R, R), (19.6, 20.2, 19.9), 11.5), -3.0, 3.0).astype(np.float32)
    fu = Fusion(sd, 3.0, subsample_rate=4.0, knn=4, marching_cubes_step_size=2, write_warpfield=False)
    fu.initialize_canonical()
    V, F, N, _ = mc_np.marching_cubes(sd, None, 2)
    assert np.array_equal(fu._faces, F) and np.array_equal(fu._vertices, V)
    e = np.array([(np.linalg.norm(V[a] - V[b]) + np.linalg.norm(V[a] - V[c]) + np.linalg.norm(V[b] - V[c])) / 3 for a, b, c in F])
    assert fu._radius == 4.0 * np.average(e)                                # fp32 arithmetic, as on the reference's fp32 vertices
    assert abs(fu.average_edge_dist_in_face(F[7]) - e[7]) < 1e-6
    assert len(fu._nodes) > 4 and fu._kdtree is not None
