"""CPU: the numpy oracle against vectors produced by running the reference itself
(tests/golden/make_golden.py) and against the reference's own doctest constants.
Masks / weights must be identical; values agree to fp64 rounding."""
import numpy as np
import pytest

from oracle import oracle_np as O

VAL_TOL = 1e-12   # fp64 value tolerance vs the reference (observed <= 1e-14)


def test_doctest_known_answers(golden):
    g = golden("g1_primitives")
    # core/util.py:258-260
    assert np.array_equal(O.quaternion_multiply([4, 1, -2, 3], [8, -5, 6, 7]), [28, -44, -14, 48])
    assert np.array_equal(g["qmul_doctest"], [28, -44, -14, 48])
    # core/util.py:146-154 : rotation by 0.123 about x
    M = O.quaternion_matrix([0.99810947, 0.06146124, 0, 0])
    c, s = np.cos(0.123), np.sin(0.123)
    assert np.allclose(M, [[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]])
    assert np.allclose(M, g["qmat_doctest"], atol=1e-15)
    assert np.allclose(O.quaternion_matrix([1, 0, 0, 0]), np.identity(4))
    assert np.allclose(O.quaternion_matrix([0, 1, 0, 0]), np.diag([1, -1, -1, 1]))
    # core/util.py:300-304 sign pattern
    assert np.array_equal(g["dqconj_probe"], [1, -2, -3, -4, -5, 6, 7, 8])
    assert np.array_equal(O.dual_quaternion_conjugate(np.arange(1.0, 9.0)), [1, -2, -3, -4, -5, 6, 7, 8])


def test_g1_quaternion_algebra(golden):
    g = golden("g1_primitives")
    assert np.array_equal(O.quaternion_multiply(g["qmul_q1"], g["qmul_q0"]), g["qmul_out"])
    assert np.array_equal(O.dual_quaternion_multiply(g["dqmul_a"], g["dqmul_b"]), g["dqmul_out"])
    assert np.array_equal(O.dual_quaternion_conjugate(g["dqmul_a"]), g["dqconj_out"])


def test_g1_dqb_warp(golden):
    g = golden("g1_primitives")
    for tag in ("unit", "non", "f32"):
        out = O.dqb_warp(g["warp_dq_" + tag], g["warp_pos"])
        assert np.array_equal(out, g["warp_%s_out" % tag]), tag
    assert np.array_equal(O.dqb_warp_normal(g["warp_dq_non"], g["warpn_n"]), g["warpn_non_out"])
    assert np.array_equal(O.dqb_warp_normal(g["warp_dq_f32"], g["warpn_n"]), g["warpn_f32_out"])


def test_g1_dqb_warp_closed_form(golden):
    """x' = |r|^2 R(r^) p + 2 vec(d (x) r*), gauge d += lambda r."""
    g = golden("g1_primitives")
    dq, pos = g["warp_dq_non"], g["warp_pos"]
    p32 = pos.astype(np.float32).astype(np.float64)
    for d, p, ref in zip(dq, p32, g["warp_non_out"]):
        r, dd = d[:4], d[4:]
        Rm = O.quaternion_matrix(r)[:3, :3]
        t = 2 * O.quaternion_multiply(dd, r * [1, -1, -1, -1])[1:]
        assert np.allclose(np.dot(r, r) * Rm @ p + t, ref, atol=1e-11)
        d2 = d.copy(); d2[4:] += 0.37 * r
        assert np.allclose(O.dqb_warp(d2, p), ref, atol=1e-11)


def test_g1_se3(golden):
    g = golden("g1_primitives")
    for d, M, rt in zip(g["warp_dq_unit"], g["dqtse3_out"], g["se3tdq_roundtrip"]):
        assert np.allclose(O.DQTSE3(d), M, atol=1e-14)
        mine = O.SE3TDQ_from_Rt(M[:3, :3], M[:3, 3])
        assert np.allclose(mine, rt, atol=1e-9)
        assert np.allclose(mine, d, atol=1e-9)


def test_g1_projection(golden):
    g = golden("g1_primitives")
    u, v, ok = O.project_to_pixel(g["proj_K"], g["proj_pos"])
    assert np.array_equal(ok, g["proj_ok"])
    assert not ok[:4].any()
    assert np.allclose(u[ok], g["proj_u"][ok], rtol=0, atol=1e-12)
    assert np.allclose(v[ok], g["proj_v"][ok], rtol=0, atol=1e-12)


def test_g1_interpolate(golden):
    g = golden("g1_primitives")
    out, valid = O.interpolate_tsdf(g["interp_pos"], g["interp_vol"])
    assert np.array_equal(valid, g["interp_valid"])
    assert (~valid).sum() == 6
    assert np.array_equal(out[valid], g["interp_out"][valid])
    # swapped y/z fractions: f = 9x+3y+z at (0.5,0.25,1.75) -> 8.0 (true trilinear 7.0)
    lin = np.fromfunction(lambda x, y, z: 9 * x + 3 * y + z, (3, 3, 3))
    o, ok = O.interpolate_tsdf(np.array([0.5, 0.25, 1.75]), lin)
    assert ok and o == 8.0 and g["interp_probe"] == 8.0


def test_g1_losses(golden):
    g = golden("g1_primitives")
    assert np.array_equal(O.huber_loss(g["loss_x"], 0.7), g["huber_out"])
    assert np.array_equal(O.tukey_biweight_loss(g["loss_x"], 1.3), g["tukey_out"])


def test_g2_fuse_depths(golden):
    g = golden("g2_fuse_depths")
    R = int(g["R"]); K = g["K"]; Kinv = np.linalg.inv(K)
    T = np.zeros((R, R, R)) + float(g["tdist"]); W = np.zeros((R, R, R))
    margin = [None]
    for i in range(5):
        O.fuse_depths(g["dms"][i], g["lws"][i], K, Kinv, T, W, float(g["tdist"]), scale=float(g["scale"]),
                      center=g["center"], wmax=float(g["wmax"]), margin_out=margin)
        assert margin[0] > 1e-9          # fixture is tie-free: parity is well defined
        if i == 0:
            assert np.array_equal(W, g["W_after1"])
            assert np.abs(T - g["T_after1"]).max() <= VAL_TOL
    assert np.array_equal(W, g["W_after5"])
    assert (W == 3).any() and (W == 0).any()
    assert np.abs(T - g["T_after5"]).max() <= VAL_TOL


def test_g2_fuse_depths_slabs(golden):
    """slab sweep [a,b) along axis 0 == full sweep restricted to those planes."""
    g = golden("g2_fuse_depths")
    R = int(g["R"]); K = g["K"]; Kinv = np.linalg.inv(K)
    T = np.zeros((R, R, R)) + float(g["tdist"]); W = np.zeros((R, R, R))
    for a, b in ((0, 7), (7, 13), (13, 20)):
        O.fuse_depths(g["dms"][0], g["lws"][0], K, Kinv, T, W, float(g["tdist"]), scale=float(g["scale"]),
                      center=g["center"], wmax=float(g["wmax"]), x_range=(a, b))
    assert np.array_equal(W, g["W_after1"])
    assert np.abs(T - g["T_after1"]).max() <= VAL_TOL


def test_g3_rigid(golden):
    g = golden("g3_rigid")
    T, W = g["T0"].copy(), g["W0"].copy()
    for r in range(4):
        O.update_tsdf_rigid(T, W, g["lives"][r], g["lw"], float(g["tdist"]), wmax=float(g["wmax"]))
        if r == 0:
            assert np.array_equal(W, g["W_after1"])
            assert np.abs(T - g["T_after1"]).max() <= VAL_TOL
    assert np.array_equal(W, g["W_after4"])
    assert np.abs(T - g["T_after4"]).max() <= VAL_TOL


def test_g4_dqb(golden):
    g = golden("g4_dqb")
    k = int(g["knn"])
    T, W = g["T0"].copy(), g["W0"].copy()
    for r in range(3):
        O.update_tsdf_dqb(T, W, g["lives"][r], g["node_pos"], g["node_dq"], g["node_w"], k, g["lw"],
                          float(g["tdist"]), wmax=float(g["wmax"]))
        if r == 0:
            assert np.array_equal(W != g["W0"], g["W_after1"] != g["W0"])
            assert np.abs(W - g["W_after1"]).max() <= VAL_TOL
            assert np.abs(T - g["T_after1"]).max() <= VAL_TOL
    assert np.abs(W - g["W_after3"]).max() <= VAL_TOL
    assert np.abs(T - g["T_after3"]).max() <= VAL_TOL


def test_g4_warp_blend(golden):
    g = golden("g4_dqb")
    k = int(g["knn"])
    P, Nn = g["warp_P"], g["warp_N"]
    loc = O.knn_bruteforce(P, g["node_pos"], k)
    assert np.array_equal(loc, g["warp_loc"])          # KDTree.query(k+1)[:-1] ordering
    bl = O.dq_blend(P, g["node_dq"][loc], g["node_pos"][loc], g["node_w"][loc])
    assert np.abs(bl - g["blend_out"]).max() <= 1e-14
    pw, nw = O.warp(P, g["node_dq"][loc], g["node_pos"][loc], g["node_w"][loc], normal=Nn, m_lw=g["lw"])
    assert np.abs(pw - g["warp_pos_out"]).max() <= VAL_TOL
    assert np.abs(nw - g["warp_nrm_out"]).max() <= VAL_TOL
    bd = O.dq_blend(P[:8], g["node_dq"][loc[:8]], g["node_pos"][loc[:8]], g["node_w"][loc[:8]], dmax=3.5)
    assert np.abs(bd - g["blend_dmax_out"]).max() <= 1e-14
    # zero-blend guard (core/fusion.py:544-549)
    z = O.dq_blend(P[:2], np.zeros((2, k, 8)), g["node_pos"][loc[:2]], g["node_w"][loc[:2]])
    assert np.array_equal(z, np.tile([1, 0, 0, 0, 0, 0, 0, 0], (2, 1)))


def test_g5_residuals(golden):
    g = golden("g5_residuals")
    f = O.computef(g["node_dq"].flatten(), g["verts"], g["norms"], g["corr"], g["nbr"], g["vidx"],
                   g["node_pos"], g["node_w"], g["lw"], float(g["rw"]))
    V, N, k = len(g["verts"]), len(g["node_pos"]), int(g["knn"])
    assert len(f) == V + 3 * k * N == len(g["computef_out"])          # core/fusion.py:374
    assert np.abs(f - g["computef_out"]).max() <= VAL_TOL
    assert abs(0.5 * f @ f - float(g["cost"])) <= 1e-9
    fl = O.computef_data(g["node_dq"], g["verts"], g["norms"], g["corr"], g["nbr"], g["node_pos"],
                         g["node_w"], g["lw2"])
    assert np.abs(fl - g["computef_lw_out"]).max() <= VAL_TOL
    keep = g["rigid_keep"]
    fr = O.computef_lw_rigid(g["rigid_x"], g["verts"][keep], g["norms"][keep], g["corr"][keep])
    assert np.abs(fr - g["rigid_out"]).max() <= VAL_TOL


def test_g6_config1(golden):
    """BASELINE config 1: 64^3, one 320x240 frame -- whole-volume update mask bit-exact."""
    from dynamicfusion_body_amd import scene
    g = golden("g6_config1")
    R = int(g["R"])
    H, W_, fx, cx, cy = scene.CAMERAS["C1"]
    K = scene.intrinsics(fx, cx, cy)
    lw = scene.view_extrinsic(0.0)
    dm = scene.render_depth(K, lw, H, W_)
    T = np.zeros((R, R, R)) + float(g["tdist"]); W = np.zeros((R, R, R))
    O.fuse_depths(dm, lw, K, np.linalg.inv(K), T, W, float(g["tdist"]), scale=float(g["scale"]), center=g["center"])
    assert int((W > 0).sum()) == int(g["updated"])
    assert np.array_equal(np.packbits((W > 0).reshape(-1)), g["mask_packed"])
    assert W.sum() == float(g["sumW"])
    assert abs(T.sum() - float(g["sumT"])) <= 1e-8
    assert np.abs(T.reshape(-1)[g["sample_idx"]] - g["sample_T"]).max() <= VAL_TOL
    assert np.array_equal(W.reshape(-1)[g["sample_idx"]], g["sample_W"])


def test_g7_correspondences(golden):
    """setupCorrespondences (both classes) with marching cubes patched out: the selection loop."""
    g = golden("g7_correspondences")
    k = int(g["knn"])
    vp = O.dqb_warp(g["lw"], g["verts"]); wn = O.dqb_warp_normal(g["lw"], g["norms"])
    for tol in (1.0, 0.35):
        best, cost, keep = O.closest_correspondences(vp, wn, g["lverts"], k, tol)
        assert np.array_equal(np.nonzero(keep)[0], g["dm_corridx_%g" % tol])
        assert np.array_equal(best[keep], g["dm_corr_%g" % tol])
    assert 0 < len(g["dm_corridx_0.35"]) < len(g["dm_corridx_1"])
    nbr = g["nbr"]
    vp2, wn2 = O.warp(g["verts"], g["node_dq"][nbr], g["node_pos"][nbr], g["node_w"][nbr], normal=g["norms"], m_lw=g["lw"])
    best2, _, _ = O.closest_correspondences(vp2, wn2, g["lverts"], k, 0.2)
    assert np.array_equal(best2, g["nr_corr"])          # prune_result=False keeps every row (fusion.py:276)
