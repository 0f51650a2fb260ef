"""CPU: the C restatement (oracle/oracle_c.c) bit for bit against the numpy oracle and against
the reference's own outputs (golden g2, g6)."""
import numpy as np

from oracle import oracle_c as C
from oracle import oracle_np as O


def test_c_oracle_matches_reference_golden_g2(golden):
    g = golden("g2_fuse_depths")
    R = int(g["R"]); K = g["K"]; Kinv = np.linalg.inv(K)
    T = np.zeros((R, R, R)) + float(g["tdist"]); W = np.zeros((R, R, R))
    for i in range(5):
        C.fuse_depths(g["dms"][i], g["lws"][i], K, Kinv, T, W, float(g["tdist"]), scale=float(g["scale"]),
                      center=g["center"], wmax=float(g["wmax"]))
        if i == 0:
            assert np.array_equal(W, g["W_after1"]) and np.abs(T - g["T_after1"]).max() <= 1e-12
    assert np.array_equal(W, g["W_after5"]) and np.abs(T - g["T_after5"]).max() <= 1e-12


def test_c_oracle_matches_numpy_oracle_and_g6(golden):
    from dynamicfusion_body_amd import scene
    g = golden("g6_config1")
    R = int(g["R"])
    H, W_, fx, cx, cy = scene.CAMERAS["C1"]
    K = scene.intrinsics(fx, cx, cy); Kinv = np.linalg.inv(K)
    Tc = np.zeros((R, R, R)) + float(g["tdist"]); Wc = np.zeros((R, R, R))
    Tn, Wn = Tc.copy(), Wc.copy()
    for a, dt in ((0.0, np.float64), (35.0, np.float32)):
        lw = scene.view_extrinsic(a)
        dm = scene.render_depth(K, lw, H, W_, dtype=dt)
        n = C.fuse_depths(dm, lw, K, Kinv, Tc, Wc, float(g["tdist"]), scale=float(g["scale"]), center=g["center"], n_threads=3)
        _, _, mask = O.fuse_depths(dm, lw, K, Kinv, Tn, Wn, float(g["tdist"]), scale=float(g["scale"]), center=g["center"],
                                   return_mask=True)
        assert n == int(mask.sum())
        assert np.array_equal(Wc, Wn) and np.array_equal(Tc, Tn)          # same IEEE operations in the same order
        if a == 0.0:
            assert int((Wc > 0).sum()) == int(g["updated"])
            assert np.array_equal(np.packbits((Wc > 0).reshape(-1)), g["mask_packed"])
    # slab + general K (skew) + ragged shape
    K2 = K.copy(); K2[0, 1] = 0.4
    res = (10, 7, 13)
    T1 = np.full(res, 0.3); W1 = np.zeros(res); T2, W2 = T1.copy(), W1.copy()
    lw = scene.view_extrinsic(-20.0)
    dm = scene.render_depth(K2, lw, H, W_)
    for xr in ((0, 4), (4, 10)):
        C.fuse_depths(dm, lw, K2, np.linalg.inv(K2), T1, W1, 0.3, tsdf_res=12, scale=0.12, center=scene.SPHERE_C, wmax=2.0, x_range=xr)
    O.fuse_depths(dm, lw, K2, np.linalg.inv(K2), T2, W2, 0.3, tsdf_res=12, scale=0.12, center=scene.SPHERE_C, wmax=2.0)
    assert np.array_equal(T1, T2) and np.array_equal(W1, W2) and (W2 > 0).any()
    assert C.threads() >= 1
