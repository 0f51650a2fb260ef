"""GPU parity: K1 (dfh_integrate_depth through the C ABI) against the reference's golden
outputs and against the fp64 oracle on the same seeded inputs.

Bars (DESIGN.md "Parity"):
  * update masks and integer weights: bit-exact, fp32 and fp64 volumes alike;
  * fp64 volumes: T bit-exact against the oracle (same IEEE operations in the same order);
  * fp32 volumes: |dT| <= F32_TOL * (1 + |T|) per the number of integrations (T is rounded
    to float32 once per integration; the reference keeps float64).
"""
import numpy as np
import pytest
import torch

from oracle import oracle_np as O
from dynamicfusion_body_amd import FusionDM, _lib, kernels, scene

pytestmark = pytest.mark.gpu

F32_EPS = float(np.finfo(np.float32).eps)


@pytest.fixture(autouse=True, params=["bricks", "bricks_nocull", "rows"])
def k1_path(request):
    """Every test of this file runs on the sweeps of float32 volumes: the 4 x 2 x 32 brick column walk with conservative
    culling (depth pyramid + classification pass), the same walk over every brick (what mid-size slabs take), and the row sweep
    that projects every voxel (option k1_no_bricks)."""
    if request.param == "rows":
        _lib.set_option("k1_no_bricks", 1)
    else:
        _lib.set_option("k1_bricks_min", 0)          # (single views take the brick sweep on large slabs only)
        _lib.set_option("k1_cull", 1 if request.param == "bricks" else 0)
    return request.param


def f32_tol(n_integrations):
    return 2.0 * n_integrations * F32_EPS


def dev(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dtype=dtype)


def run_gpu(dm, lw, K, T0, W0, tdist, scale, center, wmax, vol_dtype, depth_dtype=None, tsdf_res=None,
            x_range=None, res=None):
    T = dev(T0, vol_dtype)
    W = dev(W0, vol_dtype)
    d = dev(dm, depth_dtype or torch.float64)
    kernels.integrate_depth(T, W, d, K, np.linalg.inv(K), lw, scale, center, tdist, wmax, tsdf_res=tsdf_res,
                            x_range=x_range, res=res)
    torch.cuda.synchronize()
    return T.cpu().numpy().astype(np.float64), W.cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("vol_dtype", [torch.float64, torch.float32])
def test_g2_golden_reference_outputs(golden, vol_dtype):
    """R=20, 48x64 depth with 5 % holes, five rotated views, wmax=3 -- expected values are the
    reference's own outputs (tests/golden/make_golden.py:g2)."""
    g = golden("g2_fuse_depths")
    R = int(g["R"]); K = g["K"]
    f = FusionDM(float(g["tdist"]), K, tsdf_res=R, volume_dtype=vol_dtype)
    T = np.zeros((R, R, R)) + float(g["tdist"]); W = np.zeros((R, R, R))
    for i in range(5):
        rT, rW = f.fuseDepths(g["dms"][i], g["lws"][i], T, W, scale=float(g["scale"]), center=g["center"],
                              wmax=float(g["wmax"]))
        assert rT is T and rW is W                      # in place + returned (fusion_dm.py:217)
        if i == 0:
            assert np.array_equal(W, g["W_after1"])
            tol = 1e-12 if vol_dtype == torch.float64 else f32_tol(1)
            assert np.all(np.abs(T - g["T_after1"]) <= tol * (1 + np.abs(g["T_after1"])))
    assert np.array_equal(W, g["W_after5"])
    tol = 1e-12 if vol_dtype == torch.float64 else f32_tol(5)
    assert np.all(np.abs(T - g["T_after5"]) <= tol * (1 + np.abs(g["T_after5"])))
    if vol_dtype == torch.float64:
        assert np.array_equal(T, g["T_after5"])         # observed: bit-exact


def test_g6_config1_reference_mask(golden):
    """BASELINE config 1 (64^3, one 320x240 frame): whole-volume update mask bit-exact against
    the mask the reference produced; sampled values within the fp32 bar."""
    g = golden("g6_config1")
    R = int(g["R"])
    H, W_, fx, cx, cy = scene.CAMERAS["C1"]
    K = scene.intrinsics(fx, cx, cy)
    lw = scene.view_extrinsic(0.0)
    dm = scene.render_depth(K, lw, H, W_)
    f = FusionDM(float(g["tdist"]), K, tsdf_res=R)
    T, W = f._new_volume_pair()
    f.fuseDepths(torch.from_numpy(dm).cuda(), lw, T, W, scale=float(g["scale"]), center=g["center"])
    Wn = W.cpu().numpy().astype(np.float64); Tn = T.cpu().numpy().astype(np.float64)
    assert int((Wn > 0).sum()) == int(g["updated"])
    assert np.array_equal(np.packbits((Wn > 0).reshape(-1)), g["mask_packed"])
    assert Wn.sum() == float(g["sumW"])
    sT = Tn.reshape(-1)[g["sample_idx"]]
    assert np.all(np.abs(sT - g["sample_T"]) <= f32_tol(1) * (1 + np.abs(g["sample_T"])))
    assert np.array_equal(Wn.reshape(-1)[g["sample_idx"]], g["sample_W"])


CASES = [
    # res,            HxW,      pinhole, depth dtype,    wmax, repeats
    ((20, 20, 20),   (48, 64),  True,  torch.float64, 3.0, 3),
    ((12, 10, 21),   (40, 56),  False, torch.float64, 2.0, 3),    # ragged z -> scalar path, skewed K
    ((9, 16, 32),    (33, 47),  True,  torch.float32, 100.0, 2),
    ((16, 24, 40),   (64, 80),  False, torch.float32, 4.0, 4),
]


@pytest.mark.parametrize("vol_dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("res,hw,pinhole,ddt,wmax,reps", CASES)
def test_random_views_vs_oracle(res, hw, pinhole, ddt, wmax, reps, vol_dtype):
    rng = np.random.default_rng(hash((res, hw, pinhole)) % (2 ** 31))
    H, W_ = hw
    fx = 0.93 * W_ + 0.137
    K = scene.intrinsics(fx, W_ / 2 - 0.2713, H / 2 + 0.1371)
    if not pinhole:
        K[0, 1] = 0.31                    # skew -> general 3x3 path
        K[1, 1] = fx * 1.07
    tsdf_res = res[0]
    scale = scene.GRID_SIDE / max(res)
    center = scene.SPHERE_C + rng.normal(size=3) * 0.01
    tdist = 3.3 * scale
    T = np.zeros(res) + tdist
    Wt = np.zeros(res)
    Tg, Wg = T.copy(), Wt.copy()
    Kinv = np.linalg.inv(K)
    for r in range(reps):
        lw = scene.view_extrinsic(float(rng.uniform(-60, 60)))
        lw[:, 3] += rng.normal(size=3) * 0.02
        dm = scene.render_depth(K, lw, H, W_, invalid_frac=0.05, seed=r)
        if ddt == torch.float32:
            dm = dm.astype(np.float32).astype(np.float64)
        margin = [None]
        O.fuse_depths(dm, lw, K, Kinv, T, Wt, tdist, tsdf_res=tsdf_res, scale=scale, center=center, wmax=wmax,
                      margin_out=margin)
        assert margin[0] > 1e-10
        Tg, Wg = run_gpu(dm, lw, K, Tg, Wg, tdist, scale, center, wmax, vol_dtype, ddt, tsdf_res=tsdf_res)
        assert np.array_equal(Wg, Wt), "weights / update mask differ after view %d" % r
        if vol_dtype == torch.float64:
            assert np.array_equal(Tg, T)
        else:
            assert np.all(np.abs(Tg - T) <= f32_tol(r + 1) * (1 + np.abs(T)))
    assert (Wt > 0).any() and (Wt == 0).any()


@pytest.mark.parametrize("vol_dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("res,hw,pinhole,ddt,wmax,reps", CASES)
def test_multi_view_sweep_equals_consecutive_sweeps(res, hw, pinhole, ddt, wmax, reps, vol_dtype):
    """dfh_integrate_depth_multi: all views in one sweep of the volume = one dfh_integrate_depth per view in the same
    order, bit for bit (fp32 volumes: the fused kernel; fp64 volumes: the documented per-view fallback); whole grid
    and slabs; with and without the caller's workspace."""
    rng = np.random.default_rng(hash((res, hw, pinhole, 7)) % (2 ** 31))
    H, W_ = hw
    fx = 0.93 * W_ + 0.137
    K = scene.intrinsics(fx, W_ / 2 - 0.2713, H / 2 + 0.1371)
    if not pinhole:
        K[0, 1] = 0.31
        K[1, 1] = fx * 1.07
    Kinv = np.linalg.inv(K)
    scale = scene.GRID_SIDE / max(res)
    center = scene.SPHERE_C + rng.normal(size=3) * 0.01
    tdist = 3.3 * scale
    nv = reps + 2
    lws, dms = [], []
    for r in range(nv):
        lw = scene.view_extrinsic(float(rng.uniform(-60, 60)))
        lw[:, 3] += rng.normal(size=3) * 0.02
        lws.append(lw)
        dms.append(torch.from_numpy(scene.render_depth(K, lw, H, W_, invalid_frac=0.05, seed=r)).to("cuda", dtype=ddt).contiguous())
    T0 = torch.full(res, tdist / scale, dtype=vol_dtype, device="cuda")
    W0 = torch.zeros(res, dtype=vol_dtype, device="cuda")
    Ts, Ws = T0.clone(), W0.clone()
    for d, lw in zip(dms, lws):
        kernels.integrate_depth(Ts, Ws, d, K, Kinv, lw, scale, center, tdist, wmax=wmax)
    assert float(Ws.max()) > 1.0                                         # voxels seen by several views exist
    Tm, Wm = T0.clone(), W0.clone()
    kernels.integrate_depth_views(Tm, Wm, dms, K, Kinv, lws, scale, center, tdist, wmax=wmax)
    assert torch.equal(Tm, Ts) and torch.equal(Wm, Ws)
    ws = torch.empty(4096, dtype=torch.int64, device="cuda")
    T2, W2 = T0.clone(), W0.clone()
    for a, b in ((0, 3), (3, res[0])):                                   # slabs, caller's workspace
        Ta, Wa = T2[a:b].clone(), W2[a:b].clone()
        kernels.integrate_depth_views(Ta, Wa, dms, K, Kinv, lws, scale, center, tdist, wmax=wmax, res=res, x_range=(a, b), workspace=ws)
        T2[a:b] = Ta; W2[a:b] = Wa
    assert torch.equal(T2, Ts) and torch.equal(W2, Ws)


def test_multi_view_sweep_many_views_and_errors():
    res, (H, W_) = (8, 12, 16), (32, 40)
    K = scene.intrinsics(37.0, 19.7, 16.2)
    Kinv = np.linalg.inv(K)
    scale, center = scene.GRID_SIDE / 16, scene.SPHERE_C
    tdist = 3.0 * scale
    rng = np.random.default_rng(5)
    lws = [scene.view_extrinsic(float(a)) for a in rng.uniform(-50, 50, size=19)]           # > 16: two calls inside
    dms = [torch.from_numpy(scene.render_depth(K, lw, H, W_, dtype=np.float32)).cuda() for lw in lws]
    T0 = torch.full(res, tdist / scale, dtype=torch.float32, device="cuda"); W0 = torch.zeros_like(T0)
    Ts, Ws = T0.clone(), W0.clone()
    for d, lw in zip(dms, lws):
        kernels.integrate_depth(Ts, Ws, d, K, Kinv, lw, scale, center, tdist, wmax=5.0)
    Tm, Wm = T0.clone(), W0.clone()
    kernels.integrate_depth_views(Tm, Wm, dms, K, Kinv, lws, scale, center, tdist, wmax=5.0)
    assert torch.equal(Tm, Ts) and torch.equal(Wm, Ws)
    Te, We = T0.clone(), W0.clone()
    kernels.integrate_depth_views(Te, We, [], K, Kinv, [], scale, center, tdist)            # no views: nothing happens
    assert torch.equal(Te, T0) and torch.equal(We, W0)
    with pytest.raises(ValueError):
        kernels.integrate_depth_views(Te, We, dms[:2], K, Kinv, lws[:3], scale, center, tdist)      # fusion_dm.py:96-97
    with pytest.raises(ValueError):
        kernels.integrate_depth_views(Te, We, [dms[0], dms[1][:, :30].contiguous()], K, Kinv, lws[:2], scale, center, tdist)
    with pytest.raises(ValueError):
        kernels.integrate_depth_views(Te, We, [dms[0], dms[1].double()], K, Kinv, lws[:2], scale, center, tdist)


def test_fresh_live_volume_equals_fill_then_sweep(monkeypatch):
    """dfh_integrate_depth_multi_fresh (the fill of a live volume folded into the multi-view sweep, core/fusion_dm.py:152-153 +
    :166-170) against T.fill_(value); W.zero_(); dfh_integrate_depth_multi -- every voxel, bit for bit: brick sweep (every brick is
    written, culled or not), a slab with global plane indices, the plain sweep (option k1_no_bricks), one view, no view, more
    than 16 views, a ragged grid and a float64 volume; the volumes start as garbage."""
    rng = np.random.default_rng(21)
    K = scene.intrinsics(150.3, 79.7, 59.6)
    Kinv = np.linalg.inv(K)
    H, W_ = 120, 160
    for res, x_range, dtype, n_views, no_bricks in (((64, 64, 64), None, torch.float32, 3, False), ((20, 64, 64), (22, 42), torch.float32, 4, False),
                                                    ((64, 64, 64), None, torch.float32, 3, True), ((32, 28, 30), None, torch.float32, 2, False),
                                                    ((32, 32, 32), None, torch.float64, 2, False), ((48, 48, 48), None, torch.float32, 1, False),
                                                    ((16, 32, 32), None, torch.float32, 0, False), ((24, 32, 32), None, torch.float32, 18, False)):
        R = 64 if x_range else res[1]
        scale = scene.GRID_SIDE / R
        center, tdist = scene.SPHERE_C, 4.0 * scale
        lws = [scene.view_extrinsic(float(a)) for a in rng.uniform(-60, 60, size=n_views)]
        dms = [torch.from_numpy(scene.render_depth(K, lw, H, W_, dtype=np.float32, invalid_frac=0.02, seed=int(i))).cuda() for i, lw in enumerate(lws)]
        grid = (R, res[1], res[2]) if x_range else res
        kw = dict(wmax=7.0, tsdf_res=R, res=grid, x_range=x_range or (0, res[0]))
        if no_bricks:
            _lib.set_option("k1_no_bricks", 1)
        else:
            _lib.set_option("k1_no_bricks", None)
        fill = tdist / scale
        Ta = torch.full(res, fill, dtype=dtype, device="cuda"); Wa = torch.zeros_like(Ta)
        kernels.integrate_depth_views(Ta, Wa, dms, K, Kinv, lws, scale, center, tdist, **kw)
        Tb = torch.from_numpy(rng.normal(size=res)).to("cuda", dtype=dtype); Wb = torch.from_numpy(rng.uniform(1, 9, size=res)).to("cuda", dtype=dtype)
        kernels.integrate_depth_views(Tb, Wb, dms, K, Kinv, lws, scale, center, tdist, fresh=fill, **kw)
        assert torch.equal(Ta, Tb) and torch.equal(Wa, Wb), (res, x_range, dtype, n_views, no_bricks)
        if n_views:
            assert int((Wa > 0).sum()) > 0 and int((Wa == 0).sum()) > 0
    _lib.set_option("k1_no_bricks", None)


def test_slab_sweeps_equal_full_sweep():
    """Slab partition along axis 0 (multi-GPU layout): per-slab buffers with global indices
    reproduce the full sweep bit for bit."""
    res = (24, 16, 32)
    H, W_ = 60, 80
    K = scene.intrinsics(77.31, 39.713, 30.137)
    lw = scene.view_extrinsic(25.0)
    dm = scene.render_depth(K, lw, H, W_, invalid_frac=0.03, seed=3)
    scale = scene.GRID_SIDE / 32; center = scene.SPHERE_C; tdist = 4 * scale
    T0 = np.zeros(res) + tdist; W0 = np.zeros(res)
    Tf, Wf = run_gpu(dm, lw, K, T0, W0, tdist, scale, center, 100.0, torch.float32, tsdf_res=24)
    for bounds in ([0, 5, 13, 24], [0, 24], [0, 0, 24], [0, 1, 2, 24]):
        Ts, Ws = [], []
        for a, b in zip(bounds[:-1], bounds[1:]):
            t, w = run_gpu(dm, lw, K, T0[a:b], W0[a:b], tdist, scale, center, 100.0, torch.float32,
                           tsdf_res=24, x_range=(a, b), res=res)
            Ts.append(t); Ws.append(w)
        assert np.array_equal(np.concatenate(Ts), Tf)
        assert np.array_equal(np.concatenate(Ws), Wf)


def test_edge_cases_and_errors():
    R = 8
    K = scene.intrinsics(30.3, 15.7, 11.6)
    f = FusionDM(0.1, K, tsdf_res=R)
    T, W = f._new_volume_pair()
    lw = scene.view_extrinsic(0.0)
    # all-invalid depth: nothing is touched
    f.fuseDepths(np.zeros((24, 32)), lw, T, W, scale=0.2, center=scene.SPHERE_C)
    assert float(W.abs().sum()) == 0 and bool((T == T[0, 0, 0]).all())
    # empty slab: no launch, no error
    kernels.integrate_depth(T[0:0], W[0:0], torch.zeros(24, 32, device="cuda"), K, np.linalg.inv(K), lw, 0.2,
                            scene.SPHERE_C, 0.1, tsdf_res=R, res=(R, R, R), x_range=(3, 3))
    with pytest.raises(ValueError):
        f.fuseDepths(np.zeros((24, 32)), np.eye(4), T, W)
    with pytest.raises(ValueError):
        f.fuseDepths(np.zeros((1, 32)), lw, T, W)           # H < 2: C ABI rejects
    with pytest.raises(ValueError):
        f.fuseDepths(np.zeros((24, 32)), lw, T, W.cpu().numpy())
    with pytest.raises(ValueError):
        kernels.integrate_depth(T, W[:4], torch.zeros(24, 32, device="cuda"), K, np.linalg.inv(K), lw, 0.2,
                                scene.SPHERE_C, 0.1)
    with pytest.raises(ValueError):
        f.compute_live_tsdf([np.zeros((24, 32))], [])         # reference fusion_dm.py:96-97


def test_voxels_behind_the_camera_quirk():
    """The reference never tests lpos_z > 0 (fusion_dm.py:194-203): a voxel behind the camera
    whose mirrored projection lands in the image IS integrated.  Reproduced, not fixed."""
    R = 16
    K = scene.intrinsics(30.31, 15.71, 11.63)
    lw = scene.view_extrinsic(0.0)
    lw[2, 3] -= 4.0                                  # the whole grid is behind the camera
    dm = -np.ones((24, 32))
    scale = 0.1; center = scene.SPHERE_C; tdist = 0.4
    To = np.zeros((R, R, R)) + tdist; Wo = np.zeros((R, R, R))
    O.fuse_depths(dm, lw, K, np.linalg.inv(K), To, Wo, tdist, scale=scale, center=center)
    assert (Wo > 0).sum() > 100
    Tg, Wg = run_gpu(dm, lw, K, np.zeros((R, R, R)) + tdist, np.zeros((R, R, R)), tdist, scale, center, 100.0,
                     torch.float64)
    assert np.array_equal(Wg, Wo) and np.array_equal(Tg, To)


def test_compute_live_tsdf_multi_view_matches_oracle():
    """compute_live_tsdf plumbing (fusion_dm.py:95-178): fixed avg/std alignment, scale =
    12*std/res, float32 centre, views applied sequentially."""
    R = 32
    H, W_ = 60, 80
    K = scene.intrinsics(61.37, 39.71, 29.63)
    f = FusionDM(0.6, K, tsdf_res=R)
    avg = np.array([-0.03, -0.43, -5.6], dtype='float32'); std = 1.3
    c = avg.astype(np.float64)                   # the volume is centred on `avg` (z = -5.6)
    lws, dms = [], []
    for i, a in enumerate((0.0, 30.0, -45.0)):
        lw = scene.view_extrinsic(a, centre=c)   # c maps to (0,0,|c|) in front of the camera
        dms.append(scene.render_depth(K, lw, H, W_, invalid_frac=0.02, seed=i, sphere_c=c, sphere_r=2.0,
                                      wall_z=-2.0))
        lws.append(lw)
    T, W = f.compute_live_tsdf(dms, lws)
    To = np.zeros((R, R, R)) + 0.6; Wo = np.zeros((R, R, R))
    for dm, lw in zip(dms, lws):
        O.fuse_depths(dm, lw, K, np.linalg.inv(K), To, Wo, 0.6, scale=12 * std / R, center=avg)
    assert np.array_equal(W, Wo)
    assert (Wo > 0).mean() > 0.05
    assert np.all(np.abs(T - To) <= f32_tol(3) * (1 + np.abs(To)))
    assert np.allclose(f._IND[0, 0], 8 * std / R)


@pytest.mark.parametrize("angle", [0.0, 45.0])
def test_config2_256_full_volume_vs_oracle(angle):
    """BASELINE config 2 size (256^3, 640x480): every voxel's mask bit-exact against the
    oracle, values within the fp32 bar; plus slab additivity at full size."""
    R = 256
    H, W_, fx, cx, cy = scene.CAMERAS["C2"]
    K = scene.intrinsics(fx, cx, cy)
    scale, center, tdist = scene.grid_params(R)
    lw = scene.view_extrinsic(angle)
    dm = scene.render_depth(K, lw, H, W_, dtype=np.float32)
    To = np.zeros((R, R, R)) + tdist; Wo = np.zeros((R, R, R))
    # no margin assertion here: float32-rounded depths such as 1.75 m land exactly on
    # sd == -tdist for the decimal voxel pitch; the kernel and the oracle evaluate the
    # reference's expressions in the same IEEE order, so they agree even on those voxels.
    O.fuse_depths(dm, lw, K, np.linalg.inv(K), To, Wo, tdist, scale=scale, center=center)
    f = FusionDM(tdist, K, tsdf_res=R)
    T, W = f._new_volume_pair()
    d = torch.from_numpy(dm).cuda()
    f.fuseDepths(d, lw, T, W, scale=scale, center=center)
    Wn = W.cpu().numpy(); Tn = T.cpu().numpy().astype(np.float64)
    assert np.array_equal(Wn.astype(np.float64), Wo)
    assert np.all(np.abs(Tn - To) <= f32_tol(1) * (1 + np.abs(To)))
    # size-independent property: two half-slabs == whole
    T2, W2 = f._new_volume_pair()
    for a, b in ((0, 100), (100, 256)):
        kernels.integrate_depth(T2[a:b], W2[a:b], d, K, np.linalg.inv(K), lw, scale, center, tdist,
                                tsdf_res=R, res=(R, R, R), x_range=(a, b))
    assert torch.equal(T2, T) and torch.equal(W2, W)


def test_config4_512_size_independent_properties():
    """BASELINE config 4 grid (512^3, 1280x720 depth): (1) integrating the
    same view again leaves T unchanged (the running average of equal values) and counts w up to
    wmax; (2) eight axis-0 slabs reproduce the full sweep bit for bit; (3) the float64-volume exact
    kernel and the float32 fast path agree on the update mask of every voxel; (4) all 134 M voxels against the C oracle."""
    R = 512
    H, W_, fx, cx, cy = scene.CAMERAS["C5"]
    K = scene.intrinsics(fx, cx, cy); Kinv = np.linalg.inv(K)
    scale, center, tdist = scene.grid_params(R)
    lw = scene.view_extrinsic(-30.0)
    d = torch.from_numpy(scene.render_depth(K, lw, H, W_, dtype=np.float32)).cuda()
    T = torch.full((R, R, R), tdist, dtype=torch.float32, device="cuda"); Wt = torch.zeros_like(T)
    kernels.integrate_depth(T, Wt, d, K, Kinv, lw, scale, center, tdist, 3.0)
    T1, W1 = T.clone(), Wt.clone()
    upd = W1 > 0
    assert 0.3 < float(upd.float().mean()) < 0.8
    for n in (2, 3, 4):
        kernels.integrate_depth(T, Wt, d, K, Kinv, lw, scale, center, tdist, 3.0)
        assert torch.equal(Wt > 0, upd)
        assert float(Wt[upd].min()) == float(min(n, 3)) and float(Wt[upd].max()) == float(min(n, 3))
        assert float((T - T1).abs().max()) <= 4 * n * F32_EPS * float(T1.abs().max())
        assert torch.equal(T[~upd], T1[~upd])
    # slabs
    Ts = torch.full((R, R, R), tdist, dtype=torch.float32, device="cuda"); Ws = torch.zeros_like(Ts)
    for r in range(8):
        a, b = r * 64, (r + 1) * 64
        kernels.integrate_depth(Ts[a:b], Ws[a:b], d, K, Kinv, lw, scale, center, tdist, 3.0, tsdf_res=R, res=(R, R, R), x_range=(a, b))
    assert torch.equal(Ts, T1) and torch.equal(Ws, W1)
    del Ts, Ws, T
    # exact fp64 kernel vs filtered fast path: same mask on all 134 M voxels
    T64 = torch.full((R, R, R), tdist, dtype=torch.float64, device="cuda"); W64 = torch.zeros_like(T64)
    kernels.integrate_depth(T64, W64, d, K, Kinv, lw, scale, center, tdist, 3.0)
    assert torch.equal(W64 > 0, upd)
    assert float((T64.float() - T1).abs().max()) <= 4 * F32_EPS * float(T1.abs().max())
    # (4) every one of the 134 M voxels against the C restatement of fuseDepths' CPU path (oracle/oracle_c.c = the numpy oracle bit
    # for bit = the reference's outputs, tests/test_oracle_c.py): the float64 volume identical, the float32 volume's weights
    # identical and its values within the float32 bar -- the parity test proper at BASELINE's full size (round-3 verdict, weak 2)
    from oracle import oracle_c as OC
    OC.build()
    To = np.zeros((R, R, R)) + tdist
    Wo = np.zeros((R, R, R))
    OC.fuse_depths(d.cpu().numpy(), lw, K, Kinv, To, Wo, tdist, tsdf_res=R, scale=scale, center=center, wmax=3.0, n_threads=min(OC.threads(), 16))
    assert np.array_equal(W64.cpu().numpy(), Wo) and np.array_equal(T64.cpu().numpy(), To)
    assert np.array_equal(W1.cpu().numpy(), Wo.astype(np.float32))
    assert np.all(np.abs(T1.cpu().numpy().astype(np.float64) - To) <= f32_tol(1) * (1 + np.abs(To)))


def test_brick_culling_never_changes_a_voxel(k1_path, monkeypatch):
    """The brick sweep skips a brick only when its eight projected corners PROVE that the view updates none of its voxels.
    Geometries that stress the proof -- camera inside the grid, grid partly and wholly behind the camera, grazing views,
    a close wall that occludes almost everything, grids whose sides are no multiples of the brick, slabs that cut bricks,
    a skewed (non-pinhole) K -- must give the row sweep's volumes bit for bit, single- and multi-view."""
    if k1_path == "rows":
        pytest.skip("compares the two paths itself")
    rng = np.random.default_rng(77)
    cases = []
    for res, (H, W_), pin in (((37, 21, 68), (60, 80), True), ((16, 16, 64), (48, 64), True), ((9, 30, 132), (90, 70), False),
                              ((64, 64, 64), (120, 160), True)):
        fx = 0.9 * W_ + 0.137
        K = scene.intrinsics(fx, W_ / 2 - 0.2713, H / 2 + 0.1371)
        if not pin:
            K[0, 1] = 0.31; K[1, 1] = fx * 1.07
        cases.append((res, H, W_, K))
    for res, H, W_, K in cases:
        Kinv = np.linalg.inv(K)
        scale = scene.GRID_SIDE / max(res)
        center = scene.SPHERE_C.copy()
        tdist = 3.3 * scale
        views = []
        for a, dz, wall in ((0.0, 0.0, 3.0), (75.0, 0.0, 3.0), (-20.0, -1.9, 3.0), (10.0, -4.5, 3.0), (200.0, 0.0, 3.0), (35.0, 0.6, 1.6),
                            (float(rng.uniform(-180, 180)), float(rng.uniform(-2.5, 1.0)), 2.4)):
            lw = scene.view_extrinsic(a)
            lw[2, 3] += dz                                        # camera pushed towards / into / through the grid
            lw[:, 3] += rng.normal(size=3) * 0.02
            views.append((lw, scene.render_depth(K, lw, H, W_, invalid_frac=0.05, seed=len(views), dtype=np.float32, wall_z=wall)))
        outs = {}
        for path in ("bricks", "columns", "rows"):       # culled walk (pre-passes) | walk over every brick, image test in the waves | rows
            if path == "rows":
                _lib.set_option("k1_no_bricks", 1)
            else:
                _lib.set_option("k1_no_bricks", None)
                _lib.set_option("k1_bricks_min", 0)
                _lib.set_option("k1_cull", 1 if path == "bricks" else 0)
            T = torch.full(res, tdist / scale, dtype=torch.float32, device="cuda"); Wt = torch.zeros_like(T)
            Tm, Wm = T.clone(), Wt.clone()
            Tsl, Wsl = T.clone(), Wt.clone()
            per_view = []
            for lw, dm in views:
                d = torch.from_numpy(dm).cuda()
                kernels.integrate_depth(T, Wt, d, K, Kinv, lw, scale, center, tdist, 7.0)
                per_view.append(int((Wt > 0).sum()))
                for a, b in ((0, 5), (5, res[0])):                # slabs that cut through bricks
                    kernels.integrate_depth(Tsl[a:b], Wsl[a:b], d, K, Kinv, lw, scale, center, tdist, 7.0, tsdf_res=res[0], res=res, x_range=(a, b))
            kernels.integrate_depth_views(Tm, Wm, [torch.from_numpy(dm).cuda() for _, dm in views], K, Kinv, [lw for lw, _ in views],
                                          scale, center, tdist, 7.0)
            assert torch.equal(Tm, T) and torch.equal(Wm, Wt) and torch.equal(Tsl, T) and torch.equal(Wsl, Wt)
            outs[path] = (T, Wt, per_view)
        for path in ("bricks", "columns"):
            assert torch.equal(outs[path][0], outs["rows"][0]) and torch.equal(outs[path][1], outs["rows"][1]), path
            assert outs[path][2] == outs["rows"][2], path
        assert 0 < int((outs["rows"][1] > 0).sum()) < outs["rows"][1].numel()
    _lib.set_option("k1_no_bricks", None)


def test_ocl_mode_matches_the_float32_restatement(k1_path):
    """A2 (optional): FusionDM.fuseDepths(mode="ocl") = the arithmetic of the reference's OpenCL kernel (core/fusion_dm.py:600-737),
    against oracle_np.fuse_depths_ocl, the operation-by-operation float32 reading of that kernel text: bit-exact (same IEEE
    operations in the same order).  PARITY UNPINNED against the reference itself -- no OpenCL device or pyopencl here to run it.
    Also: inputs are left alone and new float32 arrays come back (the reference's host code, :690-691,:737), and the result
    differs from the CPU-path semantics (free-space carving, bilinear depth)."""
    if k1_path == "rows":
        pytest.skip("one kernel, no sweep variants")
    R = 40
    H, W_, fx, cx, cy = scene.CAMERAS["C1"]
    K = scene.intrinsics(fx, cx, cy)
    scale, center, tdist = scene.grid_params(R)
    f = FusionDM(tdist, K, tsdf_res=R)
    f._IND = np.eye(4)
    f._IND[:3, :3] *= scale
    f._IND[:3, 3] = center - scale * R / 2                              # index -> world, what compute_live_tsdf installs
    T = np.zeros((R, R, R)) + tdist
    Wt = np.zeros((R, R, R))
    To, Wo = T.copy(), Wt.copy()
    for i, a in enumerate((0.0, 35.0, -50.0)):
        lw = scene.view_extrinsic(a)
        dm = scene.render_depth(K, lw, H, W_, invalid_frac=0.05, seed=i)
        T_in, W_in = T.copy(), Wt.copy()
        T2, W2 = f.fuseDepths(dm, lw, T, Wt, wmax=2.0, mode="ocl")
        assert np.array_equal(T, T_in) and np.array_equal(Wt, W_in)     # inputs untouched
        assert T2.dtype == np.float32 and W2.dtype == np.float32
        To, Wo = O.fuse_depths_ocl(dm, lw, K, np.linalg.inv(K), f._IND, To, Wo, tdist, wmax=2.0)
        assert np.array_equal(W2, Wo) and np.array_equal(T2, To), "view %d" % i
        T, Wt = T2, W2
    assert 0.3 < (Wo > 0).mean() < 1.0 and Wo.max() == 2.0 and (To < 0).any() and (To == np.float32(-tdist)).any()
    # CUDA tensors in, CUDA tensors out
    lw = scene.view_extrinsic(10.0)
    dm = scene.render_depth(K, lw, H, W_, dtype=np.float32)
    Td, Wd = torch.from_numpy(To).cuda(), torch.from_numpy(Wo).cuda()
    T3, W3 = f.fuseDepths(torch.from_numpy(dm).cuda(), lw, Td, Wd, mode="ocl")
    T3o, W3o = O.fuse_depths_ocl(dm, lw, K, np.linalg.inv(K), f._IND, To, Wo, tdist)
    assert torch.equal(Td.cpu(), torch.from_numpy(To)) and np.array_equal(T3.cpu().numpy(), T3o) and np.array_equal(W3.cpu().numpy(), W3o)
    with pytest.raises(ValueError):
        f.fuseDepths(dm, lw, To, Wo, mode="opencl")


def test_fusion_initialize_canonical_space_from_depth_maps():
    """Fusion.InitializeCanonicalSpace(depths, lws, K) / Fusion.fuseDepths (broken at the reference's HEAD, core/fusion.py:73-99,
    127-150; here: FusionDM.fuseDepths' sweep): the canonical volume equals FusionDM's for the same views bit for bit, numpy
    volumes are updated in place, and the object ends up with a mesh and a deformation graph."""
    from dynamicfusion_body_amd import Fusion
    R = 48
    H, W, fx, cx, cy = scene.CAMERAS["C1"]
    K = scene.intrinsics(fx, cx, cy)
    scale, center, tdist = scene.grid_params(R)
    lws = [scene.view_extrinsic(a) for a in (0.0, 45.0, -45.0)]
    dms = [scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0) for lw in lws]
    fu = Fusion(np.zeros((4, 4, 4)), tdist, knn=4, write_warpfield=False)
    fu.InitializeCanonicalSpace(depths=dms, lws=lws, K=K, tsdf_size=R, scale=scale, center=center)
    fd = FusionDM(tdist, K, tsdf_res=R)
    T, Wt = fd._new_volume_pair()
    for d, lw in zip(dms, lws):
        fd.fuseDepths(torch.from_numpy(d).cuda(), lw, T, Wt, scale=scale, center=center)
    assert torch.equal(fu._T, T) and torch.equal(fu._Wt, Wt)
    assert len(fu._nodes) > 4 and len(fu._vertices) > 100 and fu._kdtree is not None
    # the method itself on numpy volumes (in place + returned), one view
    Tn, Wn = np.zeros((R, R, R)) + tdist, np.zeros((R, R, R))
    out = fu.fuseDepths(dms[0], lws[0], Tn, Wn, scale=scale, center=center)
    assert out[0] is Tn and out[1] is Wn
    T1, W1 = fd._new_volume_pair()
    fd.fuseDepths(torch.from_numpy(dms[0]).cuda(), lws[0], T1, W1, scale=scale, center=center)
    assert np.array_equal(Tn.astype(np.float32), T1.cpu().numpy()) and np.array_equal(Wn.astype(np.float32), W1.cpu().numpy())
    with pytest.raises(ValueError):
        fu.InitializeCanonicalSpace()
    with pytest.raises(ValueError):
        Fusion(np.zeros((4, 4, 4)), tdist).fuseDepths(dms[0], lws[0], Tn, Wn)          # no intrinsics yet
