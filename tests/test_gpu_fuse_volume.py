"""GPU parity: K2 (dfh_fuse_volume_rigid = FusionDM.updateTSDF) and K3 (dfh_fuse_volume_dqb =
Fusion.updateTSDF) through the C ABI, against the reference's golden outputs
(tests/golden/g3_rigid.npz, g4_dqb.npz) and the fp64 oracle.

Bars: update masks identical; integer weights identical; fp64 volumes: K2 bit-exact
(only + - * / in the reference's order), K3 <= 1e-12 (exp() of the blend weights differs in
the last ulp between libm and the device library); fp32 volumes: |dT| <= 2*n*eps32*(1+|T|).
"""
import numpy as np
import pytest
import torch

from oracle import oracle_np as O
from dynamicfusion_body_amd import Fusion, FusionDM, kernels, scene

pytestmark = pytest.mark.gpu

F32_EPS = float(np.finfo(np.float32).eps)


def f32_tol(n):
    return 2.0 * n * F32_EPS


def dev(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dtype=dtype)


def sphere_volume(R, centre, radius, tdist):
    shape = (R, R, R) if np.isscalar(R) else R
    g = np.stack(np.meshgrid(*[np.arange(s, dtype=np.float64) for s in shape], indexing="ij"), axis=-1)
    return np.clip(np.linalg.norm(g - centre, axis=-1) - radius, -tdist * 1.5, tdist * 1.5)


def small_dq(rng, rot=0.05, trans=0.5, scale=1.0):
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
    ang = rng.normal() * rot
    q = np.append(np.cos(ang / 2), np.sin(ang / 2) * ax)
    t = rng.normal(size=3) * trans
    qe = 0.5 * O.quaternion_multiply(np.array([0.0, t[0], t[1], t[2]]), q)
    return np.append(q, qe) * scale


# ------------------------------------------------------------------------------ K2
@pytest.mark.parametrize("vol_dtype", [np.float64, np.float32])
def test_g3_rigid_golden(golden, vol_dtype):
    """R=20, non-unit `_lw`, 4 live volumes, wmax=5: expected values are the reference's."""
    g = golden("g3_rigid")
    f = FusionDM(float(g["tdist"]), np.eye(3), tsdf_res=20, volume_dtype=vol_dtype)
    f._tsdf = g["T0"]; f._tsdfw = g["W0"]; f._lw = g["lw"]
    for r in range(4):
        f.updateTSDF(g["lives"][r], wmax=float(g["wmax"]))
        if r == 0:
            assert np.array_equal(f._tsdfw, g["W_after1"])
            tol = 1e-13 if vol_dtype == np.float64 else f32_tol(1)
            assert np.all(np.abs(f._tsdf - g["T_after1"]) <= tol * (1 + np.abs(g["T_after1"])))
    assert np.array_equal(f._tsdfw, g["W_after4"])
    tol = 1e-13 if vol_dtype == np.float64 else f32_tol(4)
    assert np.all(np.abs(f._tsdf - g["T_after4"]) <= tol * (1 + np.abs(g["T_after4"])))


@pytest.mark.parametrize("res,live_res", [((16, 12, 21), (16, 12, 21)), ((12, 12, 16), (14, 10, 18)), ((8, 8, 32), (8, 8, 32))])
def test_rigid_vs_oracle_bit_exact(res, live_res):
    rng = np.random.default_rng(sum(res) * 7 + sum(live_res))
    tdist = 1.7
    T0 = sphere_volume(res, np.array(res) / 2.0 + 0.3, min(res) / 3.0, tdist)
    W0 = (rng.random(res) < 0.5) * rng.integers(1, 5, size=res).astype(np.float64)
    for trial, lw in enumerate([np.array([1.0, 0, 0, 0, 0, 0, 0, 0]), small_dq(rng, 0.1, 0.7, 1.02), small_dq(rng, 0.3, 1.5, 0.97)]):
        live = sphere_volume(live_res, np.array(live_res) / 2.0 - 0.4, min(live_res) / 3.2, tdist) + 0.01 * rng.normal(size=live_res)
        To, Wo = T0.copy(), W0.copy()
        O.update_tsdf_rigid(To, Wo, live, lw, tdist, wmax=4.0)
        T, W = dev(T0, torch.float64), dev(W0, torch.float64)
        kernels.fuse_volume_rigid(T, W, dev(live, torch.float64), lw, tdist, 4.0)
        assert np.array_equal(W.cpu().numpy(), Wo), trial
        assert np.array_equal(T.cpu().numpy(), To), trial
        # float32 volumes + float32 live: masks still identical when the live values are f32-exact
        live32 = live.astype(np.float32).astype(np.float64)
        To, Wo = T0.astype(np.float32).astype(np.float64), W0.copy()
        O.update_tsdf_rigid(To, Wo, live32, lw, tdist, wmax=4.0)
        T, W = dev(T0, torch.float32), dev(W0, torch.float32)
        kernels.fuse_volume_rigid(T, W, dev(live32, torch.float32), lw, tdist, 4.0)
        assert np.array_equal(W.cpu().numpy().astype(np.float64), Wo)
        assert np.all(np.abs(T.cpu().numpy().astype(np.float64) - To) <= f32_tol(1) * (1 + np.abs(To)))


def test_rigid_identity_structural_ties():
    """lw = identity samples exactly on voxel centres; live values exactly equal to -tdist must
    NOT update (s > -tdist is strict) -- only exact arithmetic gets this right."""
    R = 12
    tdist = 1.0
    live = np.full((R, R, R), -tdist)
    live[:, :, ::2] = -tdist + 1e-9
    live[0] = tdist * 3
    T0 = np.zeros((R, R, R)); W0 = np.ones((R, R, R))
    To, Wo = T0.copy(), W0.copy()
    O.update_tsdf_rigid(To, Wo, live, np.array([1.0, 0, 0, 0, 0, 0, 0, 0]), tdist)
    assert (Wo == 1).any() and (Wo == 2).any()
    T, W = dev(T0, torch.float64), dev(W0, torch.float64)
    kernels.fuse_volume_rigid(T, W, dev(live, torch.float64), [1, 0, 0, 0, 0, 0, 0, 0], tdist)
    assert np.array_equal(W.cpu().numpy(), Wo) and np.array_equal(T.cpu().numpy(), To)


def test_rigid_slabs_and_errors():
    rng = np.random.default_rng(5)
    res = (20, 8, 16)
    tdist = 1.5
    lw = small_dq(rng, 0.1, 0.5)
    live = dev(sphere_volume(res, np.array([9.0, 4.0, 8.0]), 3.0, tdist), torch.float32)
    T0 = sphere_volume(res, np.array([10.0, 4.2, 7.5]), 3.2, tdist); W0 = np.ones(res)
    Tf, Wf = dev(T0, torch.float32), dev(W0, torch.float32)
    kernels.fuse_volume_rigid(Tf, Wf, live, lw, tdist)
    Ts, Ws = dev(T0, torch.float32), dev(W0, torch.float32)
    for a, b in ((0, 7), (7, 7), (7, 20)):
        kernels.fuse_volume_rigid(Ts[a:b], Ws[a:b], live, lw, tdist, res=res, x_range=(a, b))
    assert torch.equal(Ts, Tf) and torch.equal(Ws, Wf)
    f = FusionDM(tdist, np.eye(3), tsdf_res=8)
    with pytest.raises(ValueError):
        f.updateTSDF(np.zeros((4, 4)))
    with pytest.raises(ValueError):
        f.updateTSDF([[1.0]])
    with pytest.raises(ValueError):
        kernels.fuse_volume_rigid(Tf, Wf[:3], live, lw, tdist)


@pytest.mark.parametrize("res", [(64, 64, 256), (40, 36, 252), (33, 30, 64)])
def test_rigid_non_temporal_and_strided_paths_give_the_same_bits(res):
    """K2's float32 kernel has four shapes: a lane's four voxels 64 apart along z (z rows of whole 256-voxel runs) or next to each
    other, T / w through the caches or non-temporal (option k2_nt; the library's own choice: slabs whose pair of volumes exceeds the
    Infinity Cache).  One expression, one set of bits -- whole grid and slabs."""
    from dynamicfusion_body_amd import _lib
    rng = np.random.default_rng(17)
    tdist = 4.0
    c = np.array(res) / 2.0
    live = dev(sphere_volume(res, c + np.array([0.4, -0.3, 0.6]), 0.3 * min(res), tdist), torch.float32)
    T0 = sphere_volume(res, c, 0.31 * min(res), tdist); W0 = rng.integers(0, 4, size=res).astype(np.float64)
    lw = small_dq(rng, 0.02, 0.4)
    ref = None
    for nt, no_strided in ((0, 0), (1, 0), (0, 1), (1, 1)):
        _lib.set_option("k2_nt", nt)
        _lib.set_option("k2_no_strided", no_strided)
        T, W = dev(T0, torch.float32), dev(W0, torch.float32)
        kernels.fuse_volume_rigid(T, W, live, lw, tdist)
        Ts, Ws = dev(T0, torch.float32), dev(W0, torch.float32)
        a = res[0] // 3
        for x0, x1 in ((0, a), (a, res[0])):
            kernels.fuse_volume_rigid(Ts[x0:x1], Ws[x0:x1], live, lw, tdist, res=res, x_range=(x0, x1))
        assert torch.equal(Ts, T) and torch.equal(Ws, W)
        if ref is None:
            ref = (T, W)
            assert int((W != dev(W0, torch.float32)).sum()) > 0.5 * W.numel()          # (the call does update most voxels)
        else:
            assert torch.equal(T, ref[0]) and torch.equal(W, ref[1]), (nt, no_strided)


# ------------------------------------------------------------------------------ K3
def make_fusion(g, vol_dtype):
    fu = Fusion(g["T0"], float(g["tdist"]), knn=int(g["knn"]), volume_dtype=vol_dtype)
    fu._tsdfw = g["W0"]
    fu._lw = g["lw"]
    fu._nodes = [(0, g["node_pos"][i], g["node_dq"][i], float(g["node_w"][i])) for i in range(len(g["node_pos"]))]
    return fu


@pytest.mark.parametrize("vol_dtype", [np.float64, np.float32])
def test_g4_dqb_golden(golden, vol_dtype):
    """R=14, N=24, k=4, non-unit `_lw`, 3 live volumes, wmax=9: the reference's outputs."""
    g = golden("g4_dqb")
    fu = make_fusion(g, vol_dtype)
    for r in range(3):
        fu.updateTSDF(g["lives"][r], wmax=float(g["wmax"]))
        if r == 0:
            W = fu._tsdfw
            W0_stored = g["W0"].astype(vol_dtype).astype(np.float64)               # what the volume held
            assert np.array_equal(W != W0_stored, g["W_after1"] != g["W0"])        # update mask
            tol = 1e-12 if vol_dtype == np.float64 else f32_tol(1)
            assert np.all(np.abs(W - g["W_after1"]) <= tol * (1 + np.abs(g["W_after1"])))
            assert np.all(np.abs(fu._tsdf - g["T_after1"]) <= tol * (1 + np.abs(g["T_after1"])))
    tol = 1e-12 if vol_dtype == np.float64 else f32_tol(3)
    assert np.all(np.abs(fu._tsdfw - g["W_after3"]) <= tol * (1 + np.abs(g["W_after3"])))
    assert np.all(np.abs(fu._tsdf - g["T_after3"]) <= tol * (1 + np.abs(g["T_after3"])))


def test_fusion_update_stored_neighbourhoods_follow_the_graph():
    """Fusion.updateTSDF keeps every voxel's node indices / blend weights between calls: new node DQs must reuse
    them, new node radii or positions must not (each call is checked against the oracle)."""
    rng = np.random.default_rng(23)
    res, N, k, tdist = (12, 10, 18), 14, 4, 2.0
    node_pos = rng.uniform(0, np.array(res) - 1, size=(N, 3))
    node_w = rng.uniform(2.0, 5.0, size=N)
    lw = small_dq(rng, 0.03, 0.2, 1.0)
    T0 = sphere_volume(res, np.array(res) / 2.0, min(res) / 3.0, tdist)
    fu = Fusion(T0, tdist, knn=k, volume_dtype=np.float64)
    fu._lw = lw
    To, Wo = T0.copy(), np.zeros(res)
    for step in range(4):
        node_dq = np.array([small_dq(rng, 0.06, 0.3, 1.0) for _ in range(N)])
        if step == 2:
            node_w = node_w * 1.3                          # same positions, other radii: the stored weights are stale
        if step == 3:
            node_pos = node_pos[::-1].copy()               # other positions: indices are stale as well
        fu._nodes = [(0, node_pos[i], node_dq[i], float(node_w[i])) for i in range(N)]
        live = sphere_volume(res, np.array(res) / 2.0 + 0.3 * step, min(res) / 3.1, tdist)
        ws_before = fu._workspace if hasattr(fu, "_workspace") else None
        fu.updateTSDF(live, wmax=20.0)
        if step == 1:
            assert fu._workspace is ws_before              # step 1 reused step 0's workspace ...
        if step in (2, 3):
            assert fu._workspace is not ws_before          # ... steps 2 and 3 rebuilt it
        O.update_tsdf_dqb(To, Wo, live, node_pos, node_dq, node_w, k, lw, tdist, wmax=20.0)
        assert np.abs(fu._tsdfw - Wo).max() <= 1e-12 and np.abs(fu._tsdf - To).max() <= 1e-12


@pytest.mark.parametrize("res,N,k", [((24, 20, 40), 60, 4), ((16, 16, 16), 9, 3), ((12, 28, 33), 420, 8), ((8, 8, 16), 5, 1)])
def test_dqb_vs_oracle(res, N, k):
    """Random graphs, ragged grids, knn 1..8; includes bricks far from every node (large candidate
    radius) and N > candidate capacity (brute-force fallback inside the brick)."""
    rng = np.random.default_rng(N * 31 + k)
    tdist = 2.0
    node_pos = rng.uniform(0, np.array(res) - 1, size=(N, 3))
    if N >= 100:                       # cluster: > kCap (256) candidates around one brick -> brute-force fallback
        node_pos[:320] = np.array(res) / 2.0 + rng.normal(size=(320, 3)) * 1.5
    node_dq = np.array([small_dq(rng, 0.08, 0.4, 1 + 0.02 * rng.normal()) for _ in range(N)])
    node_w = rng.uniform(2.0, 5.0, size=N)
    lw = small_dq(rng, 0.05, 0.3, 0.99)
    T0 = sphere_volume(res, np.array(res) / 2.0, min(res) / 3.0, tdist)
    W0 = (rng.random(res) < 0.6) * rng.uniform(0.5, 3.0, size=res)
    live = sphere_volume(res, np.array(res) / 2.0 + 0.4, min(res) / 3.1, tdist) + 0.01 * rng.normal(size=res)
    To, Wo = T0.copy(), W0.copy()
    _, _, mask = O.update_tsdf_dqb(To, Wo, live, node_pos, node_dq, node_w, k, lw, tdist, wmax=7.0, return_mask=True)
    T, W = dev(T0, torch.float64), dev(W0, torch.float64)
    kernels.fuse_volume_dqb(T, W, dev(live, torch.float64), node_pos, node_dq, node_w, k, lw, tdist, 7.0)
    Tn, Wn = T.cpu().numpy(), W.cpu().numpy()
    assert np.array_equal((Tn != T0) | (Wn != W0), mask)
    assert np.abs(Wn - Wo).max() <= 1e-12 and np.abs(Tn - To).max() <= 1e-12
    assert mask.any() and (~mask).any()
    # slabs + reused workspace
    T2, W2 = dev(T0, torch.float64), dev(W0, torch.float64)
    for a, b in ((0, 5), (5, res[0])):
        ws = kernels.dqb_workspace(res, (a, b))
        for rebuild in (True, False):
            Ts, Ws = dev(T0[a:b], torch.float64), dev(W0[a:b], torch.float64)
            kernels.fuse_volume_dqb(Ts, Ws, dev(live, torch.float64), node_pos, node_dq, node_w, k, lw, tdist, 7.0,
                                    res=res, x_range=(a, b), workspace=ws, rebuild_candidates=rebuild)
        T2[a:b] = Ts; W2[a:b] = Ws
    assert torch.equal(T2, T) and torch.equal(W2, W)
    # workspace that also keeps the voxels' node indices: the call after the rebuild skips the search and must give
    # the same bits -- also when the node DQs changed in between (the nearest nodes do not depend on them)
    T3, W3 = dev(T0, torch.float64), dev(W0, torch.float64)
    other_dq = np.array([small_dq(rng, 0.05, 0.2, 1.0) for _ in range(N)])
    for (a, b), level in (((0, 5), 1), ((5, res[0]), 2)):          # level 1: indices; level 2: indices + blend weights
        ws = kernels.dqb_workspace(res, (a, b), knn=k, n_nodes=N, level=level)
        assert ws.numel() * 4 >= kernels.dqb_workspace(res, (a, b)).numel() * 4 + (b - a) * res[1] * res[2] * (k * 2 + (level - 1) * 8 * (k + 1))
        Ts, Ws = dev(T0[a:b], torch.float64), dev(W0[a:b], torch.float64)
        kernels.fuse_volume_dqb(Ts, Ws, dev(live, torch.float64), node_pos, other_dq, node_w, k, lw, tdist, 7.0,
                                res=res, x_range=(a, b), workspace=ws, rebuild_candidates=True)
        Ts, Ws = dev(T0[a:b], torch.float64), dev(W0[a:b], torch.float64)
        kernels.fuse_volume_dqb(Ts, Ws, dev(live, torch.float64), node_pos, node_dq, node_w, k, lw, tdist, 7.0,
                                res=res, x_range=(a, b), workspace=ws, rebuild_candidates=False)
        T3[a:b] = Ts; W3[a:b] = Ws
    assert torch.equal(T3, T) and torch.equal(W3, W)


def test_dqb_zero_blend_and_errors():
    """All-zero node DQs -> |b|_8 == 0 -> identity blend (core/fusion.py:544-549)."""
    rng = np.random.default_rng(9)
    res = (8, 8, 16)
    tdist = 1.0
    node_pos = rng.uniform(0, 7, size=(6, 3)); node_dq = np.zeros((6, 8)); node_w = np.full(6, 3.0)
    lw = np.array([1.0, 0, 0, 0, 0, 0.1, 0, 0])
    T0 = rng.normal(size=res); W0 = np.zeros(res)
    live = rng.uniform(-0.5, 0.5, size=res)
    To, Wo = T0.copy(), W0.copy()
    O.update_tsdf_dqb(To, Wo, live, node_pos, node_dq, node_w, 4, lw, tdist)
    T, W = dev(T0, torch.float64), dev(W0, torch.float64)
    kernels.fuse_volume_dqb(T, W, dev(live, torch.float64), node_pos, node_dq, node_w, 4, lw, tdist)
    assert np.abs(W.cpu().numpy() - Wo).max() <= 1e-12 and np.abs(T.cpu().numpy() - To).max() <= 1e-12
    assert (Wo > 0).any()                      # first touch: w == 0 -> wt = wi (:186-187)
    with pytest.raises(ValueError):
        kernels.fuse_volume_dqb(T, W, dev(live, torch.float64), node_pos, node_dq, node_w, 9, lw, tdist)   # knn > 8
    with pytest.raises(ValueError):
        kernels.fuse_volume_dqb(T, W, dev(live, torch.float64), node_pos[:3], node_dq[:3], node_w[:3], 4, lw, tdist)
    with pytest.raises(ValueError):
        kernels.fuse_volume_dqb(T, W, dev(live, torch.float64), node_pos, node_dq[:, :7], node_w, 4, lw, tdist)
    fu = Fusion(T0, tdist)
    with pytest.raises(ValueError):
        fu.updateTSDF()                        # 'tsdf of live frame has not been loaded' (:158-159)
    with pytest.raises(ValueError):
        fu.updateTSDF(np.zeros((3, 3)))
    with pytest.raises(ValueError):
        Fusion(np.zeros((3, 3)), 1.0)


def test_dqb_config3_scale_properties():
    """256^3 + 512 Fibonacci nodes (BASELINE config 3 shape): identity warp field and identity
    lw reduce K3 to K2's sampling, so masks must equal the rigid kernel's; plus determinism."""
    R, N, k = 256, 512, 4
    scale, center, tdist_m = scene.grid_params(R)
    tdist = 4.0
    node_pos, node_w = scene.fibonacci_nodes(N, R)
    node_dq = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0, 0]), (N, 1))
    ident = np.array([1.0, 0, 0, 0, 0, 0, 0, 0])
    g = torch.arange(R, device="cuda", dtype=torch.float32)
    d = torch.sqrt((g[:, None, None] - R / 2) ** 2 + (g[None, :, None] - R / 2) ** 2 + (g[None, None, :] - R / 2) ** 2)
    live = torch.clamp(d - 80.0, -1.5 * tdist, 1.5 * tdist)
    T = torch.full((R, R, R), tdist, dtype=torch.float32, device="cuda"); W = torch.ones_like(T)
    T2, W2 = T.clone(), W.clone()
    kernels.fuse_volume_dqb(T, W, live, node_pos, node_dq, node_w, k, ident, tdist)
    kernels.fuse_volume_rigid(T2, W2, live, ident, tdist)
    assert torch.equal(W != 1, W2 != 1)
    # T mask: with the identity warp the sample is live[i] itself.  The two kernels average it with different weights, so a
    # sample within a few ulp of tdist may round back to tdist in one of them only; everywhere else the masks must agree
    sel = (live >= tdist) | (live < tdist - 1e-3)
    assert torch.equal((T != tdist)[sel], (T2 != tdist)[sel])
    assert int(((T != tdist) & sel).sum()) > 500000
    # same sampled value, different weight rule: T = (T*1 + m*wi)/(wi+1) vs (T*1 + m)/2
    T3, W3 = torch.full_like(T, tdist), torch.ones_like(T)
    kernels.fuse_volume_dqb(T3, W3, live, node_pos, node_dq, node_w, k, ident, tdist)
    assert torch.equal(T3, T) and torch.equal(W3, W)           # deterministic


@pytest.mark.parametrize("field", ["random", "identity", "random-1500-nodes"])
def test_dqb_float32_paths_give_the_same_bits(field):
    """float32 volumes, knn = 4: the fast path in every mode -- search every call, search + store, stored neighbourhoods on
    the persistent LDS kernel (undecidable voxels deferred to the redo list) and on the plain kernel -- gives the same bits;
    the fp64 chain on the same volumes (option k3_exact) gives the same update mask and values within one float32 rounding.
    `identity`: every sample lands on a lattice point, i.e. EVERY voxel takes the exact chain (all of them go through the
    redo list of the LDS kernel).  1 500 nodes: the node table no longer fits 64 KB of LDS -- unpadded rows, one workgroup per CU."""
    from dynamicfusion_body_amd import _lib
    from dynamicfusion_body_amd.dq import twist_exp_dq
    R, N, k, tdist = 64, (1500 if field.endswith("nodes") else 150), 4, 3.0
    g = torch.arange(R, device="cuda", dtype=torch.float32)
    d = torch.sqrt((g[:, None, None] - R / 2) ** 2 + (g[None, :, None] - R / 2) ** 2 + (g[None, None, :] - R / 2) ** 2)
    live = torch.clamp(d - 0.3125 * R + 0.7, -1.5 * tdist, 1.5 * tdist).contiguous()
    T0 = torch.clamp(d - 0.3125 * R, -tdist, tdist).contiguous()
    W0 = (torch.rand((R, R, R), device="cuda") < 0.7).float() * 2.0           # (zeros: the first-touch rule)
    rng = np.random.default_rng(5)
    node_pos, node_w = scene.fibonacci_nodes(N, R)
    if field.startswith("random"):
        dqs = twist_exp_dq(rng.normal(size=(N, 6)) * np.array([.02, .02, .02, .4, .4, .4]))
        lw = twist_exp_dq(np.array([0.01, -0.02, 0.015, 0.3, -0.2, 0.1]))
    else:
        dqs = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0, 0]), (N, 1))
        lw = np.array([1.0, 0, 0, 0, 0, 0, 0, 0])

    def run(ws, rebuild, opts=()):
        for o in opts:
            _lib.set_option(o, 1)
        T, W = T0.clone(), W0.clone()
        kernels.fuse_volume_dqb(T, W, live, node_pos, dqs, node_w, k, lw, tdist, 9.0, workspace=ws, rebuild_candidates=rebuild)
        torch.cuda.synchronize()
        for o in opts:
            _lib.set_option(o, None)
        return T, W
    ref = run(kernels.dqb_workspace((R, R, R)), True)
    ws2 = kernels.dqb_workspace((R, R, R), knn=k, n_nodes=N, level=2)
    for name, out in (("search + store", run(ws2, True)), ("stored, LDS kernel", run(ws2, False)), ("stored, LDS kernel again", run(ws2, False)),
                      ("stored, plain kernel", run(ws2, False, ("k3_no_lds",)))):
        assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]), name
    ex = run(kernels.dqb_workspace((R, R, R)), True, ("k3_exact",))
    assert torch.equal((ex[0] != T0) | (ex[1] != W0), (ref[0] != T0) | (ref[1] != W0))
    assert float(((ex[0] - ref[0]).abs() / (1 + ref[0].abs())).max()) <= f32_tol(1)
    assert float(((ex[1] - ref[1]).abs() / (1 + ref[1].abs())).max()) <= f32_tol(1)
    assert int(((ref[0] != T0) | (ref[1] != W0)).sum()) > 10000


# ------------------------------------------------------------------------------ K3: the constant-live skip (round 4)
def _skip_scene(res, N, tdist, field, seed=3):
    from dynamicfusion_body_amd.dq import twist_exp_dq
    rng = np.random.default_rng(seed)
    gx, gy, gz = (torch.arange(r, device="cuda", dtype=torch.float32) for r in res)
    c = [r / 2 for r in res]
    d = torch.sqrt((gx[:, None, None] - c[0]) ** 2 + (gy[None, :, None] - c[1]) ** 2 + (gz[None, None, :] - c[2]) ** 2)
    rad = 0.3 * min(res)
    # a live volume as K1 leaves it: the truncation value away from the surface (in front of it and where nothing was seen)
    live = torch.clamp(d - rad + 0.7, -tdist, tdist).contiguous()
    T0 = torch.clamp(d - rad, -tdist, tdist).contiguous()
    W0 = (torch.rand(res, device="cuda") < 0.7).float() * 2.0                  # (zeros: the first-touch rule)
    W0[: res[0] // 3] = 9.0                                                    # saturated, T at tdist far out: the stream's "nothing changes" shortcut
    node_pos, node_w = scene.fibonacci_nodes(N, min(res))
    node_pos = node_pos * (np.array(res) / min(res))
    amp = {"gentle": (.002, .15), "wild": (.05, 3.0), "identity": (0.0, 0.0)}[field]
    dqs = twist_exp_dq(rng.normal(size=(N, 6)) * np.array([amp[0]] * 3 + [amp[1]] * 3))
    return live, T0, W0, node_pos, node_w, dqs


@pytest.mark.parametrize("res,tdist,field,live_dtype", [((128, 128, 128), 4.0, "gentle", torch.float32), ((64, 62, 128), 3.0, "gentle", torch.float64),
                                                        ((64, 64, 64), 4.0, "wild", torch.float32), ((64, 64, 64), 2.0, "identity", torch.float32)])
def test_dqb_constant_live_skip_gives_the_same_bits(res, tdist, field, live_dtype):
    """Steady state of the float32 / knn = 4 path with m_lw = identity: bricks that provably sample only live voxels holding
    exactly tdist skip the warp (dqb_stream_kernel), the others go through the LDS kernel from the skip's sub-lists.  Same bits as
    with the skip off -- whole grid and a slab that starts inside the grid, float32 and float64 live volumes, a ragged y extent
    (cells cut by the volume's face never count as constant), tdist a power of two (the saturated shortcut) and not; a gentle
    field skips a share of the bricks (the sphere's shell is thick against these small grids), a wild one (translations of voxels: bounds beyond 7) none or few, and the identity
    field sends every remaining voxel through the redo list."""
    from dynamicfusion_body_amd import _lib
    N, k = 150, 4
    live, T0, W0, node_pos, node_w, dqs = _skip_scene(res, N, tdist, field)
    live = live.to(live_dtype)
    ident = np.array([1.0, 0, 0, 0, 0, 0, 0, 0])
    for a, b in ((0, res[0]), (res[0] // 4, res[0] // 4 * 3)):
        outs = {}
        for skip in (1, 0):
            ws = kernels.dqb_workspace(res, (a, b), knn=k, n_nodes=N, level=2)
            T, W = T0[a:b].clone(), W0[a:b].clone()
            _lib.set_option("k3_skip", skip)
            kernels.fuse_volume_dqb(T, W, live, node_pos, dqs, node_w, k, ident, tdist, 9.0, res=res, x_range=(a, b), workspace=ws, rebuild_candidates=True)
            Tm, Wm = T.clone(), W.clone()
            for _ in range(2):                                                  # steady state, twice (w reaches wmax on the way)
                kernels.fuse_volume_dqb(T, W, live, node_pos, dqs, node_w, k, ident, tdist, 9.0, res=res, x_range=(a, b), workspace=ws,
                                        rebuild_candidates=False)
            torch.cuda.synchronize()
            outs[skip] = (T, W, Tm, Wm)
            if skip:
                tabs = kernels.dqb_skip_tables(ws, res, res, N, x_range=(a, b))
                assert tabs["ok"]
                share = float(tabs["S"].float().mean())
        assert torch.equal(outs[1][2], outs[0][2]) and torch.equal(outs[1][3], outs[0][3])          # (the store pass: no skip yet)
        assert torch.equal(outs[1][0], outs[0][0]) and torch.equal(outs[1][1], outs[0][1]), (res, field, (a, b))
        assert not torch.equal(outs[1][0], outs[1][2])                                               # the steady-state calls did something
        if field == "gentle":
            assert share > 0.05, share
        if field == "wild":
            assert share < 0.2, share


def test_dqb_skip_bound_holds_for_every_voxel():
    """The proof obligation of the skip, checked on the device: for every brick whose bound is finite no voxel's warped position
    (the reference's chain, dfh_warp_points, with the voxel's k nearest nodes) lies further from the voxel than the brick's bound.
    Unit and non-unit node DQs (the 8-norm's scale term), rotations up to a few degrees, translations up to voxels."""
    from dynamicfusion_body_amd import _lib
    from dynamicfusion_body_amd.solve import sample_knn, warp_points
    from dynamicfusion_body_amd.dq import twist_exp_dq
    R, N, k, tdist = 64, 120, 4, 4.0
    rng = np.random.default_rng(17)
    live, T0, W0, node_pos, node_w, _ = _skip_scene((R, R, R), N, tdist, "gentle")
    ident = np.array([1.0, 0, 0, 0, 0, 0, 0, 0])
    g = torch.stack(torch.meshgrid(*[torch.arange(R, device="cuda", dtype=torch.float64)] * 3, indexing="ij"), dim=-1).reshape(-1, 3).contiguous()
    nbr, _ = sample_knn(g, node_pos, node_w, k)
    tightest = []
    for rot, trans, scale in ((.003, .3, 1.0), (.03, 1.5, 1.0), (.01, .5, 1.03)):
        dqs = twist_exp_dq(rng.normal(size=(N, 6)) * np.array([rot] * 3 + [trans] * 3)) * scale
        ws = kernels.dqb_workspace((R, R, R), knn=k, n_nodes=N, level=2)
        T, W = T0.clone(), W0.clone()
        _lib.set_option("k3_skip", 2)                                           # every brick's bound, also where the live volume rules the skip out
        for rebuild in (True, False):
            kernels.fuse_volume_dqb(T, W, live, node_pos, dqs, node_w, k, ident, tdist, 9.0, workspace=ws, rebuild_candidates=rebuild)
        torch.cuda.synchronize()
        bound = kernels.dqb_skip_tables(ws, (R, R, R), (R, R, R), N)["bound"].double().view(R // 4, R // 4, R // 16)
        wp, _ = warp_points(g, None, ident, nbr=nbr, node_dq=dqs, node_pos=node_pos, node_w=node_w)
        disp = (wp - g).norm(dim=1).view(R // 4, 4, R // 4, 4, R // 16, 16).amax(dim=(1, 3, 5))
        finite = torch.isfinite(bound)
        assert float(finite.float().mean()) > 0.9
        assert bool((bound >= 0).all())
        assert bool((disp[finite] <= bound[finite]).all()), float((disp[finite] / bound[finite]).max())
        tightest.append(float((disp[finite] / bound[finite]).max()))
    assert max(tightest) > 0.3, tightest                                       # ... and the bound is not vacuous
