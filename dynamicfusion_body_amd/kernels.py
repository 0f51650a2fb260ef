"""Tensor-level entry points: device tensors in, updated in place, asynchronous on the
current HIP stream.  Each function is one call through the C ABI (include/dfusion_hip.h);
the reference-shaped classes in fusion_dm.py / fusion.py are built on these."""
import ctypes
import numpy as np
import torch

from . import _lib
from .device import current_stream_ptr, dtype_code, require_gpu


def _check_volume_pair(T, Wt, res, x_range):
    if not (isinstance(T, torch.Tensor) and isinstance(Wt, torch.Tensor)):
        raise ValueError("volumes must be torch tensors on the GPU")
    if not (T.is_cuda and Wt.is_cuda):
        raise ValueError("volumes must live on the GPU")
    if T.dtype != Wt.dtype:
        raise ValueError("tsdf and weight volumes must share a dtype")
    if not (T.is_contiguous() and Wt.is_contiguous()):
        raise ValueError("volumes must be contiguous [x][y][z]")
    x0, x1 = x_range
    want = (x1 - x0, res[1], res[2])
    if tuple(T.shape) != want or tuple(Wt.shape) != want:
        raise ValueError("volume shape %s / %s does not match slab %s of grid %s"
                         % (tuple(T.shape), tuple(Wt.shape), want, tuple(res)))


_ws_cache = {}


def integrate_workspace(n_views, H, W, res, x_range=None, device=None):
    """Scratch tensor for integrate_depth / integrate_depth_views on planes `x_range` of a `res` grid (the views'
    parameters, depth pyramids and per-brick view masks); cached per (device, stream, n_views, H, W, grid, slab size):
    launches on one stream are ordered, so consecutive calls may share it."""
    lib = _lib.load()
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if x_range is None:
        x_range = (0, res[0])
    nx = int(x_range[1]) - int(x_range[0])
    key = (dev.index, current_stream_ptr(), int(n_views), int(H), int(W), int(res[1]), int(res[2]), nx)
    ws = _ws_cache.get(key)
    if ws is None:
        nbytes = lib.dfh_integrate_workspace_bytes(int(n_views), int(H), int(W), _lib.iarr(res), 0, nx)
        ws = torch.empty((nbytes + 15) // 16 * 2, dtype=torch.int64, device=dev)
        if len(_ws_cache) > 64:
            _ws_cache.clear()
        _ws_cache[key] = ws
    return ws


K1_PATHS = {0: "exact", 1: "rows", 2: "columns", 3: "columns_culled"}       # include/dfusion_hip.h: DFH_K1_PATH_*
BRICK = (4, 2, 32)                                                            # voxels of a brick of the column sweep (csrc/dfh_integrate.hip)


def integrate_path(T, depth, res=None, x_range=None, workspace=True):
    """Name of the sweep integrate_depth takes for this slab and depth map (dfh_integrate_depth_path; no launch)."""
    lib = _lib.load()
    if res is None:
        res = tuple(T.shape)
    if x_range is None:
        x_range = (0, res[0])
    H, W = depth.shape
    code = lib.dfh_integrate_depth_path(dtype_code(T), _lib.iarr(res), int(x_range[0]), int(x_range[1]), int(H), int(W), 1 if workspace else 0)
    if code < 0:
        _lib.check(code, "dfh_integrate_depth_path")
    return K1_PATHS[code]


def brick_masks(workspace, res, x_range=None):
    """The per-brick 16-bit view masks the last culled sweep through `workspace` left behind (bit v = view v may update a voxel
    of the brick), as a (bricks_x, bricks_y, bricks_z) int16 view: the LAST region of the workspace (include/dfusion_hip.h:
    dfh_integrate_workspace_bytes).  Measurement code counts surviving bricks with it."""
    if x_range is None:
        x_range = (0, res[0])
    nb = (-(-(int(x_range[1]) - int(x_range[0])) // BRICK[0]), -(-int(res[1]) // BRICK[1]), -(-int(res[2]) // BRICK[2]))
    n = nb[0] * nb[1] * nb[2]
    raw = workspace.view(torch.int16)
    tail = (2 * n + 15) // 16 * 8                       # the mask region is padded to whole 16-byte units
    return raw[raw.numel() - tail:raw.numel() - tail + n].view(nb)


def integrate_depth(T, Wt, depth, K, Kinv, lw, scale, center, tdist, wmax=100.0, tsdf_res=None,
                    res=None, x_range=None, workspace=None):
    """K1 = FusionDM.fuseDepths (reference core/fusion_dm.py:180-217) on device tensors.

    T, Wt : (x1-x0, Y, Z) float32/float64 CUDA tensors holding planes [x0,x1) of a `res`
            grid (default: the whole grid).  depth: (H, W) float32/float64 CUDA tensor of
            negative depths.  Runs asynchronously on the current stream.  workspace: scratch from
            integrate_workspace(1, H, W) (default: a cached one); False = sweep without brick culling."""
    require_gpu()
    lib = _lib.load()
    if res is None:
        res = tuple(T.shape)
    if x_range is None:
        x_range = (0, res[0])
    if tsdf_res is None:
        tsdf_res = res[0]
    _check_volume_pair(T, Wt, res, x_range)
    if x_range[1] == x_range[0]:
        return T, Wt
    if not (isinstance(depth, torch.Tensor) and depth.is_cuda and depth.dim() == 2 and depth.is_contiguous()):
        raise ValueError("depth must be a contiguous 2-D CUDA tensor")
    H, W = depth.shape
    if workspace is None and T.dtype == torch.float32:
        workspace = integrate_workspace(1, H, W, res, x_range, T.device)
    ws_ptr, ws_bytes = (workspace.data_ptr(), workspace.numel() * workspace.element_size()) if isinstance(workspace, torch.Tensor) else (0, 0)
    rc = lib.dfh_integrate_depth(T.data_ptr(), Wt.data_ptr(), dtype_code(T), _lib.iarr(res), int(tsdf_res),
                                 int(x_range[0]), int(x_range[1]), depth.data_ptr(), dtype_code(depth),
                                 int(H), int(W), _lib.darr(K, 9), _lib.darr(Kinv, 9), _lib.darr(lw, 12),
                                 float(scale), _lib.darr(np.asarray(center, dtype=np.float64), 3),
                                 float(tdist), float(wmax), ws_ptr, ws_bytes, current_stream_ptr())
    _lib.check(rc, "dfh_integrate_depth")
    return T, Wt


def integrate_depth_ocl(T, Wt, depth, proj, kinv_row2, tdist, wmax=100.0, res=None, x_range=None):
    """A2: the arithmetic of the reference's OpenCL kernel (core/fusion_dm.py:630-674; dfh_integrate_depth_ocl) on float32
    device volumes, in place.  proj: float32 3x4 index -> pixel map (K lw IND, :695); kinv_row2: third row of K^-1 (float32)."""
    require_gpu()
    lib = _lib.load()
    if res is None:
        res = tuple(T.shape)
    if x_range is None:
        x_range = (0, res[0])
    _check_volume_pair(T, Wt, res, x_range)
    if T.dtype != torch.float32:
        raise ValueError("the OpenCL arithmetic is float32: volumes must be float32")
    if not (isinstance(depth, torch.Tensor) and depth.is_cuda and depth.dim() == 2 and depth.is_contiguous() and depth.dtype == torch.float32):
        raise ValueError("depth must be a contiguous 2-D float32 CUDA tensor")
    if x_range[1] == x_range[0]:
        return T, Wt
    pr = (ctypes.c_float * 12)(*np.asarray(proj, dtype=np.float32).reshape(12).tolist())
    kr = (ctypes.c_float * 3)(*np.asarray(kinv_row2, dtype=np.float32).reshape(3).tolist())
    H, W = depth.shape
    _lib.check(lib.dfh_integrate_depth_ocl(T.data_ptr(), Wt.data_ptr(), _lib.iarr(res), int(x_range[0]), int(x_range[1]), depth.data_ptr(),
                                           int(H), int(W), pr, kr, ctypes.c_float(float(tdist)), ctypes.c_float(float(wmax)),
                                           current_stream_ptr()), "dfh_integrate_depth_ocl")
    return T, Wt


def integrate_depth_views(T, Wt, depths, K, Kinv, lws, scale, center, tdist, wmax=100.0, tsdf_res=None, res=None,
                          x_range=None, workspace=None, fresh=None):
    """Several views in one sweep of the volume (dfh_integrate_depth_multi): same result, bit for bit, as
    integrate_depth called once per view in this order (what the reference's loops over fuseDepths do,
    core/fusion_dm.py:152-154,166-170), with T and w read and written once.  depths: list of (H, W) CUDA tensors of one
    shape and dtype; lws: list of 3x4 extrinsics.  More than 16 views are taken 16 at a time.
    fresh=value: T and Wt are first set to (value, 0) -- a live volume from scratch, core/fusion_dm.py:152-153 -- as part of the
    same sweep (dfh_integrate_depth_multi_fresh): what T.fill_(value); Wt.zero_() in front of this call give, bit for bit."""
    require_gpu()
    lib = _lib.load()
    depths, lws = list(depths), list(lws)
    if len(depths) != len(lws):
        raise ValueError('length of camera matrix array must equal that of depth maps')        # core/fusion_dm.py:96-97
    if res is None:
        res = tuple(T.shape)
    if x_range is None:
        x_range = (0, res[0])
    if tsdf_res is None:
        tsdf_res = res[0]
    _check_volume_pair(T, Wt, res, x_range)
    if x_range[1] == x_range[0]:
        return T, Wt
    if not depths:
        if fresh is not None:
            T.fill_(float(fresh))
            Wt.zero_()
        return T, Wt
    for d in depths:
        if not (isinstance(d, torch.Tensor) and d.is_cuda and d.dim() == 2 and d.is_contiguous()):
            raise ValueError("depth must be a contiguous 2-D CUDA tensor")
        if d.shape != depths[0].shape or d.dtype != depths[0].dtype:
            raise ValueError("all depth maps of one call must have the same shape and dtype")
    H, W = depths[0].shape
    for i in range(0, len(depths), 16):
        dd, ll = depths[i:i + 16], lws[i:i + 16]
        n = len(dd)
        nbytes = lib.dfh_integrate_workspace_bytes(n, int(H), int(W), _lib.iarr(res), int(x_range[0]), int(x_range[1]))
        ws = workspace if (workspace is not None and workspace.numel() * workspace.element_size() >= nbytes) else \
            integrate_workspace(n, H, W, res, x_range, T.device)
        ptrs = (ctypes.c_void_p * n)(*[d.data_ptr() for d in dd])
        lw_flat = np.concatenate([np.asarray(l, dtype=np.float64).reshape(12) for l in ll])
        tail = (n, ptrs, dtype_code(dd[0]), int(H), int(W), _lib.darr(K, 9), _lib.darr(Kinv, 9), _lib.darr(lw_flat, 12 * n), float(scale),
                _lib.darr(np.asarray(center, dtype=np.float64), 3), float(tdist), float(wmax), ws.data_ptr(),
                ws.numel() * ws.element_size(), current_stream_ptr())
        head = (T.data_ptr(), Wt.data_ptr(), dtype_code(T), _lib.iarr(res), int(tsdf_res), int(x_range[0]), int(x_range[1]))
        if fresh is not None and i == 0:
            _lib.check(lib.dfh_integrate_depth_multi_fresh(*head, float(fresh), *tail), "dfh_integrate_depth_multi_fresh")
        else:
            _lib.check(lib.dfh_integrate_depth_multi(*head, *tail), "dfh_integrate_depth_multi")
    return T, Wt


def _check_live(live):
    if not (isinstance(live, torch.Tensor) and live.is_cuda and live.dim() == 3 and live.is_contiguous()):
        raise ValueError("live TSDF must be a contiguous 3-D CUDA tensor")


def fuse_volume_rigid(T, Wt, live, lw_dq, tdist, wmax=100.0, res=None, x_range=None):
    """K2 = FusionDM.updateTSDF (reference core/fusion_dm.py:300-316) on device tensors.
    T, Wt: planes [x0,x1) of the canonical grid `res`; live: the whole live volume."""
    require_gpu()
    lib = _lib.load()
    if res is None:
        res = tuple(T.shape)
    if x_range is None:
        x_range = (0, res[0])
    _check_volume_pair(T, Wt, res, x_range)
    _check_live(live)
    if x_range[1] == x_range[0]:
        return T, Wt
    rc = lib.dfh_fuse_volume_rigid(T.data_ptr(), Wt.data_ptr(), dtype_code(T), _lib.iarr(res), int(x_range[0]),
                                   int(x_range[1]), live.data_ptr(), dtype_code(live), _lib.iarr(live.shape),
                                   _lib.darr(lw_dq, 8), float(tdist), float(wmax), current_stream_ptr())
    _lib.check(rc, "dfh_fuse_volume_rigid")
    return T, Wt


def dqb_workspace(res, x_range=None, device=None, knn=None, n_nodes=None, level=2):
    """Scratch tensor for fuse_volume_dqb: the per-brick candidate node lists and, when `knn` and `n_nodes` are
    given, per voxel the knn node indices (level 1: 2*knn bytes) and blend weights (level 2: + 8*(knn+1) bytes);
    calls with rebuild_candidates=False then skip the node search / the weight computation."""
    require_gpu()
    lib = _lib.load()
    if x_range is None:
        x_range = (0, res[0])
    if knn is not None and n_nodes is not None:
        nbytes = lib.dfh_dqb_workspace_bytes_cached(_lib.iarr(res), int(x_range[0]), int(x_range[1]), int(knn), int(n_nodes), int(level))
    else:
        nbytes = lib.dfh_dqb_workspace_bytes(_lib.iarr(res), int(x_range[0]), int(x_range[1]))
    return torch.empty(max(1, (nbytes + 3) // 4), dtype=torch.int32, device=device or "cuda")


def dqb_skip_tables(workspace, res, live_res, n_nodes, x_range=None, knn=4):
    """Views of the constant-live skip's per-call tables inside a level-2 dqb_workspace (dfh_dqb_skip_layout), as left by the last
    steady-state fuse_volume_dqb call through it: {"U": live-cell mask words, "S": uint8 per brick (1 = its voxels took the constant-live stream), "reach": uint8 per
    brick, "bound": float32 per brick (voxels; -1 = not computed: the live volume ruled the skip out; option k3_skip = 2 computes all), "used": int16 (bricks, 16) node ids, "n_listed": 16-voxel rows left to the warp kernel, "n_runs": all such rows, "ok": sizes admit the skip}.  For tests and
    measurement code."""
    import ctypes
    lib = _lib.load()
    if x_range is None:
        x_range = (0, res[0])
    out = (ctypes.c_size_t * 13)()
    _lib.check(lib.dfh_dqb_skip_layout(_lib.iarr(res), int(x_range[0]), int(x_range[1]), _lib.iarr(live_res), int(knn), int(n_nodes), out),
               "dfh_dqb_skip_layout")
    raw = workspace.view(torch.uint8)
    nx = int(x_range[1]) - int(x_range[0])
    nb = (-(-nx // 4)) * (-(-int(res[1]) // 4)) * (-(-int(res[2]) // 16))
    CX, CY, WZ, SCX, SCY = (int(out[i]) for i in (6, 7, 8, 9, 10))
    n_runs = nx * int(res[1]) * (int(res[2]) // 16)                      # 16-voxel rows

    def region(i, nbytes, dtype):
        return raw[int(out[i]):int(out[i]) + nbytes].view(dtype)
    tabs = {"ok": bool(out[11]), "U": region(0, CX * CY * WZ * 8, torch.int64).view(CX, CY, WZ), "S": region(1, nb, torch.uint8),
            "reach": region(2, nb, torch.uint8), "bound": region(3, nb * 4, torch.float32), "used": region(12, nb * 32, torch.int16).view(nb, 16)}
    tabs["n_runs"] = n_runs
    tabs["n_listed"] = int((tabs["S"] == 0).sum()) * 16                 # rows left to the warp kernel (whole bricks)
    return tabs


def dqb_build_candidates(workspace, res, node_pos, knn, x_range=None):
    """Fill the per-brick candidate node lists of a dqb_workspace (what fuse_volume_dqb does itself on a call with
    rebuild_candidates=True); needed up front only by solve.sample_knn(..., bricks=...)."""
    require_gpu()
    lib = _lib.load()
    if x_range is None:
        x_range = (0, res[0])
    P = node_pos if isinstance(node_pos, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(node_pos, dtype=np.float64)))
    P = P.to(device="cuda", dtype=torch.float64).contiguous()
    _lib.check(lib.dfh_dqb_build_candidates(_lib.iarr(res), int(x_range[0]), int(x_range[1]), P.data_ptr(), int(P.shape[0]), int(knn),
                                            workspace.data_ptr(), workspace.numel() * 4, current_stream_ptr()), "dfh_dqb_build_candidates")
    return workspace


def _node_tensors(node_pos, node_dq, node_w):
    def prep(a, shape_tail):
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64)))
        t = t.to(device="cuda", dtype=torch.float64).contiguous()
        if tuple(t.shape[1:]) != shape_tail:
            raise ValueError("node array has shape %s, expected (N,%s)" % (tuple(t.shape), ",".join(map(str, shape_tail))))
        return t
    P, Q, Wn = prep(node_pos, (3,)), prep(node_dq, (8,)), prep(node_w, ())
    if not (P.shape[0] == Q.shape[0] == Wn.shape[0]):
        raise ValueError("node_pos / node_dq / node_w disagree on the number of nodes")
    return P, Q, Wn


def fuse_volume_dqb(T, Wt, live, node_pos, node_dq, node_w, knn, lw_dq, tdist, wmax=100.0, res=None,
                    x_range=None, workspace=None, rebuild_candidates=True):
    """K3 = Fusion.updateTSDF (reference core/fusion.py:153-198) on device tensors.
    node_pos (N,3), node_dq (N,8), node_w (N,) fp64 (numpy or CUDA).  `workspace` (from
    dqb_workspace) may be kept across calls; pass rebuild_candidates=False while the node
    positions, knn and slab are unchanged."""
    require_gpu()
    lib = _lib.load()
    if res is None:
        res = tuple(T.shape)
    if x_range is None:
        x_range = (0, res[0])
    _check_volume_pair(T, Wt, res, x_range)
    _check_live(live)
    P, Q, Wn = _node_tensors(node_pos, node_dq, node_w)
    if x_range[1] == x_range[0]:
        return T, Wt
    if workspace is None:
        workspace = dqb_workspace(res, x_range)
        rebuild_candidates = True
    rc = lib.dfh_fuse_volume_dqb(T.data_ptr(), Wt.data_ptr(), dtype_code(T), _lib.iarr(res), int(x_range[0]),
                                 int(x_range[1]), live.data_ptr(), dtype_code(live), _lib.iarr(live.shape),
                                 P.data_ptr(), Q.data_ptr(), Wn.data_ptr(), int(P.shape[0]), int(knn),
                                 _lib.darr(lw_dq, 8), float(tdist), float(wmax), workspace.data_ptr(),
                                 workspace.numel() * 4, 1 if rebuild_candidates else 0, current_stream_ptr())
    _lib.check(rc, "dfh_fuse_volume_dqb")
    return T, Wt
