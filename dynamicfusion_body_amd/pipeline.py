"""Per-frame orchestration of the hot path: surface samples from the canonical volume, their
node neighbourhoods, then GN iterations (associate -> build -> all-reduce -> PCG -> update), then
the TSDF update.  Mirrors the reference's frame loop (test.py:116-131: setupCorrespondences ->
solve -> updateTSDF) with projective association in place of marching cubes + KD-tree.

`extract_surface_samples` runs the HIP band-compaction kernels of csrc/dfh_extract.hip (the
first "next" row of SURVEY.md §8(f)); `extract_surface_samples_torch` is the same computation on
torch ops, kept for CPU-side tests of the sample definition."""
import numpy as np
import torch

from . import _lib, kernels
from .device import HostScalar, current_stream_ptr, dtype_code, require_gpu
from .solve import WarpSolver, sample_knn


def extract_surface_samples(T, Wt, band, x0=0, max_samples=None):
    """Band voxels (w > 0, |T| < band; T in voxel units as fuseDepths stores it) of a slab starting
    at global plane x0 -> (surface points (S,3) in global index space, unit normals), fp64 CUDA
    tensors in voxel order.  Three HIP launches (count, scan, emit) and one 8-byte read-back of the
    sample count.  max_samples < the number of band voxels keeps an even subsample in voxel order (sample i iff it is the
    first one with slot floor(i * max_samples / n)), i.e. the whole surface at a lower density, never a prefix."""
    require_gpu()
    lib = _lib.load()
    if not (isinstance(T, torch.Tensor) and T.is_cuda and T.dim() == 3 and T.is_contiguous() and Wt.shape == T.shape
            and Wt.is_cuda and Wt.is_contiguous() and Wt.dtype == T.dtype):
        raise ValueError("T and Wt must be contiguous 3-D CUDA tensors of the same shape and dtype")
    res = _lib.iarr(T.shape)
    nbytes = lib.dfh_surface_workspace_bytes(res)
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=T.device)
    total = HostScalar(torch.int64)                    # (the scan kernel stores the count straight into pinned host memory)
    _lib.check(lib.dfh_surface_count(T.data_ptr(), Wt.data_ptr(), dtype_code(T), res, float(band), ws.data_ptr(),
                                     ws.numel() * 8, total.ptr(), current_stream_ptr()), "dfh_surface_count")
    n = total.get()
    cap = n if max_samples is None else min(n, int(max_samples))
    pos = torch.empty((cap, 3), dtype=torch.float64, device=T.device)
    nrm = torch.empty((cap, 3), dtype=torch.float64, device=T.device)
    _lib.check(lib.dfh_surface_emit(T.data_ptr(), Wt.data_ptr(), dtype_code(T), res, int(x0), float(band), ws.data_ptr(),
                                    pos.data_ptr(), nrm.data_ptr(), cap, current_stream_ptr()), "dfh_surface_emit")
    return pos, nrm


def extract_surface_samples_torch(T, Wt, band, x0=0, max_samples=None):
    """Same samples on torch ops (works on CPU tensors).  Band voxels of a slab
    starting at global plane x0 -> (surface points (S,3) in global index space, unit normals).
    Normals are central differences of T inside the slab (one-sided at its faces); points are the
    voxel centres moved onto the zero level set by one Newton step, centre - T grad / |grad|^2, of length <= |T|."""
    Tf = T.to(torch.float64)
    mask = (Wt > 0) & (Tf.abs() < band)
    idx = mask.nonzero(as_tuple=False)
    if max_samples is not None and idx.shape[0] > max_samples:
        n, cap = idx.shape[0], int(max_samples)                  # the device rule: first sample of every slot floor(i * cap / n)
        idx = idx[(torch.arange(cap, device=idx.device, dtype=torch.int64) * n + cap - 1) // cap]
    X, Y, Z = T.shape
    ix, iy, iz = idx[:, 0], idx[:, 1], idx[:, 2]

    def diff(axis_idx, n, take):
        lo, hi = (axis_idx - 1).clamp(min=0), (axis_idx + 1).clamp(max=n - 1)
        return (take(hi) - take(lo)) / (hi - lo).clamp(min=1).to(torch.float64)      # (a one-voxel-thick axis: 0 / 1, as the kernels)
    gx = diff(ix, X, lambda a: Tf[a, iy, iz])
    gy = diff(iy, Y, lambda a: Tf[ix, a, iz])
    gz = diff(iz, Z, lambda a: Tf[ix, iy, a])
    g = torch.stack([gx, gy, gz], dim=1)
    nrm = g.norm(dim=1, keepdim=True)
    ok = nrm[:, 0] > 1e-6
    idx, g, nrm = idx[ok], g[ok], nrm[ok]
    n = g / nrm
    pos = idx.to(torch.float64)
    pos[:, 0] += x0
    pos = pos - (Tf[idx[:, 0], idx[:, 1], idx[:, 2]][:, None] / nrm.clamp(min=1.0)) * n   # one Newton step: T grad / |grad|^2, no longer than |T|
    return pos.contiguous(), n.contiguous()


class FrameSolver:
    """Warp-field estimation of one live depth frame against the canonical volume."""

    def __init__(self, K, scale, center, half, knn=4, pcg_iters=10, distributed=True):
        self.K = np.asarray(K, dtype=np.float64)
        self.Kinv = np.linalg.inv(self.K)
        self.scale, self.center, self.half = float(scale), np.asarray(center, dtype=np.float64), float(half)
        self.solver = WarpSolver(knn=knn, pcg_iters=pcg_iters, distributed=distributed)
        self.knn = knn
        self.lw = np.array([1.0, 0, 0, 0, 0, 0, 0, 0])

    def set_graph(self, node_pos, node_dq, node_w):
        node_nbr, _ = sample_knn(node_pos, node_pos, node_w, self.knn)        # a node's own k nearest nodes
        self.solver.set_graph(node_pos, node_dq, node_w, node_nbr=node_nbr)

    def set_canonical(self, T, Wt, band=1.0, x0=0, max_samples=None, knn_bricks=None):
        """knn_bricks: see solve.sample_knn (candidate lists of the K3 workspace: the samples' node search skips its
        per-workgroup bounding-box pass)."""
        pos, nrm = extract_surface_samples(T, Wt, band, x0=x0, max_samples=max_samples)
        self.solver.set_samples(pos, nrm, knn_bricks=knn_bricks)
        return pos.shape[0]

    def gn_iteration(self, depth, lw_cam, rw=5.0, lm_abs=10.0, lm_rel=1e-2, max_dist=2.0, huber=0.0, n_iters=1, n_global=0, global_lm=0.1):
        """associate -> build (+ all-reduce) -> PCG -> twist update, n_iters times; asynchronous.  depth / lw_cam may be lists
        (several live views: every sample associates with the view in which it lies closest to the observed surface).
        n_global: that many rigid-mode steps first (global_iteration), in the same library call on one GPU."""
        self.solver.iterate_associated(depth, self.K, self.Kinv, lw_cam, self.scale, self.center, self.half, self.lw, rw, max_dist, huber,
                                       lm_abs, lm_rel, n_iters=n_iters, n_global=n_global, global_lm=global_lm)

    def global_iteration(self, depth, lw_cam, rw=5.0, max_dist=2.0, huber=0.0, lm_rel=0.1, n_iters=1, built=False, stride=1):
        """The rigid mode alone -- one twist shared by all nodes -- n_iters times; asynchronous.  Worth several node iterations where
        the live frame has moved as a whole: ten truncated PCG iterations hardly touch that mode.  Default: straight from the
        samples (WarpSolver.global_sampled: data rows only, two short launches a step); built=True: from the built normal
        equations, regulariser included (associate -> build (+ all-reduce) -> WarpSolver.global_step)."""
        if not built:
            self.solver.global_sampled(depth, self.K, self.Kinv, lw_cam, self.scale, self.center, self.half, self.lw, max_dist, huber, lm_rel,
                                       n_steps=n_iters, stride=stride)
            return
        for _ in range(int(n_iters)):
            self.solver.build_associated(depth, self.K, self.Kinv, lw_cam, self.scale, self.center, self.half, self.lw, rw, max_dist, huber)
            self.solver.global_step(lm_rel)

    def solve(self, depth, lw_cam, rw=5.0, iters=10, **kw):
        costs = []
        for _ in range(iters):
            self.gn_iteration(depth, lw_cam, rw, **kw)
            costs.append(self.solver.cost())
        return costs


class SlabFrame:
    """One rank's share of the per-frame loop (reference test.py:116-131) with the canonical volume cut into
    axis-0 slabs: live depth -> this rank's live slab (K1, no exchange) -> all-gather of the live volume ->
    GN iterations (samples of this slab; one all-reduce of the normal equations per iteration) -> canonical
    slab <- live volume through the warp field (K3) -> this slab's samples for the next frame.
    With one rank it is the single-GPU frame."""

    RELAX = 0.8          # default of step(relax=...): see there
    GLOBAL_ITERS = 2     # default of step(global_iters=...): rigid-mode steps in front of the node iterations
    GLOBAL_STRIDE = 4    # ... each fitted to every 4th 128-sample tile (step(global_stride=...))

    def __init__(self, K, scale, center, res, tdist_vox, node_pos, node_w, knn=4, pcg_iters=10, band=4.0, volume_dtype=torch.float32,
                 distributed=True, solve_mode="auto"):
        """solve_mode (several ranks): "sharded" = every rank builds the normal equations of its own slab's samples, one
        all-reduce per GN iteration (BASELINE north star); "replicated" = the slabs' samples are all-gathered once per frame
        and every rank solves the whole system, no per-iteration collective (bit-identical warp fields on all ranks, and the
        single-GPU loop's bits WHEN both run the same PCG path -- _lib dfh_pcg_path: ranks that share one GPU, a rehearsal, take
        the two-launch kernels, one rank per GPU the persistent kernel like a single-GPU run; tests/test_gpu_dist_gloo.py pins it at
        256^3 / 512 nodes); "auto" = dist.solve_mode's latency model, decided at the first sample refresh."""
        from . import dist as D
        self.D = D
        self.distributed = bool(distributed)
        if solve_mode not in ("auto", "sharded", "replicated"):
            raise ValueError("solve_mode must be 'auto', 'sharded' or 'replicated'")
        self.solve_mode = solve_mode
        self.R = int(res)
        self.K = np.asarray(K, dtype=np.float64)
        self.Kinv = np.linalg.inv(self.K)
        self.scale, self.center, self.tvox = float(scale), np.asarray(center, dtype=np.float64), float(tdist_vox)
        self.tdist_world = self.tvox * self.scale
        self.rank, self.ws = D.world() if self.distributed else (0, 1)     # distributed=False: the whole grid, no collectives
        self.a, self.b = D.slab_range(self.R, self.rank, self.ws)
        R = self.R
        self.T = torch.full((self.b - self.a, R, R), self.tvox, dtype=volume_dtype, device="cuda")
        self.Wt = torch.zeros_like(self.T)
        self.live = torch.empty_like(self.T)
        self.live_w = torch.empty_like(self.T)
        self.band, self.knn = float(band), int(knn)
        # (replicated / undecided: the solver itself runs no collective; a decision for "sharded" switches it on)
        self.fs = FrameSolver(K, scale, center, R / 2, knn=knn, pcg_iters=pcg_iters,
                              distributed=self.distributed and self.ws > 1 and solve_mode == "sharded")
        N = len(node_pos)
        ident = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0, 0]), (N, 1))
        self.fs.set_graph(node_pos, ident, node_w)
        self.ws_dqb = kernels.dqb_workspace((R, R, R), (self.a, self.b), knn=knn, n_nodes=N)
        self.ws_views = None                     # dfh_integrate_depth_multi's scratch (parameters + depth pyramids): sized on first use
        self._side = None                        # side stream of step(): the live-volume sweep beside the plan build
        self.updated = None                      # event recorded by step() right after the TSDF update
        self.knn_bricks = None
        if self.b > self.a:
            kernels.dqb_build_candidates(self.ws_dqb, (R, R, R), node_pos, knn, (self.a, self.b))
            self.knn_bricks = ((R, R, R), (self.a, self.b), self.ws_dqb)       # the samples' node search uses the same lists
        self._first = True
        self.ident_lw = np.array([1.0, 0, 0, 0, 0, 0, 0, 0])

    def integrate(self, depth, lw_cam):
        """Fuse a depth map into this rank's canonical slab (initial frames)."""
        R = self.R
        kernels.integrate_depth(self.T, self.Wt, depth, self.K, self.Kinv, lw_cam, self.scale, self.center, self.tdist_world,
                                tsdf_res=R, res=(R, R, R), x_range=(self.a, self.b))

    def refresh_samples(self):
        """Samples of this slab.  Normals are central differences of T, also across the slab faces: the
        neighbours' face planes come as a halo whose weight is 0 (they feed gradients, never samples), so the
        union over ranks is exactly the whole-grid sample set."""
        if self.ws == 1:
            return self.fs.set_canonical(self.T, self.Wt, band=self.band, x0=self.a, knn_bricks=self.knn_bricks)
        lo, hi = self.D.halo_planes(self.T, self.R)
        Tp, Wp, x0 = [self.T], [self.Wt], self.a
        if lo is not None:
            Tp.insert(0, lo[None]); Wp.insert(0, torch.zeros_like(lo)[None]); x0 -= 1
        if hi is not None:
            Tp.append(hi[None]); Wp.append(torch.zeros_like(hi)[None])
        Tp, Wp = torch.cat(Tp).contiguous(), torch.cat(Wp).contiguous()
        if self.solve_mode == "sharded":
            return self.fs.set_canonical(Tp, Wp, band=self.band, x0=x0, knn_bricks=self.knn_bricks)
        # replicated (or still to be decided): this slab's samples with their node tables, then all slabs' in rank order --
        # the whole-grid sample set in the whole-grid order (slabs are x ranges, samples are emitted x-major)
        sv = self.fs.solver
        pos, nrm = extract_surface_samples(Tp, Wp, self.band, x0=x0)
        nbr, wts = sample_knn(pos, sv.node_pos, sv.node_w, self.knn, bricks=self.knn_bricks) if pos.shape[0] else \
            (torch.empty((0, self.knn), dtype=torch.int32, device="cuda"), torch.empty((0, self.knn), dtype=torch.float64, device="cuda"))
        packed = torch.cat([pos, nrm, wts, nbr.to(torch.float64)], dim=1)        # (node indices are exact in fp64)
        dev = self.D.collective_device()
        allp = self.D.allgather_ragged(packed.to(dev)).to("cuda")
        k = self.knn
        if self.solve_mode == "auto":
            S_all, N = int(allp.shape[0]), int(sv.N)
            mode = self.D.solve_mode(S_all, N, 12 * N, self.ws)               # (~11.5 blocks per node row in practice)
            self.solve_mode = mode
            if mode == "sharded":                                             # every rank decides alike: same inputs
                self.fs.solver.distributed = True
                return self.fs.set_canonical(Tp, Wp, band=self.band, x0=x0, knn_bricks=self.knn_bricks)
        sv.set_samples(allp[:, 0:3].contiguous(), allp[:, 3:6].contiguous(), nbr=allp[:, 6 + k:6 + 2 * k].to(torch.int32).contiguous(),
                       weights=allp[:, 6:6 + k].contiguous())
        return int(allp.shape[0])

    def update_graph(self, radius=None):
        """Deformation-graph maintenance after a TSDF update (reference Fusion.update_graph, core/fusion.py:201-239) on the
        device path, with this loop's surface points -- the band samples of the canonical slab -- in the role of the mesh
        vertices: samples no node supports (min over their knn nodes of |node - p| / w >= 1) are radius-subsampled into new
        nodes whose DQs are the blend of the old graph at their positions; the solver's graph, K3's stored neighbourhoods
        and candidate lists and the samples' node table are rebuilt when nodes were inserted.  Every rank inserts the same
        nodes (the unsupported points are gathered in rank order before the greedy subsampling).  Returns the number of
        inserted nodes.  radius defaults to half the nodes' weight (w = 2 radius, core/fusion.py:116)."""
        from . import graph as _graph
        sv = self.fs.solver
        if radius is None:
            radius = 0.5 * float(sv.node_w[0])
        pts = sv.spos if sv.S > 0 else torch.zeros((0, 3), dtype=torch.float64, device="cuda")
        gather = None
        if self.ws > 1 and self.solve_mode == "sharded":         # (replicated: every rank already holds every sample)
            def gather(uns):                             # (ragged all-gather on one device: dist.gather_rows)
                return self.D.gather_rows(np.asarray(uns, dtype=np.float64).reshape(-1, 3))
        # the samples are stored sorted by node tuple; the greedy subsampling depends on the order of its input, so it is
        # fed in a canonical order (by position) that does not depend on the slab partition
        def canon(uns):
            if gather is not None:
                uns = gather(uns)
            if len(uns) == 0:
                return uns
            return uns[np.lexsort((uns[:, 2], uns[:, 1], uns[:, 0]))]
        vidx, P2, Q2, W2, lookup, n_new = _graph.update_graph_device(sv.node_pos, sv.node_dq, sv.node_w, pts, radius, self.knn,
                                                                    gather_unsupported=canon)
        if n_new == 0:
            return 0
        self.fs.set_graph(P2, Q2, W2)                    # resets the block pattern; node-node table of the regulariser rebuilt
        N = int(P2.shape[0])
        R = self.R
        self.ws_dqb = kernels.dqb_workspace((R, R, R), (self.a, self.b), knn=self.knn, n_nodes=N)
        self.knn_bricks = None
        if self.b > self.a:
            kernels.dqb_build_candidates(self.ws_dqb, (R, R, R), P2, self.knn, (self.a, self.b))
            self.knn_bricks = ((R, R, R), (self.a, self.b), self.ws_dqb)
        self._first = True                               # K3 stores its per-voxel neighbourhoods again on the next call
        self.refresh_samples()
        return n_new

    def step(self, depth, lw_cam, gn_iters=10, rw=5.0, lm_abs=10.0, lm_rel=1e-2, max_dist=2.0, huber=0.5, stage_ms=None,
             update_graph=False, on_updated=None, data_views=None, relax=None, global_iters=None, global_lm=0.1, global_stride=None):
        """Defaults (regulariser weight, LM damping, association gate and Huber threshold in voxels, the per-frame decay of the warp
        field `relax` = 0.8) are the ones under which the loop follows a +-0.6 voxel oscillation of the bench scene with a BOUNDED
        warp field: max node translation 0.85 voxel at frame 400, 1.01 at frame 1 200, sample count constant (tools/soak.py;
        tests/test_gpu_pipeline.py::test_soak_300_frames runs 300 of them).  Without the decay (relax = 1, round 3) nothing
        pulls a node back: 2.3 voxels at frame 400 and growing, the |T| < band shell thickens and the sample count doubles.  With
        a 4-voxel gate and weak damping, nodes without data support wander and the TSDF update corrupts the canonical volume.
        stage_ms: optional dict; when given, the device is synchronised after every stage and the stage's wall
        time (ms) is added under its name (for breakdowns only: the syncs cost throughput).
        on_updated: optional callable, called once the launches of this frame's TSDF update are queued and `self.updated` is
        recorded -- the place where a consumer of the updated canonical slab (mesh extraction on another stream) queues its
        first launches, ahead of the sample refresh's."""
        import time as _t
        t0 = [_t.perf_counter()]

        def mark(name):
            if stage_ms is not None:
                torch.cuda.synchronize()
                now = _t.perf_counter()
                stage_ms[name] = stage_ms.get(name, 0.0) + (now - t0[0]) * 1e3
                t0[0] = now
        R = self.R
        # `depth` / `lw_cam` may be lists: the live volume is then fused from all of them (as the reference's
        # compute_live_tsdf does, core/fusion_dm.py:166-170) and the warp solve associates every sample against all of them
        # (the view in which it lies closest to the observed surface: dfh_gn_associate_views; data_views=1 keeps round 2's
        # first-view-only data term)
        depth_list = list(depth) if isinstance(depth, (list, tuple)) else [depth]
        lw_list = list(lw_cam) if isinstance(depth, (list, tuple)) else [lw_cam]
        if len(depth_list) != len(lw_list):
            raise ValueError('length of camera matrix array must equal that of depth maps')
        depth, lw_cam = depth_list[0], lw_list[0]
        nd = len(depth_list) if data_views is None else max(1, min(int(data_views), len(depth_list)))
        solve_depth, solve_lw = (depth_list[:nd], lw_list[:nd]) if nd > 1 else (depth, lw_cam)
        def sweep_live():
            if self.ws_views is None:
                self.ws_views = kernels.integrate_workspace(min(len(depth_list), 16), depth.shape[0], depth.shape[1], (R, R, R), (self.a, self.b), self.live.device)
            kernels.integrate_depth_views(self.live, self.live_w, depth_list, self.K, self.Kinv, lw_list, self.scale, self.center,
                                          self.tdist_world, tsdf_res=R, res=(R, R, R), x_range=(self.a, self.b), workspace=self.ws_views,
                                          fresh=self.tvox)          # (the fill of the live volume is part of the sweep)
        # The solve reads the depth maps (projective association), not the live volume; the live volume depends on the depth maps
        # only and is first read by the TSDF update.  So the live-volume sweep (bandwidth-bound) runs on a side stream beside the
        # plan's launches AND the whole solve (bound by latency, ten waves per CU), and the streams join in front of the TSDF
        # update (round 4; rounds 2-3 joined before the first GN iteration).  With stage timing the order is sequential.
        joined = True
        if stage_ms is None and not _lib.opt_on("py_no_side_stream"):
            if self._side is None:
                self._side = torch.cuda.Stream()
            main = torch.cuda.current_stream()
            self._side.wait_stream(main)             # the previous frame's TSDF update has read the live volume
            with torch.cuda.stream(self._side):
                sweep_live()
            self.fs.solver.prepare()
            joined = False
        else:
            sweep_live()
        mark("live_tsdf")
        # the rigid mode first (two steps: one twist shared by all nodes, fitted to the data rows -- FrameSolver.global_iteration),
        # then the node iterations (one host call: nothing between their launches depends on the host)
        ng = self.GLOBAL_ITERS if global_iters is None else int(global_iters)
        if ng > 0:
            # (every `global_stride`-th tile of THIS rank's samples: with the samples sharded over ranks a stride > 1 thins another
            # subsample than one GPU does -- the same estimator on other data; stride 1 is partition-independent)
            self.fs.global_iteration(solve_depth, solve_lw, max_dist=max_dist, huber=huber, lm_rel=global_lm, n_iters=ng,
                                     stride=self.GLOBAL_STRIDE if global_stride is None else int(global_stride))
        self.fs.gn_iteration(solve_depth, solve_lw, rw=rw, lm_abs=lm_abs, lm_rel=lm_rel, max_dist=max_dist, huber=huber, n_iters=gn_iters)
        mark("solve")
        if not joined:
            torch.cuda.current_stream().wait_stream(self._side)
        live_full = self.D.allgather_planes(self.live, R) if self.ws > 1 else self.live
        mark("allgather")
        sv = self.fs.solver
        kernels.fuse_volume_dqb(self.T, self.Wt, live_full, sv.node_pos, sv.node_dq, sv.node_w, self.knn, self.ident_lw, self.tvox,
                                res=(R, R, R), x_range=(self.a, self.b), workspace=self.ws_dqb, rebuild_candidates=self._first)
        self._first = False
        # the warp field decays towards the identity once the TSDF update has used it (dfh_relax_twists: every node's motion
        # scaled by `relax` along its own screw).  Fusion.updateTSDF writes most of the motion into the canonical volume every
        # frame, and without a term that pulls a node back the field random-walks over hundreds of frames
        # (tools/soak.py; tests/test_gpu_pipeline.py::test_soak_300_frames).  relax = 1 keeps round 3's behaviour.
        rx = self.RELAX if relax is None else float(relax)
        if rx != 1.0:
            _lib.check(_lib.load().dfh_relax_twists(sv.node_dq.data_ptr(), int(sv.N), rx, current_stream_ptr()), "dfh_relax_twists")
        if self.updated is None:
            self.updated = torch.cuda.Event()
        self.updated.record()                  # the canonical slab of this frame is final from here on (a consumer on another
        if on_updated is not None:             # stream, e.g. mesh extraction, need not wait for the sample refresh below)
            on_updated()
        mark("tsdf_update")
        n = self.refresh_samples()
        mark("samples")
        self.fs.solver.check_status(completed_only=True)   # the sample count's read-back has synchronised: a timed-out PCG raises here
        if update_graph:
            if self.update_graph():
                n = self.fs.solver.S
            mark("graph")
        return n
