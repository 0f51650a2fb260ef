"""Host side of the warp-field solve: device buffers, the block-sparse pattern of J^T J and the
Gauss-Newton / Levenberg-Marquardt loop around the HIP kernels of csrc/dfh_solve.hip.

The reference's solve is scipy.optimize.least_squares on `computef` with finite-difference
Jacobians (core/fusion.py:327-412); what is kept from it is the residual definition, the
hyper-parameters (`regularization_weight`, its /8 relaxation while the cost reduction stays in
(5 %, 90 %), :405-412) and the call surface (see fusion.py / fusion_dm.py in this package).
torch is used for allocation and for the once-per-frame index bookkeeping (sorting samples by
node tuple, unique node pairs); every per-iteration step is a HIP kernel behind the C ABI.
"""

import ctypes

import numpy as np
import torch

from . import _lib, dist as _dist
from .device import HostScalar, current_stream_ptr, dtype_code, require_gpu


def _f64(a, shape_tail=None):
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64)))
    t = t.to(device="cuda", dtype=torch.float64).contiguous()
    if shape_tail is not None and tuple(t.shape[1:]) != tuple(shape_tail):
        raise ValueError("array has shape %s, expected (n,%s)" % (tuple(t.shape), ",".join(map(str, shape_tail))))
    return t


def _i32(a, shape_tail=None):
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(a)))
    t = t.to(device="cuda", dtype=torch.int32).contiguous()
    if shape_tail is not None and tuple(t.shape[1:]) != tuple(shape_tail):
        raise ValueError("index array has shape %s, expected (n,%s)" % (tuple(t.shape), ",".join(map(str, shape_tail))))
    return t


# ------------------------------------------------------------------------------ residual evaluators
def residual_rigid(x, verts, normals, corr):
    """FusionDM.computef_lw rows (reference core/fusion_dm.py:285-297) -> CUDA fp64 tensor."""
    require_gpu()
    lib = _lib.load()
    V, Nn, C = _f64(verts, (3,)), _f64(normals, (3,)), _f64(corr, (3,))
    if not (V.shape[0] == Nn.shape[0] == C.shape[0]):
        raise ValueError("vertices / normals / correspondences disagree in length")
    out = torch.empty(V.shape[0], dtype=torch.float64, device="cuda")
    _lib.check(lib.dfh_residual_rigid(V.data_ptr(), Nn.data_ptr(), C.data_ptr(), V.shape[0], _lib.darr(x, 8),
                                      out.data_ptr(), current_stream_ptr()), "dfh_residual_rigid")
    return out


def residual_data(node_dq, verts, normals, corr, nbr, node_pos, node_w, lw_dq):
    """Data rows of Fusion.computef / computef_lw (reference core/fusion.py:444-473)."""
    require_gpu()
    lib = _lib.load()
    V, Nn, C = _f64(verts, (3,)), _f64(normals, (3,)), _f64(corr, (3,))
    Q, P, Wn = _f64(node_dq, (8,)), _f64(node_pos, (3,)), _f64(node_w, ())
    nb = _i32(nbr)
    if nb.dim() != 2 or nb.shape[0] != V.shape[0]:
        raise ValueError("neighbour table must be (n_vertices, knn)")
    if not (V.shape[0] == Nn.shape[0] == C.shape[0]):
        raise ValueError("Please first call setupCorrespondences to compute point to point correspondences "
                         "between canonical and live frame vertices!")         # core/fusion.py:337-338
    if int(nb.max()) >= Q.shape[0] or int(nb.min()) < 0:
        raise ValueError("neighbour table refers to a node that does not exist")
    out = torch.empty(V.shape[0], dtype=torch.float64, device="cuda")
    _lib.check(lib.dfh_residual_data(V.data_ptr(), Nn.data_ptr(), C.data_ptr(), nb.data_ptr(), V.shape[0], nb.shape[1],
                                     Q.data_ptr(), P.data_ptr(), Wn.data_ptr(), Q.shape[0], _lib.darr(lw_dq, 8),
                                     out.data_ptr(), current_stream_ptr()), "dfh_residual_data")
    return out


def residual_reg(node_dq, node_nbr, node_pos, node_w, rw):
    """Regularisation rows of Fusion.computef (reference core/fusion.py:475-484), node-major."""
    require_gpu()
    lib = _lib.load()
    Q, P, Wn = _f64(node_dq, (8,)), _f64(node_pos, (3,)), _f64(node_w, ())
    nb = _i32(node_nbr)
    if nb.dim() != 2 or nb.shape[0] != Q.shape[0]:
        raise ValueError("node neighbour table must be (n_nodes, knn)")
    if int(nb.max()) >= Q.shape[0] or int(nb.min()) < 0:
        raise ValueError("node neighbour table refers to a node that does not exist")
    out = torch.empty(Q.shape[0] * nb.shape[1] * 3, dtype=torch.float64, device="cuda")
    _lib.check(lib.dfh_residual_reg(nb.data_ptr(), Q.shape[0], nb.shape[1], Q.data_ptr(), P.data_ptr(), Wn.data_ptr(),
                                    float(rw), out.data_ptr(), current_stream_ptr()), "dfh_residual_reg")
    return out


def warp_points(verts, normals, lw_dq, nbr=None, node_dq=None, node_pos=None, node_w=None):
    """Batch Fusion.warp (reference core/fusion.py:502-520): (warped points, warped normals) as CUDA
    fp64 tensors.  nbr=None applies only the global `lw_dq` (the FusionDM case)."""
    require_gpu()
    lib = _lib.load()
    V = _f64(verts, (3,))
    Nn = None if normals is None else _f64(normals, (3,))
    out_p = torch.empty_like(V)
    out_n = None if Nn is None else torch.empty_like(V)
    if nbr is not None:
        nb = _i32(nbr)
        Q, P, Wn = _f64(node_dq, (8,)), _f64(node_pos, (3,)), _f64(node_w, ())
        if nb.dim() != 2 or nb.shape[0] != V.shape[0]:
            raise ValueError("neighbour table must be (n_vertices, knn)")
        args = (nb.data_ptr(), V.shape[0], nb.shape[1], Q.data_ptr(), P.data_ptr(), Wn.data_ptr(), Q.shape[0])
    else:
        args = (0, V.shape[0], 1, 0, 0, 0, 0)
    _lib.check(lib.dfh_warp_points(V.data_ptr(), 0 if Nn is None else Nn.data_ptr(), *args, _lib.darr(lw_dq, 8),
                                   out_p.data_ptr(), 0 if out_n is None else out_n.data_ptr(), current_stream_ptr()),
               "dfh_warp_points")
    return out_p, out_n


def closest_correspondences(warped_pos, warped_nrm, live_verts, knn, tolerance):
    """Selection loop of setupCorrespondences (reference core/fusion_dm.py:229-244): returns
    (best live point (V,3), best cost (V,), keep (V,) uint8) as CUDA tensors."""
    require_gpu()
    lib = _lib.load()
    P, Nn, Lv = _f64(warped_pos, (3,)), _f64(warped_nrm, (3,)), _f64(live_verts, (3,))
    if P.shape[0] != Nn.shape[0]:
        raise ValueError("warped positions and normals disagree in length")
    corr = torch.empty_like(P)
    cost = torch.empty(P.shape[0], dtype=torch.float64, device="cuda")
    keep = torch.empty(P.shape[0], dtype=torch.uint8, device="cuda")
    _lib.check(lib.dfh_closest_correspondences(P.data_ptr(), Nn.data_ptr(), P.shape[0], Lv.data_ptr(), Lv.shape[0], int(knn),
                                               float(tolerance), corr.data_ptr(), cost.data_ptr(), keep.data_ptr(),
                                               current_stream_ptr()), "dfh_closest_correspondences")
    return corr, cost, keep


def sample_knn(sample_pos, node_pos, node_w, knn, bricks=None):
    """(nbr (S,k) int32, weights (S,k) fp64) of arbitrary points: k nearest nodes, nearest first.
    bricks = (res, (x0, x1), workspace): look the candidates up in the per-brick lists of a fuse_volume_dqb workspace
    whose lists were built for these node positions and this knn (kernels.dqb_build_candidates); same result."""
    require_gpu()
    lib = _lib.load()
    Sp, P, Wn = _f64(sample_pos, (3,)), _f64(node_pos, (3,)), _f64(node_w, ())
    S = Sp.shape[0]
    nbr = torch.empty((S, knn), dtype=torch.int32, device="cuda")
    wts = torch.empty((S, knn), dtype=torch.float64, device="cuda")
    if bricks is not None:
        res, (x0, x1), ws = bricks
        _lib.check(lib.dfh_sample_knn_bricks(Sp.data_ptr(), S, P.data_ptr(), Wn.data_ptr(), P.shape[0], int(knn), _lib.iarr(res),
                                             int(x0), int(x1), ws.data_ptr(), ws.numel() * ws.element_size(), nbr.data_ptr(),
                                             wts.data_ptr(), current_stream_ptr()), "dfh_sample_knn_bricks")
        return nbr, wts
    _lib.check(lib.dfh_sample_knn(Sp.data_ptr(), S, P.data_ptr(), Wn.data_ptr(), P.shape[0], int(knn), nbr.data_ptr(),
                                  wts.data_ptr(), current_stream_ptr()), "dfh_sample_knn")
    return nbr, wts


def solve_rigid_gn(x0, verts, normals, corr, valid=None, iters=10, lm=0.0):
    """Gauss-Newton on 0.5*|FusionDM.computef_lw(x)|^2 over the 6-DoF left twist of x.
    Returns (x, [cost before each step])."""
    require_gpu()
    lib = _lib.load()
    V, Nn, C = _f64(verts, (3,)), _f64(normals, (3,)), _f64(corr, (3,))
    val = None if valid is None else valid.to(device="cuda", dtype=torch.uint8).contiguous()
    out = torch.empty(44, dtype=torch.float64, device="cuda")
    x = np.asarray(x0, dtype=np.float64).copy()
    costs = []
    from .dq import dq_mul, twist_exp_dq
    for _ in range(iters):
        _lib.check(lib.dfh_gn_build_rigid(V.data_ptr(), Nn.data_ptr(), C.data_ptr(), 0 if val is None else val.data_ptr(),
                                          V.shape[0], _lib.darr(x, 8), out.data_ptr(), current_stream_ptr()),
                   "dfh_gn_build_rigid")
        h = out.cpu().numpy()
        A, g = h[:36].reshape(6, 6), h[36:42]
        costs.append(float(h[42]))
        dx = np.linalg.solve(A + lm * np.eye(6), -g)
        x = dq_mul(twist_exp_dq(dx), x)
    return x, costs


class _LazyLong:
    """int32 device permutation, widened to int64 (what tensor indexing wants) only if somebody indexes with it."""

    def __init__(self, idx32):
        self._i32, self._i64 = idx32, None

    def long(self):
        if self._i64 is None:
            self._i64 = self._i32.long()
        return self._i64


class _AsyncScalar:
    """A device scalar on its way to the host: the copy is queued now (into pinned memory, behind an event), the host waits
    only in get().  Whatever is launched in between runs while the host would otherwise sit in a stream synchronisation --
    after `.item()` the device idles for the ~35 us the host needs to issue its next launch."""
    _pool = {}

    def __init__(self, dev_scalar):
        key = dev_scalar.dtype
        free = _AsyncScalar._pool.setdefault(key, [])
        self._host = free.pop() if free else torch.empty(1, dtype=key).pin_memory()
        self._host.copy_(dev_scalar.reshape(1), non_blocking=True)
        self._event = torch.cuda.Event()
        self._event.record()

    def get(self):
        self._event.synchronize()
        v = int(self._host[0])
        _AsyncScalar._pool[self._host.dtype].append(self._host)
        self._host = None
        return v


# ------------------------------------------------------------------------------ non-rigid solver
class WarpSolver:
    """Gauss-Newton / LM solver for the node dual quaternions.

    set_graph()   node positions, DQs, weights and the node-node table of the regulariser
    set_samples() canonical sample points + normals; computes (or takes) their k nearest nodes and
                  static blend weights, sorts the samples by node tuple and builds the 6x6
                  block pattern of J^T J
    set_correspondences() / associate_depth()  fixed correspondences or projective association
    step()        one iteration: build -> PCG -> twist update (asynchronous)
    """

    def __init__(self, knn=4, pcg_iters=10, distributed=True):
        require_gpu()
        self.lib = _lib.load()
        self.knn = int(knn)
        self.pcg_iters = int(pcg_iters)
        self.distributed = bool(distributed)      # False: ignore an initialised process group
        self.force_collective = False             # True: run the sharded solve's pack / all-reduce / unpack even in a group of one
                                                  # (tools/rccl_capture_check.py: the RCCL path rehearsed on a one-GPU box)
        self.node_nbr = None
        self.S = 0
        if _dist.ranks_share_a_gpu():                  # (whether or not THIS solver runs collectives: ADVICE round 2)
            # several processes time-share this GPU (a rehearsal of the multi-GPU job on one card): co-residency of the
            # persistent PCG's workgroups is not guaranteed across processes -> two launches per iteration, no grid barrier
            _lib.check(self.lib.dfh_pcg_set_mode(2), "dfh_pcg_set_mode")

    def check_status(self, completed_only=False):
        """Raise DfhError if a persistent PCG solve since the last check timed out in its grid barrier (x = NaN; the twist
        update is all or nothing, so node_dq is what it was before the timed-out solve -- later iterations of the same call
        start from that field).  Synchronises; called where the host synchronises anyway (cost()).  completed_only: look only at
        the solves that have already completed and do not touch the device when none of them timed out (the end of
        SlabFrame.step, right after the sample count's read-back).  After a time-out this process takes the multi-launch
        PCG path."""
        import ctypes
        n = ctypes.c_long(0)
        fn = self.lib.dfh_pcg_status_peek if completed_only else self.lib.dfh_pcg_status
        rc = fn(current_stream_ptr(), ctypes.byref(n))
        if rc != 0:
            self.lib.dfh_pcg_set_mode(2)
            _lib.check(rc, "dfh_pcg_status")

    # -- graph -------------------------------------------------------------------------------
    def set_graph(self, node_pos, node_dq, node_w, node_nbr=None):
        self.node_pos = _f64(node_pos, (3,))
        self.node_dq = _f64(node_dq, (8,)).clone()
        self.node_w = _f64(node_w, ())
        self.N = self.node_pos.shape[0]
        if not (self.node_dq.shape[0] == self.N == self.node_w.shape[0]):
            raise ValueError("node arrays disagree on the number of nodes")
        self.node_nbr = None if node_nbr is None else _i32(node_nbr, (self.knn,))
        self._pattern = None
        self._pattern_keys = None

    # -- samples -----------------------------------------------------------------------------
    def set_samples(self, pos, nrm, nbr=None, weights=None, sort=True, knn_bricks=None):
        pos, nrm = _f64(pos, (3,)), _f64(nrm, (3,))
        if pos.shape[0] != nrm.shape[0]:
            raise ValueError("sample positions and normals disagree in length")
        if nbr is None:
            nbr, weights = sample_knn(pos, self.node_pos, self.node_w, self.knn, bricks=knn_bricks)
        else:
            nbr = _i32(nbr, (self.knn,))
            if weights is None:
                d = pos[:, None, :] - self.node_pos[nbr.long()]
                dist = torch.sqrt((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2])
                weights = torch.exp(-1.0 * (dist / (2 * self.node_w[nbr.long()])) ** 2)
            weights = _f64(weights, (self.knn,))
        self.order = None
        self._tuple_key = None
        if sort and pos.shape[0] > 0:
            # group samples with the same ordered node tuple: few runs per 256-sample tile.  The tuple is packed
            # into one int64 (base-N digits, lexicographic order preserved) so that a 1-D radix sort does it.
            if float(self.N) ** self.knn < 2.0 ** 62 and not _lib.opt_on("py_plan_torch"):
                pos, nrm, nbr, weights = self._sort_samples_device(pos, nrm, nbr, weights)      # pack + sort + permute behind one call
            else:
                key = self._pack_tuples(nbr)
                if key is not None:
                    key, self.order = torch.sort(key, stable=True)
                    self._tuple_key = key
                else:
                    _, inv = torch.unique(nbr, dim=0, return_inverse=True)
                    self.order = torch.argsort(inv, stable=True)
                pos, nrm, nbr, weights = self._permute(pos, nrm, nbr, weights)
        self.spos, self.snrm = pos.contiguous(), nrm.contiguous()
        self.snbr, self.swts = nbr.contiguous(), weights.contiguous()
        self.S = pos.shape[0]
        # (corr: every association path writes all S rows -- 0 where invalid -- and the builds read it for valid samples only: no
        # fill needed, 63 MB per frame at config 5; valid IS cleared: a build before any association must see no rows)
        self.corr = torch.empty((self.S, 3), dtype=torch.float64, device="cuda")
        self.valid = torch.zeros(self.S, dtype=torch.uint8, device="cuda")
        self._pattern = None

    def _sort_samples_device(self, pos, nrm, nbr, weights):
        """dfh_gn_sort_samples: the samples in the (stable) order of their node tuples; sets order and _tuple_key."""
        S = pos.shape[0]
        pos, nrm, nbr, weights = pos.contiguous(), nrm.contiguous(), nbr.contiguous(), weights.contiguous()
        out = (torch.empty_like(pos), torch.empty_like(nrm), torch.empty_like(nbr), torch.empty_like(weights))
        key = torch.empty(S, dtype=torch.int64, device="cuda")
        order = torch.empty(S, dtype=torch.int32, device="cuda")
        nbytes = self.lib.dfh_gn_sort_workspace_bytes(S)
        ws = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device="cuda")
        _lib.check(self.lib.dfh_gn_sort_samples(pos.data_ptr(), nrm.data_ptr(), nbr.data_ptr(), weights.data_ptr(), S, self.knn, self.N,
                                                out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), key.data_ptr(),
                                                order.data_ptr(), ws.data_ptr(), ws.numel() * 8, current_stream_ptr()), "dfh_gn_sort_samples")
        self._tuple_key = key
        self._order32 = order
        self.order = _LazyLong(order)
        return out

    def _permute(self, pos, nrm, nbr, weights):
        """The four per-sample arrays in `self.order`, one fused pass (dfh_permute_samples)."""
        pos, nrm, nbr, weights = pos.contiguous(), nrm.contiguous(), nbr.contiguous(), weights.contiguous()
        out = (torch.empty_like(pos), torch.empty_like(nrm), torch.empty_like(nbr), torch.empty_like(weights))
        order = self.order.contiguous()
        _lib.check(self.lib.dfh_permute_samples(order.data_ptr(), pos.shape[0], self.knn, pos.data_ptr(), nrm.data_ptr(), nbr.data_ptr(),
                                                weights.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                                out[3].data_ptr(), current_stream_ptr()), "dfh_permute_samples")
        return out

    def _pack_tuples(self, nbr):
        """(S,k) node ids -> int64 keys whose order is the lexicographic order of the tuples; None if k digits
        in base N do not fit."""
        if float(self.N) ** self.knn >= 2.0 ** 62:
            return None
        key = nbr[:, 0].long()
        for j in range(1, self.knn):
            key = key * self.N + nbr[:, j].long()
        return key

    def _unpack_tuples(self, key):
        cols = []
        for _ in range(self.knn):
            cols.append(key % self.N)
            key = key // self.N
        return torch.stack(cols[::-1], dim=1)

    def prepare(self, overlap=()):
        """Build whatever the next build() needs (block pattern, data plan) now instead of inside the first build().
        overlap: up to two callables that launch device work independent of the solver (the frame loop passes the clearing
        of the live volume and the live-volume sweep); they are called at the two points where the host has to wait for a
        count from the device, so that the device has work while the host catches up.  Not all of them need be called."""
        if self._pattern is None:
            self._build_pattern(overlap)

    def _build_pattern(self, overlap=()):
        """Block pattern of J^T J (diagonal, node pairs sharing a sample, regularisation pairs) and everything sized by
        it.  The pattern only GROWS while the graph stays: if the new samples' node pairs are all in the current
        pattern (the usual case from one frame to the next) it is kept -- blocks without contributions are exact
        zeros and change no result -- and only the data plan is rebuilt."""
        N, k = self.N, self.knn
        new = []
        old = getattr(self, "_pattern_keys", None)
        if old is not None:
            # the data plan looks every row's node pairs up in the pattern anyway: build it against the current
            # pattern and learn from the look-up whether all pairs were there (one flag, one synchronisation)
            nbytes = self.lib.dfh_pcg_workspace_bytes(N, self.pcg_iters)          # pcg_iters may have been raised
            if self.pcg_ws.numel() * 8 < nbytes:
                self.pcg_ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device="cuda")
            covered = self._build_plan(old, reg=False, overlap=overlap)
            if (_dist.all_ranks(covered) if self.distributed else covered):
                self._pattern = True
                return
            if self.S > 0:
                new.append(self._pair_keys_of_rows())
            new.append(old)
        elif self.S > 0:
            # distinct node tuples only (the pattern is a function of the tuples, not of the samples)
            if self._tuple_key is not None:
                tup = self._unpack_tuples(torch.unique_consecutive(self._tuple_key))
            else:
                tup = torch.unique(self.snbr, dim=0).long()
            new.append((tup[:, :, None] * N + tup[:, None, :]).reshape(-1))
        if old is None:
            new.append(torch.arange(N, device="cuda", dtype=torch.int64) * (N + 1))      # diagonal
            if self.node_nbr is not None:
                i = torch.arange(N, device="cuda", dtype=torch.int64)[:, None].expand(N, k)
                j = self.node_nbr.long()
                new += [(i * N + j).reshape(-1), (j * N + i).reshape(-1)]
        if self.S > 0 and N <= 4096 and old is not None:
            # the pattern had to grow, so the sample set is moving: add head-room once, so that the next sets are
            # covered and the pattern is kept -- every node pair no further apart than 1.15 x the widest pair that
            # shares a sample (all-zero blocks change no result and cost the PCG ~1 us per 1000 blocks and iteration;
            # rebuilding pattern and plans costs 0.5 ms per frame).  A solver that is set up once never comes here.
            pk = new[0]
            pi, pj = pk // N, pk % N
            lim = ((self.node_pos[pi] - self.node_pos[pj]) ** 2).sum(dim=1).max() * (1.15 * 1.15)
            dd = ((self.node_pos[:, None, :] - self.node_pos[None, :, :]) ** 2).sum(dim=2)
            near = torch.nonzero(dd <= lim)
            new.append(near[:, 0] * N + near[:, 1])
        keys = torch.cat(new)
        keys = _dist.union_sorted_keys(keys) if self.distributed else torch.unique(keys)   # sorted; same on every rank
        self._pattern_keys = keys
        rows = (keys // N).to(torch.int32)
        self.col = (keys % N).to(torch.int32).contiguous()
        self.row_ptr = torch.searchsorted(rows.contiguous(), torch.arange(N + 1, device="cuda", dtype=torch.int32)).to(torch.int32).contiguous()
        self.B = int(keys.numel())
        # one flat allocation {J^T J blocks | J^T r | cost, count}: a single all-reduce per iteration
        self.system = torch.zeros(self.B * 36 + 6 * N + 2, dtype=torch.float64, device="cuda")
        self.vals = self.system[:self.B * 36]
        self.rhs = self.system[self.B * 36:self.B * 36 + 6 * N]
        self.cost_count = self.system[self.B * 36 + 6 * N:]
        # the pattern's symmetry for the gather (dfh_gn_iteration_views: blk_upper): per block with col >= row its index and its
        # mirror block's (-1 on the diagonal); only those blocks' lists are walked, every sum is stored twice
        upper_m = self.col >= rows
        mkey_s = self.col.to(torch.int64) * N + rows.to(torch.int64)
        mirror_s = torch.searchsorted(keys, mkey_s).clamp(max=self.B - 1)
        self.blk_upper, self.n_upper = None, 0
        if bool((keys[mirror_s] == mkey_s).all()):                  # (symmetric by construction; if ever not: every block is walked)
            ids = torch.nonzero(upper_m).flatten()
            mir = torch.where(self.col[ids] == rows[ids], torch.full_like(ids, -1), mirror_s[ids])
            self.blk_upper = torch.stack([ids, mir], dim=1).to(torch.int32).contiguous()
            self.n_upper = int(ids.numel())
        # what the multi-GPU all-reduce carries: J^T J is symmetric, only the blocks with col >= row travel (dfh_gn_pack_upper)
        self._tri = None
        if self.distributed and (_dist.world()[1] > 1 or self.force_collective) and not _lib.opt_on("py_allreduce_full"):
            upper = self.col >= rows
            up_rank = (torch.cumsum(upper.to(torch.int64), 0) - 1)
            mkey = self.col.to(torch.int64) * N + rows.to(torch.int64)
            mirror = torch.searchsorted(keys, mkey).clamp(max=self.B - 1)
            if bool((keys[mirror] == mkey).all()):                  # (the pattern is symmetric by construction; if ever not: whole buffer)
                src = torch.where(upper, up_rank, up_rank[mirror]).to(torch.int32).contiguous()
                n_upper = int(upper.sum())
                self._tri = (rows.contiguous(), src, n_upper, torch.empty(36 * n_upper + 6 * N + 2, dtype=torch.float64, device="cuda"))
        self.dx = torch.empty(6 * N, dtype=torch.float64, device="cuda")
        nbytes = self.lib.dfh_pcg_workspace_bytes(N, self.pcg_iters)
        self.pcg_ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device="cuda")
        self._build_plan(keys)
        self._pattern = True

    def _build_plan(self, keys, reg=True, overlap=()):
        """Static part of the data term (dfh_gn_build_planned): rows = runs of equal node tuples inside a
        256-sample tile, and for every block / node the rows (and tuple slots) that contribute to it.
        reg=False keeps the regularisation lists (they depend on the pattern and the graph only).
        The data plan is built on the device (dfh_gn_plan_count / dfh_gn_plan_build: run scan, block look-up, two radix
        sorts); `keys` must be the pattern row_ptr / col describe.  Returns whether every node pair of every row is a
        block of the pattern (always True when reg=True: that call follows a pattern build)."""
        if self.S == 0 or self._tuple_key is None or _lib.opt_on("py_plan_torch"):
            return self._build_plan_torch(keys, reg)
        N, k, S, dev = self.N, self.knn, self.S, "cuda"
        tile = int(self.lib.dfh_gn_tile_samples())
        n_tiles = (S + tile - 1) // tile
        tile_off = torch.empty(n_tiles + 1, dtype=torch.int32, device=dev)
        # (the row count sizes the plan arrays: the scan kernel stores it straight into pinned host memory)
        n_rows_d = None if HostScalar.enabled else torch.empty(1, dtype=torch.int32, device=dev)
        pending = HostScalar(torch.int32) if HostScalar.enabled else None
        _lib.check(self.lib.dfh_gn_plan_count(self.snbr.data_ptr(), S, k, tile_off.data_ptr(),
                                              pending.ptr() if pending is not None else n_rows_d.data_ptr(), current_stream_ptr()),
                   "dfh_gn_plan_count")
        if pending is None:
            pending = _AsyncScalar(n_rows_d)
        if len(overlap) > 0:
            overlap[0]()
        R = pending.get()
        self.n_rows = R
        if R * k * k >= 2 ** 31:
            raise ValueError("too many sample runs for 32-bit plan entries")
        self.run_id = torch.empty(S, dtype=torch.int32, device=dev)
        self._row_first = torch.empty(R, dtype=torch.int32, device=dev)
        self.blk_ptr = torch.empty(self.B + 1, dtype=torch.int32, device=dev)
        self.blk_ent = torch.empty(R * k * k, dtype=torch.int32, device=dev)
        self.node_ptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
        self.node_ent = torch.empty(R * k, dtype=torch.int32, device=dev)
        # (the "pattern must grow" flag: one plain store of the plan's scan launch, into pinned host memory where enabled)
        unc_h = HostScalar(torch.int32) if HostScalar.enabled and not reg else None
        unc = torch.empty(1, dtype=torch.int32, device=dev) if unc_h is None else None
        nbytes = self.lib.dfh_gn_plan_workspace_bytes(R, k)
        ws = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=dev)
        _lib.check(self.lib.dfh_gn_plan_build(self.snbr.data_ptr(), S, k, N, tile_off.data_ptr(), R, self.row_ptr.data_ptr(),
                                              self.col.data_ptr(), self.B, self.run_id.data_ptr(), self._row_first.data_ptr(),
                                              self.blk_ptr.data_ptr(), self.blk_ent.data_ptr(), self.node_ptr.data_ptr(),
                                              self.node_ent.data_ptr(), unc.data_ptr() if unc_h is None else unc_h.ptr(), ws.data_ptr(), ws.numel() * 8,
                                              current_stream_ptr()), "dfh_gn_plan_build")
        ne = int(self.lib.dfh_gn_partial_doubles(k))
        self.partial = torch.empty(max(1, R * ne) + 2 * n_tiles + R, dtype=torch.float64, device=dev)   # rows | {cost, count} per tile | live flag per row
        self._pair_keys = None                                         # (computed on demand: only a growing pattern needs them)
        if reg:
            self._build_reg_plan(keys)
            return True
        pending = _AsyncScalar(unc) if unc_h is None else unc_h
        if len(overlap) > 1:
            overlap[1]()
        return pending.get() == 0

    def _pair_keys_of_rows(self):
        """Node-pair keys a * N + b of every row's tuple (what the block pattern has to contain)."""
        if getattr(self, "_pair_keys", None) is None:
            tup = self.snbr[self._row_first.long()].long()
            self._pair_keys = (tup[:, :, None] * self.N + tup[:, None, :]).reshape(-1)
        return self._pair_keys

    def _csr_lists(self, tup, keys):
        """CSR lists for the gather: per block the entries row*K^2 + sa*K + sb, per node the entries row*K + slot (torch)."""
        N, dev = self.N, "cuda"
        i32 = lambda t: t.to(torch.int32).contiguous()
        key = (tup[:, :, None] * N + tup[:, None, :]).reshape(-1)
        blk = torch.searchsorted(keys, key)
        covered = (keys[blk.clamp(max=keys.numel() - 1)] == key).all()                 # every pair is a block of the pattern?
        sblk, order = torch.sort(blk.to(torch.int32), stable=True)                     # 32-bit keys: half the sort traffic
        bp = torch.searchsorted(sblk, torch.arange(self.B + 1, device=dev, dtype=torch.int32), out_int32=True)
        snode, order2 = torch.sort(tup.reshape(-1).to(torch.int32), stable=True)
        npt = torch.searchsorted(snode, torch.arange(N + 1, device=dev, dtype=torch.int32), out_int32=True)
        return (bp, i32(order), npt, i32(order2)), key, covered

    def _build_reg_plan(self, keys):
        """Lists of the regulariser's pair rows (pattern + graph only: rebuilt with the pattern, not per frame)."""
        N, k, dev = self.N, self.knn, "cuda"
        self.partial_reg = None
        if self.node_nbr is not None:
            ii = torch.arange(N, device=dev, dtype=torch.int64)[:, None].expand(N, k).reshape(-1)
            pair = torch.stack([ii, self.node_nbr.long().reshape(-1)], dim=1)          # row t = i*k + slot
            (self.rblk_ptr, self.rblk_ent, self.rnode_ptr, self.rnode_ent), _, _ = self._csr_lists(pair, keys)
            self.partial_reg = torch.zeros(N * k * int(self.lib.dfh_gn_partial_doubles(2)), dtype=torch.float64, device=dev)

    def _build_plan_torch(self, keys, reg=True):
        """The same plan from torch ops (samples without packed tuple keys, S = 0, DFH_PLAN_TORCH=1 for A/B)."""
        N, k, S, dev = self.N, self.knn, self.S, "cuda"
        if S == 0:
            self.run_id = torch.zeros(1, dtype=torch.int32, device=dev)
            self.n_rows = 0
            tup = torch.zeros((0, k), dtype=torch.int64, device=dev)
            self._row_first = torch.zeros(0, dtype=torch.int32, device=dev)
        else:
            head = torch.empty(S, dtype=torch.bool, device=dev)
            head[1:] = (self._tuple_key[1:] != self._tuple_key[:-1]) if self._tuple_key is not None else \
                (self.snbr[1:] != self.snbr[:-1]).any(dim=1)
            head[::int(self.lib.dfh_gn_tile_samples())] = True        # a row never spans two tiles
            self.run_id = torch.cumsum(head, 0, dtype=torch.int32).sub_(1)
            self._row_first = torch.nonzero(head).reshape(-1).to(torch.int32)
            tup = self.snbr[head].long()
            self.n_rows = int(tup.shape[0])
        R = self.n_rows
        (self.blk_ptr, self.blk_ent, self.node_ptr, self.node_ent), self._pair_keys, covered = self._csr_lists(tup, keys)
        if reg:
            self._build_reg_plan(keys)
        ne = int(self.lib.dfh_gn_partial_doubles(k))
        tile = int(self.lib.dfh_gn_tile_samples())
        self.partial = torch.empty(max(1, R * ne) + 2 * ((S + tile - 1) // tile) + R, dtype=torch.float64, device=dev)   # rows | {cost, count} per tile | live flag per row
        if R * k * k >= 2 ** 31:
            raise ValueError("too many sample runs for 32-bit plan entries")
        return bool(covered) if not reg else True          # (reg=True is the call that follows a pattern build: covered by construction)

    def _order_index(self):
        return self.order.long() if isinstance(self.order, _LazyLong) else self.order

    # -- correspondences ---------------------------------------------------------------------
    def set_correspondences(self, corr, valid=None):
        c = _f64(corr, (3,))
        if c.shape[0] != self.S:
            raise ValueError("Please first call setupCorrespondences to compute point to point correspondences "
                             "between canonical and live frame vertices!")         # core/fusion.py:337-338
        self.corr = c[self._order_index()].contiguous() if self.order is not None else c
        if valid is None:
            self.valid = torch.ones(self.S, dtype=torch.uint8, device="cuda")
        else:
            v = valid if isinstance(valid, torch.Tensor) else torch.from_numpy(np.asarray(valid))
            v = v.to(device="cuda", dtype=torch.uint8)
            self.valid = v[self._order_index()].contiguous() if self.order is not None else v.contiguous()

    # -- several live views -----------------------------------------------------------------
    @staticmethod
    def _one_or_many(depth, lw_cam):
        """(depth, lw_cam) of a single view, or (list of depths, list of extrinsics) with more than one entry."""
        if isinstance(depth, (list, tuple)):
            depths, lws = list(depth), list(lw_cam)
            if len(depths) != len(lws) or not depths:
                raise ValueError('length of camera matrix array must equal that of depth maps')
            if len(depths) == 1:
                return depths[0], lws[0], False
            return depths, lws, True
        return depth, lw_cam, False

    def _views_table(self, depths, lw_cams):
        """Device table of a frame's views (dfh_gn_pack_views: extrinsics, their inverses, depth pointers), rebuilt only when
        the depth tensors or the extrinsics change (a frame's iterations share it).  Returns (table, n_views, H, W)."""
        for d in depths:
            if not (isinstance(d, torch.Tensor) and d.is_cuda and d.dim() == 2 and d.is_contiguous()):
                raise ValueError("depth must be a contiguous 2-D CUDA tensor")
            if d.shape != depths[0].shape or d.dtype != depths[0].dtype:
                raise ValueError("the depth maps of one frame must share a shape and a dtype")
        n = len(depths)
        lw = np.ascontiguousarray(np.stack([np.asarray(m, dtype=np.float64).reshape(12) for m in lw_cams]))
        key = (tuple(int(d.data_ptr()) for d in depths), lw.tobytes(), current_stream_ptr())
        if getattr(self, "_views_key", None) != key:
            ptrs = (ctypes.c_void_p * n)(*[int(d.data_ptr()) for d in depths])
            Hh, Ww = (int(v) for v in depths[0].shape)
            if depths[0].dtype == torch.float32:
                # with the per-view depth-cell tables: the fused build drops, per tile, the views that cannot hold a correspondence
                nbytes = self.lib.dfh_gn_views_bytes_cells(n, Hh, Ww)
                buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
                _lib.check(self.lib.dfh_gn_pack_views_cells(buf.data_ptr(), n, ptrs, Hh, Ww, lw.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                            current_stream_ptr()), "dfh_gn_pack_views_cells")
            else:
                nbytes = self.lib.dfh_gn_views_bytes(n)
                buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
                _lib.check(self.lib.dfh_gn_pack_views(buf.data_ptr(), n, ptrs, lw.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                      current_stream_ptr()), "dfh_gn_pack_views")
            self._views_key, self._views_buf, self._views_keep = key, buf, list(depths)
        H, W = depths[0].shape
        return self._views_buf, n, int(H), int(W)

    def associate_depth(self, depth, K, Kinv, lw_cam, scale, center, half, lw_dq, max_dist=0.0):
        """Projective association of the warped samples against a live depth map (CUDA tensor) -- or against several
        (lists of depth maps and extrinsics): each sample keeps the correspondence of the view in which it lies closest to
        the observed surface (dfh_gn_associate_views)."""
        depth, lw_cam, many = self._one_or_many(depth, lw_cam)
        if many:
            tab, n, H, W = self._views_table(depth, lw_cam)
            _lib.check(self.lib.dfh_gn_associate_views(self.spos.data_ptr(), self.snbr.data_ptr(), self.swts.data_ptr(), self.S, self.knn,
                                                       self.node_dq.data_ptr(), _lib.darr(lw_dq, 8), tab.data_ptr(), n, dtype_code(depth[0]),
                                                       H, W, _lib.darr(K, 9), _lib.darr(Kinv, 9), float(scale),
                                                       _lib.darr(np.asarray(center, dtype=np.float64), 3), float(half), float(max_dist),
                                                       self.corr.data_ptr(), self.valid.data_ptr(), current_stream_ptr()),
                       "dfh_gn_associate_views")
            return
        if not (isinstance(depth, torch.Tensor) and depth.is_cuda and depth.dim() == 2 and depth.is_contiguous()):
            raise ValueError("depth must be a contiguous 2-D CUDA tensor")
        H, W = depth.shape
        _lib.check(self.lib.dfh_gn_associate(self.spos.data_ptr(), self.snbr.data_ptr(), self.swts.data_ptr(), self.S, self.knn,
                                             self.node_dq.data_ptr(), _lib.darr(lw_dq, 8), depth.data_ptr(), dtype_code(depth),
                                             int(H), int(W), _lib.darr(K, 9), _lib.darr(Kinv, 9), _lib.darr(lw_cam, 12),
                                             float(scale), _lib.darr(np.asarray(center, dtype=np.float64), 3), float(half),
                                             float(max_dist), self.corr.data_ptr(), self.valid.data_ptr(),
                                             current_stream_ptr()), "dfh_gn_associate")

    # -- iteration ---------------------------------------------------------------------------
    def build(self, lw_dq, rw, huber=0.0):
        """J^T J (block-sparse), J^T r and the cost 0.5*|computef|^2 at the current node DQs."""
        if self._pattern is None:
            self._build_pattern()
        # data rows are sharded with the samples; the regularisation rows (one per node pair) are
        # not, so exactly one rank adds them before the all-reduce
        reg_here = (not self.distributed) or _dist.world()[0] == 0
        nn = 0 if (self.node_nbr is None or rw == 0.0 or not reg_here) else self.node_nbr.data_ptr()
        if _lib.opt_on("py_gn_atomic"):          # the atomics-based build (no plan needed), kept for A/B comparison
            _lib.check(self.lib.dfh_gn_build(self.spos.data_ptr(), self.snrm.data_ptr(), self.snbr.data_ptr(), self.swts.data_ptr(),
                                             self.corr.data_ptr(), self.valid.data_ptr(), self.S, self.knn,
                                             self.node_dq.data_ptr(), self.node_pos.data_ptr(), self.node_w.data_ptr(), nn, self.N,
                                             _lib.darr(lw_dq, 8), float(rw), self.row_ptr.data_ptr(), self.col.data_ptr(), self.B,
                                             self.vals.data_ptr(), self.rhs.data_ptr(), self.cost_count.data_ptr(),
                                             current_stream_ptr()), "dfh_gn_build")
        else:
            _lib.check(self.lib.dfh_gn_build_planned(
                self.spos.data_ptr(), self.snrm.data_ptr(), self.snbr.data_ptr(), self.swts.data_ptr(), self.corr.data_ptr(),
                self.valid.data_ptr(), self.S, self.knn, self.node_dq.data_ptr(), self.node_pos.data_ptr(), self.node_w.data_ptr(),
                nn, self.N, _lib.darr(lw_dq, 8), float(rw), self.row_ptr.data_ptr(), self.col.data_ptr(), self.B,
                self.vals.data_ptr(), self.rhs.data_ptr(), self.cost_count.data_ptr(), self.run_id.data_ptr(), self.n_rows,
                self.partial.data_ptr(), self.blk_ptr.data_ptr(), self.blk_ent.data_ptr(), self.node_ptr.data_ptr(),
                self.node_ent.data_ptr(), *((self.partial_reg.data_ptr(), self.rblk_ptr.data_ptr(), self.rblk_ent.data_ptr(),
                                             self.rnode_ptr.data_ptr(), self.rnode_ent.data_ptr())
                                            if self.partial_reg is not None else (0, 0, 0, 0, 0)),
                float(huber), current_stream_ptr()), "dfh_gn_build_planned")
        self._allreduce_system()                  # no-op on one GPU; samples are sharded by slab

    def build_associated(self, depth, K, Kinv, lw_cam, scale, center, half, lw_dq, rw, max_dist=0.0, huber=0.0):
        """associate_depth + build in one launch sequence (dfh_gn_build_planned_assoc): same corr / valid, same system, bit
        for bit; float32 depth maps and the planned build only (otherwise the two calls are made)."""
        if self._pattern is None:
            self._build_pattern()
        depth, lw_cam, many = self._one_or_many(depth, lw_cam)
        d0 = depth[0] if many else depth
        fused = (isinstance(d0, torch.Tensor) and d0.is_cuda and d0.dim() == 2 and d0.is_contiguous() and
                 d0.dtype == torch.float32 and self.S > 0 and not _lib.opt_on("py_gn_atomic") and not _lib.opt_on("py_gn_no_fused_assoc"))
        if not fused:
            self.associate_depth(depth, K, Kinv, lw_cam, scale, center, half, lw_dq, max_dist)
            return self.build(lw_dq, rw, huber)
        reg_here = (not self.distributed) or _dist.world()[0] == 0
        nn = 0 if (self.node_nbr is None or rw == 0.0 or not reg_here) else self.node_nbr.data_ptr()
        if many:
            tab, nv, H, W = self._views_table(depth, lw_cam)
            _lib.check(self.lib.dfh_gn_build_planned_assoc_views(
                self.spos.data_ptr(), self.snrm.data_ptr(), self.snbr.data_ptr(), self.swts.data_ptr(), self.corr.data_ptr(),
                self.valid.data_ptr(), self.S, self.knn, self.node_dq.data_ptr(), self.node_pos.data_ptr(), self.node_w.data_ptr(),
                nn, self.N, _lib.darr(lw_dq, 8), float(rw), self.row_ptr.data_ptr(), self.col.data_ptr(), self.B,
                self.vals.data_ptr(), self.rhs.data_ptr(), self.cost_count.data_ptr(), self.run_id.data_ptr(), self.n_rows,
                self.partial.data_ptr(), self.blk_ptr.data_ptr(), self.blk_ent.data_ptr(), self.node_ptr.data_ptr(),
                self.node_ent.data_ptr(), *((self.partial_reg.data_ptr(), self.rblk_ptr.data_ptr(), self.rblk_ent.data_ptr(),
                                             self.rnode_ptr.data_ptr(), self.rnode_ent.data_ptr())
                                            if self.partial_reg is not None else (0, 0, 0, 0, 0)),
                float(huber), tab.data_ptr(), nv, H, W, _lib.darr(K, 9), _lib.darr(Kinv, 9), float(scale),
                _lib.darr(np.asarray(center, dtype=np.float64), 3), float(half), float(max_dist),
                self.blk_upper.data_ptr() if self.blk_upper is not None else 0, self.n_upper, current_stream_ptr()),
                "dfh_gn_build_planned_assoc_views")
            self._allreduce_system()
            return
        H, W = depth.shape
        _lib.check(self.lib.dfh_gn_build_planned_assoc(
            self.spos.data_ptr(), self.snrm.data_ptr(), self.snbr.data_ptr(), self.swts.data_ptr(), self.corr.data_ptr(),
            self.valid.data_ptr(), self.S, self.knn, self.node_dq.data_ptr(), self.node_pos.data_ptr(), self.node_w.data_ptr(),
            nn, self.N, _lib.darr(lw_dq, 8), float(rw), self.row_ptr.data_ptr(), self.col.data_ptr(), self.B,
            self.vals.data_ptr(), self.rhs.data_ptr(), self.cost_count.data_ptr(), self.run_id.data_ptr(), self.n_rows,
            self.partial.data_ptr(), self.blk_ptr.data_ptr(), self.blk_ent.data_ptr(), self.node_ptr.data_ptr(),
            self.node_ent.data_ptr(), *((self.partial_reg.data_ptr(), self.rblk_ptr.data_ptr(), self.rblk_ent.data_ptr(),
                                         self.rnode_ptr.data_ptr(), self.rnode_ent.data_ptr())
                                        if self.partial_reg is not None else (0, 0, 0, 0, 0)),
            float(huber), depth.data_ptr(), int(H), int(W), _lib.darr(K, 9), _lib.darr(Kinv, 9), _lib.darr(lw_cam, 12), float(scale),
            _lib.darr(np.asarray(center, dtype=np.float64), 3), float(half), float(max_dist), current_stream_ptr()),
            "dfh_gn_build_planned_assoc")
        self._allreduce_system()

    def iterate_associated(self, depth, K, Kinv, lw_cam, scale, center, half, lw_dq, rw, max_dist=0.0, huber=0.0, lm_abs=0.0, lm_rel=0.0,
                           n_iters=1, n_global=0, global_lm=0.1):
        """n_iters GN iterations: build_associated + solve_update each.  On one GPU (no all-reduce between the halves) they
        are ONE call, dfh_gn_iteration_views (the views' table holds one or several depth maps): the iterations are queued
        back to back without returning to Python, and in each the clearing of the solve's workspace rides in the data-row
        launch -- the same bits as separate calls.  option py_gn_iter_per_call: one dfh_gn_iteration call per iteration (single
        view), as before round 3."""
        if self._pattern is None:
            self._build_pattern()
        depth, lw_cam, many = self._one_or_many(depth, lw_cam)
        d0 = depth[0] if many else depth
        one_call = (isinstance(d0, torch.Tensor) and d0.is_cuda and d0.dim() == 2 and d0.is_contiguous() and
                    d0.dtype == torch.float32 and self.S > 0 and not _lib.opt_on("py_gn_atomic") and
                    not _lib.opt_on("py_gn_no_fused_assoc") and not _lib.opt_on("py_gn_no_fused_iter") and
                    not (self.distributed and (_dist.world()[1] > 1 or self.force_collective)))
        if not one_call:
            for _ in range(int(n_global)):                # the rigid mode first (global_step): one twist for all nodes
                self.build_associated(depth, K, Kinv, lw_cam, scale, center, half, lw_dq, rw, max_dist, huber)
                self.global_step(global_lm)
            for _ in range(int(n_iters)):
                self.build_associated(depth, K, Kinv, lw_cam, scale, center, half, lw_dq, rw, max_dist, huber)
                self.solve_update(lm_abs, lm_rel)
            return
        nn = 0 if (self.node_nbr is None or rw == 0.0) else self.node_nbr.data_ptr()
        common = (self.spos.data_ptr(), self.snrm.data_ptr(), self.snbr.data_ptr(), self.swts.data_ptr(), self.corr.data_ptr(),
                  self.valid.data_ptr(), self.S, self.knn, self.node_dq.data_ptr(), self.node_pos.data_ptr(), self.node_w.data_ptr(),
                  nn, self.N, _lib.darr(lw_dq, 8), float(rw), self.row_ptr.data_ptr(), self.col.data_ptr(), self.B,
                  self.vals.data_ptr(), self.rhs.data_ptr(), self.cost_count.data_ptr(), self.run_id.data_ptr(), self.n_rows,
                  self.partial.data_ptr(), self.blk_ptr.data_ptr(), self.blk_ent.data_ptr(), self.node_ptr.data_ptr(),
                  self.node_ent.data_ptr(), *((self.partial_reg.data_ptr(), self.rblk_ptr.data_ptr(), self.rblk_ent.data_ptr(),
                                               self.rnode_ptr.data_ptr(), self.rnode_ent.data_ptr())
                                              if self.partial_reg is not None else (0, 0, 0, 0, 0)),
                  float(huber))
        tail = (self.pcg_iters, float(lm_abs), float(lm_rel), self.dx.data_ptr(), self.pcg_ws.data_ptr(), self.pcg_ws.numel() * 8, 1.0)
        if not many and _lib.opt_on("py_gn_iter_per_call"):
            for _ in range(int(n_global)):
                self.build_associated(depth, K, Kinv, lw_cam, scale, center, half, lw_dq, rw, max_dist, huber)
                self.global_step(global_lm)
            H, W = depth.shape
            for _ in range(int(n_iters)):
                _lib.check(self.lib.dfh_gn_iteration(
                    *common, depth.data_ptr(), int(H), int(W), _lib.darr(K, 9), _lib.darr(Kinv, 9), _lib.darr(lw_cam, 12), float(scale),
                    _lib.darr(np.asarray(center, dtype=np.float64), 3), float(half), float(max_dist), *tail, current_stream_ptr()),
                    "dfh_gn_iteration")
            return
        tab, nv, H, W = self._views_table(depth if many else [depth], lw_cam if many else [lw_cam])
        if int(n_global) > 0:
            if getattr(self, "_global_ws", None) is None:
                self._global_ws = torch.zeros((self.lib.dfh_gn_global_step_bytes() + 7) // 8, dtype=torch.float64, device="cuda")
                self.global_xi = torch.zeros(8, dtype=torch.float64, device="cuda")
            _lib.check(self.lib.dfh_gn_frame_solve_views(
                *common, tab.data_ptr(), nv, H, W, _lib.darr(K, 9), _lib.darr(Kinv, 9), float(scale),
                _lib.darr(np.asarray(center, dtype=np.float64), 3), float(half), float(max_dist), *tail, int(n_iters),
                self.blk_upper.data_ptr() if self.blk_upper is not None else 0, self.n_upper, int(n_global), float(global_lm),
                self.global_xi.data_ptr(), self._global_ws.data_ptr(), self._global_ws.numel() * 8, current_stream_ptr()),
                "dfh_gn_frame_solve_views")
            return
        _lib.check(self.lib.dfh_gn_iteration_views(
            *common, tab.data_ptr(), nv, H, W, _lib.darr(K, 9), _lib.darr(Kinv, 9), float(scale),
            _lib.darr(np.asarray(center, dtype=np.float64), 3), float(half), float(max_dist), *tail, int(n_iters),
            self.blk_upper.data_ptr() if self.blk_upper is not None else 0, self.n_upper, current_stream_ptr()),
            "dfh_gn_iteration_views")

    def _allreduce_system(self):
        """Sum of the normal equations over ranks, in place: the upper block triangle + J^T r + {cost, count} in one collective
        (pack, all-reduce of ~55 % of the system's bytes, unpack with the mirrored blocks transposed), or the whole flat buffer
        (DFH_ALLREDUCE_FULL=1).  One GPU / replicated solve: nothing."""
        if not self.distributed or (_dist.world()[1] == 1 and not self.force_collective):
            return
        if self._tri is None:
            _dist.allreduce_system(self.system, force=self.force_collective)
            return
        rows, src, n_upper, packed = self._tri
        _lib.check(self.lib.dfh_gn_pack_upper(self.system.data_ptr(), rows.data_ptr(), self.col.data_ptr(), src.data_ptr(), self.B, self.N,
                                              n_upper, packed.data_ptr(), current_stream_ptr()), "dfh_gn_pack_upper")
        _dist.allreduce_system(packed, force=self.force_collective)
        _lib.check(self.lib.dfh_gn_unpack_upper(self.system.data_ptr(), rows.data_ptr(), self.col.data_ptr(), src.data_ptr(), self.B, self.N,
                                                n_upper, packed.data_ptr(), current_stream_ptr()), "dfh_gn_unpack_upper")

    def solve_update(self, lm_abs=0.0, lm_rel=0.0):
        """PCG + twist update of the node DQs for the system of the last build (asynchronous)."""
        _lib.check(self.lib.dfh_pcg_solve_update(self.row_ptr.data_ptr(), self.col.data_ptr(), self.vals.data_ptr(), self.rhs.data_ptr(),
                                                 self.N, self.pcg_iters, float(lm_abs), float(lm_rel), self.dx.data_ptr(),
                                                 self.pcg_ws.data_ptr(), self.pcg_ws.numel() * 8, self.node_dq.data_ptr(), 1.0,
                                                 current_stream_ptr()), "dfh_pcg_solve_update")

    def solve_linear(self, lm_abs=0.0, lm_rel=0.0):
        _lib.check(self.lib.dfh_pcg_solve(self.row_ptr.data_ptr(), self.col.data_ptr(), self.vals.data_ptr(), self.rhs.data_ptr(),
                                          self.N, self.pcg_iters, float(lm_abs), float(lm_rel), self.dx.data_ptr(),
                                          self.pcg_ws.data_ptr(), self.pcg_ws.numel() * 8, current_stream_ptr()),
                   "dfh_pcg_solve")

    def global_sampled(self, depth, K, Kinv, lw_cam, scale, center, half, lw_dq, max_dist=0.0, huber=0.0, lm_rel=0.1, n_steps=1, stride=1):
        """n_steps rigid-mode steps straight from the samples (dfh_gn_global_sampled_views): ONE twist shared by all nodes, fitted
        to the data rows of every `stride`-th tile (no built system, no regulariser), applied to every node.  depth / lw_cam:
        one view or lists.  Samples sharded over ranks: the 29 sums are all-reduced, every rank applies the same twist.
        Asynchronous; self.global_xi holds the last step's twist | objective | valid count."""
        depth, lw_cam, many = self._one_or_many(depth, lw_cam)
        tab, nv, H, W = self._views_table(depth if many else [depth], lw_cam if many else [lw_cam])
        if getattr(self, "global_xi", None) is None or self.global_xi.numel() < 8:
            self.global_xi = torch.zeros(8, dtype=torch.float64, device="cuda")
        nbytes = self.lib.dfh_gn_global_sampled_bytes(self.S, int(stride))
        if getattr(self, "_gs_ws", None) is None or self._gs_ws.numel() * 8 < nbytes:
            self._gs_ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device="cuda")
        sharded = self.distributed and (_dist.world()[1] > 1 or self.force_collective)
        args = (self.spos.data_ptr(), self.snrm.data_ptr(), self.snbr.data_ptr(), self.swts.data_ptr(), self.S, self.knn, self.node_dq.data_ptr(),
                self.N, _lib.darr(lw_dq, 8), float(huber), tab.data_ptr(), nv, H, W, _lib.darr(K, 9), _lib.darr(Kinv, 9), float(scale),
                _lib.darr(np.asarray(center, dtype=np.float64), 3), float(half), float(max_dist), int(stride), float(lm_rel))
        if not sharded:
            _lib.check(self.lib.dfh_gn_global_sampled_views(*args, int(n_steps), self.global_xi.data_ptr(), 0, self._gs_ws.data_ptr(),
                                                            self._gs_ws.numel() * 8, current_stream_ptr()), "dfh_gn_global_sampled_views")
            return
        if getattr(self, "_gs_sums", None) is None:
            self._gs_sums = torch.zeros(32, dtype=torch.float64, device="cuda")
        for _ in range(int(n_steps)):
            _lib.check(self.lib.dfh_gn_global_sampled_views(*args, 1, 0, self._gs_sums.data_ptr(), self._gs_ws.data_ptr(),
                                                            self._gs_ws.numel() * 8, current_stream_ptr()), "dfh_gn_global_sampled_views")
            _dist.allreduce_system(self._gs_sums, force=self.force_collective)
            _lib.check(self.lib.dfh_gn_global_apply(self._gs_sums.data_ptr(), float(lm_rel), self.N, self.node_dq.data_ptr(),
                                                    self.global_xi.data_ptr(), current_stream_ptr()), "dfh_gn_global_apply")

    def global_step(self, lm_rel=0.1):
        """The rigid mode of the last build, solved on its own and applied to every node (dfh_gn_global_step): one twist shared
        by all nodes.  Asynchronous; the twist is left in self.global_xi (6 doubles on the device)."""
        if getattr(self, "_global_ws", None) is None:
            self._global_ws = torch.zeros((self.lib.dfh_gn_global_step_bytes() + 7) // 8, dtype=torch.float64, device="cuda")
            self.global_xi = torch.zeros(8, dtype=torch.float64, device="cuda")
        _lib.check(self.lib.dfh_gn_global_step(self.vals.data_ptr(), self.B, self.rhs.data_ptr(), self.N, float(lm_rel), self.node_dq.data_ptr(),
                                               self.global_xi.data_ptr(), self._global_ws.data_ptr(), self._global_ws.numel() * 8,
                                               current_stream_ptr()), "dfh_gn_global_step")

    def apply(self, step=1.0):
        _lib.check(self.lib.dfh_apply_twist(self.node_dq.data_ptr(), self.dx.data_ptr(), self.N, float(step),
                                            current_stream_ptr()), "dfh_apply_twist")

    def step(self, lw_dq, rw, lm_abs=0.0, lm_rel=0.0, huber=0.0):
        """One asynchronous GN iteration (no host synchronisation).  huber > 0: Huber IRLS weights on the data rows."""
        self.build(lw_dq, rw, huber)
        _lib.check(self.lib.dfh_pcg_solve_update(self.row_ptr.data_ptr(), self.col.data_ptr(), self.vals.data_ptr(), self.rhs.data_ptr(),
                                                 self.N, self.pcg_iters, float(lm_abs), float(lm_rel), self.dx.data_ptr(),
                                                 self.pcg_ws.data_ptr(), self.pcg_ws.numel() * 8, self.node_dq.data_ptr(), 1.0,
                                                 current_stream_ptr()), "dfh_pcg_solve_update")

    def cost(self):
        """(0.5*|r|^2, valid sample count) of the last build (synchronises; raises if a PCG solve timed out)."""
        self.check_status()
        h = self.cost_count.cpu().numpy()
        return float(h[0]), int(h[1])

    def dense_normal_equations(self):
        """Dense (6N x 6N) copy of the last build, for tests."""
        N = self.N
        A = torch.zeros((N, N, 6, 6), dtype=torch.float64, device="cuda")
        rows = torch.repeat_interleave(torch.arange(N, device="cuda"), (self.row_ptr[1:] - self.row_ptr[:-1]).long())
        A[rows, self.col.long()] = self.vals.view(self.B, 6, 6)
        return A.permute(0, 2, 1, 3).reshape(6 * N, 6 * N).cpu().numpy(), self.rhs.cpu().numpy()

    def solve_lm(self, lw_dq, rw, iters=10, lm_abs=1e-6, lm_rel=0.0, adaptive=True, huber=0.0):
        """Levenberg-Marquardt loop.  A step that raises the cost is undone and retried with 10x
        the damping; an accepted step relaxes the damping by 3x.  Returns the costs
        [initial, after step 1, ...] (0.5*|computef|^2 over valid rows; with huber > 0 the Huber objective of the data
        rows, minimised by iteratively re-weighted steps)."""
        lam_a, lam_r = float(lm_abs), float(lm_rel)
        self.build(lw_dq, rw, huber)
        c_cur, _ = self.cost()
        costs = [c_cur]
        for _ in range(iters):
            saved_dq = self.node_dq.clone()
            saved_vals, saved_rhs = self.vals.clone(), self.rhs.clone()
            accepted = False
            for _attempt in range(8):
                self.vals.copy_(saved_vals)
                self.rhs.copy_(saved_rhs)
                self.solve_linear(lam_a, lam_r)
                self.apply()
                self.build(lw_dq, rw, huber)                   # cost at the trial point = next iteration's system
                c_new, _ = self.cost()
                if not adaptive or c_new <= c_cur * (1 + 1e-12):
                    accepted = True
                    break
                self.node_dq.copy_(saved_dq)
                lam_a, lam_r = max(lam_a, 1e-9) * 10.0, max(lam_r, 1e-6) * 10.0
            if not accepted:
                self.build(lw_dq, rw, huber)
                break
            c_cur = c_new
            costs.append(c_cur)
            lam_a, lam_r = lam_a / 3.0, lam_r / 3.0
        return costs
