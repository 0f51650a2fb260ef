"""`FusionDM` with the reference's call surface (reference core/fusion_dm.py:53-354),
running its per-voxel hot methods as HIP kernels on an MI355X.

The reference's own device plug-in is a subclass overriding one numpy-in / numpy-out
method (class FusionDM_GPU, core/fusion_dm.py:563-574,600); this class keeps the same
method names, argument order, in-place + return ownership and ValueError behaviour, so
`FusionDM(tdist, K, tsdf_res=...)` / `.fuseDepths(...)` / `.compute_live_tsdf(...)` /
`.updateTSDF(...)` / `.computef_lw(...)` / `.solve(...)` call sites keep working.

What differs, on purpose:
  * volumes live on the GPU as float32 (16 B/voxel read-modify-write; pass
    `volume_dtype=np.float64` for bit-for-bit float64 volumes).  `_tsdf` / `_tsdfw` are
    properties that download on read and upload on assignment;
  * methods accept CUDA tensors wherever the reference takes numpy arrays, and then work
    in place without host round trips;
  * semantics are the reference's CPU path (fuseDepths :180-217), not its OpenCL kernel
    (:600-737), which computes something else;
  * no CPU fallback: without the HIP library / a GPU the hot methods raise.
"""
import os

import numpy as np
import torch
from numpy import linalg as la

from . import kernels, mesh as _mesh, solve as _solve
from .device import f32_exact, require_gpu, to_device, torch_dtype


def _is_tensor(a):
    return isinstance(a, torch.Tensor)


class FusionDM:
    def __init__(self, trunc_distance, K, tsdf_res=256, subsample_rate=5.0, knn=4, marching_cubes_step_size=3,
                 verbose=False, write_warpfield=True, volume_dtype=np.float32):
        # attribute set of the reference ctor, core/fusion_dm.py:57-81
        self._itercounter = 0
        self._curr_tsdf = None
        self._tdist = abs(trunc_distance)
        self._tsdf_res = tsdf_res
        self._vol_dtype = torch_dtype(volume_dtype)
        self._T = None          # device volumes, allocated lazily (ctor must work without a GPU)
        self._Wt = None
        self._lw = np.array([1, 0, 0, 0, 0, 0, 0, 0], dtype=np.float32)

        K = np.asarray(K, dtype=np.float64)
        if K.shape != (3, 3):
            raise ValueError('intrinsic matrix K must be 3x3')
        self._K = K
        self._Kinv = la.inv(K)

        self._IND = np.eye(4)
        self._INDinv = la.inv(self._IND)

        self._knn = knn
        self._marching_cubes_step_size = marching_cubes_step_size
        self._subsample_rate = subsample_rate
        self._nodes = []
        self._neighbor_look_up = []
        self._correspondences = []
        self._corridx = []
        self._vertices = None
        self._normals = None
        self._kdtree = None
        self._verbose = verbose
        self._write_warpfield = write_warpfield

    def verbose_gpu(self):
        """What the reference's device plug-in prints about its OpenCL devices (FusionDM_GPU.verbose_gpu, core/fusion_dm.py:576-598),
        for the HIP devices this process sees: name, compute units, clock, memory, LDS and workgroup limits, the library's ABI."""
        from . import _lib
        print('\n' + '=' * 60 + '\nHIP devices (libdfusion_hip ABI %d)' % _lib.ABI_VERSION)
        if not torch.cuda.is_available():
            print('    none visible')
        for i in range(torch.cuda.device_count()):
            p = torch.cuda.get_device_properties(i)
            print('=' * 60)
            print('    Device %d - Name:  %s (%s)' % (i, p.name, getattr(p, "gcnArchName", "")))
            print('    Device - Compute Units:  {0}'.format(p.multi_processor_count))
            print('    Device - Max Clock Speed:  {0:.0f} Mhz'.format(getattr(p, "clock_rate", 0) / 1e3))
            print('    Device - Local Memory:  {0:.0f} KB'.format(getattr(p, "shared_memory_per_block", 0) / 1024.0))
            print('    Device - Global Memory: {0:.0f} GB'.format(p.total_memory / 1073741824.0))
            print('    Device - Max Work Group Size: {0:.0f}'.format(getattr(p, "max_threads_per_block", 1024)))
        print('\n')

    # ------------------------------------------------------------------ volumes
    def _new_volume_pair(self, res=None):
        require_gpu()
        r = self._tsdf_res if res is None else res
        shape = (r, r, r) if np.isscalar(r) else tuple(r)
        T = torch.full(shape, self._tdist, dtype=self._vol_dtype, device="cuda")      # :61 / :100
        Wt = torch.zeros(shape, dtype=self._vol_dtype, device="cuda")                 # :62 / :101
        return T, Wt

    def _ensure_volumes(self):
        if self._T is None or self._Wt is None:
            T, Wt = self._new_volume_pair()
            self._T = T if self._T is None else self._T
            self._Wt = Wt if self._Wt is None else self._Wt

    @property
    def _tsdf(self):
        self._ensure_volumes()
        return self._T.cpu().numpy().astype(np.float64)

    @_tsdf.setter
    def _tsdf(self, value):
        self._T = to_device(value, dtype=self._vol_dtype)

    @property
    def _tsdfw(self):
        self._ensure_volumes()
        return self._Wt.cpu().numpy().astype(np.float64)

    @_tsdfw.setter
    def _tsdfw(self, value):
        self._Wt = to_device(value, dtype=self._vol_dtype)

    @property
    def tsdf_device(self):
        """The resident TSDF / weight tensors (no copy)."""
        self._ensure_volumes()
        return self._T, self._Wt

    # ------------------------------------------------------------------ A1
    def _depth_to_device(self, dm):
        if _is_tensor(dm):
            if dm.dim() != 2:
                raise ValueError('depth map must be 2-D')
            d = dm if dm.dtype in (torch.float32, torch.float64) else dm.to(torch.float32)
            return to_device(d)
        dm = np.asarray(dm)
        if dm.ndim != 2:
            raise ValueError('depth map must be 2-D')
        # float32 on device unless that would change a value (then masks could flip)
        return to_device(dm, dtype=torch.float32 if f32_exact(dm) else torch.float64)

    def fuseDepths_ocl(self, dm, lw, tsdf, tsdf_w, wmax=100.0):
        """The reference's OTHER fuseDepths: FusionDM_GPU's OpenCL kernel (core/fusion_dm.py:600-737), whose arithmetic differs
        from the CPU path (float32 throughout, bilinear depth, index -> pixel through `_IND`, pixels
        without depth carve free space, opposite sign, no `scale`).  Like the reference's host code it leaves its inputs alone
        and returns NEW float32 arrays (:690-691,:737); CUDA tensors in, CUDA tensors out."""
        lw = np.asarray(lw, dtype=np.float64)
        if lw.shape != (3, 4):
            raise ValueError('lw must be a 3x4 camera extrinsic')
        if _is_tensor(tsdf) != _is_tensor(tsdf_w):
            raise ValueError('tsdf and tsdf_w must both be numpy arrays or both CUDA tensors')
        proj = np.matmul(self._K, np.matmul(lw, self._IND)).astype(np.float32)             # :695
        kinv = self._Kinv.astype(np.float32)                                                # :697
        tdist_lit = np.float32(float("%f" % self._tdist))                                   # "#define TDIST %ff" (:682-687)
        wmax_lit = np.float32(float("%f" % wmax))
        depth = to_device(dm if _is_tensor(dm) else np.asarray(dm), dtype=torch.float32)    # dm.astype(np.float32), :698
        if depth.dim() != 2:
            raise ValueError('depth map must be 2-D')
        if _is_tensor(tsdf):
            T, Wt = tsdf.to(dtype=torch.float32).clone().contiguous(), tsdf_w.to(dtype=torch.float32).clone().contiguous()
        else:
            if tsdf.ndim != 3 or tsdf.shape != tsdf_w.shape:
                raise ValueError('tsdf and tsdf_w must be 3-D arrays of the same shape')
            T, Wt = to_device(tsdf.astype(np.float32)), to_device(tsdf_w.astype(np.float32))   # :690-691
        kernels.integrate_depth_ocl(T, Wt, depth, proj, kinv[2], tdist_lit, wmax_lit)
        if _is_tensor(tsdf):
            return (T, Wt)
        return (T.cpu().numpy(), Wt.cpu().numpy())

    def fuseDepths(self, dm, lw, tsdf, tsdf_w, scale=1.0, center=np.zeros(3), wmax=100.0, mode="cpu"):
        """Integrate one depth map into (tsdf, tsdf_w); reference core/fusion_dm.py:180-217.

        numpy volumes are updated in place AND returned (the reference mutates through
        np.nditer and returns the same arrays, :186,:217); CUDA tensors are updated in place
        with no host traffic.  mode="ocl": the arithmetic of the reference's OpenCL override instead
        (fuseDepths_ocl: new float32 arrays, `scale` / `center` unused as there)."""
        if mode == "ocl":
            return self.fuseDepths_ocl(dm, lw, tsdf, tsdf_w, wmax)
        if mode != "cpu":
            raise ValueError('mode must be "cpu" (FusionDM.fuseDepths semantics) or "ocl" (FusionDM_GPU.fuseDepths semantics)')
        lw = np.asarray(lw, dtype=np.float64)
        if lw.shape != (3, 4):
            raise ValueError('lw must be a 3x4 camera extrinsic')
        depth = self._depth_to_device(dm)
        if _is_tensor(tsdf) != _is_tensor(tsdf_w):
            raise ValueError('tsdf and tsdf_w must both be numpy arrays or both CUDA tensors')
        if _is_tensor(tsdf):
            kernels.integrate_depth(tsdf, tsdf_w, depth, self._K, self._Kinv, lw, scale, center, self._tdist,
                                    wmax, tsdf_res=self._tsdf_res)
            return (tsdf, tsdf_w)
        if tsdf.ndim != 3 or tsdf.shape != tsdf_w.shape:
            raise ValueError('tsdf and tsdf_w must be 3-D arrays of the same shape')
        T = to_device(tsdf, dtype=self._vol_dtype)
        Wt = to_device(tsdf_w, dtype=self._vol_dtype)
        kernels.integrate_depth(T, Wt, depth, self._K, self._Kinv, lw, scale, center, self._tdist, wmax,
                                tsdf_res=self._tsdf_res)
        tsdf[...] = T.cpu().numpy()
        tsdf_w[...] = Wt.cpu().numpy()
        return (tsdf, tsdf_w)

    # ------------------------------------------------------------------ A3
    def _live_to_device(self, curr_tsdf):
        if _is_tensor(curr_tsdf):
            if curr_tsdf.dim() != 3:
                raise ValueError('Only accept 3D array as tsdf')
            t = curr_tsdf if curr_tsdf.dtype in (torch.float32, torch.float64) else curr_tsdf.to(torch.float32)
            return to_device(t)
        if type(curr_tsdf) is not np.ndarray or curr_tsdf.ndim != 3:
            raise ValueError('Only accept 3D np array as tsdf')
        return to_device(curr_tsdf, dtype=torch.float32 if f32_exact(curr_tsdf) else torch.float64)

    def updateTSDF(self, curr_tsdf, wmax=100.0):
        """Rigidly warp every canonical voxel by `_lw` into the live TSDF, sample it and fuse;
        reference core/fusion_dm.py:300-316.  Mutates `_tsdf` / `_tsdfw` (on the GPU)."""
        live = self._live_to_device(curr_tsdf)
        self._ensure_volumes()
        kernels.fuse_volume_rigid(self._T, self._Wt, live, np.asarray(self._lw, dtype=np.float64), self._tdist, wmax)

    # ------------------------------------------------------------------ A9
    def _corr_state(self):
        if self._vertices is None or self._normals is None:
            raise ValueError('canonical vertices / normals have not been set')
        idx = np.asarray(self._corridx, dtype=np.int64)
        C = np.asarray(self._correspondences, dtype=np.float64).reshape(-1, 3)
        if len(idx) != len(C):
            raise ValueError('_corridx and _correspondences disagree in length')
        V = np.asarray(self._vertices, dtype=np.float64)[idx]
        Nn = np.asarray(self._normals, dtype=np.float64)[idx]
        return V, Nn, C

    def computef_lw(self, x):
        """Point-to-plane residual of the kept correspondences under the global transform x;
        reference core/fusion_dm.py:285-297."""
        V, Nn, C = self._corr_state()
        return _solve.residual_rigid(x, V, Nn, C).cpu().numpy()

    # ------------------------------------------------------------------ surface + correspondences
    def marching_cubes(self, tsdf=None, step_size=1):
        """Mesh vertices / faces / normals of the level set; reference core/fusion_dm.py:319-331
        (`measure.marching_cubes_lewiner(volume, step_size=step_size)`: no level is passed, so skimage's
        default (min + max) / 2 applies -- reproduced).  With `tsdf` returns (verts, faces, normals,
        values) as numpy arrays; without, fills `_vertices` / `_faces` / `_normals` from the canonical
        volume.  Runs on the GPU (csrc/dfh_mesh.hip); conventions and what is pinned: mesh.py."""
        if step_size < 1:
            step_size = self._marching_cubes_step_size
        if tsdf is not None:
            return _mesh.marching_cubes(self._live_to_device(tsdf), None, step_size, as_numpy=True)
        self._ensure_volumes()
        self._vertices, self._faces, self._normals, values = _mesh.marching_cubes(self._T, None, step_size, as_numpy=True)
        if self._verbose:
            print("Marching Cubes result: number of extracted vertices is %d" % (len(self._vertices)))

    def surface_samples(self, tsdf=None, band=1.0):
        """Dense alternative to the mesh vertices for the solve (not in the reference): every band
        voxel moved onto the zero level set along the TSDF gradient (csrc/dfh_extract.hip), with the
        normalised gradient as normal.  Same shapes as marching_cubes, faces / values = None."""
        from .pipeline import extract_surface_samples
        if tsdf is not None:
            live = self._live_to_device(tsdf)
            pos, nrm = extract_surface_samples(live, torch.ones_like(live), band)
            return pos.cpu().numpy(), None, nrm.cpu().numpy(), None
        self._ensure_volumes()
        pos, nrm = extract_surface_samples(self._T, self._Wt, band)
        self._vertices, self._faces, self._normals = pos.cpu().numpy(), None, nrm.cpu().numpy()

    def write_canonical_mesh(self, path, filename):
        """OBJ file of the canonical surface in world coordinates; reference core/fusion_dm.py:339-354
        (level 0, step 1, degenerate faces dropped; `v` / `vn` / `f a//a b//b c//c`, 1-based)."""
        self._ensure_volumes()
        verts, faces, normals, values = _mesh.marching_cubes(self._T, 0.0, 1, as_numpy=True)
        _mesh.write_obj(os.path.join(path, filename), verts, faces, normals, ind=self._IND)

    def write_warp_field(self, path, filename):
        """Reference core/fusion_dm.py:334-336: pickle of `_nodes` into <path>/<filename>__<itercounter>.p."""
        from . import io as _io
        return _io.write_warp_field(self._nodes, path, filename, self._itercounter)

    def write_live_frame_mesh(self, path, filename, warpfield_path):
        """Reference core/fusion_dm.py:357-358: an empty stub there too."""
        pass

    def average_edge_dist_in_face(self, f):
        """Reference core/fusion_dm.py:360-364."""
        v1, v2, v3 = (np.asarray(self._vertices[i]) for i in f[:3])
        d = lambda a, b: np.linalg.norm(a - b)
        return (d(v1, v2) + d(v1, v3) + d(v2, v3)) / 3

    def setupCorrespondences(self, curr_tsdf, prune_result=True, tolerance=1.0, live_vertices=None):
        """Closest-point correspondences of the canonical vertices in the live surface; reference
        core/fusion_dm.py:219-244 (warp by `_lw`, knn nearest live vertices, smallest point-to-plane
        cost, keep if <= tolerance).  `live_vertices` replaces the marching-cubes call on curr_tsdf."""
        if self._vertices is None or self._normals is None:
            raise ValueError('canonical vertices / normals have not been set (call marching_cubes())')
        lverts = self.marching_cubes(curr_tsdf, step_size=1)[0] if live_vertices is None else np.asarray(live_vertices, dtype=np.float64)
        if len(lverts) < self._knn:
            raise ValueError('fewer live vertices than knn')
        lw = np.asarray(self._lw, dtype=np.float64)
        vp, wn = _solve.warp_points(self._vertices, self._normals, lw)
        corr, cost, keep = _solve.closest_correspondences(vp, wn, lverts, self._knn, tolerance)
        keep = keep.cpu().numpy().astype(bool)
        self._corridx = list(np.nonzero(keep)[0])
        self._correspondences = list(corr.cpu().numpy()[keep])

    def solve(self, curr_tsdf=None, iterations=10):
        """Rigid alignment `_lw`; call surface of reference core/fusion_dm.py:264-282: three rounds
        of [setupCorrespondences(curr_tsdf) -> minimise 0.5*|computef_lw|^2 from the current `_lw`].
        The reference minimises with scipy's least_squares (finite differences); here Gauss-Newton
        on the 6-DoF left twist with analytic Jacobians (HIP kernels).  curr_tsdf=None keeps the
        correspondences already stored in `_corridx` / `_correspondences` (one round)."""
        self._itercounter += 1
        self.last_costs = []
        for _ in range(3 if curr_tsdf is not None else 1):             # iteration = 3, :265
            if curr_tsdf is not None:
                self.setupCorrespondences(curr_tsdf)
            V, Nn, C = self._corr_state()
            x, costs = _solve.solve_rigid_gn(np.asarray(self._lw, dtype=np.float64), V, Nn, C, iters=iterations)
            self._lw = x
            self.last_costs.append(costs)

    # ------------------------------------------------------------------ driver
    def _auto_alignment(self, depths, lws):
        """Centre / spread of the back-projected depth pixels (intent of
        core/fusion_dm.py:110-134; that loop cannot run for more than one view at HEAD
        because `avgs` is turned into an ndarray inside it, :131)."""
        avgs, stds = [], []
        for dm, A in zip(depths, lws):
            dm = dm.cpu().numpy() if _is_tensor(dm) else np.asarray(dm)
            A = np.asarray(A, dtype=np.float64)
            rows, cols = np.nonzero(dm)
            uv = -1 * dm[rows, cols][:, None] * np.stack([cols, rows, np.ones_like(rows)], axis=1).astype(float)
            pos3 = uv @ self._Kinv.T
            Rinv = la.inv(A[:, :3])
            pts = (pos3 - A[:, 3]) @ Rinv.T
            avgs.append(np.average(pts, axis=0))
            stds.append(np.std(pts, axis=0))
        return np.average(np.array(avgs), axis=0), float(np.average(np.array(stds)))

    def compute_live_tsdf(self, depths, lws, UseAutoAlignment=False, useICP=False, outputMesh=False,
                          as_numpy=True, mesh_path='.'):
        """Fuse a set of depth maps into a fresh volume; reference core/fusion_dm.py:95-178.
        The volume stays on the GPU across views; `as_numpy=False` returns the CUDA
        tensors instead of downloading them."""
        if len(depths) != len(lws):
            raise ValueError('length of camera matrix array Ks must equal that of depth maps')   # :96-97
        avg = np.array([-0.03, -0.43, -5.6], dtype='float32')       # :106-107
        std = 1.3
        if UseAutoAlignment:
            avg, std = self._auto_alignment(depths, lws)

        res = self._tsdf_res
        scale = 8 * std / res                                       # :136-141
        self._IND[0, 0] = scale
        self._IND[1, 1] = scale
        self._IND[2, 2] = scale
        self._IND[0:3, 3] = avg - scale * res / 2
        self._INDinv = la.inv(self._IND)

        if useICP:                                                  # :149-164
            for idx in range(len(depths)):
                T, Wt = self._new_volume_pair()
                self.fuseDepths(depths[idx], lws[idx], T, Wt, scale=10 * std / res, center=avg)
                if idx == 0:
                    self._T, self._Wt = T, Wt
                    self.marching_cubes()
                else:
                    self._lw = np.array([1, 0, 0, 0, 0, 0, 0, 0], dtype=np.float32)
                    self.solve(T)
                    self.updateTSDF(T)
            T, Wt = self._T, self._Wt
        else:
            dev = [self._depth_to_device(d) for d in depths]
            if len(dev) > 1 and len({(tuple(d.shape), d.dtype) for d in dev}) == 1 and \
                    all(np.asarray(l).shape == (3, 4) for l in lws):
                # the loop below as ONE sweep of the volume, the initial values (:100-101) included (same bits:
                # kernels.integrate_depth_views with fresh=)
                r = self._tsdf_res
                T = torch.empty((r, r, r), dtype=self._vol_dtype, device="cuda")
                Wt = torch.empty_like(T)
                kernels.integrate_depth_views(T, Wt, dev, self._K, self._Kinv, [np.asarray(l, dtype=np.float64) for l in lws],
                                              12 * std / res, avg, self._tdist, 100.0, tsdf_res=self._tsdf_res, fresh=self._tdist)
                self._depthidx = len(depths) - 1
            else:
                T, Wt = self._new_volume_pair()
                for idx in range(len(depths)):                      # :166-170
                    self._depthidx = idx
                    self.fuseDepths(dev[idx], lws[idx], T, Wt, scale=12 * std / res, center=avg)
            self._T, self._Wt = T, Wt
        if outputMesh:                                              # :174-176
            np.save(os.path.join(mesh_path, 'tsdf_temp'), self._T.cpu().numpy())     # the reference's `np.save('tsdf_temp', self._tsdf)` (:175), in mesh_path
            self.write_canonical_mesh(mesh_path, 'test.obj')
        if as_numpy:
            return (self._tsdf, self._tsdfw)
        return (T, Wt)
