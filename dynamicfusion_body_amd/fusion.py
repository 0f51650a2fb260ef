"""`Fusion` (non-rigid DynamicFusion) with the reference's call surface
(reference core/fusion.py:49-598), its per-voxel / per-vertex hot methods running as HIP
kernels on an MI355X.

The reference constructor cannot run at HEAD (it reads an undefined `tsdf`,
core/fusion.py:51); callers pass the canonical TSDF first (test.py:74,110), which is the
signature kept here.  Marching cubes, correspondence search and graph maintenance are not
on this hot path: the deformation graph is supplied as the reference's own
`_nodes` list of 4-tuples (vertex index, position, dual quaternion, weight = 2*radius,
core/fusion.py:107-116), and `_vertices/_normals/_correspondences/_neighbor_look_up` are
plain arrays the caller fills.
"""
import numpy as np
import torch

from . import kernels
from .device import f32_exact, require_gpu, to_device, torch_dtype


def _is_tensor(a):
    return isinstance(a, torch.Tensor)


class Fusion:
    def __init__(self, tsdf, trunc_distance, subsample_rate=5.0, knn=4, marching_cubes_step_size=3, verbose=False,
                 use_cnn=False, write_warpfield=True, volume_dtype=np.float32):
        if not _is_tensor(tsdf) and (type(tsdf) is not np.ndarray or tsdf.ndim != 3):
            raise ValueError('Only 3D numpy array is accepted as tsdf')          # core/fusion.py:51-52
        if _is_tensor(tsdf) and tsdf.dim() != 3:
            raise ValueError('Only 3D numpy array is accepted as tsdf')
        if use_cnn:
            raise NotImplementedError('the CNN correspondence branch (core/sdf.py:75-150) is outside this hot path')
        self._itercounter = 0
        self._curr_tsdf = None
        self._tdist = abs(trunc_distance)
        self._lw = np.array([1, 0, 0, 0, 0, 0.1, 0, 0], dtype=np.float32)        # :57
        self._knn = knn
        self._marching_cubes_step_size = marching_cubes_step_size
        self._subsample_rate = subsample_rate
        self._nodes = []
        self._neighbor_look_up = []
        self._correspondences = []
        self._vertices = None
        self._normals = None
        self._kdtree = None
        self._verbose = verbose
        self._write_warpfield = write_warpfield
        self._sess = None
        self._vol_dtype = torch_dtype(volume_dtype)
        self._T = None
        self._Wt = None
        self._tsdf_host = tsdf                   # uploaded on first use (ctor works without a GPU)
        self._workspace = None
        self._workspace_key = None

    # ------------------------------------------------------------------ volumes
    def _ensure_volumes(self):
        if self._T is None:
            require_gpu()
            self._T = to_device(self._tsdf_host, dtype=self._vol_dtype)
            self._tsdf_host = None
        if self._Wt is None:
            self._Wt = torch.zeros_like(self._T)                                 # InitializeCanonicalSpace, :74

    @property
    def _tsdf(self):
        self._ensure_volumes()
        return self._T.cpu().numpy().astype(np.float64)

    @_tsdf.setter
    def _tsdf(self, value):
        self._T = to_device(value, dtype=self._vol_dtype)
        self._tsdf_host = None

    @property
    def _tsdfw(self):
        self._ensure_volumes()
        return self._Wt.cpu().numpy().astype(np.float64)

    @_tsdfw.setter
    def _tsdfw(self, value):
        self._Wt = to_device(value, dtype=self._vol_dtype)

    @property
    def tsdf_device(self):
        self._ensure_volumes()
        return self._T, self._Wt

    # ------------------------------------------------------------------ graph
    def node_arrays(self):
        """(pos (N,3), dq (N,8), w (N,), vertex index (N,)) from the `_nodes` 4-tuples."""
        if len(self._nodes) == 0:
            raise ValueError('the deformation graph is empty: fill _nodes first')
        vidx = np.array([int(n[0]) for n in self._nodes], dtype=np.int64)
        pos = np.array([np.asarray(n[1], dtype=np.float64) for n in self._nodes])
        dq = np.array([np.asarray(n[2], dtype=np.float64) for n in self._nodes])
        w = np.array([float(n[3]) for n in self._nodes], dtype=np.float64)
        return pos, dq, w, vidx

    def _live_to_device(self, curr_tsdf):
        if _is_tensor(curr_tsdf):
            if curr_tsdf.dim() != 3:
                raise ValueError('Only accept 3D np array as tsdf')
            t = curr_tsdf if curr_tsdf.dtype in (torch.float32, torch.float64) else curr_tsdf.to(torch.float32)
            return to_device(t)
        if type(curr_tsdf) is not np.ndarray:
            raise ValueError('Only accept 3D np array as tsdf')                  # :160-161
        if curr_tsdf.ndim != 3:
            raise ValueError('Only accept 3D np array as tsdf')                  # :162-163
        return to_device(curr_tsdf, dtype=torch.float32 if f32_exact(curr_tsdf) else torch.float64)

    # ------------------------------------------------------------------ A4
    def updateTSDF(self, curr_tsdf=None, wmax=100.0):
        """Warp every canonical voxel through the DQB warp field and `_lw` into the live TSDF,
        sample and fuse; reference core/fusion.py:153-198 (the debug prints :192-195 are not
        reproduced)."""
        if curr_tsdf is not None:
            self._curr_tsdf = curr_tsdf
        if self._curr_tsdf is None:
            raise ValueError('tsdf of live frame has not been loaded')           # :158-159
        live = self._live_to_device(self._curr_tsdf)
        self._ensure_volumes()
        pos, dq, w, _ = self.node_arrays()
        res = tuple(self._T.shape)
        key = (res, pos.tobytes(), self._knn)
        rebuild = key != self._workspace_key
        if rebuild:
            self._workspace = kernels.dqb_workspace(res)
            self._workspace_key = key
        kernels.fuse_volume_dqb(self._T, self._Wt, live, pos, dq, w, self._knn,
                                np.asarray(self._lw, dtype=np.float64), self._tdist, wmax,
                                workspace=self._workspace, rebuild_candidates=rebuild)
