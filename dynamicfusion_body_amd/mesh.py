"""Mesh extraction and the canonical-mesh file: the reference's `marching_cubes` /
`write_canonical_mesh` (core/fusion_dm.py:319-331,339-354, core/fusion.py:554-568).

`marching_cubes` runs the HIP kernels of csrc/dfh_mesh.hip (count -> scan -> vertices -> faces) and
has skimage's call shape: (verts, faces, normals, values), with `level=None` meaning
(min + max) / 2 (skimage's documented default, which is what the reference's `marching_cubes()`
gets since it passes no level)."""
import numpy as np
import torch

from . import _lib
from .device import HostScalar, current_stream_ptr, dtype_code, require_gpu


def marching_cubes(volume, level=None, step_size=1, as_numpy=False, order="reference", visit_all_tiles=False):
    """volume: 3-D CUDA tensor (fp32 / fp64, contiguous).  Returns verts (V,3) fp32 in array-index
    coordinates, faces (F,3) int32, unit normals (V,3) fp32 pointing down the gradient, values (V,)
    fp32 -- CUDA tensors, or numpy arrays with as_numpy=True.  Zero-area faces are not emitted
    (allow_degenerate=False, the only mode the reference uses for its outputs).
    order="reference": skimage's numbering (faces cube by cube, vertices by first use, unused vertices
    dropped); order="lattice": vertices by owning lattice point (skips the renumbering pass).
    visit_all_tiles: emit passes over every tile instead of the compacted list (same result; for tests)."""
    return marching_cubes_begin(volume, level, step_size).finish(as_numpy=as_numpy, order=order, visit_all_tiles=visit_all_tiles)


class marching_cubes_begin:
    """marching_cubes in two halves: the constructor launches the count pass (on the current stream) and returns; finish() waits
    for the totals, launches the emit passes and returns the mesh.  A caller with other work to queue in between (a frame loop
    that has just updated the canonical volume) starts the count early and collects the mesh later.
    Ordering: the count pass's stream is remembered and an event is recorded behind it; finish() may run on any stream -- it
    makes that stream wait for the event before the emit passes read the count pass's workspace.  The volume must not change
    between the two halves (the totals that size the outputs were counted on it): finish() checks the volume tensor's
    version counter and refuses a volume that was written through torch in between (a write through the C ABI is the caller's
    responsibility)."""

    def __init__(self, volume, level=None, step_size=1):
        require_gpu()
        self.lib = lib = _lib.load()
        if not (isinstance(volume, torch.Tensor) and volume.is_cuda and volume.dim() == 3 and volume.is_contiguous()):
            raise ValueError("volume must be a contiguous 3-D CUDA tensor")
        step = int(step_size)
        if step < 1:
            raise ValueError("step_size must be at least 1")                         # skimage raises ValueError too
        if min(volume.shape) < 2:
            raise ValueError("Input array must be at least 2x2x2.")
        if level is None:
            level = 0.5 * (float(volume.min()) + float(volume.max()))
        self.volume, self.step, self.level = volume, step, float(level)
        self.res = _lib.iarr(volume.shape)
        nbytes = lib.dfh_mc_workspace_bytes(self.res, step)
        self.ws = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=volume.device)
        self.totals = HostScalar(torch.int64, 3)      # (vertices, faces, active tiles: stored into pinned host memory by the scan)
        _lib.check(lib.dfh_mc_count(volume.data_ptr(), dtype_code(volume), self.res, step, self.level, self.ws.data_ptr(),
                                    self.ws.numel() * 8, self.totals.ptr(), current_stream_ptr()), "dfh_mc_count")
        self._count_stream = current_stream_ptr()
        self._counted = torch.cuda.Event()
        self._counted.record()
        self._version = volume._version

    def finish(self, as_numpy=False, order="reference", visit_all_tiles=False):
        if order not in ("reference", "lattice"):
            raise ValueError("order must be 'reference' or 'lattice'")
        lib, volume, ws = self.lib, self.volume, self.ws
        if volume._version != self._version:
            raise RuntimeError("the volume was modified between marching_cubes_begin() and finish(): the counted totals no longer "
                               "describe it")
        if current_stream_ptr() != self._count_stream:
            torch.cuda.current_stream().wait_event(self._counted)       # the emit passes read the count pass's workspace
        nv, nf, nactive = self.totals.get()
        if nv >= (1 << 29) or nf >= (1 << 31) // 3:
            raise ValueError("surface too large for 32-bit mesh indices (%d vertices, %d faces)" % (nv, nf))
        dev = volume.device
        verts = torch.empty((nv, 3), dtype=torch.float32, device=dev)
        normals = torch.empty((nv, 3), dtype=torch.float32, device=dev)
        values = torch.empty((nv,), dtype=torch.float32, device=dev)
        faces = torch.empty((nf, 3), dtype=torch.int32, device=dev)
        _lib.check(lib.dfh_mc_emit(volume.data_ptr(), dtype_code(volume), self.res, self.step, self.level, ws.data_ptr(), ws.numel() * 8,
                                   verts.data_ptr(), normals.data_ptr(), values.data_ptr(), faces.data_ptr(), nv, nf,
                                   -1 if visit_all_tiles else nactive, current_stream_ptr()), "dfh_mc_emit")
        if order == "reference":
            nbytes = lib.dfh_mc_reorder_workspace_bytes(nv, nf)
            ws2 = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=dev)
            v2, n2, val2 = torch.empty_like(verts), torch.empty_like(normals), torch.empty_like(values)
            used = HostScalar(torch.int64)
            _lib.check(lib.dfh_mc_reorder(verts.data_ptr(), normals.data_ptr(), values.data_ptr(), faces.data_ptr(), nv, nf,
                                          v2.data_ptr(), n2.data_ptr(), val2.data_ptr(), used.ptr(), ws2.data_ptr(),
                                          ws2.numel() * 8, current_stream_ptr()), "dfh_mc_reorder")
            nu = used.get()
            verts, normals, values = v2[:nu], n2[:nu], val2[:nu]
        if as_numpy:
            return verts.cpu().numpy(), faces.cpu().numpy(), normals.cpu().numpy(), values.cpu().numpy()
        return verts, faces, normals, values


def write_obj(fpath, verts, faces, normals, ind=None):
    """The reference's OBJ layout (core/fusion_dm.py:339-354): `v x y z` rows, then `vn`, then
    `f a//a b//b c//c` with 1-based indices, `%f` formatting; `ind` (4x4) maps index space to world
    (`self._IND`: rotation applied to normals, rotation + translation to vertices)."""
    verts = np.asarray(verts, dtype=np.float64)
    normals = np.asarray(normals, dtype=np.float64)
    faces = np.asarray(faces)
    if ind is not None:
        ind = np.asarray(ind, dtype=np.float64)
        rot, trans = ind[:3, :3], ind[:3, 3]
        verts = verts @ rot.T + trans
        normals = normals @ rot.T
    with open(fpath, "w") as f:
        f.write("".join("v %f %f %f\n" % (v[0], v[1], v[2]) for v in verts))
        f.write("".join("vn %f %f %f\n" % (n[0], n[1], n[2]) for n in normals))
        f.write("".join("f %d//%d %d//%d %d//%d\n" % (a + 1, a + 1, b + 1, b + 1, c + 1, c + 1) for a, b, c in faces))


def read_obj(fpath):
    """v / vn / f rows of an OBJ file -> (verts, faces as stored, normals); text parsing only."""
    V, N, F = [], [], []
    with open(fpath) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                V.append([float(x) for x in t[1:4]])
            elif t[0] == "vn":
                N.append([float(x) for x in t[1:4]])
            elif t[0] == "f":
                F.append([int(x.split("/")[0]) for x in t[1:4]])
    return (np.array(V, dtype=np.float64).reshape(-1, 3), np.array(F, dtype=np.int64).reshape(-1, 3),
            np.array(N, dtype=np.float64).reshape(-1, 3))
