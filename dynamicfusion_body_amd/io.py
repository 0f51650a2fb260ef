"""File formats on either side of the hot path, host-side like in the
reference: the `.dist` signed-distance volumes (reference core/sdf.py:24-69), the `proj*.txt`
camera matrices (core/util.py:330-335) and the warp-field pickle (core/fusion.py:571-573)."""
import os
import pickle

import numpy as np


def load_sdf(file_path, read_closest_points=False, verbose=False):
    """Reference core/sdf.py:24-69.  Layout: int32 -res_x, int32 -res_y, int32 res_z (the first two
    are stored negated), 3 float64 b_min, 3 float64 b_max, (1+res)^3 float32 distances stored
    z-major (reshaped (1+res_z, 1+res_y, 1+res_x) and swapped to x-major), optionally 3 float32
    closest-point coordinates per grid vertex.  Returns (b_min, b_max, volume, closest_points)."""
    with open(file_path, 'rb') as fp:
        head = np.fromfile(fp, dtype=np.int32, count=3)
        if head.size != 3:
            raise ValueError('%s: truncated .dist header' % file_path)
        res_x, res_y, res_z = -int(head[0]), -int(head[1]), int(head[2])
        if verbose:
            print("resolution: %d %d %d" % (res_x, res_y, res_z))
        b_min = np.fromfile(fp, dtype=np.float64, count=3)
        b_max = np.fromfile(fp, dtype=np.float64, count=3)
        grid_num = (1 + res_x) * (1 + res_y) * (1 + res_z)
        volume = np.fromfile(fp, dtype=np.float32, count=grid_num)
        if b_min.size != 3 or b_max.size != 3 or volume.size != grid_num:
            raise ValueError('%s: truncated .dist body' % file_path)
        volume = np.swapaxes(volume.reshape(((1 + res_z), (1 + res_y), (1 + res_x))), 0, 2)
        closest_points = None
        if read_closest_points:
            cp = np.fromfile(fp, dtype=np.float32, count=grid_num * 3)
            if cp.size != grid_num * 3:
                raise ValueError('%s: truncated closest-point block' % file_path)
            closest_points = np.swapaxes(cp.reshape(((1 + res_z), (1 + res_y), (1 + res_x), 3)), 0, 2)
    return b_min, b_max, volume, closest_points


def write_sdf(file_path, b_min, b_max, volume, closest_points=None):
    """Inverse of load_sdf (the reference only reads this format)."""
    vol = np.asarray(volume, dtype=np.float32)
    rx, ry, rz = (s - 1 for s in vol.shape)
    with open(file_path, 'wb') as fp:
        np.array([-rx, -ry, rz], dtype=np.int32).tofile(fp)
        np.asarray(b_min, dtype=np.float64).tofile(fp)
        np.asarray(b_max, dtype=np.float64).tofile(fp)
        np.ascontiguousarray(np.swapaxes(vol, 0, 2)).tofile(fp)
        if closest_points is not None:
            np.ascontiguousarray(np.swapaxes(np.asarray(closest_points, dtype=np.float32), 0, 2)).tofile(fp)


def read_proj_matrix(fpath):
    """Reference core/util.py:330-335: whitespace-free, single-space separated rows of floats
    (every line must end with a newline, which the reference strips blindly)."""
    arr = []
    with open(fpath, 'r') as f:
        for line in f:
            arr.append(line[:-1].split(' '))
    return np.array(arr, dtype='float')


def write_warp_field(nodes, path, filename, itercounter):
    """Reference core/fusion.py:571-573: pickle of the `_nodes` list of 4-tuples
    (vertex index, position, dual quaternion, weight) to <path>/<filename>__<iter>.p."""
    fname = os.path.join(path, filename + '__' + str(itercounter) + '.p')
    with open(fname, 'wb') as f:
        pickle.dump(nodes, f)
    return fname


def read_warp_field(fname):
    """Files written by write_warp_field of THIS package (never unpickle untrusted files)."""
    with open(fname, 'rb') as f:
        return pickle.load(f)
