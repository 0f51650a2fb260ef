"""Deterministic synthetic inputs for the hot path: no files, no network.

Scene: sphere (centre (0,0,2) m, radius 0.5 m) in front of a wall z = 3 m, seen by a
pinhole camera orbiting the grid centre.  Depth maps follow the reference's storage
convention: NEGATIVE depth along the camera z axis, 0 = no measurement
(core/fusion_dm.py:196-197, test.py:194-195).
"""
import math

import numpy as np

SPHERE_C = np.array([0.0, 0.0, 2.0])
SPHERE_R = 0.5
WALL_Z = 3.0
GRID_SIDE = 1.6

# config -> (H, W, f, cx, cy).  SURVEY §8(d) suggests 300.3/159.7/119.6 etc.; on the decimal
# voxel grid those still give exact .5-pixel ties in exact arithmetic (e.g. 600.3*6/261 +
# 319.7 = 333.5), where the reference's round() outcome depends on its platform BLAS.  The
# 4-decimal values below have prime numerators (3003113, 1597033, ...) coprime to every
# grid denominator, so no voxel of the unrotated view sits on a tie.
CAMERAS = {
    "C1": (240, 320, 300.3113, 159.7033, 119.6003),
    "C2": (480, 640, 600.3113, 319.7009, 239.6029),
    "C5": (720, 1280, 1100.3107, 639.7003, 359.6003),
}


def intrinsics(f, cx, cy):
    return np.array([[f, 0.0, cx], [0.0, f, cy], [0.0, 0.0, 1.0]])


def view_extrinsic(angle_deg, centre=SPHERE_C):
    """3x4 world->camera matrix `lw` (the reference's K^-1 P, test.py:154-155) for a
    camera rotated about the grid centre's y axis; the centre stays 2 m in front."""
    a = math.radians(angle_deg)
    R = np.array([[math.cos(a), 0.0, math.sin(a)], [0.0, 1.0, 0.0], [-math.sin(a), 0.0, math.cos(a)]])
    t = -R @ centre + np.array([0.0, 0.0, float(np.linalg.norm(centre))])
    return np.concatenate([R, t[:, None]], axis=1)


def render_depth(K, lw, H, W, invalid_frac=0.02, seed=1234, dtype=np.float64,
                 sphere_c=SPHERE_C, sphere_r=SPHERE_R, wall_z=WALL_Z, sphere_offset=None):
    """Analytic ray-sphere / ray-plane depth, stored negative; `invalid_frac` of the
    pixels are zeroed with np.random.default_rng(seed).  wall_z=None: no back wall (pixels off the
    sphere carry no measurement)."""
    Kinv = np.linalg.inv(K)
    v, u = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    d = np.stack([u, v, np.ones_like(u)], axis=-1) @ Kinv.T          # ray dirs, d_z == 1
    R, t = lw[:, :3], lw[:, 3]
    c = np.asarray(sphere_c, dtype=np.float64)
    if sphere_offset is not None:
        c = c + np.asarray(sphere_offset)
    cc = R @ c + t
    dd = np.sum(d * d, axis=-1)
    dc = d @ cc
    disc = dc * dc - dd * (cc @ cc - sphere_r * sphere_r)
    ts = np.where(disc >= 0, (dc - np.sqrt(np.maximum(disc, 0))) / dd, np.inf)
    ts = np.where(ts > 0, ts, np.inf)
    if wall_z is None:
        tw = np.full_like(ts, np.inf)
    else:
        n_c = R @ np.array([0.0, 0.0, 1.0])
        p_c = R @ np.array([0.0, 0.0, wall_z]) + t
        den = d @ n_c
        tw = np.where(np.abs(den) > 1e-12, (n_c @ p_c) / np.where(np.abs(den) > 1e-12, den, 1.0), np.inf)
        tw = np.where(tw > 0, tw, np.inf)
    depth = np.minimum(ts, tw)
    depth = np.where(np.isfinite(depth), depth, 0.0)
    if invalid_frac > 0:
        rng = np.random.default_rng(seed)
        depth = np.where(rng.random((H, W)) < invalid_frac, 0.0, depth)
    return (-depth).astype(dtype)


def grid_params(res, side=GRID_SIDE, centre=SPHERE_C):
    """(scale, center, tdist) of SURVEY §8(d): cube of `side` m centred at the sphere,
    truncation = 4 voxels."""
    scale = side / res
    return scale, np.asarray(centre, dtype=np.float64).copy(), 4.0 * scale


def fibonacci_nodes(n, res, radius_vox=None):
    """First n points of a Fibonacci lattice on the sphere, mapped to voxel-index space
    of a res^3 grid (SURVEY §8(d) warp-field nodes).  Returns (pos (n,3), node_w (n,))."""
    if radius_vox is None:
        radius_vox = SPHERE_R / (GRID_SIDE / res)
    i = np.arange(n, dtype=np.float64) + 0.5
    phi = np.arccos(1 - 2 * i / n)
    theta = math.pi * (1 + 5 ** 0.5) * i
    p = np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], axis=-1)
    pos = p * radius_vox + res / 2.0
    spacing = radius_vox * math.sqrt(4 * math.pi / n)       # mean node spacing on the sphere
    return pos, np.full(n, 2.0 * spacing)
