"""Small host-side dual-quaternion helpers (numpy, fp64) used by the solver's host loop and by
callers that need SE(3) <-> DQ conversions with the reference's conventions (w-first layout,
basis (1,i,j,k,e,ei,ej,ek), core/util.py:78-89)."""
import numpy as np


def qmul(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw], axis=-1)


def dq_mul(a, b):
    """Dual-quaternion product (core/util.py:275-282)."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.concatenate([qmul(a[..., :4], b[..., :4]),
                           qmul(a[..., :4], b[..., 4:]) + qmul(a[..., 4:], b[..., :4])], axis=-1)


def twist_exp_dq(xi):
    """Unit dual quaternion of the twist xi = (omega, v): rotation exp(omega), translation v."""
    xi = np.asarray(xi, dtype=np.float64)
    om, v = xi[..., :3], xi[..., 3:]
    th = np.sqrt(np.sum(om * om, axis=-1))
    small = th < 1e-8
    s = np.where(small, 0.5 - th * th / 48.0, np.sin(0.5 * th) / np.where(small, 1.0, th))
    q = np.concatenate([np.cos(0.5 * th)[..., None], s[..., None] * om], axis=-1)
    vq = np.concatenate([np.zeros(v.shape[:-1] + (1,)), v], axis=-1)
    return np.concatenate([q, 0.5 * qmul(vq, q)], axis=-1)


def DQTSE3(q):
    """Dual quaternion -> 4x4 rigid matrix (core/util.py:86-89; rotation of the normalised real part)."""
    q = np.asarray(q, dtype=np.float64)
    r = q[:4] / np.linalg.norm(q[:4])
    w, x, y, z = r
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    t = qmul(2 * q[4:], q[:4] * np.array([1.0, -1.0, -1.0, -1.0]))[1:]
    M = np.identity(4)
    M[:3, :3] = R
    M[:3, 3] = t
    return M


def SE3TDQ(M):
    """4x4 rigid matrix -> unit dual quaternion with w >= 0 (core/util.py:79-84)."""
    M = np.asarray(M, dtype=np.float64)
    R, t = M[:3, :3], M[:3, 3]
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    q = q / np.linalg.norm(q)
    if q[0] < 0:
        q = -q
    return np.concatenate([q, 0.5 * qmul(np.array([0.0, t[0], t[1], t[2]]), q)])


# ---- the reference's small helpers beside its DQ algebra (core/util.py) ------------------------------------------
def huber_loss(x, c):
    """Huber's rho of core/util.py:50-54: x^2 / 2 up to |x| = c, c (|x| - c / 2) beyond; scalars or arrays.  (The solver
    applies it as IRLS weights on the data rows: dfh_gn_build_planned's huber_delta.)"""
    a = np.abs(x)
    return np.where(a <= c, 0.5 * np.square(x), c * (a - 0.5 * c)) if np.ndim(x) else (0.5 * x * x if a <= c else c * (a - 0.5 * c))


def tukey_biweight_loss(x, c):
    """core/util.py:56-60 as written there: x (1 - (x / c)^2)^2 inside |x| <= c, 0 outside (the biweight's influence
    function rather than its rho; reproduced, not corrected); scalars or arrays."""
    if np.ndim(x):
        x = np.asarray(x, dtype=np.float64)
        return np.where(np.abs(x) > c, 0.0, x * np.square(1.0 - np.square(x / c)))
    return 0 if abs(x) > c else x * (1 - (x / c) ** 2) ** 2


def inverse_rigid_matrix(A):
    """Inverse of a 3 x 4 rigid transform [R | t] as a 3 x 4 matrix [R^-1 | -R^-1 t] (core/util.py:338-346)."""
    A = np.asarray(A, dtype=np.float64)
    if A.shape != (3, 4):
        raise ValueError("a 3 x 4 matrix [R | t] is expected")
    Rinv = np.linalg.inv(A[:, :3])
    return np.concatenate([Rinv, -(Rinv @ A[:, 3])[:, None]], axis=1)
