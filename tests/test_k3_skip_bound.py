"""CPU: the displacement bound behind K3's constant-live skip (csrc/dfh_fuse_volume.hip, dqb_bound_kernel), restated in numpy and
checked against the ORACLE's dq_blend + dqb_warp (oracle/oracle_np.py = the reference's chain) on random blends:
    x1 - p = N(b, p) / |b|_8^2,  N quadratic in b with symmetric bilinear form G,
    |x1 - p| <= max_jk [ |G(dq_j, dq_k, c)| + (|v_j| + |v_k| + 2 |r_j - 1| |r_k - 1| + |d_j . d_k|) rho ] / min_jk r_j . r_k
for every convex blend of the nodes' dual quaternions and every p within rho of c."""
import numpy as np

from oracle import oracle_np as O


def _qmul(a, b):
    return O.quaternion_multiply(a, b)


def _conj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def _G(b1, b2, p):
    r1, d1, r2, d2 = b1[:4], b1[4:], b2[:4], b2[4:]
    P = np.array([0.0, *p])
    return (0.5 * (_qmul(_qmul(r1, P), _conj(r2))[1:] + _qmul(_qmul(r2, P), _conj(r1))[1:]) - (r1 @ r2 + d1 @ d2) * p
            + _qmul(d1, _conj(r2))[1:] + _qmul(d2, _conj(r1))[1:])


def brick_bound(dqs, c, rho):
    e = np.array([1.0, 0, 0, 0])
    num, n2min = 0.0, np.inf
    for j in range(len(dqs)):
        for k in range(j, len(dqs)):
            r1, r2, d1, d2 = dqs[j, :4], dqs[k, :4], dqs[j, 4:], dqs[k, 4:]
            lam = np.linalg.norm(r1[1:]) + np.linalg.norm(r2[1:]) + 2 * np.linalg.norm(r1 - e) * np.linalg.norm(r2 - e) + abs(d1 @ d2)
            num = max(num, np.linalg.norm(_G(dqs[j], dqs[k], c)) + lam * rho)
            n2min = min(n2min, r1 @ r2)
    return num / n2min


def test_quadratic_form_is_the_warp_displacement():
    rng = np.random.default_rng(0)
    for _ in range(20):
        b = rng.normal(size=8)
        b[0] += 3.0
        p = (rng.normal(size=3) * 50).astype(np.float32).astype(np.float64)     # (dqb_warp rounds its point to float32, core/util.py:69)
        x1 = O.dqb_warp(b / np.linalg.norm(b), p)
        assert np.abs(_G(b, b, p) / (b @ b) - (x1 - p)).max() <= 1e-11


def test_bound_covers_every_convex_blend():
    rng = np.random.default_rng(1)
    e = np.array([1.0, 0, 0, 0])
    worst = 0.0
    for trial in range(60):
        n = int(rng.integers(2, 9))
        c = rng.uniform(0, 512, 3)
        rho = 7.8
        dqs = np.array([np.concatenate([e + rng.normal(size=4) * rng.choice([1e-3, 1e-2, 0.1]), rng.normal(size=4) * rng.choice([0.05, 0.3, 1.0])])
                        for _ in range(n)])
        D = brick_bound(dqs, c, rho)
        for _ in range(60):
            w = rng.dirichlet(np.ones(n) * rng.choice([0.2, 1, 5]))
            if rng.random() < 0.3:                                               # blends of two nodes: the simplex's edges
                w[:] = 0
                idx = rng.choice(n, size=2, replace=False)
                w[idx] = rng.dirichlet(np.ones(2))
            pp = rng.normal(size=3)
            p = (c + pp / np.linalg.norm(pp) * rho * rng.random()).astype(np.float32).astype(np.float64)
            if np.linalg.norm(p - c) > rho:
                continue
            b = w @ dqs                                                           # Fusion.dq_blend: weights, then the 8-norm (core/fusion.py:527-551)
            x1 = O.dqb_warp(b / np.linalg.norm(b), p)
            disp = np.linalg.norm(x1 - p)
            assert disp <= D * (1 + 1e-9) + 1e-9, (disp, D)
            worst = max(worst, disp / D)
    assert worst > 0.5                                                            # tight enough to be useful
