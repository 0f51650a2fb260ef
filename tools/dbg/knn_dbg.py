import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dynamicfusion_body_amd import solve
from oracle import oracle_np as O
rng = np.random.default_rng(1)
for N, k, dup, tail in ((1500, 4, True, True), (1500, 4, False, False), (1500, 4, False, True), (1500, 4, True, False), (300, 4, False, False)):
    npos = rng.uniform(0, 120, size=(N, 3)); nw = rng.uniform(2, 6, size=N)
    if dup: npos[N // 2:N // 2 + 20] = npos[:20]
    z = np.arange(900, dtype=np.float64)
    pts = np.stack([20.0 + (z // 300) + 0.3 * np.sin(z), 33.0 + 0.2 * np.cos(z), (z % 300) * 0.4], axis=1)
    if tail: pts = np.concatenate([pts, rng.uniform(0, 120, size=(300, 3))])
    nbr, wts = solve.sample_knn(pts, npos, nw, k)
    loc = O.knn_bruteforce(pts, npos, k)
    bad = np.nonzero((nbr.cpu().numpy() != loc).any(1))[0]
    print(N, k, dup, tail, "mismatch rows", len(bad), bad[:10], (nbr.cpu().numpy()[bad[:2]], loc[bad[:2]]) if len(bad) else "")
