import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dynamicfusion_body_amd import solve
from oracle import oracle_np as O
rng = np.random.default_rng(2)
for N in (40, 300, 600, 1500):
    npos = rng.uniform(0, 120, size=(N, 3)); nw = rng.uniform(2, 6, size=N)
    npos[N // 2:N // 2 + 5] = npos[:5]
    pts = npos[:5] + 0.25            # nearest nodes: the duplicate pairs
    for rep in (1, 300):
        P = np.repeat(pts, rep, axis=0)
        nbr, _ = solve.sample_knn(P, npos, nw, 4)
        print(N, rep, nbr.cpu().numpy()[::rep][:5, :2].tolist(), O.knn_bruteforce(pts, npos, 4)[:, :2].tolist())
