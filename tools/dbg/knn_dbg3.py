import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dynamicfusion_body_amd import solve
from oracle import oracle_np as O
rng = np.random.default_rng(1)
N, k = 1500, 4
npos = rng.uniform(0, 120, size=(N, 3)); nw = rng.uniform(2, 6, size=N)
npos[N // 2:N // 2 + 20] = npos[:20]
z = np.arange(900, dtype=np.float64)
pts = np.stack([20.0 + (z // 300) + 0.3 * np.sin(z), 33.0 + 0.2 * np.cos(z), (z % 300) * 0.4], axis=1)
nbr = solve.sample_knn(pts, npos, nw, k)[0].cpu().numpy()
loc = O.knn_bruteforce(pts, npos, k)
bad = np.nonzero((nbr != loc).any(1))[0]
r = bad[0]
print("row", r, nbr[r], loc[r])
d = pts[r] - npos; d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
for n in set(nbr[r]) | set(loc[r]): print(n, repr(d2[n]), npos[n])
one = solve.sample_knn(pts[r:r + 1], npos, nw, k)[0].cpu().numpy()
print("alone:", one)
blk = (r // 256) * 256
print("block rows", blk, "mismatches in block:", [int(x) for x in bad if blk <= x < blk + 256][:20])
