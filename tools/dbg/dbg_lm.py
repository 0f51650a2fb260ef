import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from conftest import load_golden
import test_gpu_solve as T
from oracle import gn_np as G, oracle_np as O
g, verts, norms, corr, nbr, vidx, npos, ndq_true, nw, lw, rw = T.load(load_golden)
N=len(npos)
ndq_true = ndq_true / np.sqrt(np.sum(ndq_true[:, :4] ** 2, axis=1, keepdims=True))
target,_ = O.warp(verts, ndq_true[nbr], npos[nbr], nw[nbr], normal=norms, m_lw=lw)
ident = np.tile(np.array([1.0,0,0,0,0,0,0,0]),(N,1))
sv = T.make_solver(npos, ident, nw, nbr, vidx, verts, norms, target, nbr.shape[1], pcg_iters=800)
costs = sv.solve_lm(lw, 1e-3, iters=10, lm_abs=1.0, lm_rel=0.0, adaptive=False)
dqs=ident.copy(); oc=[]
for it in range(10):
    dqs,c,dx = G.gn_step(dqs, verts, norms, target, nbr, vidx, npos, nw, lw, 1e-3, lm=1.0/3.0**it); oc.append(c)
for a,b in zip(costs[:10],oc): print("%.12e %.12e rel %.2e"%(a,b,abs(a-b)/abs(b)))
