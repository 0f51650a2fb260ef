import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dynamicfusion_body_amd import scene, kernels, solve
from dynamicfusion_body_amd.pipeline import extract_surface_samples
R, N, k = 256, 512, 4
H, W, fx, cx, cy = scene.CAMERAS["C2"]
K = scene.intrinsics(fx, cx, cy); Kinv = np.linalg.inv(K)
scale, center, tdist = scene.grid_params(R)
T = torch.full((R, R, R), tdist, dtype=torch.float32, device="cuda")
Wt = torch.zeros((R, R, R), dtype=torch.float32, device="cuda")
for a in (0.0, 40.0, -40.0):
    lw = scene.view_extrinsic(a)
    d = torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda()
    kernels.integrate_depth(T, Wt, d, K, Kinv, lw, scale, center, tdist)
pos, nrm = extract_surface_samples(T, Wt, 4.0)
node_pos, node_w = scene.fibonacci_nodes(N, R)
nbr, wts = solve.sample_knn(pos, node_pos, node_w, k)
S = nbr.shape[0]
key = torch.zeros(S, dtype=torch.int64, device="cuda")
for j in range(k):
    key = key * 1024 + nbr[:, j].long()
def rows(keys, tile):
    head = torch.ones_like(keys, dtype=torch.bool)
    head[1:] = keys[1:] != keys[:-1]
    head[::tile] = True
    return int(head.sum())
for tile in (256,):
    print("samples", S, "tiles", (S + tile - 1) // tile)
    print("emission order, no sort:        rows", rows(key, tile))
    print("global sort:                    rows", rows(torch.sort(key).values, tile))
    tid = torch.arange(S, device="cuda") // tile
    print("sort inside every tile:         rows", rows(torch.sort(key + (tid << 44)).values - 0, tile))
    for span in (1024, 4096, 16384):
        tid = torch.arange(S, device="cuda") // span
        print("sort inside spans of %5d:      rows" % span, rows(torch.sort(key + (tid << 44)).values, tile))
