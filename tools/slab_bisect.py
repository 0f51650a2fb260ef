"""Stage-by-stage comparison of the slab path with the whole-grid path of the frame loop, in ONE process (no process
group): for P slabs of the bench's frame scene (R^3 grid, N nodes, 3 views) every stage a rank of pipeline.SlabFrame runs is
run on each slab with the whole-grid run's inputs and compared bit for bit with the whole-grid result:
  canonical K1 | live K1 (fresh multi-view sweep) | K3 (first call and steady state) | halo-padded sample extraction |
  the samples' node search through the slab's brick lists.
Usage: python tools/slab_bisect.py [R] [N] [P] [frames]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dynamicfusion_body_amd import kernels, scene                      # noqa: E402
from dynamicfusion_body_amd import dist as D                           # noqa: E402
from dynamicfusion_body_amd.pipeline import SlabFrame, extract_surface_samples   # noqa: E402
from dynamicfusion_body_amd.solve import sample_knn                    # noqa: E402


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    P = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    nframes = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    H, W, fx, cx, cy = scene.CAMERAS["C2" if R <= 256 else "C5"]
    K = scene.intrinsics(fx, cx, cy)
    Kinv = np.linalg.inv(K)
    scale, center, tdist = scene.grid_params(R)
    tvox = tdist / scale
    node_pos, node_w = scene.fibonacci_nodes(N, R)
    band, knn = 4.0, 4
    angles = (0.0, 40.0, -40.0)
    lws = [scene.view_extrinsic(a) for a in angles]
    whole = SlabFrame(K, scale, center, R, tvox, node_pos, node_w, knn=knn, pcg_iters=10, band=band, distributed=False)
    ranges = [D.slab_range(R, r, P) for r in range(P)]
    Ts = [torch.full((b - a, R, R), tvox, dtype=torch.float32, device="cuda") for a, b in ranges]
    Ws = [torch.zeros_like(t) for t in Ts]
    lives = [(torch.empty_like(t), torch.empty_like(t)) for t in Ts]
    wss = [kernels.dqb_workspace((R, R, R), rg, knn=knn, n_nodes=N) for rg in ranges]
    for ws, rg in zip(wss, ranges):
        kernels.dqb_build_candidates(ws, (R, R, R), node_pos, knn, rg)
    ws_views = [None] * P
    bad = []

    def cmp(name, x, y):
        same = bool(torch.equal(x, y))
        nd = 0 if same else int((x != y).sum())
        print("%-44s %s%s" % (name, "same" if same else "DIFFERENT", "" if same else "  (%d elements, max |d| %.3g)" %
                              (nd, float((x.double() - y.double()).abs().max()))), flush=True)
        if not same:
            bad.append(name)

    for a_, lw in zip(angles, lws):
        d = torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda()
        whole.integrate(d, lw)
        for (a, b), T, Wt in zip(ranges, Ts, Ws):
            kernels.integrate_depth(T, Wt, d, K, Kinv, lw, scale, center, tdist, tsdf_res=R, res=(R, R, R), x_range=(a, b))
    for r, (a, b) in enumerate(ranges):
        cmp("canonical K1 slab %d T" % r, Ts[r], whole.T[a:b])
        cmp("canonical K1 slab %d w" % r, Ws[r], whole.Wt[a:b])

    def slab_samples(tag):
        parts = []
        for r, (a, b) in enumerate(ranges):
            Tp, Wp, x0 = [Ts[r]], [Ws[r]], a
            if r > 0:
                lo = Ts[r - 1][-1]
                Tp.insert(0, lo[None]); Wp.insert(0, torch.zeros_like(lo)[None]); x0 -= 1
            if r < P - 1:
                hi = Ts[r + 1][0]
                Tp.append(hi[None]); Wp.append(torch.zeros_like(hi)[None])
            Tp, Wp = torch.cat(Tp).contiguous(), torch.cat(Wp).contiguous()
            pos, nrm = extract_surface_samples(Tp, Wp, band, x0=x0)
            nbr, wts = sample_knn(pos, node_pos, node_w, knn, bricks=((R, R, R), (a, b), wss[r]))
            parts.append((pos, nrm, nbr, wts))
        pos, nrm, nbr, wts = [torch.cat([p[i] for p in parts]) for i in range(4)]
        wp, wn = extract_surface_samples(whole.T, whole.Wt, band, x0=0)
        wnbr, wwts = sample_knn(wp, node_pos, node_w, knn, bricks=whole.knn_bricks)
        print("%s: samples slabs %s = %d, whole %d" % (tag, [int(p[0].shape[0]) for p in parts], pos.shape[0], wp.shape[0]), flush=True)
        if pos.shape[0] != wp.shape[0]:
            bad.append(tag + " sample count")
            # which planes differ?
            cs = torch.bincount(pos[:, 0].round().long().clamp(0, R - 1), minlength=R)
            cw = torch.bincount(wp[:, 0].round().long().clamp(0, R - 1), minlength=R)
            dd = (cs != cw).nonzero().flatten().tolist()
            print("   planes (by rounded x) whose counts differ: %s" % dd[:40], flush=True)
            return
        cmp(tag + " sample pos", pos, wp)
        cmp(tag + " sample nrm", nrm, wn)
        cmp(tag + " sample knn idx", nbr, wnbr)
        cmp(tag + " sample knn wts", wts, wwts)
        nb2, wt2 = sample_knn(wp, node_pos, node_w, knn)
        cmp(tag + " whole bricks vs brute idx", wnbr, nb2)
        cmp(tag + " whole bricks vs brute wts", wwts, wt2)

    whole.refresh_samples()
    slab_samples("initial")
    first = True
    for f in range(nframes):
        off = np.array([0.10, -0.07, 0.05]) * (f + 1) * scale
        depths = [torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0, sphere_offset=off,
                                                      sphere_r=scene.SPHERE_R * (1.0 + 0.004 * (f + 1)))).cuda() for lw in lws]
        # live volume
        wsv = kernels.integrate_workspace(3, H, W, (R, R, R), (0, R), whole.live.device)
        kernels.integrate_depth_views(whole.live, whole.live_w, depths, K, Kinv, lws, scale, center, tdist, tsdf_res=R, res=(R, R, R),
                                      x_range=(0, R), workspace=wsv, fresh=tvox)
        for r, (a, b) in enumerate(ranges):
            if ws_views[r] is None:
                ws_views[r] = torch.empty_like(kernels.integrate_workspace(3, H, W, (R, R, R), (a, b), whole.live.device))
            kernels.integrate_depth_views(lives[r][0], lives[r][1], depths, K, Kinv, lws, scale, center, tdist, tsdf_res=R,
                                          res=(R, R, R), x_range=(a, b), workspace=ws_views[r], fresh=tvox)
            cmp("frame %d live K1 slab %d T" % (f, r), lives[r][0], whole.live[a:b])
            cmp("frame %d live K1 slab %d w" % (f, r), lives[r][1], whole.live_w[a:b])
        # solve on the whole grid's samples
        whole.fs.gn_iteration(depths, lws, rw=5.0, lm_abs=10.0, lm_rel=1e-2, max_dist=2.0, huber=0.5, n_iters=10)
        sv = whole.fs.solver
        dq = sv.node_dq.clone()
        kernels.fuse_volume_dqb(whole.T, whole.Wt, whole.live, sv.node_pos, dq, sv.node_w, knn, whole.ident_lw, tvox, res=(R, R, R),
                                x_range=(0, R), workspace=whole.ws_dqb, rebuild_candidates=first)
        for r, (a, b) in enumerate(ranges):
            kernels.fuse_volume_dqb(Ts[r], Ws[r], whole.live, sv.node_pos, dq, sv.node_w, knn, whole.ident_lw, tvox, res=(R, R, R),
                                    x_range=(a, b), workspace=wss[r], rebuild_candidates=first)
            cmp("frame %d K3 slab %d T" % (f, r), Ts[r], whole.T[a:b])
            cmp("frame %d K3 slab %d w" % (f, r), Ws[r], whole.Wt[a:b])
        first = False
        whole.refresh_samples()
        slab_samples("frame %d" % f)
        c, n = sv.cost()
        print("frame %d: cost %.9f valid %d samples %d" % (f, c, n, sv.S), flush=True)
    print("DIFFERENT STAGES: %s" % (bad if bad else "none"))


if __name__ == "__main__":
    main()
