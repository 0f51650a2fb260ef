import os, socket, sys
import numpy as np, torch, torch.distributed as dist, torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

def worker(rank, ws, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from dynamicfusion_body_amd import scene
    from dynamicfusion_body_amd import dist as D
    from dynamicfusion_body_amd.pipeline import SlabFrame
    torch.cuda.set_device(0)
    R, N = 63, 40
    H, W, fx, cx, cy = scene.CAMERAS["C1"]
    K = scene.intrinsics(fx, cx, cy)
    scale, center, tdist = scene.grid_params(R)
    node_pos, node_w = scene.fibonacci_nodes(N, R)
    lw_cam = scene.view_extrinsic(0.0)
    d1 = torch.from_numpy(scene.render_depth(K, lw_cam, H, W, dtype=np.float32, sphere_offset=np.array([0.3, -0.2, 0.15]) * scale, sphere_r=scene.SPHERE_R * 1.01)).cuda()
    for mode in ("sharded", "whole"):
        sf = SlabFrame(K, scale, center, R, tdist / scale, node_pos, node_w, knn=4, pcg_iters=300, band=2.0, distributed=(mode == "sharded"))
        for ang in (0.0, 40.0, -40.0):
            lw = scene.view_extrinsic(ang)
            d = torch.from_numpy(scene.render_depth(K, lw, H, W, dtype=np.float32, invalid_frac=0.0)).cuda()
            sf.integrate(d, lw)
        S = sf.refresh_samples()
        tot = torch.tensor([float(S)], dtype=torch.float64)
        if mode == "sharded": dist.all_reduce(tot)
        log = ["%s rank %d: samples %d (total %d)" % (mode, rank, S, int(tot.item()))]
        for it in range(3):
            sf.fs.gn_iteration(d1, lw_cam, rw=0.05, lm_abs=float(os.environ.get('LM','1.0')), lm_rel=1e-2, max_dist=4.0)
            c, n = sf.fs.solver.cost()
            log.append("  it %d cost %.6f valid %d |dx| %.6f |dq-I| %.6f" % (it, c, n, float(sf.fs.solver.dx.norm()), float((sf.fs.solver.node_dq[:, 1:]).norm())))
        print("\n".join(log), flush=True)
    dist.destroy_process_group()

if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, 2, port)) for r in range(2)]
    [p.start() for p in ps]; [p.join(200) for p in ps]
