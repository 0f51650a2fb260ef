"""CPU, world_size 2, gloo: the multi-GPU host logic -- slab partition, union of the block
pattern, single all-reduce of the flat normal-equation buffer, max-over-ranks timing."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dynamicfusion_body_amd import dist as D


def test_slab_range_covers_exactly():
    for n in (1, 7, 64, 256, 513):
        for ws in (1, 2, 3, 8):
            cur = 0
            sizes = []
            for r in range(ws):
                a, b = D.slab_range(n, r, ws)
                assert a == cur and b >= a
                sizes.append(b - a)
                cur = b
            assert cur == n and max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        D.slab_range(10, 2, 2)
    assert D.world() == (0, 1)
    t = torch.arange(4.0)
    assert D.allreduce_system(t) is t and D.max_over_ranks([1.0, 2.0]) == [1.0, 2.0]
    assert torch.equal(D.union_sorted_keys(torch.tensor([5, 1, 5, 3])), torch.tensor([1, 3, 5]))
    v = torch.zeros(4, 2, 2)
    assert D.allgather_planes(v, 4) is v and D.halo_planes(v, 4) == (None, None)
    assert D.all_ranks(True) is True and D.all_ranks(False) is False


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        assert D.world() == (rank, ws)
        # slabs of a 10-plane grid: each rank contributes the node pairs its own samples touch
        a, b = D.slab_range(10, rank, ws)
        keys = torch.tensor([[0, 4, 9], [4, 7, 11, 12]][rank], dtype=torch.int64)
        u = D.union_sorted_keys(keys)
        assert torch.equal(u, torch.tensor([0, 4, 7, 9, 11, 12]))
        # normal equations: partial sums per slab add up to the full system
        full = torch.arange(20, dtype=torch.float64) * 0.5
        part = torch.zeros(20, dtype=torch.float64)
        part[2 * a:2 * b] = full[2 * a:2 * b]
        D.allreduce_system(part)
        assert torch.equal(part, full)
        mx = D.max_over_ranks([float(rank), 1.0 - rank])
        assert mx == [float(ws - 1), 1.0]
        # collective decisions (keep / rebuild the block pattern): true only if true on every rank
        assert D.all_ranks(True) is True and D.all_ranks(rank == 0) is False and D.all_ranks(False) is False
        # live volume: slabs -> whole volume on every rank (even and uneven plane counts)
        for planes in (10, 7):
            vol = torch.arange(planes * 3 * 2, dtype=torch.float32).reshape(planes, 3, 2)
            pa, pb = D.slab_range(planes, rank, ws)
            assert torch.equal(D.allgather_planes(vol[pa:pb].clone(), planes), vol)
        for planes in (10, 7):
            vol = torch.arange(planes * 3 * 2, dtype=torch.float32).reshape(planes, 3, 2)
            pa, pb = D.slab_range(planes, rank, ws)
            lo, hi = D.halo_planes(vol[pa:pb].clone(), planes)
            assert (lo is None) == (pa == 0) and (hi is None) == (pb == planes)
            assert lo is None or torch.equal(lo, vol[pa - 1])
            assert hi is None or torch.equal(hi, vol[pb])
        # ragged rows (the unsupported surface points of update_graph): rank-ordered concatenation, empty contributions
        import numpy as np
        mine = [np.arange(15, dtype=np.float64).reshape(5, 3), np.zeros((0, 3))][rank]
        allr = D.gather_rows(mine)
        assert allr.shape == (5, 3) and np.array_equal(allr, np.arange(15, dtype=np.float64).reshape(5, 3))
        mine = np.full((2 + 3 * rank, 3), float(rank))
        allr = D.gather_rows(mine, device="cpu")                   # (the device choice is one argument: "cuda" under RCCL)
        assert allr.shape == (7, 3) and (allr[:2] == 0).all() and (allr[2:] == 1).all()
        assert D.collective_device().type == "cpu"                 # gloo
        try:
            D.allgather_planes(torch.zeros(1, 3, 2), 10)
            raise AssertionError("wrong slab size not rejected")
        except ValueError:
            pass
        out[rank] = 1
    finally:
        dist.destroy_process_group()


def test_two_ranks_gloo():
    ws = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Array("i", [0] * ws)
    procs = [ctx.Process(target=_worker, args=(r, ws, port, out)) for r in range(ws)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs)
    assert list(out) == [1] * ws


def test_spawn_ranks_runs_a_gloo_job(tmp_path):
    """launch.spawn_ranks (what `python bench.py --gpus N` uses to start its own ranks): N fresh processes with the
    torch.distributed environment, rank 0's stdout relayed, worst exit code returned, survivors of a failed rank ended."""
    import subprocess
    import sys
    script = tmp_path / "job.py"
    script.write_text(
        "import os, sys, torch, torch.distributed as dist\n"
        "dist.init_process_group('gloo')\n"
        "t = torch.tensor([float(dist.get_rank() + 1)])\n"
        "dist.all_reduce(t)\n"
        "print('SUM', int(t.item()), os.environ['WORLD_SIZE'], os.environ['LOCAL_RANK'])\n"
        "dist.destroy_process_group()\n"
        "sys.exit(int(sys.argv[1]) if dist.is_available() and os.environ['RANK'] == '1' else 0)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); from dynamicfusion_body_amd import launch; "
            "sys.exit(launch.spawn_ranks([%r, sys.argv[1]], 3, timeout=120))" % (root, str(script)))
    r = subprocess.run([sys.executable, "-c", code, "0"], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if not l.startswith("[Gloo]")]     # (gloo's own banner)
    assert lines == ["SUM 6 3 0"]                                       # only rank 0's stdout comes through
    r = subprocess.run([sys.executable, "-c", code, "7"], capture_output=True, text=True, timeout=180)
    assert r.returncode == 7
