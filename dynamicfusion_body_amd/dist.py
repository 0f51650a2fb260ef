"""Multi-GPU layout of the hot path: one process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI), the voxel grid cut into contiguous slabs along array axis 0 (the slowest axis of
[x][y][z]; BASELINE north_star calls it the "Z-slab"), no exchange for TSDF integration, and ONE
all-reduce(sum) per Gauss-Newton iteration over the flat buffer {J^T J blocks | J^T r | cost, count}
(a few MB: latency-bound on xGMI, so it is a single collective, never one per tensor).
The TSDF->TSDF update (K2/K3) samples the LIVE volume at warped positions, i.e. across slab faces: the
live slabs are all-gathered once per frame (`allgather_planes`: one collective of the whole volume,
bandwidth-bound: 67 MB at 256^3).

Every function here also works on CPU tensors with the gloo backend (tests/test_dist_gloo.py)."""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def ranks_share_a_gpu():
    """True when this node runs more ranks than it has GPUs (several processes time-sharing one card)."""
    import os
    _, ws = world()
    if ws == 1 or not torch.cuda.is_available():
        return False
    local = int(os.environ.get("LOCAL_WORLD_SIZE", ws))
    return local > torch.cuda.device_count()


def slab_range(n_planes, rank, world_size):
    """Planes [a, b) of axis 0 owned by `rank`: contiguous, covering, sizes differ by at most 1."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank %d / world size %d" % (rank, world_size))
    base, rem = divmod(int(n_planes), world_size)
    a = rank * base + min(rank, rem)
    return a, a + base + (1 if rank < rem else 0)


def union_sorted_keys(keys):
    """Union over ranks of sorted unique int64 key tensors (block pattern of J^T J): every rank
    ends up with the identical sorted tensor.  Once per frame."""
    rank, ws = world()
    keys = torch.unique(keys)
    if ws == 1:
        return keys
    n = torch.tensor([keys.numel()], dtype=torch.int64, device=keys.device)
    sizes = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(sizes, n)
    m = int(max(int(s) for s in sizes))
    pad = torch.full((m,), -1, dtype=torch.int64, device=keys.device)
    pad[:keys.numel()] = keys
    parts = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(parts, pad)
    allk = torch.cat(parts)
    return torch.unique(allk[allk >= 0])


def allgather_planes(local, n_planes):
    """Slabs of axis 0 (rank r holds planes slab_range(n_planes, r, world)) -> the whole volume on every
    rank.  One all-gather of equal-sized (padded by at most one plane) buffers straight into the result."""
    rank, ws = world()
    if ws == 1:
        return local
    a, b = slab_range(n_planes, rank, ws)
    if local.shape[0] != b - a:
        raise ValueError("rank %d holds %d planes, its slab has %d" % (rank, local.shape[0], b - a))
    plane = local.shape[1:]
    m = -(-int(n_planes) // ws)                                   # largest slab
    if n_planes % ws == 0:
        full = torch.empty((n_planes,) + tuple(plane), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(full, local.contiguous())
        return full
    buf = torch.zeros((m,) + tuple(plane), dtype=local.dtype, device=local.device)
    buf[:b - a] = local
    gathered = torch.empty((ws * m,) + tuple(plane), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, buf)
    full = torch.empty((n_planes,) + tuple(plane), dtype=local.dtype, device=local.device)
    for r in range(ws):
        ra, rb = slab_range(n_planes, r, ws)
        full[ra:rb] = gathered[r * m:r * m + (rb - ra)]
    return full


def halo_planes(local, n_planes):
    """(plane a-1, plane b) of the neighbouring slabs, None at the ends of the grid: every rank contributes
    its first and last plane to one all-gather (2 planes per rank: 0.5 MB at 256^2 fp32)."""
    rank, ws = world()
    if ws == 1:
        return None, None
    a, b = slab_range(n_planes, rank, ws)
    if local.shape[0] != b - a:
        raise ValueError("rank %d holds %d planes, its slab has %d" % (rank, local.shape[0], b - a))
    mine = torch.zeros((2,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if b > a:
        mine[0] = local[0]
        mine[1] = local[-1]
    allp = torch.empty((2 * ws,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(allp, mine)
    lo = hi = None
    for r in range(rank - 1, -1, -1):                 # nearest non-empty slab below / above
        ra, rb = slab_range(n_planes, r, ws)
        if rb > ra:
            lo = allp[2 * r + 1]
            break
    for r in range(rank + 1, ws):
        ra, rb = slab_range(n_planes, r, ws)
        if rb > ra:
            hi = allp[2 * r]
            break
    return lo, hi


def collective_device(device=None):
    """The device small host-side values travel on: RCCL ("nccl") moves GPU tensors only, gloo moves CPU ones (and GPU
    ones through a host copy).  `device` overrides (tests rehearse the GPU plumbing over gloo)."""
    if device is not None:
        return torch.device(device)
    return torch.device("cuda" if (dist.get_backend() == "nccl" and torch.cuda.is_available()) else "cpu")


def gather_rows(rows, device=None):
    """Concatenation over ranks, in rank order, of per-rank (n_r, C) float64 numpy arrays (n_r differs from rank to rank):
    every rank gets the same (sum n_r, C) array.  Sizes, padded buffers and outputs all live on ONE device chosen once
    (`collective_device`), so the same code runs over RCCL and over gloo."""
    import numpy as np
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    if rows.ndim != 2:
        raise ValueError("gather_rows expects a 2-D array, got shape %s" % (rows.shape,))
    _, ws = world()
    if ws == 1:
        return rows
    dev = collective_device(device)
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(sizes, n)
    sizes = [int(s_) for s_ in sizes]
    m = max(1, max(sizes))
    buf = torch.zeros((m, rows.shape[1]), dtype=torch.float64, device=dev)
    if rows.shape[0]:
        buf[:rows.shape[0]] = torch.from_numpy(rows).to(dev)
    parts = [torch.empty_like(buf) for _ in range(ws)]
    dist.all_gather(parts, buf)
    return np.concatenate([p_[:n_].cpu().numpy() for p_, n_ in zip(parts, sizes)])


def allgather_ragged(t):
    """Rank-ordered concatenation of 2-D tensors whose row counts differ from rank to rank (same dtype, same columns, all on
    the collective's device): sizes (one tiny all-gather), then ONE all-gather of buffers padded to the longest.  Every rank
    gets the same (sum n_r, C) tensor.  Used once per frame by the replicated solve (the slabs' samples)."""
    _, ws = world()
    if ws == 1:
        return t
    if t.dim() != 2:
        raise ValueError("allgather_ragged expects a 2-D tensor")
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(sizes, n)
    sizes = [int(s_) for s_ in sizes]
    m = max(1, max(sizes))
    buf = torch.zeros((m, t.shape[1]), dtype=t.dtype, device=t.device)
    buf[:t.shape[0]] = t
    out = torch.empty((ws * m, t.shape[1]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, buf)
    return torch.cat([out[r * m:r * m + sizes[r]] for r in range(ws)])


# ---- which way to run the warp solve on P GPUs -------------------------------------------------------------------------
# Per GN iteration (microseconds; single-GPU parts measured on MI355X: profiles/r2j_kernel_stats.csv, r3_*; the collective's
# figures are ESTIMATES from the link rates -- no multi-GPU node was available to any round of this build):
#   rows(S)      the data-row launch: 31 us at 335 k samples, linear in the local sample count
#   gather(S)    block walk: ~14 us at 335 k samples; the rows it walks are the local ones
#   pcg(N)       43 us at 512 nodes, 55 at 2 048 (10 iterations, persistent kernel), run by every rank redundantly
#   allreduce    alpha + bytes * 2 (P - 1) / P / (links * rate): alpha ~ 35 us (RCCL launch + cross-GPU hand-offs of a
#                latency-bound message), 7 xGMI links at ~48 GB/s sustained each way
#   eager        ~10 us: a sharded iteration is two library calls with torch's collective between them instead of one call per
#                frame.  MEASURED with a 1-rank RCCL group on one MI355X (tools/rccl_capture_check.py,
#                profiles/r3_rccl_capture_1rank.json): +17 us per iteration eager (110.5 against 93.2), of which 7 us are device
#                time (pack + a 1-rank all-reduce + unpack: 98.6 against 91.5 us as hipGraph replays -- the all-reduce CAPTURES
#                into a hipGraph with the solve's kernels around it and replays to the same bits)
# sharded(P)    = rows / P + gather / P + allreduce + pcg + eager
# replicated(P) = rows + gather + pcg                 (+ one all-gather of the samples per FRAME: 96 B each)
# At config 3 (335 k samples, 512 nodes, 0.93 MB of upper triangle) sharded(8) ~ 4 + 2 + 40 + 43 + 10 = 99 us against 96 replicated
# and 88 on one GPU: the collective costs what the sample-parallel work saves -- GN-iters/s would not rise with the GPU count.  At
# config 4 (828 k samples, 2 048 nodes, 3.7 MB) ~ 10 + 4 + 54 + 55 + 10 = 133 against 187 (166 on one GPU): sharded wins from P = 2 up.  So the
# default is the replicated solve unless the local work outweighs the collective (DESIGN.md section 6 has the table).
def solve_mode(n_samples_total, n_nodes, n_blocks, world_size):
    """'replicated' or 'sharded' for the warp solve on world_size GPUs, by the latency model above."""
    P = int(world_size)
    if P <= 1:
        return "replicated"
    rows = 31.0 * n_samples_total / 335e3
    gather = 14.0 * n_samples_total / 335e3
    pcg = 43.0 + 12.0 * max(0.0, (n_nodes - 512) / 1536.0)
    nbytes = 8.0 * (36.0 * (n_blocks + n_nodes) / 2.0 + 6.0 * n_nodes + 2.0)    # upper block triangle | J^T r | cost, count
    allreduce = 35.0 + nbytes * 2.0 * (P - 1) / P / (7 * 48e3)          # bytes / (MB/s -> us): 48 GB/s = 48e3 bytes per us
    eager = 10.0
    frame_gather = 96.0 * n_samples_total * (P - 1) / P / (7 * 48e3) / 10.0      # per iteration of a 10-iteration frame
    sharded = (rows + gather) / P + allreduce + pcg + eager
    replicated = rows + gather + pcg + frame_gather
    return "sharded" if sharded < replicated else "replicated"


def all_ranks(flag):
    """True iff `flag` is true on every rank (one tiny all-reduce; ranks must take collective decisions alike)."""
    _, ws = world()
    if ws == 1:
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=collective_device())
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t[0]))


def allreduce_system(flat, force=False):
    """Sum the flat normal-equation buffer over ranks (in place, one collective).  force: issue the collective in a group of one
    too (a rehearsal of the RCCL path on one GPU)."""
    _, ws = world()
    if ws > 1 or (force and dist.is_initialized()):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def max_over_ranks(values):
    """Element-wise MAX of a small list of floats over ranks (timing)."""
    _, ws = world()
    if ws == 1:
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64, device=collective_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t]
