"""Multi-GPU layout of the hot path: one process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI), the voxel grid cut into contiguous slabs along array axis 0 (the slowest axis of
[x][y][z]; BASELINE north_star calls it the "Z-slab"), no exchange for TSDF integration, and ONE
all-reduce(sum) per Gauss-Newton iteration over the flat buffer {J^T J blocks | J^T r | cost, count}
(a few MB: latency-bound on xGMI, so it is a single collective, never one per tensor).

Every function here also works on CPU tensors with the gloo backend (tests/test_dist_gloo.py)."""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


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


def allreduce_system(flat):
    """Sum the flat normal-equation buffer over ranks (in place, one collective)."""
    _, ws = world()
    if ws > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def max_over_ranks(values):
    """Element-wise MAX of a small list of floats over ranks (timing)."""
    _, ws = world()
    if ws == 1:
        return list(values)
    dev = "cuda" if (dist.get_backend() == "nccl" and torch.cuda.is_available()) else "cpu"
    t = torch.tensor(list(values), dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t]
