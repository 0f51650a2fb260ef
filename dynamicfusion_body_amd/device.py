"""Device-memory plumbing on PyTorch-ROCm tensors (torch is the allocator / stream
provider here, not the compute path)."""
import numpy as np
import torch

from . import _lib


def require_gpu():
    if not torch.cuda.is_available():
        raise _lib.DfhError("no HIP device visible: this path only runs on the GPU (no CPU fallback)")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def current_stream_ptr():
    """hipStream_t of torch's current stream on the current device (every launch asks: torch.cuda.current_stream() builds a
    Stream object, ~8 us; the raw queries are well under 1 us)."""
    if _raw_stream is None or _raw_device is None:
        return torch.cuda.current_stream().cuda_stream
    return _raw_stream(_raw_device())


def dtype_code(t):
    if t.dtype == torch.float32:
        return _lib.F32
    if t.dtype == torch.float64:
        return _lib.F64
    raise ValueError("only float32 / float64 device tensors are supported, got %s" % t.dtype)


def torch_dtype(d):
    if isinstance(d, torch.dtype):
        if d not in (torch.float32, torch.float64):
            raise ValueError("volume dtype must be float32 or float64")
        return d
    return {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}[np.dtype(d)]


def to_device(a, dtype=None, device=None):
    """numpy array or tensor -> contiguous device tensor (copies only when needed)."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    if isinstance(a, torch.Tensor):
        t = a.to(device=device, dtype=dtype or a.dtype)
    else:
        t = torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dtype)
    return t.contiguous()


def f32_exact(a):
    """True if every value of the float array survives a round trip through float32."""
    a = np.asarray(a)
    if a.dtype == np.float32:
        return True
    return bool(np.array_equal(a.astype(np.float32).astype(a.dtype), a))
