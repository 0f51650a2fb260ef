"""Device-memory plumbing on PyTorch-ROCm tensors (torch is the allocator / stream
provider here, not the compute path)."""
import os
import time

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


class HostScalar:
    """A non-negative integer that a kernel stores straight into pinned host memory (the pointer is handed to the C-ABI where it
    asks for a device scalar: pinned memory is device-accessible at its host address).  The host pre-sets -1 and get() spins on
    the word until the kernel's store lands -- a few microseconds after the kernel ends, where a queued device-to-host copy
    behind an event took ~30 us during which the device had nothing to do (profiles/r2_gn_experiments.txt section 6).
    Only for scalars written by ONE plain store of the launch sequence's last writer (no atomics across PCIe).
    Option py_no_host_scalars (read once, at the first HostScalar) goes back to a device scalar and .item()."""
    _pool = {}
    enabled = True           # (set False by tests, or once at import by the option py_no_host_scalars -- see below the class)

    _option_read = False

    def __init__(self, dtype=torch.int32, n=1):
        """n > 1: that many words (each written once by the launch sequence); get() then returns a tuple."""
        self._dtype, self._n = dtype, int(n)
        self._np = None
        if not HostScalar._option_read:                # (once per process: the library's option table, not the environment)
            HostScalar._option_read = True
            if _lib.opt_on("py_no_host_scalars"):
                HostScalar.enabled = False
        if HostScalar.enabled:
            free = HostScalar._pool.setdefault((dtype, self._n), [])
            self._t = free.pop() if free else torch.empty(self._n, dtype=dtype).pin_memory()
            self._np = self._t.numpy()
            self._np[:] = -1
        else:
            self._t = torch.empty(self._n, dtype=dtype, device="cuda")

    def ptr(self):
        return self._t.data_ptr()

    def get(self, timeout=2.0):
        if self._np is None:                           # (a device scalar: created while the host-visible form was off)
            v = self._t.tolist()
            return int(v[0]) if self._n == 1 else tuple(int(x) for x in v)
        a = self._np
        spins = 0
        t_end = None
        while (a[0] == -1) if self._n == 1 else bool((a == -1).any()):
            spins += 1
            if spins & 0x3ff == 0:                     # (look at the clock every ~1000 reads only)
                now = time.monotonic()
                if t_end is None:
                    t_end = now + timeout
                elif now > t_end:
                    torch.cuda.synchronize()           # raises if a launch of the sequence failed
                    if (a == -1).any():
                        raise _lib.DfhError("a kernel's host-visible scalar never arrived")
                    # it arrived with the synchronisation only (pinned memory that is not host-coherent in this process, or a
                    # device busy for seconds): correct, but not worth spinning for again
                    HostScalar.enabled = False
        v = int(a[0]) if self._n == 1 else tuple(int(x) for x in a)
        HostScalar._pool[(self._dtype, self._n)].append(self._t)
        self._t = self._np = None
        return v


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
