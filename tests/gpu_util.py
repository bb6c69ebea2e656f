"""Helpers for the -m gpu parity tests: device buffers come from torch (plumbing only),
every compute call goes through the C-ABI of libmckernels.so."""
import numpy as np


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def zeros(*shape):
    import torch
    return torch.zeros(*shape, dtype=torch.float32, device="cuda")


def nans(*shape):
    """scaleTargets == 0 targets must be write-only: poison them."""
    import torch
    return torch.full(shape, float("nan"), dtype=torch.float32, device="cuda")


def ptr(t):
    return t.data_ptr()


def host(t):
    import torch
    torch.cuda.synchronize()
    return t.cpu().numpy()


def relerr(got, exp):
    """max |got - exp| relative to the RMS magnitude of exp (fp32 contraction tolerance metric)."""
    exp = exp.astype(np.float64)
    scale = max(float(np.sqrt((exp ** 2).mean())), 1e-30)
    return float(np.abs(got.astype(np.float64) - exp).max() / scale)


_KEEP = []


def keep(t):
    """Hold a reference so a temporary device buffer is not recycled by torch's caching
    allocator before the (asynchronous) kernel that reads it has run."""
    _KEEP.append(t)
    if len(_KEEP) > 64:
        import torch
        torch.cuda.synchronize()
        del _KEEP[:32]
    return t
