"""Device-resident constants, created once per (key, device, dtype).

Creating a tensor from host data (`torch.tensor(list, device=...)`,
`from_numpy().to(dev)`) is a pageable H2D copy: a hidden host sync in eager mode
and illegal inside hipGraph capture.  Every small constant of the hot path goes
through this cache instead (first touch happens during the un-captured warm-up
steps)."""
import numpy as np
import torch

_CACHE = {}


def const(key, values, device, dtype=torch.float32):
    dev = torch.device(device)
    k = (key, dev.type, dev.index, dtype)
    t = _CACHE.get(k)
    if t is None:
        t = torch.as_tensor(np.asarray(values), dtype=dtype).to(dev)
        _CACHE[k] = t
    return t


def array_const(arr, device, dtype=torch.float32):
    """Cache keyed by the array's content (for small arrays such as mean sizes)."""
    a = np.ascontiguousarray(np.asarray(arr))
    return const(("arr", a.shape, a.tobytes()), a, device, dtype)
