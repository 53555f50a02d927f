"""Seeded synthetic scenes: no ScanNet/ScanRefer data is needed.

`scene_xyz` is shared by the parity tests and bench.py so both see the same
point distributions, including the adversarial rows the reference's tie/skip
rules care about (exact duplicates, |p|^2 <= 1e-3 points, all-zero rows).
"""
import numpy as np


def scene_xyz(batch, n, seed=42, mode="volume", adversarial=True):
    """(batch, n, 3) float32 room-like point sets.

    mode "volume": uniform in [-3,3]x[-3,3]x[0,2.5] m (ball queries never exit
    early = worst case).  mode "surface": floor + 4 walls + faces of 20 random
    boxes (denser balls, early exits like real scans).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.empty((batch, n, 3), np.float32)
    for b in range(batch):
        if mode == "volume":
            p = rng.uniform([-3, -3, 0], [3, 3, 2.5], size=(n, 3))
        else:
            p = _surface(rng, n)
        p = p.astype(np.float32)
        if adversarial and n >= 256:
            k = max(1, min(64, n // 64))
            src = rng.integers(0, n, k)
            dst = rng.integers(0, n, k)
            p[dst] = p[src]                      # exact duplicates
            near = rng.integers(1, n, 8)
            p[near] = rng.uniform(-0.015, 0.015, size=(8, 3)).astype(np.float32)
            p[rng.integers(1, n, 8)] = 0.0        # all-zero rows
        out[b] = p
    return out


def _surface(rng, n):
    parts = []
    n_floor = n // 3
    f = rng.uniform([-3, -3, 0], [3, 3, 0], size=(n_floor, 3))
    parts.append(f)
    n_wall = n // 3
    w = rng.uniform([-3, -3, 0], [3, 3, 2.5], size=(n_wall, 3))
    side = rng.integers(0, 4, n_wall)
    w[side == 0, 0] = -3
    w[side == 1, 0] = 3
    w[side == 2, 1] = -3
    w[side == 3, 1] = 3
    parts.append(w)
    n_box = n - n_floor - n_wall
    centers = rng.uniform([-2.5, -2.5, 0.3], [2.5, 2.5, 1.2], size=(20, 3))
    sizes = rng.uniform(0.3, 1.2, size=(20, 3))
    which = rng.integers(0, 20, n_box)
    q = rng.uniform(-0.5, 0.5, size=(n_box, 3))
    face = rng.integers(0, 3, n_box)
    sign = rng.choice([-0.5, 0.5], n_box)
    q[np.arange(n_box), face] = sign
    parts.append(centers[which] + q * sizes[which])
    return np.concatenate(parts, 0)
