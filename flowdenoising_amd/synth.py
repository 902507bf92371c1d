"""Synthetic volumes for tests and bench.py (BASELINE.md section 3 / SURVEY.md 8d).

~32 random 3-D Gaussian blobs (radius 4-24 voxels, amplitude U(0.5, 1.5)) plus a slow 3-D
sinusoid, a per-slice sub-pixel drift (0.3 sin(2 pi z/97), 0.3 cos(2 pi z/131)) px so that
the optical flow is non-trivial, plus additive N(0, 0.25^2) noise; float32.

`amplitude` scales the whole volume.  OpenCV's Farneback solve is regularised with an
absolute +1e-3 on the determinant, so on unit-range data every flow collapses to ~0 and
the warp degenerates to the identity; microscopy intensities are 8/16-bit-like, so the
default scale is 100 to keep the flow path honest (flows of a few tenths of a pixel).

Works with numpy (tests, CPU) or torch (bench.py generates straight into HBM).
"""
import math

import numpy as np


def blob_params(shape, seed, nblobs=32):
    Z, Y, X = shape
    rng = np.random.default_rng(seed)
    c = rng.uniform(0, 1, size=(nblobs, 3)) * np.array([Z, Y, X])
    rad = rng.uniform(4, 24, size=nblobs)
    amp = rng.uniform(0.5, 1.5, size=nblobs)
    return c, rad, amp


def make_volume(shape, seed=1234, amplitude=100.0, noise=0.25, nblobs=32, xp=np, device=None, z0=0, zlen=None,
                noise_seed=None):
    """Return a (zlen, Y, X) float32 slab [z0, z0+zlen) of the synthetic volume of `shape`.

    The deterministic part depends only on (shape, seed), so slabs generated on different
    ranks tile the same volume; the noise of slab z0 is seeded with noise_seed + z0."""
    Z, Y, X = shape
    zlen = Z - z0 if zlen is None else zlen
    c, rad, amp = blob_params(shape, seed, nblobs)
    is_torch = xp is not np
    if is_torch:
        import torch
        kw = dict(dtype=torch.float32, device=device)
        z = torch.arange(z0, z0 + zlen, **kw).view(-1, 1, 1)
        y = torch.arange(Y, **kw).view(1, -1, 1)
        x = torch.arange(X, **kw).view(1, 1, -1)
        exp, sin, cos = torch.exp, torch.sin, torch.cos
        vol = torch.zeros((zlen, Y, X), **kw)
    else:
        z = np.arange(z0, z0 + zlen, dtype=np.float32).reshape(-1, 1, 1)
        y = np.arange(Y, dtype=np.float32).reshape(1, -1, 1)
        x = np.arange(X, dtype=np.float32).reshape(1, 1, -1)
        exp, sin, cos = np.exp, np.sin, np.cos
        vol = np.zeros((zlen, Y, X), dtype=np.float32)
    dx = 0.3 * sin(z * (2 * math.pi / 97.0))   # drift of slice z, in pixels
    dy = 0.3 * cos(z * (2 * math.pi / 131.0))
    for (cz, cy, cx), r, a in zip(c, rad, amp):
        inv = -0.5 / (r * r)
        gz = exp((z - cz) ** 2 * inv)
        gy = exp((y - cy - dy) ** 2 * inv)
        gx = exp((x - cx - dx) ** 2 * inv)
        vol += float(a) * gz * gy * gx
    vol += 0.5 * sin(z * (2 * math.pi / max(Z, 1))) * sin((y - dy) * (2 * math.pi / max(Y, 1) * 2)) \
        * cos((x - dx) * (2 * math.pi / max(X, 1) * 3))
    ns = (seed if noise_seed is None else noise_seed) * 1000003 + z0
    if is_torch:
        import torch
        g = torch.Generator(device=device)
        g.manual_seed(int(ns))
        vol += noise * torch.randn((zlen, Y, X), generator=g, **kw)
        vol *= amplitude
        return vol
    rng = np.random.default_rng(ns)
    vol += (noise * rng.standard_normal((zlen, Y, X))).astype(np.float32)
    vol *= np.float32(amplitude)
    return vol.astype(np.float32)
