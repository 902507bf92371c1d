"""Test infrastructure: par's calling pattern for the injected pair operators, written axis-generically.

src/flowdenoising.py hands `get_flow` / `warp_slice` to `FlowDenoising(P, vol, l, w, get_flow, warp_slice)` (par:506) and
then calls them from P pool threads at once: `PoolExecutor(max_workers=P).map(filter_along_Z_chunk, ...)` over P contiguous
chunks of `dim // P` target slices plus a remainder round of single slices (par:181-206), every thread running
`filter_along_*_slice` (par:306-373) on its own targets: wrap-around neighbours taken as VIEWS of the shared volume
(`vol[:, (y + i - ks2) % Y, :]` is row-strided, `vol[:, :, (x + i - ks2) % X]` element-strided), a chain of in-place flows
per side, the result stored in an array of the volume's dtype (par:131) and copied back after the pass (par:287-289).
All three passes are kept (par itself loses its X pass, par:290 + par:520).
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def _slice_of(vol, axis, s):
    """The view par passes: vol[s, :, :], vol[:, s, :] or vol[:, :, s]."""
    return vol[s, :, :] if axis == 0 else vol[:, s, :] if axis == 1 else vol[:, :, s]


def filter_slice(vol, filtered_vol, axis, s, kernel, l, w, get_flow, warp_slice):
    """par:306-327 (Z), 329-350 (Y), 352-373 (X)."""
    ks2 = kernel.size // 2
    n = vol.shape[axis]
    target = _slice_of(vol, axis, s)
    tmp_slice = np.zeros_like(target).astype(np.float32)
    for side in (range(ks2 - 1, -1, -1), range(ks2 + 1, kernel.size)):
        if side.start > ks2:
            tmp_slice += target * kernel[ks2]
        prev_flow = np.zeros(shape=target.shape + (2,), dtype=np.float32)
        for i in side:
            reference = _slice_of(vol, axis, (s + i - ks2) % n)
            flow = get_flow(reference, target, l, w, prev_flow)
            prev_flow = flow
            tmp_slice += warp_slice(reference, flow) * kernel[i]
    if axis == 0:
        filtered_vol[s, :, :] = tmp_slice
    elif axis == 1:
        filtered_vol[:, s, :] = tmp_slice
    else:
        filtered_vol[:, :, s] = tmp_slice


def par_sweep(get_flow, warp_slice, vol, kernels, l, w, P):
    """par's passes with its own scheduler: P chunks of dim // P targets on P threads, then the remainder round (par:181-206).
    P = 1 runs the same loops on the calling thread."""
    vol = vol.copy()
    for axis, kernel in enumerate(kernels):
        if kernel is None:
            continue
        filtered_vol = np.zeros_like(vol)
        dim = vol.shape[axis]

        def chunk(index, size, offset):
            for q in range(size):
                filter_slice(vol, filtered_vol, axis, index * size + q + offset, kernel, l, w, get_flow, warp_slice)
            return index

        size = dim // P
        rounds = [[(i, size, 0) for i in range(P)]]
        if dim % P:
            rounds.append([(i, 1, size * P) for i in range(dim % P)])
        for jobs in rounds:
            if P == 1:
                for j in jobs:
                    chunk(*j)
                continue
            with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
                for _ in ex.map(lambda j: chunk(*j), jobs):
                    pass
        vol[...] = filtered_vol
    return vol
