"""Out-of-core OF_filter: volumes that do not fit the GPU.

The volume stays in host memory (numpy array or memory-mapped MRC).  Every axis pass is cut into
chunks of target slices ALONG ITS OWN AXIS -- the same decomposition the multi-GPU engine uses
(distributed.py), executed chunk after chunk on one GPU with host memory as the exchange medium:

    for axis in Z, Y, X:   for chunk of slices along axis:
        stack = chunk + K//2 halo slices either side (mean-padded or wrapped at the volume ends),
                re-oriented so that slices are outermost (seq:255, seq:333)      -> H2D
        fdn_sweep_stack_dev(stack)                                               -> D2H into `out`

Results are bit-identical to OF_filter on the whole volume: target slices are independent, each
chain restarts from zero flow (seq:94,109), and the pad value is the mean of the whole volume
(seq:420, numpy's float32 reduction through fdn_mean_host).  The idea generalises the resident
"chunk + kernel.size" slices of tests/flowdenoising_reviewer_solution2.py:496-508 in the reference.
"""
import numpy as np

from . import _lib
from .operators import OF_ITERS, OF_POLY_N, OF_POLY_SIGMA, handle

# device bytes per pixel of a target slice of a pass: stack + output + polynomial expansion + two flow
# buffers, with room for a pyramid and the general path's matrices
_BYTES_PER_TARGET_PIXEL = 96


def auto_chunk(shape, axis, halo, free_bytes):
    """Target slices per chunk such that a pass over it fits `free_bytes` of device memory."""
    img = int(np.prod([s for a, s in enumerate(shape) if a != axis]))
    budget = int(free_bytes * 0.7) - 2 * halo * img * 24      # halo slices: stack + expansion only
    return int(max(1, min(shape[axis], budget // (img * _BYTES_PER_TARGET_PIXEL))))


def filter_streamed(vol, kernels, l, w, chunk_slices=None, use_of=True, border_mode=_lib.BORDER_MEAN_PAD,
                    chained=True, device=0, mean=None, workers=2):
    """OF_filter / no_OF_filter (seq:419-431) with at most `chunk_slices` target slices of a pass per worker on
    the GPU (None: as many as fit).  `workers` chunks are in flight at once, each on its own handle and
    stream, so that one chunk's host copies and transfers overlap another's kernels.
    `vol`: (Z, Y, X) array-like on the host; returns a new float32 array."""
    src = np.asarray(vol)
    if src.ndim != 3:
        raise ValueError(f"expected a (Z, Y, X) volume, got shape {src.shape}")
    if src.dtype != np.float32:
        src = src.astype(np.float32)
    if mean is None:
        mean = _lib.mean_host(src)                      # seq:420
    wrap = border_mode == _lib.BORDER_WRAP
    params = _lib.SweepParams(int(l), int(w), OF_ITERS, OF_POLY_N, OF_POLY_SIGMA, int(border_mode), int(bool(chained)),
                              int(bool(use_of)))
    h = handle(device)
    free = None
    if chunk_slices is None:
        import torch                                     # only to ask for the free device memory
        free = torch.cuda.mem_get_info(device)[0] if torch.cuda.is_available() else 64 << 30
        if src.size * (_BYTES_PER_TARGET_PIXEL + 8) < 0.7 * free:      # everything fits: the resident path
            h2 = handle(device)
            d_in = h2.malloc(src.nbytes)
            try:
                d_out = h2.malloc(src.nbytes)
                try:
                    h2.h2d(d_in, np.ascontiguousarray(src))
                    h2.filter_3d_dev(d_in, d_out, src.shape, kernels, mean, params)
                    res = np.empty(src.shape, dtype=np.float32)
                    h2.d2h(res, d_out)
                    return res
                finally:
                    h2.free(d_out)
            finally:
                h2.free(d_in)
    import threading
    from concurrent.futures import ThreadPoolExecutor
    tls = threading.local()
    made = []

    def worker_handle():              # one handle (stream, workspaces) per worker thread
        if not hasattr(tls, "h"):
            tls.h = _lib.Handle(device)
            made.append(tls.h)
        return tls.h

    pool = ThreadPoolExecutor(max_workers=max(1, int(workers)))
    try:
        cur = _passes(src, kernels, chunk_slices, None if free is None else free // max(1, int(workers)), mean, wrap, params,
                      pool, worker_handle)
    finally:
        pool.shutdown(wait=True)
        for hh in made:
            hh.close()
    return cur if cur is not src else src.copy()


def _passes(src, kernels, chunk_slices, free, mean, wrap, params, pool, worker_handle):
    cur = src
    for axis in (0, 1, 2):
        k = kernels[axis]
        if k is None:
            continue
        k = np.ascontiguousarray(k, dtype=np.float64)
        r = k.size // 2
        n = cur.shape[axis]
        other = [a for a in range(3) if a != axis]
        H, W = cur.shape[other[0]], cur.shape[other[1]]   # the pass's images (seq:255, seq:333)
        HW = H * W
        out = np.empty(cur.shape, dtype=np.float32)
        if chunk_slices is None:
            step = auto_chunk(cur.shape, axis, r, free)
        else:
            step = max(1, int(chunk_slices))
        def chunk_job(s0, s1, cur=cur, out=out, axis=axis, k=k, r=r, n=n, H=H, W=W, HW=HW):
            h = worker_handle()
            S = s1 - s0
            # source slices of stack positions 0 .. S+2r-1 as runs of consecutive slices (host order)
            if wrap:
                idx = np.arange(s0 - r, s1 + r) % n
                cuts = [0] + [i + 1 for i in range(idx.size - 1) if idx[i + 1] != idx[i] + 1] + [idx.size]
                runs = [(cuts[j], int(idx[cuts[j]]), cuts[j + 1] - cuts[j]) for j in range(len(cuts) - 1)]
            else:
                lo, hi = max(0, s0 - r), min(n, s1 + r)
                runs = [(lo - (s0 - r), lo, hi - lo)]
            d_stack = h.malloc((S + 2 * r) * HW * 4)
            d_out = d_blk = None
            try:
                d_out = h.malloc(S * HW * 4)
                d_blk = h.malloc((S + 2 * r) * HW * 4) if axis else None
                if not wrap:                              # seq:88-89: slices beyond the volume ends hold the mean
                    p0, _, cnt = runs[0]
                    if p0:
                        h.memset_f32(d_stack, mean, p0 * HW)
                    if p0 + cnt < S + 2 * r:
                        h.memset_f32(d_stack + (p0 + cnt) * HW * 4, mean, (S + 2 * r - p0 - cnt) * HW)
                for p0, g0, cnt in runs:                  # upload in the volume's own layout, re-orient on the GPU
                    sl = [slice(None)] * 3
                    sl[axis] = slice(g0, g0 + cnt)
                    blk = np.ascontiguousarray(cur[tuple(sl)])
                    dst = d_stack + p0 * HW * 4
                    if axis == 0:
                        h.h2d(dst, blk)
                    else:
                        h.h2d(d_blk, blk)
                        if axis == 1:                     # (Z, cnt, X) -> (cnt, Z, X)
                            h.permute_dev(d_blk, dst, (cnt, H, W), (W, cnt * W, 1))
                        else:                             # (Z, Y, cnt) -> (cnt, Z, Y)
                            h.permute_dev(d_blk, dst, (cnt, H, W), (1, W * cnt, cnt))
                h.sweep_stack_dev(d_stack, d_out, S, H, W, k, params)
                sl = [slice(None)] * 3
                sl[axis] = slice(s0, s1)
                if axis == 0:
                    res = np.empty((S, H, W), dtype=np.float32)
                    h.d2h(res, d_out)
                elif axis == 1:                           # (S, Z, X) -> (Z, S, X)
                    h.permute_dev(d_out, d_blk, (H, S, W), (W, H * W, 1))
                    res = np.empty((H, S, W), dtype=np.float32)
                    h.d2h(res, d_blk)
                else:                                     # (S, Z, Y) -> (Z, Y, S)
                    h.permute_dev(d_out, d_blk, (H, W, S), (W, 1, H * W))
                    res = np.empty((H, W, S), dtype=np.float32)
                    h.d2h(res, d_blk)
                out[tuple(sl)] = res
            finally:
                for d in (d_blk, d_out, d_stack):
                    if d:
                        h.free(d)
        list(pool.map(lambda se: chunk_job(*se), [(s0, min(n, s0 + step)) for s0 in range(0, n, step)]))
        cur = out
    return cur


def OF_filter_streamed(vol, kernels, l, w, chunk_slices=None, **kw):
    return filter_streamed(vol, kernels, l, w, chunk_slices, use_of=True, **kw)


def no_OF_filter_streamed(vol, kernels, chunk_slices=None, **kw):
    return filter_streamed(vol, kernels, 0, 5, chunk_slices, use_of=False, **kw)
