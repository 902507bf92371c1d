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
    from .operators import integer_semantics
    src = np.asarray(vol)
    if src.ndim != 3:
        raise ValueError(f"expected a (Z, Y, X) volume, got shape {src.shape}")
    wrap = border_mode == _lib.BORDER_WRAP
    params = _lib.SweepParams(int(l), int(w), OF_ITERS, OF_POLY_N, OF_POLY_SIGMA, int(border_mode), int(bool(chained)),
                              int(bool(use_of)))
    params = integer_semantics(src, params)             # an integer MRC keeps its dtype in the reference (seq:513)
    if params.warp_mode == _lib.WARP_F64_PADDED:
        mean = np.float32(params.pad64)
    if src.dtype != np.float32:
        src = src.astype(np.float32)
    if not src.flags["C_CONTIGUOUS"]:
        src = np.ascontiguousarray(src)
    if mean is None:
        mean = _lib.mean_host(src)                      # seq:420
    h = handle(device)
    free = None
    if chunk_slices is None:
        free = h.mem_info()[0]
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

    def worker_handle():              # one handle (stream, workspaces) and one set of device buffers per worker thread
        if not hasattr(tls, "h"):
            tls.h = _lib.Handle(device)
            tls.bufs = {}
            made.append((tls.h, tls.bufs))
        h, bufs = tls.h, tls.bufs

        def dev(name, nbytes):        # grows, never shrinks within the call
            p, cap = bufs.get(name, (0, 0))
            if cap < nbytes:
                if p:
                    h.free(p)
                p = h.malloc(nbytes)
                bufs[name] = (p, nbytes)
            return p
        return h, dev
    worker_handle.device = device

    pool = ThreadPoolExecutor(max_workers=max(1, int(workers)))
    try:
        cur = _passes(src, kernels, chunk_slices, None if free is None else free // max(1, int(workers)), mean, wrap, params,
                      pool, worker_handle)
    finally:
        pool.shutdown(wait=True)
        for hh, bb in made:
            for p, _ in bb.values():
                hh.free(p)
            hh.close()
    return cur.copy() if cur is src else cur


def _passes(src, kernels, chunk_slices, free, mean, wrap, params, pool, worker_handle):
    """Z, Y, X passes over a host volume.  A chunk's slices travel as they lie in the host array -- contiguous for Z,
    strided 2-D copies for Y (Z rows of cnt * X floats) and X (Z * Y rows of cnt floats), fdn_memcpy2d_* -- between
    page-locked host arrays (the input and two ping-pong result arrays are registered once) and per-worker device
    buffers that live for the whole call; the re-orientation happens on the GPU (fdn_permute_dev)."""
    cur = src
    bufs = [None, None]                # host ping-pong results
    locked = []
    main = _lib.Handle(worker_handle.device)
    try:
        if src.flags["WRITEABLE"] and main.host_register(src):      # a read-only memory map cannot be page-locked: copied pageable
            locked.append(src)
        npass = 0
        for axis in (0, 1, 2):
            k = kernels[axis]
            if k is None:
                continue
            k = np.ascontiguousarray(k, dtype=np.float64)
            r = k.size // 2
            Z, Y, X = cur.shape
            n = cur.shape[axis]
            other = [a for a in range(3) if a != axis]
            H, W = cur.shape[other[0]], cur.shape[other[1]]   # the pass's images (seq:255, seq:333)
            HW = H * W
            if bufs[npass & 1] is None:
                bufs[npass & 1] = np.empty(cur.shape, dtype=np.float32)
                if main.host_register(bufs[npass & 1]):
                    locked.append(bufs[npass & 1])
            out = bufs[npass & 1]
            npass += 1
            if chunk_slices is None:
                step = auto_chunk(cur.shape, axis, r, free)
            else:
                step = max(1, int(chunk_slices))

            def chunk_job(s0, s1, cur=cur, out=out, axis=axis, k=k, r=r, n=n, H=H, W=W, HW=HW, Z=Z, Y=Y, X=X, step=step):
                h, dev = worker_handle()
                S = s1 - s0
                # source slices of stack positions 0 .. S+2r-1 as runs of consecutive slices (host order)
                if wrap:
                    idx = np.arange(s0 - r, s1 + r) % n
                    cuts = [0] + [i + 1 for i in range(idx.size - 1) if idx[i + 1] != idx[i] + 1] + [idx.size]
                    runs = [(cuts[j], int(idx[cuts[j]]), cuts[j + 1] - cuts[j]) for j in range(len(cuts) - 1)]
                else:
                    lo, hi = max(0, s0 - r), min(n, s1 + r)
                    runs = [(lo - (s0 - r), lo, hi - lo)]
                d_stack = dev("stack", (step + 2 * r) * HW * 4)
                d_out = dev("out", step * HW * 4)
                d_blk = dev("blk", (step + 2 * r) * HW * 4) if axis else None
                if not wrap:                              # seq:88-89: slices beyond the volume ends hold the mean
                    p0, _, cnt = runs[0]
                    if p0:
                        h.memset_f32(d_stack, mean, p0 * HW)
                    if p0 + cnt < S + 2 * r:
                        h.memset_f32(d_stack + (p0 + cnt) * HW * 4, mean, (S + 2 * r - p0 - cnt) * HW)
                base = cur.ctypes.data
                for p0, g0, cnt in runs:                  # upload in the volume's own layout, re-orient on the GPU
                    dst = d_stack + p0 * HW * 4
                    if axis == 0:
                        h.h2d_2d(dst, cnt * HW * 4, base + g0 * HW * 4, cnt * HW * 4, cnt * HW * 4, 1)
                    elif axis == 1:                       # volume[:, g0:g0+cnt, :]: Z rows of cnt * X floats -> (cnt, Z, X)
                        h.h2d_2d(d_blk, cnt * X * 4, base + g0 * X * 4, Y * X * 4, cnt * X * 4, Z)
                        h.permute_dev(d_blk, dst, (cnt, H, W), (W, cnt * W, 1))
                    else:                                 # volume[:, :, g0:g0+cnt]: Z * Y rows of cnt floats -> (cnt, Z, Y)
                        h.h2d_2d(d_blk, cnt * 4, base + g0 * 4, X * 4, cnt * 4, Z * Y)
                        h.permute_dev(d_blk, dst, (cnt, H, W), (1, W * cnt, cnt))
                pc = params
                if params.warp_mode == _lib.WARP_F64_PADDED:      # which stack slices are pad slices (float64 mean there)
                    pc = params.copy()
                    pc.pad_lo, pc.pad_hi = runs[0][0], S + 2 * r - runs[0][0] - runs[0][2]
                h.sweep_stack_dev(d_stack, d_out, S, H, W, k, pc)
                obase = out.ctypes.data
                if axis == 0:
                    h.d2h_2d(obase + s0 * HW * 4, S * HW * 4, d_out, S * HW * 4, S * HW * 4, 1)
                elif axis == 1:                           # (S, Z, X) -> (Z, S, X) -> out[:, s0:s1, :]
                    h.permute_dev(d_out, d_blk, (H, S, W), (W, H * W, 1))
                    h.d2h_2d(obase + s0 * X * 4, Y * X * 4, d_blk, S * X * 4, S * X * 4, Z)
                else:                                     # (S, Z, Y) -> (Z, Y, S) -> out[:, :, s0:s1]
                    h.permute_dev(d_out, d_blk, (H, W, S), (W, 1, H * W))
                    h.d2h_2d(obase + s0 * 4, X * 4, d_blk, S * 4, S * 4, Z * Y)
            list(pool.map(lambda se: chunk_job(*se), [(s0, min(n, s0 + step)) for s0 in range(0, n, step)]))
            cur = out
    finally:
        for a in locked:
            main.host_unregister(a)
        main.close()
    return cur


def OF_filter_streamed(vol, kernels, l, w, chunk_slices=None, **kw):
    return filter_streamed(vol, kernels, l, w, chunk_slices, use_of=True, **kw)


def no_OF_filter_streamed(vol, kernels, chunk_slices=None, **kw):
    return filter_streamed(vol, kernels, 0, 5, chunk_slices, use_of=False, **kw)
