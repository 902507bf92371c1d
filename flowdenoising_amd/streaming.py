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


def _populate(a):
    """Fault the pages of a fresh array in -- by the kernel itself (MADV_POPULATE_WRITE, Linux 5.14), else by touching one
    element per page.  hipHostRegister of an UNTOUCHED 2 GiB array takes the page faults inside the runtime, 0.09 s during
    which it holds a lock every kernel launch and copy of the process waits for; of a populated one, 4 ms."""
    import ctypes
    addr, n = a.ctypes.data, a.nbytes
    lo = (addr + 4095) & ~4095
    hi = (addr + n) & ~4095
    try:
        libc = ctypes.CDLL(None, use_errno=True)
        if hi > lo and libc.madvise(ctypes.c_void_p(lo), ctypes.c_size_t(hi - lo), ctypes.c_int(23)) == 0:
            return
    except (OSError, AttributeError):
        pass
    flat = a.reshape(-1)
    flat[::1024] = 0


def filter_streamed(vol, kernels, l, w, chunk_slices=None, use_of=True, border_mode=_lib.BORDER_MEAN_PAD,
                    chained=True, device=0, mean=None, workers=3):
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
    import os
    import queue
    import threading
    import time
    from concurrent.futures import ThreadPoolExecutor
    nworkers = max(1, int(workers))
    timing = {} if os.environ.get("FDN_STREAM_TIMING") else None      # phase record (tools/stream_profile.py)
    t_start = time.perf_counter()
    # One handle (stream, workspaces) and one set of device buffers per worker, made HERE, before the first pass -- handles are
    # shared-safe (every entry point takes the handle's lock), a worker thread just takes one from the queue for a chunk.
    # Every buffer a chunk will need -- its stack, its output, the strided block, and inside the library the polynomial
    # expansions and flows of the largest pass -- is allocated now (fdn_reserve_stack): no chunk waits for gigabytes of
    # hipMalloc (16 ms per GB), and no allocation takes the address-space lock while result arrays are being page-locked.
    per_worker_free = None if free is None else free // nworkers
    plan = _plan(src.shape, kernels, chunk_slices, per_worker_free)
    free_handles = queue.SimpleQueue()
    with _workers_lock:
        made = _take_workers(device, nworkers)       # this call's workers: kept from an earlier call, or new
    try:
        for hw, bufs in made:
            for name in ("stack", "out", "blk"):
                nbytes = max((pp[name] for pp in plan.values()), default=0)
                p0, cap = bufs.get(name, (0, 0))
                if nbytes > cap:
                    if p0:
                        hw.free(p0)
                    bufs[name] = (hw.malloc(nbytes), nbytes)
            for axis, pp in plan.items():
                hw.reserve_stack(pp["step"], pp["H"], pp["W"], pp["K"], params)
            free_handles.put((hw, bufs))
        if timing is not None:
            timing["setup_workers_s"] = time.perf_counter() - t_start

        def worker_handle():
            hw, bufs = free_handles.get()

            def dev(name, nbytes):        # (reserved above; grows only if a caller's chunking differs from the plan)
                p, cap = bufs.get(name, (0, 0))
                if cap < nbytes:
                    if p:
                        hw.free(p)
                    p = hw.malloc(nbytes)
                    bufs[name] = (p, nbytes)
                return p
            return hw, dev, lambda: free_handles.put((hw, bufs))
        worker_handle.device = device

        pool = ThreadPoolExecutor(max_workers=nworkers)
        try:
            t1 = time.perf_counter()
            cur = _passes(src, kernels, plan, mean, wrap, params, pool, worker_handle, timing)
            if timing is not None:
                timing["passes_call_s"] = time.perf_counter() - t1
        finally:
            t1 = time.perf_counter()
            pool.shutdown(wait=True)
            if timing is not None:
                timing["pool_shutdown_s"] = time.perf_counter() - t1
    finally:
        t0 = time.perf_counter()
        with _workers_lock:                          # the workers, their streams and their device buffers stay for the next call
            _kept.setdefault(device, []).extend(made)
        if timing is not None:
            timing["teardown_s"] = time.perf_counter() - t0
            timing["total_s"] = time.perf_counter() - t_start
            print("stream timing:", {k: round(v, 3) for k, v in timing.items()}, flush=True)
    return cur.copy() if cur is src else cur


# Worker handles of the out-of-core mode, kept between calls like the process-wide handle of operators.handle() keeps its
# buffers: a worker's device buffers are several GB (stack, output, expansions, flows), and allocating them costs 16-60 ms per
# GB -- a third of a second per call on configs[2].  release_workers() gives everything back.
_kept = {}
_kept_host = {}        # (device, shape) -> the page-locked intermediate result array of the last call
_workers_lock = __import__("threading").Lock()


def _take_workers(device, n):
    have = _kept.get(device, [])
    mine, _kept[device] = have[:n], have[n:]
    while len(mine) < n:
        hw = _lib.Handle(device)
        hw.set_option("sub_batches", 1)      # the workers' chunks already overlap on streams of their own: no sub-batches on top
        mine.append((hw, {}))
    return mine


def release_workers():
    """Close the out-of-core mode's kept worker handles and free their device buffers."""
    with _workers_lock:
        kept = [w for ws in _kept.values() for w in ws]
        _kept.clear()
        hosts = list(_kept_host.values())
        _kept_host.clear()
    for a in hosts:
        _lib.load().fdn_host_unregister(None, __import__("ctypes").c_void_p(a.ctypes.data))
    for hw, bufs in kept:
        for p, _ in bufs.values():
            hw.free(p)
        hw.close()


def _plan(shape, kernels, chunk_slices, free):
    """Per pass (axis -> dict): target slices per chunk, image size, taps, and the bytes of a worker's three device buffers."""
    plan = {}
    for axis in (0, 1, 2):
        k = kernels[axis]
        if k is None:
            continue
        K = int(np.asarray(k).size)
        r = K // 2
        other = [a for a in range(3) if a != axis]
        H, W = shape[other[0]], shape[other[1]]
        n = shape[axis]
        step = min(n, auto_chunk(shape, axis, r, free) if chunk_slices is None else max(1, int(chunk_slices)))
        HW = H * W
        plan[axis] = {"step": step, "H": H, "W": W, "K": K, "stack": (step + 2 * r) * HW * 4, "out": step * HW * 4,
                      "blk": (step + 2 * r) * HW * 4 if axis else 0}
    return plan


def _passes(src, kernels, plan, mean, wrap, params, pool, worker_handle, timing=None):
    """Z, Y, X passes over a host volume.  A chunk's slices travel as they lie in the host array -- contiguous for Z,
    strided 2-D copies for Y (Z rows of cnt * X floats) and X (Z * Y rows of cnt floats), fdn_memcpy2d_* -- between
    page-locked host arrays (the input and two ping-pong result arrays are registered once) and per-worker device
    buffers that live for the whole call; the re-orientation happens on the GPU (fdn_permute_dev)."""
    import threading
    import time
    cur = src
    locked = []
    t_enter = time.perf_counter()
    main = _lib.Handle(worker_handle.device)
    if timing is not None:
        timing["main_handle_s"] = time.perf_counter() - t_enter
    npasses = sum(1 for a in (0, 1, 2) if kernels[a] is not None)
    # The two ping-pong result arrays are made and page-locked by a helper thread WHILE the first pass runs: a fresh 2 GiB
    # array costs 0.09 s of first-touch page faults inside hipHostRegister.  The first pass's chunks upload and compute
    # meanwhile and wait for `ready[0]` only before their download; nothing else allocates then (the workers' buffers were
    # reserved before), so nobody queues for the address-space lock -- what undid this overlap in round 5.
    results = [None, None]
    ready = [threading.Event(), threading.Event()]
    was_locked = [False, False]

    failed = []
    abort = threading.Event()

    final_slot = (npasses - 1) & 1     # the array the last pass writes is the caller's result: always a fresh one
    kept_key = (worker_handle.device, tuple(src.shape))
    from_cache = [False, False]

    def make_results():
        try:
            for i in range(min(2, npasses)):
                a = None
                if i != final_slot:        # the intermediate array: kept between calls, page-locked (release_workers() frees it)
                    with _workers_lock:
                        a = _kept_host.pop(kept_key, None)
                if a is not None:
                    from_cache[i] = True
                    was_locked[i] = True
                else:
                    a = np.empty(src.shape, dtype=np.float32)
                    _populate(a)
                    was_locked[i] = main.host_register(a)
                results[i] = a
                ready[i].set()
        except BaseException as e:      # noqa: BLE001 -- handed to the pass that waits for the array
            failed.append(e)
            for ev in ready:
                ev.set()
    helper = threading.Thread(target=make_results, daemon=True)
    try:
        t0 = time.perf_counter()
        if src.flags["WRITEABLE"] and main.host_register(src):      # a read-only memory map cannot be page-locked: copied pageable
            locked.append(src)
        if timing is not None:
            timing["lock_input_s"] = time.perf_counter() - t0
            timing["input_locked"] = float(len(locked))
        helper.start()
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
            t0 = time.perf_counter()
            slot = npass & 1
            npass += 1
            step = plan[axis]["step"]

            def out_array(slot=slot):          # the pass's result array, once the helper has made it
                ready[slot].wait()
                if failed:
                    raise failed[0]
                return results[slot]

            def chunk_job(s0, s1, cur=cur, out_array=out_array, axis=axis, k=k, r=r, n=n, H=H, W=W, HW=HW, Z=Z, Y=Y, X=X, step=step):
                if abort.is_set():                        # a chunk has failed: the ones still queued do nothing
                    return
                h, dev, give_back = worker_handle()
                try:
                    _chunk(h, dev, s0, s1, cur, out_array, axis, k, r, n, H, W, HW, Z, Y, X, step)
                except BaseException:
                    abort.set()
                    raise
                finally:
                    give_back()

            def _chunk(h, dev, s0, s1, cur, out_array, axis, k, r, n, H, W, HW, Z, Y, X, step):
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
                obase = out_array().ctypes.data
                if axis == 0:
                    h.d2h_2d(obase + s0 * HW * 4, S * HW * 4, d_out, S * HW * 4, S * HW * 4, 1)
                elif axis == 1:                           # (S, Z, X) -> (Z, S, X) -> out[:, s0:s1, :]
                    h.permute_dev(d_out, d_blk, (H, S, W), (W, H * W, 1))
                    h.d2h_2d(obase + s0 * X * 4, Y * X * 4, d_blk, S * X * 4, S * X * 4, Z)
                else:                                     # (S, Z, Y) -> (Z, Y, S) -> out[:, :, s0:s1]
                    h.permute_dev(d_out, d_blk, (H, W, S), (W, 1, H * W))
                    h.d2h_2d(obase + s0 * 4, X * 4, d_blk, S * 4, S * 4, Z * Y)
            futs = [pool.submit(chunk_job, s0, min(n, s0 + step)) for s0 in range(0, n, step)]
            try:
                for f in futs:
                    f.result()
            except BaseException:
                # a chunk failed: the ones in flight still copy to and from the page-locked arrays -- they end before those
                # arrays are released below; the ones that have not started never do (`abort`)
                abort.set()
                for f in futs:
                    f.cancel()
                from concurrent.futures import wait
                wait(futs)
                raise
            cur = out_array()
            if timing is not None:
                timing[f"pass_{axis}_s"] = time.perf_counter() - t0
    finally:
        t0 = time.perf_counter()
        if helper.is_alive() or helper.ident is not None:
            helper.join()
        if timing is not None:
            timing["helper_join_s"] = time.perf_counter() - t0
        # The ping-pong array that is not the result stays for the next call of the same shape, page-locked as it is
        # (unmapping 2 GiB of touched pages costs 0.07 s, under the interpreter lock whichever thread does it, and making
        # it again 0.09 s); a result array of its own is what every call returns.
        keep = None
        other = 1 - final_slot
        if npasses > 1 and results[other] is not None and results[other] is not cur and was_locked[other]:
            keep = results[other]
        for i in range(2):
            if was_locked[i] and results[i] is not keep:
                locked.append(results[i])
        for a in locked:
            main.host_unregister(a)
        main.close()
        if keep is not None:
            with _workers_lock:                        # one intermediate array at a time (the latest shape)
                dropped = [v for v in _kept_host.values() if v is not keep]
                _kept_host.clear()
                _kept_host[kept_key] = keep
            for old in dropped:                        # released BEFORE its memory goes back (a registration must not outlive it)
                _lib.load().fdn_host_unregister(None, __import__("ctypes").c_void_p(old.ctypes.data))
            del dropped
        results[:] = [None, None]
        if timing is not None:
            timing["unlock_s"] = time.perf_counter() - t0
    return cur


def OF_filter_streamed(vol, kernels, l, w, chunk_slices=None, **kw):
    return filter_streamed(vol, kernels, l, w, chunk_slices, use_of=True, **kw)


def no_OF_filter_streamed(vol, kernels, chunk_slices=None, **kw):
    return filter_streamed(vol, kernels, 0, 5, chunk_slices, use_of=False, **kw)
