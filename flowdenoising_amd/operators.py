"""Host-side mirror of the reference's operator interface for the hot path.

Same names, argument order and meaning as src/flowdenoising_sequential.py ("seq") and
src/flowdenoising.py ("par") in the reference tree, so that code (and tests) written
against the reference read the same here.  All arithmetic runs in libflowdn.so on the
GPU; this module only marshals numpy arrays.  There is no CPU fallback.

    get_gaussian_kernel(sigma)                      seq:30-41
    get_flow(reference, target, l, w, prev_flow)    seq:59-67 / par:65-87
    get_flow_without_prev_flow(...)                 par:89-114
    warp_slice(reference, flow)                     seq:51-57
    OF_filter_along_Z/Y/X(vol, kernel, l, w, mean)  seq:78-130 / 235-288 / 313-364
    no_OF_filter_along_Z/Y/X(vol, kernel, mean)     seq:171-192 / 290-311 / 396-417
    OF_filter(vol, kernel, l, w)                    seq:419-424
    no_OF_filter(vol, kernel)                       seq:426-431
    FlowDenoising(P, vol, l, w, ...).filter(kernels)   par:297-304, 285-290
"""
import threading

import numpy as np

from . import _lib

OF_LEVELS = 0          # seq:44  (par:48 uses 3)
OF_WINDOW_SIZE = 5     # seq:45
OF_ITERS = 3           # seq:46
OF_POLY_N = 5          # seq:47
OF_POLY_SIGMA = 1.2    # seq:48
SIGMA = 2.0            # seq:49

_handles = {}
_handles_lock = threading.Lock()


def handle(device=0):
    """Process-wide fdn handle for `device` (created on first use; the CLI's reservation thread and its main thread may
    both be the first)."""
    with _handles_lock:
        h = _handles.get(device)
        if h is None:
            # the command-line script opens the GPU context in a thread of its own while the package is being imported
            # (flowdenoising.py: a handle created and destroyed there); the first real handle waits for that thread
            for t in threading.enumerate():
                if t.name == "fdn-warm-gpu" and t is not threading.current_thread():
                    t.join(30.0)
            h = _handles[device] = _lib.Handle(device)
        return h


# The pair operators' handles.  par calls get_flow / warp_slice from its P pool threads at once (par:187-193, par:306-327).
# One handle serves them correctly -- every entry point takes the handle's lock -- but one after another, and one pair of
# images fills a twelfth of the GPU for the 1.5 ms a workgroup needs to march down 1024 rows.  So concurrent callers each take
# a handle (its own stream, workspaces and bounce buffer) from a small pool: the first is the process-wide handle (a
# single-threaded caller never sees another), more are made when a call finds every handle in use, up to FDN_PAIR_HANDLES
# (default 8); beyond that callers wait for a free one.
class _PairHandles:
    def __init__(self, device):
        import os
        import queue
        self.device = device
        self.free = queue.LifoQueue()          # last in, first out: a lone caller keeps getting the same handle
        self.made = 0
        self.limit = max(1, int(os.environ.get("FDN_PAIR_HANDLES", "8")))
        self.lock = threading.Lock()
        self.extra = []

    def take(self):
        h = self._take()
        main = _handles.get(self.device)
        if main is not None and h is not main and h.options != main.options:
            # what the caller has set on the process-wide handle (strict_order, opencv_fma, remap_model ...) holds for the
            # pair operators whichever handle serves them
            try:
                for k, v in main.options.items():
                    if h.options.get(k) != v:
                        h.set_option(k, v)
            except BaseException:
                self.free.put(h)
                raise
        return h

    def _take(self):
        import queue
        try:
            return self.free.get_nowait()
        except queue.Empty:
            pass
        with self.lock:
            make = self.made < self.limit
            first = self.made == 0
            if make:
                self.made += 1
        if not make:
            return self.free.get()
        try:
            if first:
                return handle(self.device)
            h = _lib.Handle(self.device)
            with self.lock:
                self.extra.append(h)
            return h
        except BaseException:
            with self.lock:
                self.made -= 1
            raise

    def give_back(self, h):
        self.free.put(h)


_pair_pools = {}


def _pair_pool(device):
    with _handles_lock:
        pool = _pair_pools.get(device)
        if pool is None:
            pool = _pair_pools[device] = _PairHandles(device)
        return pool


def release_pair_handles():
    """Close the extra handles the pair operators made for concurrent callers (the process-wide handle stays)."""
    with _handles_lock:
        pools = list(_pair_pools.values())
        _pair_pools.clear()
    for pool in pools:
        for h in pool.extra:
            h.close()


def _params(l, w, use_of=True, border_mode=_lib.BORDER_MEAN_PAD, chained=True):
    return _lib.SweepParams(int(l), int(w), OF_ITERS, OF_POLY_N, OF_POLY_SIGMA, int(border_mode),
                            int(bool(chained)), int(bool(use_of)))


def get_gaussian_kernel(sigma=1):
    return _lib.gaussian_kernel(sigma)


def get_flow(reference, target, l=OF_LEVELS, w=OF_WINDOW_SIZE, prev_flow=None, device=0):
    """cv2.calcOpticalFlowFarneback(prev=target, next=reference, flow=prev_flow, 0.5, l, w, 3, 5, 1.2,
    OPTFLOW_USE_INITIAL_FLOW): `prev_flow` is the initial guess and is overwritten in place."""
    if prev_flow is None:
        raise ValueError("OPTFLOW_USE_INITIAL_FLOW needs prev_flow (cv2 asserts the same)")
    pool = _pair_pool(device)
    h = pool.take()
    try:
        return h.farneback(target, reference, prev_flow, l, w, OF_ITERS, OF_POLY_N, OF_POLY_SIGMA, _lib.USE_INITIAL_FLOW)
    finally:
        pool.give_back(h)


get_flow_with_prev_flow = get_flow  # par:65


def get_flow_without_prev_flow(reference, target, l=OF_LEVELS, w=OF_WINDOW_SIZE, prev_flow=None, device=0):
    """par:89-114: flags=0, flow=None."""
    pool = _pair_pool(device)
    h = pool.take()
    try:
        return h.farneback(target, reference, None, l, w, OF_ITERS, OF_POLY_N, OF_POLY_SIGMA, 0)
    finally:
        pool.give_back(h)


def warp_slice(reference, flow, device=0):
    pool = _pair_pool(device)
    h = pool.take()
    try:
        return h.warp(reference, flow)
    finally:
        pool.give_back(h)


def _filter_axis(vol, axis, kernel, l, w, mean, use_of, border_mode, chained, device):
    p = _params(l, w, use_of, border_mode, chained)
    if border_mode == _lib.BORDER_MEAN_PAD and np.asarray(mean).dtype not in (np.float32, np.float16):
        # seq:88: np.full(..., fill_value=mean) takes the DTYPE OF `mean`: anything but a numpy float32 (vol.mean() of an
        # integer volume, a Python float) makes the padded volume float64 -- for a float32 `vol` too
        p.warp_mode = _lib.WARP_F64_PADDED
        p.pad64 = float(mean)
    elif border_mode != _lib.BORDER_MEAN_PAD:
        p = integer_semantics(vol, p)
    return handle(device).filter_axis(vol, axis, kernel, mean, p)


def OF_filter_along_Z(vol, kernel, l, w, mean, border_mode=_lib.BORDER_MEAN_PAD, chained=True, device=0):
    return _filter_axis(vol, 0, kernel, l, w, mean, True, border_mode, chained, device)


def OF_filter_along_Y(vol, kernel, l, w, mean, border_mode=_lib.BORDER_MEAN_PAD, chained=True, device=0):
    return _filter_axis(vol, 1, kernel, l, w, mean, True, border_mode, chained, device)


def OF_filter_along_X(vol, kernel, l, w, mean, border_mode=_lib.BORDER_MEAN_PAD, chained=True, device=0):
    return _filter_axis(vol, 2, kernel, l, w, mean, True, border_mode, chained, device)


def no_OF_filter_along_Z(vol, kernel, mean, device=0):
    return _filter_axis(vol, 0, kernel, 0, OF_WINDOW_SIZE, mean, False, _lib.BORDER_MEAN_PAD, True, device)


def no_OF_filter_along_Y(vol, kernel, mean, device=0):
    return _filter_axis(vol, 1, kernel, 0, OF_WINDOW_SIZE, mean, False, _lib.BORDER_MEAN_PAD, True, device)


def no_OF_filter_along_X(vol, kernel, mean, device=0):
    return _filter_axis(vol, 2, kernel, 0, OF_WINDOW_SIZE, mean, False, _lib.BORDER_MEAN_PAD, True, device)


def integer_semantics(vol, params, mean_on_device=False):
    """What the reference does with a volume that is not float32 (an int8/int16/uint16 MRC keeps its dtype, seq:513,
    par:472), as fdn_sweep_params fields (include/flowdn.h, FDN_WARP_*).  Returns params (a copy when changed).
      mean-padded (seq): vol.mean() is a float64, hence a float64 padded volume (seq:88-89): cv2.remap weights in
        double, the pad slices hold the float64 mean;
      wrap-around (par): neighbour slices are integer images: cv2.remap rounds and saturates, each pass is truncated
        into the integer volume (par:131, 287-289).  cv2.remap has no CV_8S path (the reference raises there); its
        CV_8U path interpolates in 8-bit fixed point (FDN_WARP_FIXED_U8)."""
    vol = np.asarray(vol)
    if not np.issubdtype(vol.dtype, np.integer):
        return params
    if vol.dtype.itemsize > 2:
        raise ValueError(f"{vol.dtype} volumes are not supported: float32 holds 8- and 16-bit integers exactly, not wider ones")
    p = params.copy()
    if params.border_mode == _lib.BORDER_MEAN_PAD:
        p.warp_mode = _lib.WARP_F64_PADDED
        # seq:420 on an integer array: numpy's float64 mean = (exact integer sum) / count.  mean_on_device: the caller takes
        # it from the device copy instead (fdn_stats_dev: the float64 sum of 8- or 16-bit integers is exact as well; half a
        # second of numpy on a 1 GiB int16 tomogram) and fills pad64 in before the first launch
        p.pad64 = float("nan") if mean_on_device else float(vol.mean())
    else:
        if params.use_of and vol.dtype == np.int8:
            raise ValueError("cv2.remap does not accept 8-bit signed images (the reference fails on a mode-0 MRC here)")
        info = np.iinfo(vol.dtype)
        # uint8 images are interpolated in 8-bit fixed point by cv2.remap (FDN_WARP_FIXED_U8); 16-bit ones in float, then rounded
        p.warp_mode = _lib.WARP_FIXED_U8 if vol.dtype == np.uint8 else _lib.WARP_ROUND_INT
        p.round_lo, p.round_hi = float(info.min), float(info.max)
    return p


def _as_f32(vol):
    vol = np.asarray(vol)
    if vol.ndim != 3:
        raise ValueError(f"expected a (Z, Y, X) volume, got shape {vol.shape}")
    return np.ascontiguousarray(vol, dtype=np.float32)


def filter_3d_own_mean(vol, kernel, params, device=0, stats=None, float32_semantics=False, tiff_downcast=False, timing=None,
                       sink=None, wait_for=None, mapped_out=None):
    """Upload, take vol.mean() (seq:420) on the GPU -- fdn_mean_dev reproduces numpy's float32 reduction bit
    for bit, and a 2 GiB volume costs numpy 0.3 s on the host -- run the passes, download.
    An 8- or 16-bit integer volume travels as it is and becomes float32 on the device (fdn_convert_dev: exact).
    stats: a dict that receives {"in": {...}, "out": {...}} (min, max, mean, std of the input and output volumes,
    fdn_stats_dev) -- what the reference's CLI logs and mrcfile puts into the output header, taken on the GPU.
    float32_semantics: the integer array stands for `vol.astype(np.float32)` (a TIFF stack, seq:517): plain float32
    arithmetic, but the conversion happens on the device.
    tiff_downcast: return what seq:566-571 writes into a TIFF -- `filtered.astype(np.uint8)` if its maximum is below
    256, else `np.uint16` -- cast on the device (fdn_truncate_dev), so that a quarter or half of the bytes come back.
    timing: a dict that receives wall seconds of the phases: "h2d", "compute", "d2h" (tools/cli_wall.py).
    sink: `sink(dtype, stats_out)` returns an object with write_slab(array) / close() (io.VolumeWriter): the result is
    then downloaded in slabs of Z slices and every slab is handed to it, in order, from a second thread -- the file
    write of one slab overlaps the download of the next (the reference writes after its last pass, seq:558-571).
    mapped_out: a callable returning an io.MappedMrcWriter (the float32 MRC output file mapped into memory): it is made
    after the upload (the input may be the same file) and prepared -- pages faulted in and, where allowed, page-locked --
    in a thread of its own while the passes run; the result is then copied from the GPU straight into the file's pages.
    wait_for: called after the upload and before anything is launched on the process-wide handle -- the CLI passes the
    join of the thread in which that handle reserves its buffers (fdn_reserve_3d); the upload itself then runs on a
    handle of its own, concurrently with it."""
    import time
    tick = [time.perf_counter()]

    def lap(name):
        if timing is not None:
            h.synchronize()
            now = time.perf_counter()
            timing[name] = timing.get(name, 0.0) + now - tick[0]
            tick[0] = now

    if not float32_semantics:
        params = integer_semantics(vol, params, mean_on_device=True)
    vol = np.asarray(vol)
    if vol.ndim != 3:
        raise ValueError(f"expected a (Z, Y, X) volume, got shape {vol.shape}")
    raw_int = vol.dtype.kind in "iu" and vol.dtype.itemsize <= 2 and vol.dtype.isnative
    src = np.ascontiguousarray(vol) if raw_int else _as_f32(vol)
    h = handle(device)
    hu = _lib.Handle(device) if wait_for is not None else h       # the upload's own handle and stream while `h` is being prepared
    nbytes = src.size * 4
    d_in = hu.malloc(nbytes)
    try:
        d_out = hu.malloc(nbytes)
        try:
            out = np.empty(src.shape, dtype=np.float32)
            big = src.nbytes >= (8 << 20)
            # DMA at PCIe speed instead of staged copies.  A read-only memory map of the input file (the CLI's MRC path) is
            # page-locked where the runtime allows it: the upload then reads the page cache directly -- no host copy at all
            try:
                pin_in = big and hu.host_register(src)
                if timing is not None:
                    timing["in_pinned"] = bool(pin_in)
                try:
                    hu.h2d(d_out if raw_int else d_in, src)      # raw integers into the (larger) output buffer first
                finally:
                    if pin_in:
                        hu.host_unregister(src)
                if hu is not h:
                    hu.synchronize()
            finally:
                if hu is not h:              # the upload's own handle goes away on error paths too; d_in / d_out outlive it
                    hu.close()
            if wait_for is not None:
                wait_for()
            if raw_int:
                h.convert_dev(d_out, src.dtype, d_in, src.size)      # float32 from there into d_in
            lap("h2d")
            writer = prep = hw = None
            # (statistics from per-slice reductions added in slice order, fdn_stats_slices_dev: the form a multi-GPU run
            # reproduces bit for bit from its slabs, so both write the same header)
            st_in = h.stats_volume(d_in, src.shape) if (stats is not None or params.pad64 != params.pad64) else None
            if stats is not None:
                stats["in"] = st_in
            if params.pad64 != params.pad64:      # an integer volume's float64 mean (seq:420), exact from the device copy
                params.pad64 = st_in["mean"]
            # an integer volume's mean is numpy's float64 one (params.pad64); Farneback sees it as float32
            mean = np.float32(params.pad64) if params.warp_mode == _lib.WARP_F64_PADDED else h.mean_dev(d_in, src.size)
            h.filter_3d_dev(d_in, d_out, src.shape, kernel, mean, params)
            try:
                if mapped_out is not None and not tiff_downcast:
                    # The passes are ENQUEUED now (nothing above waits for them) and run for a while on their own: the output
                    # file is mapped and its pages faulted in and page-locked meanwhile.  (Not earlier: hipHostRegister holds a
                    # runtime lock that kernel launches wait for -- started before the launches it added its 0.3 s to the passes.)
                    import threading
                    try:
                        writer = mapped_out()
                        hw = _lib.Handle(device)          # page-locking goes through a handle of its own
                        ready = {}
                        prep = threading.Thread(target=lambda: ready.setdefault("pinned", writer.prepare(hw)), daemon=True)
                        prep.start()
                    except (OSError, _lib.FlowdnError):   # no mapping here, or no second handle: the slab writer below
                        if writer is not None:
                            writer.close()
                        if hw is not None:
                            hw.close()
                        writer = prep = hw = None
                if prep is not None:
                    prep.join()                           # (before the statistics' launches: they would queue behind the page-locking)
                if stats is not None or tiff_downcast or writer is not None:
                    st_out = h.stats_volume(d_out, src.shape)
                    if stats is not None:
                        stats["out"] = st_out
                d_res = d_out
                if tiff_downcast:
                    out = np.empty(src.shape, dtype=np.uint8 if st_out["max"] < 256 else np.uint16)
                    h.truncate_dev(d_out, out.dtype, d_in, src.size)      # the input's device copy is no longer needed
                    d_res = d_in
                lap("compute")
                if writer is not None:
                    if timing is not None:
                        timing["out_pinned"] = bool(ready.get("pinned"))
                    if ready.get("pinned"):
                        h.d2h(writer.data, d_res)             # one DMA into the file's own pages
                    else:                                     # pages are there but cannot be locked: slabs through the staging path
                        per = writer.data[0].nbytes
                        step = max(1, (128 << 20) // max(per, 1))
                        for z0 in range(0, src.shape[0], step):
                            h.d2h(writer.data[z0:z0 + step], d_res + z0 * per)
                    writer.finish(st_out)                     # renames the finished file over the output path
                    if stats is not None:
                        stats["streamed"] = True
                    lap("d2h")
                    return None
            finally:
                # whatever happened between the writer's creation and here -- a failed pass, a failed reduction, an interrupt --
                # the page-locking thread has ended, the mapping is closed (an unfinished temporary file removed: the output
                # path is untouched, as with the reference, which writes after its last pass) and the extra handle is gone
                if prep is not None and prep.is_alive():
                    prep.join()
                if writer is not None:
                    writer.close()
                if hw is not None:
                    hw.close()
            pin_out = big and h.host_register(out)
            try:
                if sink is None:
                    h.d2h(out, d_res)
                else:
                    _download_into(h, out, d_res, sink(out.dtype, st_out if (stats is not None or tiff_downcast) else None))
            finally:
                if pin_out:
                    h.host_unregister(out)
            lap("d2h")
            return out
        finally:
            h.free(d_out)
    finally:
        h.free(d_in)


def _download_into(h, out, d_res, writer, slab_bytes=128 << 20):
    """Device -> `out` in slabs of whole Z slices; each slab goes to writer.write_slab from a second thread as soon as it
    has arrived, so that writing the file overlaps the rest of the download.  Closes the writer."""
    import queue
    import threading
    Z = out.shape[0]
    per = out[0].nbytes
    step = max(1, int(slab_bytes // max(per, 1)))
    q = queue.Queue()
    err = []

    def drain():
        while True:
            item = q.get()
            if item is None:
                return
            try:
                if not err:
                    writer.write_slab(out[item[0]:item[1]])
            except Exception as e:      # keep draining so that the producer never blocks; re-raised below
                err.append(e)

    t = threading.Thread(target=drain, daemon=True)
    t.start()
    try:
        for z0 in range(0, Z, step):
            z1 = min(Z, z0 + step)
            h.d2h(out[z0:z1], d_res + z0 * per)
            q.put((z0, z1))
    finally:
        q.put(None)
        t.join()
        writer.close()
    if err:
        raise err[0]


def OF_filter(vol, kernel, l, w, border_mode=_lib.BORDER_MEAN_PAD, chained=True, device=0):
    """seq:419-424: mean = vol.mean(); Z, then Y, then X.  A None entry in `kernel` skips that axis."""
    return filter_3d_own_mean(vol, kernel, _params(l, w, True, border_mode, chained), device)


def no_OF_filter(vol, kernel, device=0):
    """seq:426-431."""
    return filter_3d_own_mean(vol, kernel, _params(0, OF_WINDOW_SIZE, False), device)


class GaussianDenoising:
    """par:116-295 (-n/--no_OF): wrap-around borders; `filter` leaves the result in
    `self.filtered_vol`.  The reference's thread pool (`number_of_processes`) has no GPU
    counterpart: every target slice of a pass is batched into the same kernel launches."""

    use_of = False

    def __init__(self, number_of_processes, vol):
        self.number_of_processes = number_of_processes
        self.vol = vol
        self.filtered_vol = np.zeros_like(vol)
        self.l, self.w = 0, OF_WINDOW_SIZE
        self.chained = True
        self.device = 0

    def filter(self, kernels):
        p = integer_semantics(self.vol, _params(self.l, self.w, self.use_of, _lib.BORDER_WRAP, self.chained))
        out = handle(self.device).filter_3d(_as_f32(self.vol), kernels, 0.0, p)
        # par:131,287: results are stored in arrays of the INPUT dtype
        self.filtered_vol[...] = out
        self.vol[...] = self.filtered_vol
        return None


class FlowDenoising(GaussianDenoising):
    """par:297-373.  `get_flow` selects chained (get_flow_with_prev_flow) or recomputed
    (get_flow_without_prev_flow) flow exactly like par:442-447; `warp_slice` is accepted for
    signature compatibility.  Deviation (documented in DESIGN.md): the reference discards its
    X pass (par:290 + par:520); here `vol` and `filtered_vol` both hold the full Z->Y->X result,
    as src/flowdenoising_GPU.py:460 returns it."""

    use_of = True

    def __init__(self, number_of_processes, vol, l, w, get_flow=None, warp_slice=None):
        super().__init__(number_of_processes, vol)
        self.l, self.w = l, w
        self.chained = get_flow is not get_flow_without_prev_flow
