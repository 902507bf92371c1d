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
import numpy as np

from . import _lib

OF_LEVELS = 0          # seq:44  (par:48 uses 3)
OF_WINDOW_SIZE = 5     # seq:45
OF_ITERS = 3           # seq:46
OF_POLY_N = 5          # seq:47
OF_POLY_SIGMA = 1.2    # seq:48
SIGMA = 2.0            # seq:49

_handles = {}


def handle(device=0):
    """Process-wide fdn handle for `device` (created on first use)."""
    h = _handles.get(device)
    if h is None:
        h = _handles[device] = _lib.Handle(device)
    return h


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
    return handle(device).farneback(target, reference, prev_flow, l, w, OF_ITERS, OF_POLY_N, OF_POLY_SIGMA,
                                    _lib.USE_INITIAL_FLOW)


get_flow_with_prev_flow = get_flow  # par:65


def get_flow_without_prev_flow(reference, target, l=OF_LEVELS, w=OF_WINDOW_SIZE, prev_flow=None, device=0):
    """par:89-114: flags=0, flow=None."""
    return handle(device).farneback(target, reference, None, l, w, OF_ITERS, OF_POLY_N, OF_POLY_SIGMA, 0)


def warp_slice(reference, flow, device=0):
    return handle(device).warp(reference, flow)


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


def integer_semantics(vol, params):
    """What the reference does with a volume that is not float32 (an int8/int16/uint16 MRC keeps its dtype, seq:513,
    par:472), as fdn_sweep_params fields (include/flowdn.h, FDN_WARP_*).  Returns params (a copy when changed).
      mean-padded (seq): vol.mean() is a float64, hence a float64 padded volume (seq:88-89): cv2.remap weights in
        double, the pad slices hold the float64 mean;
      wrap-around (par): neighbour slices are integer images: cv2.remap rounds and saturates, each pass is truncated
        into the integer volume (par:131, 287-289).  cv2.remap has no CV_8S path (the reference raises there) and its
        CV_8U path interpolates in fixed point, which is not restated here."""
    vol = np.asarray(vol)
    if not np.issubdtype(vol.dtype, np.integer):
        return params
    if vol.dtype.itemsize > 2:
        raise ValueError(f"{vol.dtype} volumes are not supported: float32 holds 8- and 16-bit integers exactly, not wider ones")
    p = params.copy()
    if params.border_mode == _lib.BORDER_MEAN_PAD:
        p.warp_mode = _lib.WARP_F64_PADDED
        p.pad64 = float(vol.mean())              # seq:420 on an integer array: numpy's float64 mean
    else:
        if params.use_of and vol.dtype == np.int8:
            raise ValueError("cv2.remap does not accept 8-bit signed images (the reference fails on a mode-0 MRC here)")
        if params.use_of and vol.dtype == np.uint8:
            raise NotImplementedError("cv2.remap interpolates 8-bit unsigned images in fixed point; that path is not restated")
        info = np.iinfo(vol.dtype)
        p.warp_mode = _lib.WARP_ROUND_INT
        p.round_lo, p.round_hi = float(info.min), float(info.max)
    return p


def _as_f32(vol):
    vol = np.asarray(vol)
    if vol.ndim != 3:
        raise ValueError(f"expected a (Z, Y, X) volume, got shape {vol.shape}")
    return np.ascontiguousarray(vol, dtype=np.float32)


def filter_3d_own_mean(vol, kernel, params, device=0):
    """Upload, take vol.mean() (seq:420) on the GPU -- fdn_mean_dev reproduces numpy's float32 reduction bit
    for bit, and a 2 GiB volume costs numpy 0.3 s on the host -- run the passes, download."""
    params = integer_semantics(vol, params)
    vol = _as_f32(vol)
    h = handle(device)
    d_in = h.malloc(vol.nbytes)
    try:
        d_out = h.malloc(vol.nbytes)
        try:
            out = np.empty_like(vol)
            big = vol.nbytes >= (8 << 20)
            pin_in = big and vol.flags["WRITEABLE"] and h.host_register(vol)      # DMA at PCIe speed instead of staged copies
            try:
                h.h2d(d_in, vol)
            finally:
                if pin_in:
                    h.host_unregister(vol)
            # an integer volume's mean is numpy's float64 one (params.pad64); Farneback sees it as float32
            mean = np.float32(params.pad64) if params.warp_mode == _lib.WARP_F64_PADDED else h.mean_dev(d_in, vol.size)
            h.filter_3d_dev(d_in, d_out, vol.shape, kernel, mean, params)
            pin_out = big and h.host_register(out)
            try:
                h.d2h(out, d_out)
            finally:
                if pin_out:
                    h.host_unregister(out)
            return out
        finally:
            h.free(d_out)
    finally:
        h.free(d_in)


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
