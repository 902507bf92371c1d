"""ctypes front-end of the CPU oracle (oracle/fdn_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  Function names follow the reference (src/flowdenoising_sequential.py,
"seq:N" below) so that parity tests read like calls into the reference.

Parity status: unpinned against real cv2 (see the header of fdn_oracle.c).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfdn_oracle.so")

OPTFLOW_USE_INITIAL_FLOW = 4
INTER_LINEAR = 1
INTER_AREA = 3

BOX_RUNNING = 0   # OpenCV's running sums (faithful restatement)
BOX_DIRECT = 1    # direct f64 window sums in both directions (for sensitivity studies)
BOX_VRUN_HDIRECT = 2  # OpenCV's vertical running sum + direct f64 horizontal window: the kernels' order (winsize <= 9)
BOX_VRUN_HDOUBLING = 3  # ... + the horizontal window summed by doubling (an order the one-iteration kernel used before)
BOX_VRUN_HBLOCKS = 4    # ... + the horizontal window as blocks of floor(sqrt(w)) columns: the one-iteration kernel's order (winsize >= 10)


def build(force=False):
    """Compile the oracle with oracle/Makefile (gcc only, no reference sources involved)."""
    src = os.path.join(_HERE, "fdn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libfdn_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


class _FbParams(ctypes.Structure):
    _fields_ = [("levels", ctypes.c_int), ("winsize", ctypes.c_int), ("iters", ctypes.c_int),
                ("poly_n", ctypes.c_int), ("poly_sigma", ctypes.c_double),
                ("flags", ctypes.c_int), ("box_mode", ctypes.c_int)]


class _SweepParams(ctypes.Structure):
    _fields_ = [("levels", ctypes.c_int), ("winsize", ctypes.c_int), ("border_mode", ctypes.c_int),
                ("chained", ctypes.c_int), ("use_of", ctypes.c_int), ("box_mode", ctypes.c_int),
                ("nthreads", ctypes.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.fdo_gaussian_kernel.restype = ctypes.c_int
        _lib.fdo_max_threads.restype = ctypes.c_int
    return _lib


def set_fma(mode, lanes=8):
    """How the multiply-adds of the pyramid blur and of the vertical resize pass round (fdn_oracle.c; the product's "opencv_fma"
    option): 0 = two roundings (default), 1 = fused everywhere, 2 = fused on the first (width // lanes) * lanes elements of a row
    and two roundings on its tail."""
    lib().fdo_set_fma(ctypes.c_int(int(mode)))
    lib().fdo_set_fma_lanes(ctypes.c_int(int(lanes)))


def set_remap_model(model):
    """Which cv2.remap the warps follow (the product's "remap_model" option): 0 = the classic 1/32-pixel table (default), 1 =
    unquantised float32 bilinear."""
    lib().fdo_set_remap_model(ctypes.c_int(int(model)))


def OF_filter_integer_input(vol, kernel, l, w, nthreads=1, use_of=True):
    """seq:419-424 on an INTEGER volume (an int8/int16/uint16 MRC keeps its dtype, seq:513): vol.mean() is then a float64,
    seq:88's padded volume is float64 in all three passes, cv2.remap weights in double and the pad slices hold the f64
    mean (fdn_oracle.c, fdo_set_f64_padded).  The HIP path converts such input to float32 instead; this
    restatement exists to bound that difference."""
    vol = np.asarray(vol)
    assert np.issubdtype(vol.dtype, np.integer)
    mean64 = float(vol.mean())
    lib().fdo_set_f64_padded(ctypes.c_int(1), ctypes.c_double(mean64))
    try:
        return _filter_3d(vol.astype(np.float32), kernel, l, w, use_of, 0, True, BOX_RUNNING, nthreads, mean=np.float32(mean64))
    finally:
        lib().fdo_set_f64_padded(ctypes.c_int(0), ctypes.c_double(0.0))


def filter_par_integer_input(vol, kernel, l, w, nthreads=1, use_of=True, chained=True):
    """par:285-290 on an INTEGER volume (par:472 keeps an MRC's dtype): wrap-around neighbours that are integer images --
    cv2.remap rounds half to even and saturates -- and every pass truncated into the integer volume (fdn_oracle.c,
    fdo_set_int_round).  Returns float32 holding the integer values (all three passes; the reference loses its X pass)."""
    vol = np.asarray(vol)
    assert vol.dtype in (np.int16, np.uint16, np.uint8) or (not use_of and np.issubdtype(vol.dtype, np.integer))
    info = np.iinfo(vol.dtype)
    lib().fdo_set_int_round(ctypes.c_int(2 if vol.dtype == np.uint8 else 1), ctypes.c_double(info.min), ctypes.c_double(info.max))
    try:
        return _filter_3d(vol.astype(np.float32), kernel, l, w, use_of, 1, chained, BOX_RUNNING, nthreads, mean=np.float32(0))
    finally:
        lib().fdo_set_int_round(ctypes.c_int(0), ctypes.c_double(0), ctypes.c_double(0))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def get_gaussian_kernel(sigma=1):
    """seq:30-41."""
    out = np.zeros(4096, dtype=np.float64)
    K = lib().fdo_gaussian_kernel(ctypes.c_double(float(sigma)), _p(out), ctypes.c_int(out.size))
    assert K > 0
    return out[:K].copy()


def calcOpticalFlowFarneback(prev, next, flow, pyr_scale, levels, winsize, iterations, poly_n,
                             poly_sigma, flags, box_mode=BOX_RUNNING):
    """cv2.calcOpticalFlowFarneback as called at seq:62 (pyr_scale must be 0.5).
    `flow` is updated in place when given (and returned), like cv2 does."""
    assert pyr_scale == 0.5
    prev = _f32(prev)
    next = _f32(next)
    H, W = prev.shape
    assert next.shape == (H, W)
    if flags & OPTFLOW_USE_INITIAL_FLOW:
        assert flow is not None and flow.shape == (H, W, 2) and flow.dtype == np.float32
        assert flow.flags["C_CONTIGUOUS"]
    else:
        flow = np.zeros((H, W, 2), dtype=np.float32)
    P = _FbParams(int(levels), int(winsize), int(iterations), int(poly_n), float(poly_sigma),
                  int(flags), int(box_mode))
    lib().fdo_farneback(_p(prev), _p(next), _p(flow), ctypes.c_int(H), ctypes.c_int(W), ctypes.byref(P))
    return flow


def get_flow(reference, target, l=0, w=5, prev_flow=None, box_mode=BOX_RUNNING):
    """seq:59-67: prev=target, next=reference, USE_INITIAL_FLOW, iters 3, poly 5 / 1.2."""
    return calcOpticalFlowFarneback(target, reference, prev_flow, 0.5, l, w, 3, 5, 1.2,
                                    OPTFLOW_USE_INITIAL_FLOW, box_mode)


def remap(src, map_xy):
    """cv2.remap(src, map_xy, None, INTER_LINEAR, BORDER_REPLICATE) (seq:56)."""
    src = _f32(src)
    map_xy = _f32(map_xy)
    H, W = map_xy.shape[:2]
    assert src.shape == (H, W), "oracle remap restates the same-size case the reference uses"
    dst = np.empty((H, W), dtype=np.float32)
    lib().fdo_remap_linear_replicate(_p(src), ctypes.c_int(H), ctypes.c_int(W), _p(map_xy), _p(dst))
    return dst


def remap_any(src, map_xy):
    """cv2.remap(src, map, None, INTER_LINEAR, BORDER_REPLICATE) for the image depths the reference can meet: float32;
    float64 (remapBilinear<Cast<double, double>>: weights widened, arithmetic in double); int16 / uint16 (Cast<float, T>:
    float arithmetic, then saturate_cast = round half to even, clamped); uint8 (fixed point, FixedPtCast<int, uchar, 15>).  Returns an array of src's dtype, like cv2."""
    src = np.asarray(src)
    map_xy = _f32(map_xy)
    H, W = src.shape
    if src.dtype == np.float64:
        s64 = np.ascontiguousarray(src)
        dst = np.empty((H, W), np.float64)
        lib().fdo_remap_linear_replicate_f64(_p(s64), ctypes.c_int(H), ctypes.c_int(W), _p(map_xy), _p(dst))
        return dst
    if src.dtype in (np.int16, np.uint16):
        info = np.iinfo(src.dtype)
        return np.clip(np.rint(remap(src.astype(np.float32), map_xy)), info.min, info.max).astype(src.dtype)
    if src.dtype == np.uint8:       # FixedPtCast<int, uchar, 15>: 16-bit integer weights, (sum + 2^14) >> 15
        s8 = np.ascontiguousarray(src)
        dst = np.empty((H, W), np.uint8)
        lib().fdo_remap_linear_replicate_u8(_p(s8), ctypes.c_int(H), ctypes.c_int(W), _p(map_xy), _p(dst))
        return dst
    if src.dtype != np.float32:
        raise TypeError(f"cv2.remap: unsupported depth {src.dtype} in this restatement")
    return remap(src, map_xy)


def warp_slice(reference, flow):
    """seq:51-57."""
    reference = _f32(reference)
    flow = _f32(flow)
    H, W = flow.shape[:2]
    dst = np.empty((H, W), dtype=np.float32)
    lib().fdo_warp_slice(_p(reference), _p(flow), _p(dst), ctypes.c_int(H), ctypes.c_int(W))
    return dst


def _sweep_params(l, w, border_mode, chained, use_of, box_mode, nthreads):
    return _SweepParams(int(l), int(w), int(border_mode), int(chained), int(use_of), int(box_mode),
                        int(nthreads))


def filter_along_axis(vol, axis, kernel, l, w, mean, use_of=True, border_mode=0, chained=True,
                      box_mode=BOX_RUNNING, nthreads=1):
    """OF_filter_along_Z/Y/X (seq:78-130 / 235-288 / 313-364) and the no-OF variants
    (seq:171-192 / 290-311 / 396-417); border_mode=1 gives par:306-373's wrap-around."""
    vol = _f32(vol)
    Z, Y, X = vol.shape
    kernel = np.ascontiguousarray(kernel, dtype=np.float64)
    out = np.empty_like(vol)
    sp = _sweep_params(l, w, border_mode, chained, use_of, box_mode, nthreads)
    lib().fdo_filter_axis(_p(vol), _p(out), ctypes.c_int(Z), ctypes.c_int(Y), ctypes.c_int(X),
                          ctypes.c_int(axis), _p(kernel), ctypes.c_int(kernel.size),
                          ctypes.c_float(float(mean)), ctypes.byref(sp))
    return out


def filter_axis_range(vol, axis, kernel, l, w, mean, s0, s1, use_of=True, border_mode=0, chained=True,
                      box_mode=BOX_RUNNING, nthreads=1):
    """Like filter_along_axis but only targets s0 <= s < s1 are computed (others are zero)."""
    vol = _f32(vol)
    Z, Y, X = vol.shape
    kernel = np.ascontiguousarray(kernel, dtype=np.float64)
    out = np.zeros_like(vol)
    sp = _sweep_params(l, w, border_mode, chained, use_of, box_mode, nthreads)
    lib().fdo_filter_axis_range(_p(vol), _p(out), ctypes.c_int(Z), ctypes.c_int(Y), ctypes.c_int(X),
                                ctypes.c_int(axis), _p(kernel), ctypes.c_int(kernel.size),
                                ctypes.c_float(float(mean)), ctypes.byref(sp), ctypes.c_int(s0), ctypes.c_int(s1))
    return out


def filter_axis_range_integer(vol, axis, kernel, l, w, mean64, s0, s1, semantics="seq", nthreads=1, chained=True, int_range=None):
    """filter_axis_range under the integer-volume semantics: "seq" = float64 padded volume with the float64 mean `mean64`
    (only slices outside `vol` count as pad slices), "par" = integer images in `int_range` = (lo, hi), wrap-around."""
    if semantics == "seq":
        lib().fdo_set_f64_padded(ctypes.c_int(1), ctypes.c_double(float(mean64)))
        try:
            return filter_axis_range(vol, axis, kernel, l, w, np.float32(mean64), s0, s1, nthreads=nthreads, chained=chained)
        finally:
            lib().fdo_set_f64_padded(ctypes.c_int(0), ctypes.c_double(0.0))
    lib().fdo_set_int_round(ctypes.c_int(1), ctypes.c_double(int_range[0]), ctypes.c_double(int_range[1]))
    try:
        return filter_axis_range(vol, axis, kernel, l, w, 0.0, s0, s1, border_mode=1, nthreads=nthreads, chained=chained)
    finally:
        lib().fdo_set_int_round(ctypes.c_int(0), ctypes.c_double(0), ctypes.c_double(0))


def OF_filter_along_Z(vol, kernel, l, w, mean, **kw):
    return filter_along_axis(vol, 0, kernel, l, w, mean, **kw)


def OF_filter_along_Y(vol, kernel, l, w, mean, **kw):
    return filter_along_axis(vol, 1, kernel, l, w, mean, **kw)


def OF_filter_along_X(vol, kernel, l, w, mean, **kw):
    return filter_along_axis(vol, 2, kernel, l, w, mean, **kw)


def _filter_3d(vol, kernel, l, w, use_of, border_mode, chained, box_mode, nthreads, mean=None):
    vol = _f32(vol)
    Z, Y, X = vol.shape
    if mean is None:
        mean = vol.mean()  # seq:420 -- numpy's own f32 pairwise mean, as the reference computes it
    ks = [None if k is None else np.ascontiguousarray(k, dtype=np.float64) for k in kernel]
    out = np.empty_like(vol)
    sp = _sweep_params(l, w, border_mode, chained, use_of, box_mode, nthreads)
    args = []
    for k in ks:
        if k is None:
            args += [None, ctypes.c_int(0)]
        else:
            args += [_p(k), ctypes.c_int(k.size)]
    lib().fdo_filter_3d(_p(vol), _p(out), ctypes.c_int(Z), ctypes.c_int(Y), ctypes.c_int(X),
                        *args, ctypes.c_float(float(mean)), ctypes.byref(sp))
    return out


def OF_filter(vol, kernel, l, w, border_mode=0, chained=True, box_mode=BOX_RUNNING, nthreads=1, mean=None):
    """seq:419-424.  kernel = [kz, ky, kx]; a None entry skips that axis."""
    return _filter_3d(vol, kernel, l, w, True, border_mode, chained, box_mode, nthreads, mean)


def no_OF_filter(vol, kernel, border_mode=0, nthreads=1, mean=None):
    """seq:426-431."""
    return _filter_3d(vol, kernel, 0, 5, False, border_mode, True, BOX_RUNNING, nthreads, mean)


# pieces exposed for known-answer tests -------------------------------------------------

def poly_exp(img, n=5, sigma=1.2):
    img = _f32(img)
    H, W = img.shape
    out = np.empty((H, W, 5), dtype=np.float32)
    lib().fdo_poly_exp(_p(img), _p(out), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(n),
                       ctypes.c_double(sigma))
    return out


def polyexp_consts(n=5, sigma=1.2):
    g = np.zeros(2 * n + 1, np.float32)
    xg = np.zeros_like(g)
    xxg = np.zeros_like(g)
    ig = np.zeros(4, np.float64)
    lib().fdo_polyexp_consts(ctypes.c_int(n), ctypes.c_double(sigma), _p(g), _p(xg), _p(xxg), _p(ig))
    return g, xg, xxg, ig


def gaussian_blur(img, n, sigma):
    img = _f32(img)
    H, W = img.shape
    out = np.empty_like(img)
    lib().fdo_gaussian_blur(_p(img), _p(out), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(n),
                            ctypes.c_double(sigma))
    return out


def resize(img, dh, dw, interp):
    img = _f32(img)
    cn = 1 if img.ndim == 2 else img.shape[2]
    sh, sw = img.shape[:2]
    out = np.empty((dh, dw) if img.ndim == 2 else (dh, dw, cn), dtype=np.float32)
    lib().fdo_resize(_p(img), ctypes.c_int(sh), ctypes.c_int(sw), _p(out), ctypes.c_int(dh),
                     ctypes.c_int(dw), ctypes.c_int(cn), ctypes.c_int(interp))
    return out


def update_matrices(R0, R1, flow):
    R0 = _f32(R0); R1 = _f32(R1); flow = _f32(flow)
    H, W = flow.shape[:2]
    M = np.empty((H, W, 5), dtype=np.float32)
    lib().fdo_update_matrices(_p(R0), _p(R1), _p(flow), _p(M), ctypes.c_int(H), ctypes.c_int(W))
    return M


def update_flow_blur(R0, R1, flow, M, winsize, update, box_mode=BOX_RUNNING):
    """Returns (flow, M) after one FarnebackUpdateFlow_Blur; inputs are not modified."""
    R0 = _f32(R0); R1 = _f32(R1)
    flow = _f32(flow).copy(); M = _f32(M).copy()
    H, W = flow.shape[:2]
    lib().fdo_update_flow_blur(_p(R0), _p(R1), _p(flow), _p(M), ctypes.c_int(H), ctypes.c_int(W),
                               ctypes.c_int(winsize), ctypes.c_int(int(update)), ctypes.c_int(box_mode))
    return flow, M


def max_threads():
    return lib().fdo_max_threads()
