"""ctypes binding of libflowdn.so (include/flowdn.h).

There is no CPU fallback: if the HIP library is missing or a GPU call fails, this raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libflowdn.so")

USE_INITIAL_FLOW = 4
BORDER_MEAN_PAD = 0
BORDER_WRAP = 1
TIMER_NAMES = ("polyexp", "update_matrices", "update_flow", "warp", "permute", "transfer", "fused", "iter", "collective", "mean", "chains")


class FlowdnError(RuntimeError):
    pass


class SweepParams(ctypes.Structure):
    """struct fdn_sweep_params"""
    _fields_ = [("levels", ctypes.c_int), ("winsize", ctypes.c_int), ("iters", ctypes.c_int),
                ("poly_n", ctypes.c_int), ("poly_sigma", ctypes.c_double),
                ("border_mode", ctypes.c_int), ("chained", ctypes.c_int), ("use_of", ctypes.c_int),
                # integer-volume semantics (include/flowdn.h); all zero = float32
                ("warp_mode", ctypes.c_int), ("pad_lo", ctypes.c_int), ("pad_hi", ctypes.c_int),
                ("pad64", ctypes.c_double), ("round_lo", ctypes.c_double), ("round_hi", ctypes.c_double)]

    def copy(self):
        c = SweepParams()
        ctypes.memmove(ctypes.byref(c), ctypes.byref(self), ctypes.sizeof(SweepParams))
        return c


WARP_F32, WARP_F64_PADDED, WARP_ROUND_INT, WARP_FIXED_U8 = 0, 1, 2, 3
DEPTH_F32, DEPTH_F64, DEPTH_I16, DEPTH_U16, DEPTH_I8, DEPTH_U8 = range(6)
DEPTHS = {np.dtype(np.float32): DEPTH_F32, np.dtype(np.float64): DEPTH_F64, np.dtype(np.int16): DEPTH_I16,
          np.dtype(np.uint16): DEPTH_U16, np.dtype(np.int8): DEPTH_I8, np.dtype(np.uint8): DEPTH_U8}


# every symbol include/flowdn.h declares (tests check the .so exports all of them)
EXPORTS = [
    "fdn_create", "fdn_device_count", "fdn_device_pci_id", "fdn_destroy", "fdn_last_error", "fdn_set_stream", "fdn_reset_stream", "fdn_synchronize",
    "fdn_set_workspace_limit", "fdn_workspace_bytes", "fdn_mem_info", "fdn_set_option", "fdn_get_option", "fdn_malloc", "fdn_free", "fdn_memcpy_h2d", "fdn_memcpy_d2h",
    "fdn_memcpy2d_h2d", "fdn_memcpy2d_d2h", "fdn_host_register", "fdn_host_unregister",
    "fdn_memset_f32", "fdn_gaussian_kernel", "fdn_farneback", "fdn_farneback_strided", "fdn_farneback_dev",
    "fdn_warp", "fdn_warp_strided", "fdn_warp_dev", "fdn_farneback_typed", "fdn_warp_typed",
    "fdn_filter_axis_dev", "fdn_filter_axis", "fdn_filter_3d_dev", "fdn_filter_3d",
    "fdn_mean_host", "fdn_mean_dev", "fdn_np_chunk_sums_dev", "fdn_sum_dev", "fdn_stats_dev", "fdn_stats_slices_dev", "fdn_convert_dev", "fdn_truncate_dev", "fdn_reserve_3d", "fdn_reserve_stack", "fdn_filter_3d_sharded", "fdn_sweep_stack_dev", "fdn_permute_dev",
    "fdn_enable_timers", "fdn_get_timers", "fdn_add_timer", "fdn_version",
]

class FdnMsg(ctypes.Structure):
    """fdn_msg (include/flowdn.h)."""
    _fields_ = [("d_buf", ctypes.c_void_p), ("bytes", ctypes.c_size_t), ("peer", ctypes.c_int), ("is_send", ctypes.c_int)]


EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(FdnMsg), ctypes.c_void_p)
ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)


class FdnComm(ctypes.Structure):
    """fdn_comm (include/flowdn.h)."""
    _fields_ = [("ctx", ctypes.c_void_p), ("rank", ctypes.c_int), ("world", ctypes.c_int),
                ("exchange", EXCHANGE_FN), ("allgather_host", ALLGATHER_FN)]


_lib = None
_rccl = None
RCCL_LIB_PATH = os.path.join(_HERE, "libflowdn_rccl.so")
TRANSPORT_RCCL, TRANSPORT_SHM, TRANSPORT_NULL = 0, 1, 2
# every symbol include/flowdn_rccl.h declares
RCCL_EXPORTS = ["fdn_transport_create", "fdn_transport_destroy", "fdn_transport_last_error", "fdn_transport_describe",
                "fdn_transport_comm", "fdn_transport_exchange", "fdn_transport_allgather_host", "fdn_transport_barrier",
                "fdn_transport_count", "fdn_transport_device_id", "fdn_transport_abort"]
_torch_runtime = None      # directory of the torch-bundled ROCm libraries when libflowdn.so was bound to them


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.  Two HIP runtimes in one process do not
    both see the GPU, so if a torch installation exists (torch is optional: bench.py and the
    multi-GPU engine use it for device memory and RCCL) its runtime is loaded first and
    libflowdn.so binds to that one, whichever of the two modules the program imports first.
    torch itself is NOT imported here."""
    import importlib.util
    import sys
    global _torch_runtime
    if os.environ.get("FDN_SYSTEM_ROCM") == "1" and "torch" not in sys.modules:
        return      # a process that will never import torch (the ranks of `flowdenoising.py --gpus N`): /opt/rocm's runtime
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    cand = os.path.join(libdir, "libamdhip64.so")
    if os.path.exists(cand):
        _torch_runtime = libdir
        if "torch" in sys.modules:
            return
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load libflowdn.so; raises FlowdnError if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FlowdnError(
            f"{LIB_PATH} not found: build it with `make -C flowdenoising_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback")
    _share_hip_runtime_with_torch()
    lib = ctypes.CDLL(LIB_PATH)
    lib.fdn_last_error.restype = ctypes.c_char_p
    lib.fdn_version.restype = ctypes.c_char_p
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("fdn_last_error", "fdn_version"):
            fn.restype = ctypes.c_int
    _lib = lib
    return lib


def load_rccl():
    """Load libflowdn_rccl.so (include/flowdn_rccl.h: the native transports of fdn_filter_3d_sharded).  RCCL has to sit on
    the HIP runtime libflowdn.so is bound to: where that is the torch-bundled one (a process that imports torch), the
    torch-bundled librccl -- same soname -- is loaded first and the dynamic linker resolves our dependency to it."""
    global _rccl
    if _rccl is not None:
        return _rccl
    load()
    if not os.path.exists(RCCL_LIB_PATH):
        raise FlowdnError(f"{RCCL_LIB_PATH} not found: build it with `make -C flowdenoising_amd/csrc`")
    if _torch_runtime is not None:
        cand = os.path.join(_torch_runtime, "librccl.so")
        if os.path.exists(cand):
            try:
                ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                pass
    lib = ctypes.CDLL(RCCL_LIB_PATH)
    for name in RCCL_EXPORTS:
        getattr(lib, name).restype = ctypes.c_int
    lib.fdn_transport_last_error.restype = ctypes.c_char_p
    lib.fdn_transport_describe.restype = ctypes.c_char_p
    lib.fdn_transport_describe.argtypes = [ctypes.c_void_p]
    lib.fdn_transport_comm.restype = ctypes.c_void_p
    lib.fdn_transport_comm.argtypes = [ctypes.c_void_p]
    lib.fdn_transport_count.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    lib.fdn_transport_device_id.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
    lib.fdn_transport_abort.argtypes = [ctypes.c_void_p]
    _rccl = lib
    return lib


class Transport:
    """One fdn_transport (include/flowdn_rccl.h): `kind` "rccl" (one GPU per rank, ncclSend / ncclRecv over xGMI), "shm"
    (ranks sharing a GPU, staged through shared memory: a rehearsal) or "null" (moves nothing: per-rank overhead
    emulation).  Creation is collective over the ranks of the job; `rendezvous` is the directory the launcher made."""

    KINDS = {"rccl": TRANSPORT_RCCL, "shm": TRANSPORT_SHM, "null": TRANSPORT_NULL}

    def __init__(self, kind, rank, world, device=-1, rendezvous=None):
        self._lib = load_rccl()
        self._t = ctypes.c_void_p()
        self.kind, self.rank, self.world, self.device = kind, int(rank), int(world), int(device)
        rc = self._lib.fdn_transport_create(ctypes.c_int(self.KINDS[kind]), ctypes.c_int(self.rank), ctypes.c_int(self.world),
                                            ctypes.c_int(self.device), (rendezvous or "").encode(), ctypes.byref(self._t))
        self._check(rc)

    def _check(self, rc):
        if rc < 0:
            raise FlowdnError(self._lib.fdn_transport_last_error().decode("utf-8", "replace"))

    def describe(self):
        return self._lib.fdn_transport_describe(self._t).decode()

    def count(self):
        """Ranks as the communicator itself reports them (RCCL: ncclCommCount)."""
        n = ctypes.c_int()
        self._check(self._lib.fdn_transport_count(self._t, ctypes.byref(n)))
        return n.value

    def device_id(self):
        """PCI bus id of this rank's device ("host" without one)."""
        buf = ctypes.create_string_buffer(64)
        self._check(self._lib.fdn_transport_device_id(self._t, buf, ctypes.c_int(64)))
        return buf.value.decode()

    def devices(self):
        """Every rank's device_id(), in rank order (collective)."""
        raw = self.allgather_host(self.device_id().encode().ljust(64, b"\0"))
        return [raw[i * 64:(i + 1) * 64].rstrip(b"\0").decode() for i in range(self.world)]

    def abort(self):
        """Tell the other ranks that this one cannot go on (they stop waiting); only close() may follow."""
        if self._t:
            self._lib.fdn_transport_abort(self._t)

    def comm_ptr(self):
        """const fdn_comm* for fdn_filter_3d_sharded."""
        p = self._lib.fdn_transport_comm(self._t)
        if not p:
            self._check(-1)
        return ctypes.c_void_p(p)

    def exchange(self, msgs, stream=None):
        """msgs = [(device pointer, nbytes, peer, is_send)]: one batched group on `stream` (a hipStream_t as an int)."""
        arr = (FdnMsg * max(len(msgs), 1))()
        for i, (p, n, peer, snd) in enumerate(msgs):
            arr[i] = FdnMsg(ctypes.c_void_p(int(p)), int(n), int(peer), int(bool(snd)))
        self._check(self._lib.fdn_transport_exchange(self._t, ctypes.c_int(len(msgs)), arr, ctypes.c_void_p(int(stream or 0))))

    def allgather_host(self, send):
        """bytes of every rank, concatenated in rank order."""
        send = bytes(send)
        out = ctypes.create_string_buffer(len(send) * self.world)
        self._check(self._lib.fdn_transport_allgather_host(self._t, send, out, ctypes.c_size_t(len(send))))
        return out.raw

    def allgather_array(self, arr):
        """(world,) + arr.shape array: every rank's `arr` (same shape and dtype on all ranks)."""
        arr = np.ascontiguousarray(arr)
        raw = self.allgather_host(arr.tobytes())
        return np.frombuffer(raw, dtype=arr.dtype).reshape((self.world,) + arr.shape).copy()

    def barrier(self):
        self._check(self._lib.fdn_transport_barrier(self._t))

    def close(self):
        if self._t:
            self._lib.fdn_transport_destroy(self._t)
            self._t = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def check(rc):
    if rc < 0:
        raise FlowdnError(load().fdn_last_error().decode("utf-8", "replace"))
    return rc


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def gaussian_kernel(sigma):
    """fdn_gaussian_kernel: host-only, needs no GPU."""
    lib = load()
    out = np.empty(4096, dtype=np.float64)
    K = lib.fdn_gaussian_kernel(ctypes.c_double(float(sigma)), _ptr(out), ctypes.c_int(out.size))
    if K <= 0:
        raise FlowdnError(lib.fdn_last_error().decode() or f"kernel too long ({-K})")
    return out[:K].copy()


def mean_host(vol):
    """fdn_mean_host: numpy's own float32 mean, restated in C (host only, needs no GPU)."""
    lib = load()
    vol = np.ascontiguousarray(vol, dtype=np.float32)
    m = ctypes.c_float()
    check(lib.fdn_mean_host(_ptr(vol), ctypes.c_size_t(vol.size), ctypes.byref(m)))
    return np.float32(m.value)


def device_count():
    """HIP devices visible to this process (fdn_device_count)."""
    n = ctypes.c_int()
    check(load().fdn_device_count(ctypes.byref(n)))
    return n.value


def device_pci_id(device):
    """PCI bus id of a visible HIP device (fdn_device_pci_id)."""
    buf = ctypes.create_string_buffer(64)
    check(load().fdn_device_pci_id(ctypes.c_int(int(device)), buf, ctypes.c_int(64)))
    return buf.value.decode()


def combine_slice_stats(first, count, second=None):
    """Volume statistics from per-slice ones, slices added in order (python floats: IEEE doubles, one fixed order):
    first = fdn_stats_slices_dev(centre 0) rows {min, max, sum, .}; second = the rows taken with centre = mean."""
    mn, mx, tot = first[0][0], first[0][1], 0.0
    for a, b, s, _ in first:
        mn = a if (a < mn or a != a) else mn
        mx = b if (b > mx or b != b) else mx
        tot += float(s)
    out = {"min": float(mn), "max": float(mx), "mean": tot / float(count), "std": float("nan")}
    if second is not None:
        sq = 0.0
        for row in second:
            sq += float(row[3])
        out["std"] = float(np.sqrt(sq / float(count)))
    return out


class Handle:
    """One fdn_handle: a GPU, its stream and the library-owned scratch."""

    def __init__(self, device=0):
        self._lib = load()
        self._h = ctypes.c_void_p()
        check(self._lib.fdn_create(ctypes.c_int(device), ctypes.byref(self._h)))
        self.device = device
        self.options = {}          # what set_option has been given (operators' pair-handle pool copies it to its extra handles)

    def close(self):
        if self._h:
            self._lib.fdn_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing ------------------------------------------------------------------
    def set_stream(self, stream_ptr):
        """Enqueue on this HIP stream (0/None = the legacy default stream torch uses by default)."""
        check(self._lib.fdn_set_stream(self._h, ctypes.c_void_p(stream_ptr or 0)))

    def reset_stream(self):
        check(self._lib.fdn_reset_stream(self._h))

    def synchronize(self):
        check(self._lib.fdn_synchronize(self._h))

    def set_workspace_limit(self, nbytes):
        check(self._lib.fdn_set_workspace_limit(self._h, ctypes.c_size_t(int(nbytes))))

    def workspace_bytes(self):
        """Device memory the handle owns right now (fdn_workspace_bytes)."""
        n = ctypes.c_size_t()
        check(self._lib.fdn_workspace_bytes(self._h, ctypes.byref(n)))
        return n.value

    def mem_info(self):
        """(free, total) bytes of device memory."""
        f, t = ctypes.c_size_t(), ctypes.c_size_t()
        check(self._lib.fdn_mem_info(self._h, ctypes.byref(f), ctypes.byref(t)))
        return f.value, t.value

    def set_option(self, name, value):
        """fdn_set_option: "strict_order", "path", "fused_occ", "lds_pad", "shard_loopback", "sub_batches" (see include/flowdn.h)."""
        check(self._lib.fdn_set_option(self._h, ctypes.c_char_p(name.encode()), ctypes.c_long(int(value))))
        self.options = dict(self.options, **{name: int(value)})        # (a new dict: readers on other threads see old or new, whole)

    def get_option(self, name):
        """fdn_get_option: an option's value; also "last_sub_batches" and "compute_units" (read-only)."""
        v = ctypes.c_long()
        check(self._lib.fdn_get_option(self._h, ctypes.c_char_p(name.encode()), ctypes.byref(v)))
        return v.value

    def malloc(self, nbytes):
        p = ctypes.c_void_p()
        check(self._lib.fdn_malloc(self._h, ctypes.c_size_t(int(nbytes)), ctypes.byref(p)))
        return p.value

    def free(self, dptr):
        check(self._lib.fdn_free(self._h, ctypes.c_void_p(dptr)))

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        check(self._lib.fdn_memcpy_h2d(self._h, ctypes.c_void_p(dptr), _ptr(arr), ctypes.c_size_t(arr.nbytes)))

    def d2h(self, arr, dptr):
        assert arr.flags["C_CONTIGUOUS"]
        check(self._lib.fdn_memcpy_d2h(self._h, _ptr(arr), ctypes.c_void_p(dptr), ctypes.c_size_t(arr.nbytes)))

    def h2d_2d(self, dptr, dpitch, host_ptr, spitch, width_bytes, height):
        check(self._lib.fdn_memcpy2d_h2d(self._h, ctypes.c_void_p(dptr), ctypes.c_size_t(dpitch), ctypes.c_void_p(host_ptr),
                                         ctypes.c_size_t(spitch), ctypes.c_size_t(width_bytes), ctypes.c_size_t(height)))

    def d2h_2d(self, host_ptr, dpitch, dptr, spitch, width_bytes, height):
        check(self._lib.fdn_memcpy2d_d2h(self._h, ctypes.c_void_p(host_ptr), ctypes.c_size_t(dpitch), ctypes.c_void_p(dptr),
                                         ctypes.c_size_t(spitch), ctypes.c_size_t(width_bytes), ctypes.c_size_t(height)))

    def host_register(self, arr):
        """Page-lock a numpy array's memory; returns True if it worked (a read-only mapping, say, cannot be locked)."""
        return self._lib.fdn_host_register(self._h, ctypes.c_void_p(arr.ctypes.data), ctypes.c_size_t(arr.nbytes)) == 0

    def host_unregister(self, arr):
        self._lib.fdn_host_unregister(self._h, ctypes.c_void_p(arr.ctypes.data))

    def memset_f32(self, dptr, value, count):
        check(self._lib.fdn_memset_f32(self._h, ctypes.c_void_p(dptr), ctypes.c_float(float(value)), ctypes.c_size_t(int(count))))

    def enable_timers(self, on=True):
        check(self._lib.fdn_enable_timers(self._h, ctypes.c_int(int(on))))

    def timers(self, reset=False):
        """{name: (milliseconds, launches)} measured with HIP events on the handle's stream."""
        ms = (ctypes.c_double * len(TIMER_NAMES))()
        cnt = (ctypes.c_longlong * len(TIMER_NAMES))()
        check(self._lib.fdn_get_timers(self._h, ms, cnt, ctypes.c_int(int(reset))))
        return {n: (ms[i], cnt[i]) for i, n in enumerate(TIMER_NAMES)}

    def add_timer(self, name, ms, count=1):
        """fdn_add_timer: time measured by the host layer (the multi-GPU exchanges) into the handle's timer table."""
        check(self._lib.fdn_add_timer(self._h, ctypes.c_int(TIMER_NAMES.index(name)), ctypes.c_double(float(ms)), ctypes.c_longlong(int(count))))

    # -- pair operators ------------------------------------------------------------
    @staticmethod
    def _view(img):
        """(pointer, row stride, column stride) of a 2-D float32 view, strides in elements: no copy for the slice
        views the reference passes (padded_vol[:, y + i, :], seq:255; padded_vol[:, :, x + i], seq:333)."""
        img = np.asarray(img)
        if img.dtype != np.float32 or img.ndim != 2 or img.strides[0] % 4 or img.strides[1] % 4:
            img = np.ascontiguousarray(img, dtype=np.float32)      # other dtypes: converted (cv2 converts to f32 too)
        return img, ctypes.c_void_p(img.ctypes.data), ctypes.c_ssize_t(img.strides[0] // 4), ctypes.c_ssize_t(img.strides[1] // 4)

    @staticmethod
    def _typed_view(img):
        """(array, depth, pointer, row stride, column stride) of a 2-D image in one of the depths cv2 would be handed by the
        reference (FDN_DEPTH_*); anything else is converted to float32."""
        img = np.asarray(img)
        depth = DEPTHS.get(img.dtype.newbyteorder("=")) if img.dtype.isnative else None
        isz = img.dtype.itemsize
        if depth is None or img.ndim != 2 or img.strides[0] % isz or img.strides[1] % isz:
            img, depth, isz = np.ascontiguousarray(img, dtype=np.float32), DEPTH_F32, 4
        return img, depth, ctypes.c_void_p(img.ctypes.data), ctypes.c_ssize_t(img.strides[0] // isz), ctypes.c_ssize_t(img.strides[1] // isz)

    def farneback(self, prev, next, flow, levels, winsize, iters, poly_n, poly_sigma, flags):
        if np.asarray(prev).dtype != np.float32 or np.asarray(next).dtype != np.float32:
            # slices of an integer MRC or of seq's float64 padded volume: converted like cv2's convertTo(CV_32F), in the library
            prev, pd, pp, prs, pcs = self._typed_view(prev)
            next, nd, np_, nrs, ncs = self._typed_view(next)
            H, W = prev.shape
            if next.shape != (H, W):
                raise ValueError("prev and next must have the same shape")
            if flags & USE_INITIAL_FLOW:
                if flow is None or flow.shape != (H, W, 2) or flow.dtype != np.float32 or not flow.flags["C_CONTIGUOUS"]:
                    raise ValueError("USE_INITIAL_FLOW needs a contiguous (H, W, 2) float32 flow")
            else:
                flow = np.zeros((H, W, 2), dtype=np.float32)
            check(self._lib.fdn_farneback_typed(self._h, pp, ctypes.c_int(pd), prs, pcs, np_, ctypes.c_int(nd), nrs, ncs, _ptr(flow),
                                                ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(int(levels)), ctypes.c_int(int(winsize)),
                                                ctypes.c_int(int(iters)), ctypes.c_int(int(poly_n)), ctypes.c_double(float(poly_sigma)),
                                                ctypes.c_int(int(flags))))
            return flow
        prev, pp, prs, pcs = self._view(prev)
        next, np_, nrs, ncs = self._view(next)
        H, W = prev.shape
        if next.shape != (H, W):
            raise ValueError("prev and next must have the same shape")
        if flags & USE_INITIAL_FLOW:
            if flow is None or flow.shape != (H, W, 2) or flow.dtype != np.float32 or not flow.flags["C_CONTIGUOUS"]:
                raise ValueError("USE_INITIAL_FLOW needs a contiguous (H, W, 2) float32 flow")
        else:
            flow = np.zeros((H, W, 2), dtype=np.float32)
        check(self._lib.fdn_farneback_strided(self._h, pp, prs, pcs, np_, nrs, ncs, _ptr(flow), ctypes.c_int(H), ctypes.c_int(W),
                                              ctypes.c_int(int(levels)), ctypes.c_int(int(winsize)), ctypes.c_int(int(iters)),
                                              ctypes.c_int(int(poly_n)), ctypes.c_double(float(poly_sigma)),
                                              ctypes.c_int(int(flags))))
        return flow

    def warp(self, reference, flow):
        """cv2.remap returns the image's own type: float64 for a slice of seq's float64 padded volume (weights in double),
        rounded and saturated integers for a slice of an integer volume (par); float32 otherwise."""
        ref = np.asarray(reference)
        if not ref.dtype.isnative:              # e.g. a slice of a big-endian MRC
            ref = ref.astype(ref.dtype.newbyteorder("="))
        if ref.dtype != np.float32 and ref.dtype in DEPTHS:
            ref, depth, rp, rs, cs = self._typed_view(ref)
            flow = np.ascontiguousarray(flow, dtype=np.float32)
            H, W = flow.shape[:2]
            if ref.shape != (H, W) or flow.shape != (H, W, 2):
                raise ValueError("reference (H, W) and flow (H, W, 2) shapes disagree")
            dst = np.empty((H, W), dtype=ref.dtype)
            check(self._lib.fdn_warp_typed(self._h, rp, ctypes.c_int(depth), rs, cs, _ptr(flow), _ptr(dst), ctypes.c_int(H), ctypes.c_int(W)))
            return dst
        reference, rp, rs, cs = self._view(reference)
        flow = np.ascontiguousarray(flow, dtype=np.float32)
        H, W = flow.shape[:2]
        if reference.shape != (H, W) or flow.shape != (H, W, 2):
            raise ValueError("reference (H, W) and flow (H, W, 2) shapes disagree")
        dst = np.empty((H, W), dtype=np.float32)
        check(self._lib.fdn_warp_strided(self._h, rp, rs, cs, _ptr(flow), _ptr(dst), ctypes.c_int(H), ctypes.c_int(W)))
        return dst

    def farneback_dev(self, d_prev, prev_strides, d_next, next_strides, d_flow, H, W, levels, winsize, iters, poly_n, poly_sigma, flags):
        """Device pointers; strides = (row, column) in elements; d_flow (H, W, 2) contiguous, updated in place."""
        check(self._lib.fdn_farneback_dev(self._h, ctypes.c_void_p(d_prev), ctypes.c_ssize_t(prev_strides[0]), ctypes.c_ssize_t(prev_strides[1]),
                                          ctypes.c_void_p(d_next), ctypes.c_ssize_t(next_strides[0]), ctypes.c_ssize_t(next_strides[1]),
                                          ctypes.c_void_p(d_flow), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(int(levels)),
                                          ctypes.c_int(int(winsize)), ctypes.c_int(int(iters)), ctypes.c_int(int(poly_n)),
                                          ctypes.c_double(float(poly_sigma)), ctypes.c_int(int(flags))))

    def warp_dev(self, d_reference, strides, d_flow, d_dst, H, W):
        check(self._lib.fdn_warp_dev(self._h, ctypes.c_void_p(d_reference), ctypes.c_ssize_t(strides[0]), ctypes.c_ssize_t(strides[1]),
                                     ctypes.c_void_p(d_flow), ctypes.c_void_p(d_dst), ctypes.c_int(H), ctypes.c_int(W)))

    # -- volume operators ----------------------------------------------------------
    @staticmethod
    def _kernels(kernels):
        keep = []
        ptrs = (ctypes.c_void_p * 3)()
        Ks = (ctypes.c_int * 3)()
        for a in range(3):
            k = kernels[a]
            if k is None:
                ptrs[a] = None
                Ks[a] = 0
            else:
                k = np.ascontiguousarray(k, dtype=np.float64)
                keep.append(k)
                ptrs[a] = k.ctypes.data
                Ks[a] = k.size
        return ptrs, Ks, keep

    def filter_axis(self, vol, axis, kernel, pad_value, params):
        vol = np.ascontiguousarray(vol, dtype=np.float32)
        Z, Y, X = vol.shape
        kernel = np.ascontiguousarray(kernel, dtype=np.float64)
        out = np.empty_like(vol)
        check(self._lib.fdn_filter_axis(self._h, _ptr(vol), _ptr(out), ctypes.c_int(Z), ctypes.c_int(Y), ctypes.c_int(X),
                                        ctypes.c_int(axis), _ptr(kernel), ctypes.c_int(kernel.size),
                                        ctypes.c_float(float(pad_value)), ctypes.byref(params)))
        return out

    def filter_3d(self, vol, kernels, pad_value, params, out=None):
        """`out`: a float32 C-contiguous array to receive the result (a fresh one costs its first-touch page faults:
        ~0.1 s per 2 GiB, more than the PCIe transfer)."""
        vol = np.ascontiguousarray(vol, dtype=np.float32)
        Z, Y, X = vol.shape
        ptrs, Ks, keep = self._kernels(kernels)
        if out is None:
            out = np.empty_like(vol)
        elif out.shape != vol.shape or out.dtype != np.float32 or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be a C-contiguous float32 array of the volume's shape")
        check(self._lib.fdn_filter_3d(self._h, _ptr(vol), _ptr(out), ctypes.c_int(Z), ctypes.c_int(Y), ctypes.c_int(X),
                                      ptrs, Ks, ctypes.c_float(float(pad_value)), ctypes.byref(params)))
        return out

    def filter_3d_dev(self, d_in, d_out, shape, kernels, pad_value, params):
        Z, Y, X = shape
        ptrs, Ks, keep = self._kernels(kernels)
        check(self._lib.fdn_filter_3d_dev(self._h, ctypes.c_void_p(d_in), ctypes.c_void_p(d_out), ctypes.c_int(Z),
                                          ctypes.c_int(Y), ctypes.c_int(X), ptrs, Ks,
                                          ctypes.c_float(float(pad_value)), ctypes.byref(params)))

    def reserve_3d(self, shape, Ks, params):
        """Allocate what filter_3d_dev(shape, kernels of Ks taps, params) will use; launches nothing (fdn_reserve_3d)."""
        Z, Y, X = shape
        arr = (ctypes.c_int * 3)(*[int(k) for k in Ks])
        check(self._lib.fdn_reserve_3d(self._h, ctypes.c_int(Z), ctypes.c_int(Y), ctypes.c_int(X), arr, ctypes.byref(params)))

    def filter_3d_sharded(self, d_slab_in, d_slab_out, shape, kernels, params, comm):
        """fdn_filter_3d_sharded: this rank's Z-slab in, filtered Z-slab out, the transport behind `comm`:
        an object with .rank, .world, .exchange(msgs, stream) -- msgs = [(device pointer, nbytes, peer, is_send)] --
        and .allgather_host(send: bytes) -> bytes of all ranks in rank order (distributed.TorchComm is one) -- or a
        Transport (RCCL / shared memory / null, libflowdn_rccl.so), whose C callbacks the library then calls directly."""
        Z, Y, X = shape
        ptrs, Ks, keep = self._kernels(kernels)
        if isinstance(comm, Transport):         # a native transport (libflowdn_rccl.so): no Python in the data path
            check(self._lib.fdn_filter_3d_sharded(self._h, ctypes.c_void_p(d_slab_in), ctypes.c_void_p(d_slab_out), ctypes.c_int(Z),
                                                  ctypes.c_int(Y), ctypes.c_int(X), ptrs, Ks, ctypes.byref(params), comm.comm_ptr()))
            return
        errors = []

        def _exchange(ctx, n, msgs, stream):
            try:
                comm.exchange([(msgs[i].d_buf, msgs[i].bytes, msgs[i].peer, bool(msgs[i].is_send)) for i in range(n)], stream)
                return 0
            except Exception as e:          # no exception may cross the C boundary
                errors.append(e)
                return -1

        def _allgather(ctx, send, recv, nbytes):
            try:
                out = comm.allgather_host(ctypes.string_at(send, nbytes))
                ctypes.memmove(recv, out, len(out))
                return 0
            except Exception as e:
                errors.append(e)
                return -1

        c = FdnComm(None, int(comm.rank), int(comm.world), EXCHANGE_FN(_exchange), ALLGATHER_FN(_allgather))
        rc = self._lib.fdn_filter_3d_sharded(self._h, ctypes.c_void_p(d_slab_in), ctypes.c_void_p(d_slab_out), ctypes.c_int(Z),
                                             ctypes.c_int(Y), ctypes.c_int(X), ptrs, Ks, ctypes.byref(params), ctypes.byref(c))
        if errors:
            raise errors[0]
        check(rc)

    def filter_axis_dev(self, d_in, d_out, shape, axis, kernel, pad_value, params):
        Z, Y, X = shape
        kernel = np.ascontiguousarray(kernel, dtype=np.float64)
        check(self._lib.fdn_filter_axis_dev(self._h, ctypes.c_void_p(d_in), ctypes.c_void_p(d_out), ctypes.c_int(Z),
                                            ctypes.c_int(Y), ctypes.c_int(X), ctypes.c_int(axis), _ptr(kernel),
                                            ctypes.c_int(kernel.size), ctypes.c_float(float(pad_value)),
                                            ctypes.byref(params)))

    def sweep_stack_dev(self, d_stack, d_out, S, H, W, kernel, params):
        kernel = np.ascontiguousarray(kernel, dtype=np.float64)
        check(self._lib.fdn_sweep_stack_dev(self._h, ctypes.c_void_p(d_stack), ctypes.c_void_p(d_out), ctypes.c_int(S),
                                            ctypes.c_int(H), ctypes.c_int(W), _ptr(kernel), ctypes.c_int(kernel.size),
                                            ctypes.byref(params)))

    def reserve_stack(self, S, H, W, K, params):
        """Allocate what sweep_stack_dev(S, H, W, K taps, params) will use; launches nothing (fdn_reserve_stack)."""
        check(self._lib.fdn_reserve_stack(self._h, ctypes.c_int(S), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(K), ctypes.byref(params)))

    def permute_dev(self, d_in, d_out, dims, strides):
        A, B, C = dims
        sa, sb, sc = strides
        check(self._lib.fdn_permute_dev(self._h, ctypes.c_void_p(d_in), ctypes.c_void_p(d_out), ctypes.c_int(A),
                                        ctypes.c_int(B), ctypes.c_int(C), ctypes.c_int64(sa), ctypes.c_int64(sb),
                                        ctypes.c_int64(sc)))

    def mean_dev(self, d_in, count):
        m = ctypes.c_float()
        check(self._lib.fdn_mean_dev(self._h, ctypes.c_void_p(d_in), ctypes.c_size_t(int(count)), ctypes.byref(m)))
        return np.float32(m.value)

    def stats_dev(self, d_in, count):
        """{min, max, mean, std} of a device float32 array (float64 arithmetic): the MRC header statistics and the
        volume statistics the reference logs (seq:529-532, 547-550, 562-564)."""
        out = (ctypes.c_double * 4)()
        check(self._lib.fdn_stats_dev(self._h, ctypes.c_void_p(d_in), ctypes.c_size_t(int(count)), out))
        return {"min": out[0], "max": out[1], "mean": out[2], "std": out[3]}

    def stats_volume(self, d_in, shape):
        """The same four numbers from per-slice reductions added up in slice order (fdn_stats_slices_dev): independent of
        how a volume is cut into Z-slabs, so a multi-GPU run writes the single-GPU header bit for bit."""
        n = int(shape[0])
        per = int(np.prod(shape[1:]))
        first = self.stats_slices_dev(d_in, n, per, 0.0)
        mean = combine_slice_stats(first, n * per)["mean"]
        return combine_slice_stats(first, n * per, self.stats_slices_dev(d_in, n, per, mean))

    def stats_slices_dev(self, d_in, nslices, slice_elems, centre):
        """(nslices, 4) float64: min, max, sum, sum of squared deviations from `centre` of every slice."""
        out = np.empty((int(nslices), 4), dtype=np.float64)
        check(self._lib.fdn_stats_slices_dev(self._h, ctypes.c_void_p(d_in), ctypes.c_int(int(nslices)), ctypes.c_size_t(int(slice_elems)),
                                             ctypes.c_double(float(centre)), ctypes.c_void_p(out.ctypes.data)))
        return out

    def convert_dev(self, d_src, dtype, d_dst, count):
        """d_dst (float32) = float32(d_src) for a device array of an 8- or 16-bit integer dtype."""
        depth = DEPTHS[np.dtype(dtype).newbyteorder("=")]
        check(self._lib.fdn_convert_dev(self._h, ctypes.c_void_p(d_src), ctypes.c_int(depth), ctypes.c_void_p(d_dst),
                                        ctypes.c_size_t(int(count))))

    def truncate_dev(self, d_src, dtype, d_dst, count):
        """d_dst (uint8 / uint16) = d_src.astype(dtype) for a device float32 array (numpy's truncating cast, seq:566-571)."""
        depth = DEPTHS[np.dtype(dtype)]
        check(self._lib.fdn_truncate_dev(self._h, ctypes.c_void_p(d_src), ctypes.c_int(depth), ctypes.c_void_p(d_dst),
                                         ctypes.c_size_t(int(count))))

    def np_chunk_sums_dev(self, d_in, count):
        """numpy's float32 pairwise sums of the 8192-element chunks of a device array (the last may be partial)."""
        out = np.empty((int(count) + 8191) // 8192, dtype=np.float32)
        check(self._lib.fdn_np_chunk_sums_dev(self._h, ctypes.c_void_p(d_in), ctypes.c_size_t(int(count)),
                                              ctypes.c_void_p(out.ctypes.data)))
        return out

    def sum_dev(self, d_in, count):
        s = ctypes.c_double()
        check(self._lib.fdn_sum_dev(self._h, ctypes.c_void_p(d_in), ctypes.c_size_t(int(count)), ctypes.byref(s)))
        return s.value
