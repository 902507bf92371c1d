"""The "flower" protocol of src/flowdenoising_GPU.py (gpu:92-177): an object holds the Farneback parameters, takes the
target slice once (`set_target`) and then estimates the flow towards one reference slice after another (`get_flow`),
each seeded with the previous flow (gpu:105-140, cv2.cuda's FarnebackOpticalFlow in the reference).

Here the target stays on the device between calls, like the reference's GpuMat: fdn_farneback_dev works on device images,
so a chain of K - 1 calls uploads the target once, every reference once, and the flow travels only because the protocol
returns it as a numpy array.  Results are those of cv2.calcOpticalFlowFarneback (gpu:158-168's CPU_flower): the CUDA
implementation the reference's GPU_flower wraps is a different algorithm variant, not what this library restates."""
import numpy as np

from . import _lib
from .operators import OF_ITERS, OF_POLY_N, OF_POLY_SIGMA, handle


class CPU_flower:
    """gpu:142-177.  (The name is the reference's; the work runs on the GPU.)"""

    def __init__(self, l=3, w=5, iters=OF_ITERS, polyN=OF_POLY_N, polySigma=OF_POLY_SIGMA, flags=_lib.USE_INITIAL_FLOW, device=0):
        self.l, self.w, self.iters, self.polyN, self.polySigma, self.flags = l, w, iters, polyN, polySigma, flags
        self._h = handle(device)
        self._shape = None
        self._bufs = None          # device: target | reference | flow

    def _alloc(self, shape):
        if self._shape == shape:
            return
        self.close()
        H, W = shape
        base = self._h.malloc(H * W * 4 * 4)
        self._bufs = (base, base + H * W * 4, base + 2 * H * W * 4)
        self._shape = shape

    def set_target(self, target):
        target = np.ascontiguousarray(target, dtype=np.float32)         # convertTo(CV_32F), as cv2 does with any depth
        if target.ndim != 2:
            raise ValueError("target must be a 2-D image")
        self._alloc(target.shape)
        self._h.h2d(self._bufs[0], target)

    def get_flow(self, reference, prev_flow=None):
        if self._shape is None:
            raise RuntimeError("set_target() first (gpu:105)")
        H, W = self._shape
        reference = np.ascontiguousarray(reference, dtype=np.float32)
        if reference.shape != (H, W):
            raise ValueError("reference and target must have the same shape")
        d_t, d_r, d_f = self._bufs
        self._h.h2d(d_r, reference)
        if self.flags & _lib.USE_INITIAL_FLOW:
            if prev_flow is None or prev_flow.shape != (H, W, 2):
                raise ValueError("OPTFLOW_USE_INITIAL_FLOW needs prev_flow of shape (H, W, 2)")
            self._h.h2d(d_f, np.ascontiguousarray(prev_flow, dtype=np.float32))
        # prev = target, next = reference (gpu:158-160; I0 = target, I1 = reference in gpu:129)
        self._h.farneback_dev(d_t, (W, 1), d_r, (W, 1), d_f, H, W, self.l, self.w, self.iters, self.polyN, self.polySigma, self.flags)
        flow = np.empty((H, W, 2), dtype=np.float32)
        self._h.d2h(flow, d_f)
        return flow

    def close(self):
        if self._bufs is not None:
            self._h.free(self._bufs[0])
            self._bufs, self._shape = None, None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GPU_flower(CPU_flower):
    """gpu:92-140: same protocol; positional (l, w, iters, polyN, polySigma, flags) as in the reference."""

    def __init__(self, l, w, iters, polyN, polySigma, flags=_lib.USE_INITIAL_FLOW, device=0):
        super().__init__(l, w, iters, polyN, polySigma, flags, device)
