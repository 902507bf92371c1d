#!/usr/bin/env python3
"""Generate the reference-derived golden vectors in tests/golden/.

Runs ONLY in the build container (needs /root/reference).  It imports the reference's
src/flowdenoising_sequential.py and calls the two parts of it that can run without
OpenCV: get_gaussian_kernel (seq:30-41) and no_OF_filter (seq:426-431 ->
seq:171-192, 290-311, 396-417).  Those functions use numpy/scipy only.

The reference's top-level imports of cv2 / mrcfile / skimage / tifffile / imageio
(absent from this image; an ordinary ModuleNotFoundError, not a refusal) are
satisfied by EMPTY placeholder modules: they carry the three integer constants the
module reads at import time and no functionality, and nothing that would need them
(get_flow, warp_slice, OF_filter*, file I/O) is ever called here.  So every number
written below was computed by the reference's own Python + numpy + scipy.

Outputs (data only: inputs and expected outputs):
  ref_kernels.npz   sigma list + kernels from seq.get_gaussian_kernel
  ref_no_of.npz     seeded input volume, sigmas, seq.no_OF_filter output
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference/src/flowdenoising_sequential.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_seq():
    cv2 = types.ModuleType("cv2")
    cv2.BORDER_REPLICATE = 1
    cv2.INTER_LINEAR = 1
    cv2.OPTFLOW_USE_INITIAL_FLOW = 4
    placeholders = {"cv2": cv2}
    for name in ("mrcfile", "skimage", "skimage.io", "tifffile", "imageio"):
        placeholders[name] = types.ModuleType(name)
    placeholders["skimage"].io = placeholders["skimage.io"]
    saved = {k: sys.modules.get(k) for k in placeholders}
    sys.modules.update(placeholders)
    try:
        spec = importlib.util.spec_from_file_location("fd_seq_reference", REF)
        seq = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(seq)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return seq


def main():
    seq = load_seq()
    sigmas = [0.1, 0.5, 1.0, 1.5, 2.0, 2.5, 3.0, 4.0]
    kernels = {f"k{i}": seq.get_gaussian_kernel(s) for i, s in enumerate(sigmas)}
    np.savez(os.path.join(HERE, "ref_kernels.npz"), sigmas=np.array(sigmas), **kernels)

    rng = np.random.default_rng(20261003)
    vol = (rng.standard_normal((12, 10, 14)) * 50 + 100).astype(np.float32)
    sig = [1.0, 1.5, 0.5]
    ks = [seq.get_gaussian_kernel(s) for s in sig]
    seq.l, seq.w = 0, 5  # module globals the no_OF log lines read (seq:173)
    out = seq.no_OF_filter(vol, ks)
    assert out.dtype == np.float32
    np.savez(os.path.join(HERE, "ref_no_of.npz"), vol=vol, sigmas=np.array(sig), out=out,
             mean=np.float32(vol.mean()))
    print("kernels:", [len(k) for k in kernels.values()], " no_OF out:", out.shape, out.dtype)


if __name__ == "__main__":
    main()
