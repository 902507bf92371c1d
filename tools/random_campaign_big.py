"""Randomised parity campaign on larger images (several bands, pyramids with all their levels)."""
import sys, numpy as np
sys.path.insert(0, ".")
import flowdenoising_amd as fdn
from flowdenoising_amd.synth import make_volume
from oracle import oracle
oracle.build()
import os        # the product reads these at handle creation (opencv_fma, remap_model): the oracle follows the same switches
oracle.set_fma(int(os.environ.get("FDN_OPENCV_FMA", 0)), int(os.environ.get("FDN_OPENCV_FMA_LANES", 8)))
oracle.set_remap_model(int(os.environ.get("FDN_REMAP_MODEL", 0)))
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = order = 0
worst = 0.0
for i in range(n):
    axis = int(rng.integers(0, 3))
    small, a, b = int(rng.integers(3, 8)), int(rng.integers(100, 420)), int(rng.integers(100, 640))
    shape = [0, 0, 0]; shape[axis] = small
    rest = [x for x in range(3) if x != axis]; shape[rest[0]], shape[rest[1]] = a, b
    w = int(rng.choice([3, 4, 5, 5, 6, 7, 8, 9, 11, 13, 15, 15, 21])); l = int(rng.integers(0, 4))
    sigma = float(rng.choice([0.5, 1.0])); border = int(rng.integers(0, 2)); chained = bool(rng.integers(0, 2))
    vol = make_volume(tuple(shape), seed=5000 + i, amplitude=100.0)
    k = fdn.get_gaussian_kernel(sigma); mean = vol.mean()
    fn = [fdn.OF_filter_along_Z, fdn.OF_filter_along_Y, fdn.OF_filter_along_X][axis]
    got = fn(vol, k, l, w, mean, border_mode=border, chained=chained)
    want = oracle.filter_along_axis(vol, axis, k, l, w, mean, border_mode=border, chained=chained, nthreads=16)
    if not np.array_equal(got, want):
        # the only designed difference: OpenCV's serial f64 running sum along x against the kernels' direct window sum
        want2 = oracle.filter_along_axis(vol, axis, k, l, w, mean, border_mode=border, chained=chained, box_mode=4 if w >= 10 else 2, nthreads=16)
        err = np.abs(got - want).max() / np.abs(want).max()
        kind = "f64 summation order only" if np.array_equal(got, want2) else "REAL MISMATCH"
        bad += kind == "REAL MISMATCH"
        order += kind != "REAL MISMATCH"
        worst = max(worst, err)
        print(kind, shape, axis, l, w, sigma, border, chained, err, flush=True)
print(f"{n} cases: {n - bad - order} bit-identical to the OpenCV-order oracle, {order} differ by f64 summation order only "
      f"(worst {worst:.2e} relative; bit-identical to the kernel-order oracle), {bad} real mismatches")
