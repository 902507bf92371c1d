"""One-off randomised parity campaign: many more random sweep configurations than the test suite holds.
usage: random_campaign.py [n] [seed] [wide|strict]   -- `wide`: windows 10 .. 31 (the one-iteration kernel) instead of 3 .. 15;
`strict`: strict_order on (OpenCV's own f64 summation order, segment-walked rows): every case must then be bit-identical"""
import sys, importlib.util, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
spec = importlib.util.spec_from_file_location("tp", "tests/test_gpu_parity.py"); tp = importlib.util.module_from_spec(spec); spec.loader.exec_module(tp)
import flowdenoising_amd as fdn
from oracle import oracle
oracle.build()
import os        # the product reads these at handle creation (opencv_fma, remap_model): the oracle follows the same switches
oracle.set_fma(int(os.environ.get("FDN_OPENCV_FMA", 0)), int(os.environ.get("FDN_OPENCV_FMA_LANES", 8)))
oracle.set_remap_model(int(os.environ.get("FDN_REMAP_MODEL", 0)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
bad = 0; exact = 0; order = 0
cases = tp._random_cases(n, int(sys.argv[2]) if len(sys.argv) > 2 else 777)
if len(sys.argv) > 3 and sys.argv[3] == "wide":
    rng = np.random.default_rng(99)
    cases = [(shape, axis, l, int(rng.choice([10, 11, 13, 15, 15, 17, 21, 31])), sigma, border, chained, seed)
             for (shape, axis, l, w, sigma, border, chained, seed) in cases]
if len(sys.argv) > 3 and sys.argv[3] == "strict":
    from flowdenoising_amd.operators import handle
    handle().set_option("strict_order", 1)
for i, (shape, axis, l, w, sigma, border, chained, seed) in enumerate(cases):
    vol = tp._vol(shape, seed=seed)
    k = fdn.get_gaussian_kernel(sigma)
    mean = vol.mean()
    fn = [fdn.OF_filter_along_Z, fdn.OF_filter_along_Y, fdn.OF_filter_along_X][axis]
    got = fn(vol, k, l, w, mean, border_mode=border, chained=chained)
    want = oracle.filter_along_axis(vol, axis, k, l, w, mean, border_mode=border, chained=chained, nthreads=8)
    if np.array_equal(got, want):
        exact += 1
    else:
        want2 = oracle.filter_along_axis(vol, axis, k, l, w, mean, border_mode=border, chained=chained, box_mode=4 if w >= 10 else 2, nthreads=8)
        err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)
        kind = "f64 summation order only" if np.array_equal(got, want2) else "REAL MISMATCH"
        print(kind, (shape, axis, l, w, sigma, border, chained, seed), err, flush=True)
        bad += kind == "REAL MISMATCH"
        order += kind != "REAL MISMATCH"
print(f"{len(cases)} cases: {exact} bit-identical to the OpenCV-order oracle, {order} differ by f64 summation order only, {bad} real mismatches")
